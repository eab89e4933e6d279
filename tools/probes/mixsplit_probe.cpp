// Probe: the 2-way fp16 split of the f16x3 mode with v_fma_mixlo/hi_f16 (2 VALU instructions per value: hi = f16(x * s),
// lo = f16(fma(x, s, -hi)), both halves of a register written in place) against the adopted cvt_pk / cvt back / subtract /
// cvt_pk sequence (4 per value): bit-identical pieces on 2^24 values incl. denormal results, huge and tiny scales?
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/mixsplit_probe.cpp -o tools/probes/bin/mixsplit
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_old(float x0, float x1, float s, unsigned& hi, unsigned& lo) {
  const float a = x0 * s, b = x1 * s;
  f2 v = {a, b};
  const h2 h = __builtin_convertvector(v, h2);
  const f2 u = __builtin_convertvector(h, f2);
  f2 d = {a - u.x, b - u.y};
  const h2 l = __builtin_convertvector(d, h2);
  hi = __builtin_bit_cast(unsigned, h); lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void split_mix(float x0, float x1, float s, unsigned& hi, unsigned& lo) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(hi) : "v"(x0), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(hi) : "v"(x1), "v"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(x0), "v"(s), "v"(hi));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(x1), "v"(s), "v"(hi));
}
__global__ void k(const float* x, unsigned* o, long n, float s) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  unsigned h0, l0, h1, l1;
  split_old(x[2 * i], x[2 * i + 1], s, h0, l0);
  split_mix(x[2 * i], x[2 * i + 1], s, h1, l1);
  o[4 * i] = h0; o[4 * i + 1] = l0; o[4 * i + 2] = h1; o[4 * i + 3] = l1;
}
int main() {
  const long n = 1 << 24;
  std::vector<float> h(n);
  std::mt19937 g(3);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  std::uniform_int_distribution<int> ex(-40, 4);
  for (long i = 0; i < n; ++i) h[i] = std::ldexp(u(g), ex(g));
  h[0] = 0.f; h[1] = -0.f; h[2] = 1.f; h[3] = 65504.f / 16384.f; h[4] = 1e-30f; h[5] = -1e-30f;
  float* dx; unsigned* dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 2 * 4);
  hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
  std::vector<unsigned> o(n * 2);
  for (float s : {16384.f, 1.f, 1048576.f, 3.0517578125e-05f}) {
    hipLaunchKernelGGL(k, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, dx, dout, n, s);
    hipMemcpy(o.data(), dout, n * 2 * 4, hipMemcpyDeviceToHost);
    long bad = 0, first = -1;
    for (long i = 0; i < n / 2; ++i)
      if (o[4 * i] != o[4 * i + 2] || o[4 * i + 1] != o[4 * i + 3]) { if (first < 0) first = i; ++bad; }
    printf("scale %g: %ld of %ld pairs differ", s, bad, n / 2);
    if (first >= 0) printf("  (first: x = %g %g  old %08x %08x  mix %08x %08x)", h[2 * first], h[2 * first + 1], o[4 * first], o[4 * first + 1], o[4 * first + 2], o[4 * first + 3]);
    printf("\n");
  }
  return 0;
}
