// Probe: what does "every workgroup raises one absmax word" cost?  N workgroups of 256 threads, a little work each
// (one float4 load per thread + reduction), then one of:
//   0 nothing   1 atomicMax on ONE word   2 load + conditional atomicMax on one word
//   3 atomicMax on word (block % 32) of one 128-byte line   4 atomicMax on 32 words 256 bytes apart   5 same, 4 KB apart
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/atomic_probe.cpp -o tools/probes/bin/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ x, unsigned* __restrict__ slot) {
  __shared__ unsigned wmax[4];
  const float4 v = x[(size_t)blockIdx.x * 256 + threadIdx.x];
  unsigned m = __float_as_uint(v.x) & 0x7fffffffu;
  const unsigned b = __float_as_uint(v.y) & 0x7fffffffu, c = __float_as_uint(v.z) & 0x7fffffffu, d = __float_as_uint(v.w) & 0x7fffffffu;
  m = m > b ? m : b; m = m > c ? m : c; m = m > d ? m : d;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o); m = m > t ? m : t; }
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned a = wmax[0] > wmax[1] ? wmax[0] : wmax[1], bb = wmax[2] > wmax[3] ? wmax[2] : wmax[3];
    const unsigned q = a > bb ? a : bb;
    if (MODE == 0) { if (q == 0x12345678u) slot[0] = q; }
    if (MODE == 1) atomicMax(slot, q);
    if (MODE == 2) { if (q > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, q); }
    if (MODE == 3) atomicMax(slot + (blockIdx.x & 31), q);
    if (MODE == 4) atomicMax(slot + (blockIdx.x & 31) * 64, q);
    if (MODE == 5) atomicMax(slot + (blockIdx.x & 31) * 1024, q);
  }
}
template <int MODE>
static float run(const float4* x, unsigned* slot, int n) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(n), dim3(256), 0, 0, x, slot);
  CK(hipMemset(slot, 0, 32 * 1024 * 4));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<MODE>, dim3(n), dim3(256), 0, 0, x, slot);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 20 * 1e3f;
}
int main() {
  const int NMAX = 32768;
  float4* x; unsigned* slot;
  CK(hipMalloc(&x, (size_t)NMAX * 256 * 16));
  CK(hipMalloc(&slot, 32 * 1024 * 4));
  // random magnitudes: (1) increasing maxima keep every atomic "useful" is the worst case; here values are random
  float* h = (float*)malloc((size_t)NMAX * 256 * 16);
  srand(1);
  for (size_t i = 0; i < (size_t)NMAX * 1024; ++i) h[i] = (float)rand() / RAND_MAX;
  CK(hipMemcpy(x, h, (size_t)NMAX * 256 * 16, hipMemcpyHostToDevice));
  for (int n : {608, 2048, 8192, 31033}) {
    printf("N=%5d workgroups: none %.1f us | one word %.1f | load+conditional %.1f | 32 words of one line %.1f | 32 words 256 B apart %.1f | 4 KB apart %.1f\n", n,
           run<0>(x, slot, n), run<1>(x, slot, n), run<2>(x, slot, n), run<3>(x, slot, n), run<4>(x, slot, n), run<5>(x, slot, n));
  }
  return 0;
}
