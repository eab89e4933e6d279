// Probe: where a 32-k step of the f16x3 loop goes (128x128 tile, 4 waves, 16x16x32 f16 MFMA, 48 MFMAs per wave and step,
// 2 workgroups per CU), rebuilt phase by phase like x6_phases_probe.cpp did for bf16x6.  Results are garbage on purpose;
// only the rate matters (TFLOP/s fp32-equivalent = 2 * 128 * 128 * 32 FLOP per workgroup and step).
//   0 MFMAs only (fragments in registers)            1 + fragment reads from LDS (16 ds_read_b128 per wave)
//   2 + the two barriers                              3 + split of register data + 16 ds_write_b64 per thread
//   4 + the global loads (= the production loop)
// Build: hipcc -O3 --offload-arch=gfx950 [-DRANDOM_DATA=1] tools/probes/h3_phases_probe.cpp -o tools/probes/bin/h3_phases
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#ifndef RANDOM_DATA
#define RANDOM_DATA 1
#endif
__device__ __forceinline__ unsigned pack_f16(float e0, float e1) {
  f32x2 v = {e0, e1};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ f32x2 unpack_f16(unsigned w) { return __builtin_convertvector(__builtin_bit_cast(f16x2, w), f32x2); }
#define MH(a, b) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)

template <int PH>
__global__ __launch_bounds__(256, 2) void phase_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int K, int nk) {
  constexpr int BM = 128, PITCH = 96, NP = 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * NP * BM * PITCH];
  unsigned char* As = smem;
  unsigned char* Bs = smem + NP * BM * PITCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int c4 = tid & 7, j = lane >> 3;
  const int row = (wave << 3) + ((j & 1) << 1) + ((j >> 1) & 1) + (j & 4);
  const int r = lane & 15, q = lane >> 4;
#ifdef SHARED_STRIP
  const size_t base = (size_t)(blockIdx.x % SHARED_STRIP) * 128 * K;      // the workgroups share SHARED_STRIP operand strips: L2-resident
#else
  const size_t base = (size_t)blockIdx.x * 128 * K;
#endif
  float4 ra[4], rb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    ra[i] = *reinterpret_cast<const float4*>(A + base + (size_t)(row + 32 * i) * K + c4 * 4);
    rb[i] = *reinterpret_cast<const float4*>(B + base + (size_t)(row + 32 * i) * K + c4 * 4);
  }
  auto store_op = [&](unsigned char* S, const float4* rv) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = row + 32 * i;
      float4 v = rv[i];
      v.x *= 4096.f; v.y *= 4096.f; v.z *= 4096.f; v.w *= 4096.f;
      uint2 w0, w1;
      w0.x = pack_f16(v.x, v.y); w0.y = pack_f16(v.z, v.w);
      const f32x2 b0 = unpack_f16(w0.x), b1 = unpack_f16(w0.y);
      w1.x = pack_f16(v.x - b0.x, v.y - b0.y); w1.y = pack_f16(v.z - b1.x, v.w - b1.y);
      *reinterpret_cast<uint2*>(S + rr * PITCH + c4 * 8) = w0;
      *reinterpret_cast<uint2*>(S + BM * PITCH + rr * PITCH + c4 * 8) = w1;
    }
  };
  store_op(As, ra); store_op(Bs, rb);
  __syncthreads();
  uint4 fa[4][NP], fb[4][NP];
  auto read_frags = [&]() {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t][p] = *reinterpret_cast<const uint4*>(As + p * BM * PITCH + (wm * 64 + t * 16 + r) * PITCH + q * 16);
        fb[t][p] = *reinterpret_cast<const uint4*>(Bs + p * BM * PITCH + (wn * 64 + t * 16 + r) * PITCH + q * 16);
      }
  };
  read_frags();
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][jj][e] = 0.f;
  for (int ks = 0; ks < nk; ++ks) {
    if (PH >= 4) {
      const int kk = ((ks + 1) % (K / 32)) * 32;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const float4*>(A + base + (size_t)(row + 32 * i) * K + kk + c4 * 4);
        rb[i] = *reinterpret_cast<const float4*>(B + base + (size_t)(row + 32 * i) * K + kk + c4 * 4);
      }
    }
    if (PH >= 1) { asm volatile("" ::: "memory"); read_frags(); }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        f32x4 c = acc[tm][tn];
        MH(fa[tm][1], fb[tn][0]); MH(fa[tm][0], fb[tn][1]); MH(fa[tm][0], fb[tn][0]);
        acc[tm][tn] = c;
      }
    __builtin_amdgcn_s_setprio(0);
    if (PH >= 2) __syncthreads();
    if (PH >= 3) {
      if (PH == 3) {          // keep the split inputs changing so that the arithmetic cannot be hoisted
#pragma unroll
        for (int i = 0; i < 4; ++i) { ra[i].x += 1e-3f; rb[i].y -= 1e-3f; }
      }
      store_op(As, ra); store_op(Bs, rb);
    }
    if (PH >= 2) __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += acc[i][jj][e];
  C[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int PH>
static int run(const char* name, const float* dA, const float* dB, float* dC, int K, int nk, int wgs) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((phase_kernel<PH>), dim3(wgs), dim3(256), 0, 0, dA, dB, dC, K, nk);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((phase_kernel<PH>), dim3(wgs), dim3(256), 0, 0, dA, dB, dC, K, nk);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  printf("%-60s %8.1f us  %7.1f TFLOP/s (fp32-equivalent; x3 = %6.0f executed fp16)\n", name, ms * 1e3, 2.0 * 128 * 128 * 32 * nk * wgs / (ms * 1e-3) / 1e12,
         3 * 2.0 * 128 * 128 * 32 * nk * wgs / (ms * 1e-3) / 1e12);
  return 0;
}
int main() {
  const int wgs = 1024, K = 2048, nk = 512;
  std::vector<float> h((size_t)wgs * 128 * K);
  srand(1);
  for (auto& v : h) v = RANDOM_DATA ? ((float)rand() / (float)RAND_MAX * 2.f - 1.f) : 0.5f;
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, h.size() * 4)); CK(hipMalloc(&dB, h.size() * 4)); CK(hipMalloc(&dC, (size_t)wgs * 256 * 4));
  CK(hipMemcpy(dA, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  for (auto& v : h) v = RANDOM_DATA ? ((float)rand() / (float)RAND_MAX * 2.f - 1.f) : 0.25f;
  CK(hipMemcpy(dB, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  printf("f16x3 K step, %s operands, %d workgroups x %d steps\n", RANDOM_DATA ? "random" : "constant", wgs, nk);
  if (run<0>("0 MFMAs only (3-chain per accumulator)", dA, dB, dC, K, nk, wgs)) return 1;
  if (run<1>("1 + fragment reads (16 ds_read_b128 per wave and step)", dA, dB, dC, K, nk, wgs)) return 1;
  if (run<2>("2 + two barriers per step", dA, dB, dC, K, nk, wgs)) return 1;
  if (run<3>("3 + split + 16 ds_write_b64 per thread", dA, dB, dC, K, nk, wgs)) return 1;
  if (run<4>("4 + global loads (the production loop)", dA, dB, dC, K, nk, wgs)) return 1;
  return 0;
}
