// Probe: fp32 GEMM on the bf16 matrix cores by operand splitting (a = hi + mid + lo, three bf16 pieces,
// exact for a 24-bit significand) against the fp32 MFMA, same tile structure.  C[M][N] = A[M][K] * B[N][K]^T.
//   MODE 0: v_mfma_f32_32x32x2_f32          (exact fp32 products, fp32 accumulate)
//   MODE 3: hi*hi + hi*mid + mid*hi          (error ~2^-16 per product)
//   MODE 6: + mid*mid + hi*lo + lo*hi        (dropped terms <= 2^-24 relative: fp32-level)
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/bf16x6_probe.cpp -o /tmp/bf16x6_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
__device__ __forceinline__ unsigned pack_hi(float e0, float e1) {      // (bf16 trunc(e1) << 16) | bf16 trunc(e0)
  return __builtin_amdgcn_perm(__float_as_uint(e1), __float_as_uint(e0), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float a) { return __uint_as_float(__float_as_uint(a) & 0xffff0000u); }

template <int MODE, int VAR>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                      float* __restrict__ C, int M, int N, int K) {
  constexpr int BM = 128, BN = 128, BK = 32;
  constexpr int NP = MODE == 6 ? 3 : (MODE == 3 ? 2 : 1);
  constexpr int ROWB = MODE ? 80 : 144;              // bytes per LDS row of one piece (32 k + 16 B pad)
  constexpr int OP_BYTES = NP * BM * ROWB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * OP_BYTES];
  unsigned char* As = smem;
  unsigned char* Bs = smem + OP_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int nt = N / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
  const int c4 = tid & 7, row = tid >> 3;            // 8 float4 per 32-k row, 32 rows per pass

  float4 ra[4], rb[4];
  auto load = [&](int ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + row + 32 * i) * K + ks * BK + c4 * 4);
      rb[i] = *reinterpret_cast<const float4*>(B + (size_t)(n0 + row + 32 * i) * K + ks * BK + c4 * 4);
    }
  };
  auto store_op = [&](unsigned char* S, const float4* rv) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = row + 32 * i;
      float4 v = rv[i];
      if (MODE == 0) {
        *reinterpret_cast<float4*>(S + rr * ROWB + c4 * 16) = v;
      } else {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          uint2 w;
          w.x = pack_hi(v.x, v.y);
          w.y = pack_hi(v.z, v.w);
          *reinterpret_cast<uint2*>(S + (p * BM + rr) * ROWB + c4 * 8) = w;
          if (p + 1 < NP) {
            v.x -= trunc_bf16(v.x); v.y -= trunc_bf16(v.y); v.z -= trunc_bf16(v.z); v.w -= trunc_bf16(v.w);
          }
        }
      }
    }
  };

  uint2 pa[4][3], pb[4][3];
  auto split_op = [&](uint2 (*pp)[3], const float4* rv) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = rv[i];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        pp[i][p].x = pack_hi(v.x, v.y);
        pp[i][p].y = pack_hi(v.z, v.w);
        if (p + 1 < NP) { v.x -= trunc_bf16(v.x); v.y -= trunc_bf16(v.y); v.z -= trunc_bf16(v.z); v.w -= trunc_bf16(v.w); }
      }
    }
  };
  auto write_op = [&](unsigned char* S, const uint2 (*pp)[3]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(S + (p * BM + row + 32 * i) * ROWB + c4 * 8) = pp[i][p];
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = K / BK;
  load(0);
  store_op(As, ra);
  store_op(Bs, rb);
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const bool more = ks + 1 < nk;
    if (more) load(ks + 1);
    if (MODE == 0) {
#pragma unroll
      for (int kk = 0; kk < BK / 8; ++kk) {
        float4 a4[2], b4[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          a4[t] = *reinterpret_cast<const float4*>(As + (wm * 64 + t * 32 + r) * ROWB + (kk * 8 + h * 4) * 4);
          b4[t] = *reinterpret_cast<const float4*>(Bs + (wn * 64 + t * 32 + r) * ROWB + (kk * 8 + h * 4) * 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
              const float av = j == 0 ? a4[tm].x : j == 1 ? a4[tm].y : j == 2 ? a4[tm].z : a4[tm].w;
              const float bv = j == 0 ? b4[tn].x : j == 1 ? b4[tn].y : j == 2 ? b4[tn].z : b4[tn].w;
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tm][tn], 0, 0, 0);
            }
      }
    } else {
#pragma unroll
      for (int g = 0; g < BK / 16; ++g) {
        if (VAR == 2 && g == 1 && more) split_op(pa, ra);
        bf16x8 fa[2][NP], fb[2][NP];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            fa[t][p] = *reinterpret_cast<const bf16x8*>(As + (p * BM + wm * 64 + t * 32 + r) * ROWB + g * 32 + h * 16);
            fb[t][p] = *reinterpret_cast<const bf16x8*>(Bs + (p * BM + wn * 64 + t * 32 + r) * ROWB + g * 32 + h * 16);
          }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            f32x16 c = acc[tm][tn];
            if (MODE == 6) {                           // smallest terms first
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][NP - 1], fb[tn][0], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], fb[tn][NP - 1], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][1], fb[tn][1], c, 0, 0, 0);
            }
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][1], fb[tn][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], fb[tn][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], fb[tn][0], c, 0, 0, 0);
            acc[tm][tn] = c;
          }
      }
    }
    if (MODE != 0 && VAR >= 1 && more) {
      if (VAR == 1) split_op(pa, ra);
      split_op(pb, rb);
    }
    __syncthreads();
    if (more) {
      if (MODE != 0 && VAR >= 1) { write_op(As, pa); write_op(Bs, pb); }
      else { store_op(As, ra); store_op(Bs, rb); }
    }
    __syncthreads();
  }
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        const int n = n0 + wn * 64 + tn * 32 + r;
        C[(size_t)m * N + n] = acc[tm][tn][e];
      }
}

template <int MODE, int VAR>
static void run(const char* name, const float* dA, const float* dB, float* dC, int M, int N, int K, const std::vector<float>& hA,
                const std::vector<float>& hB) {
  const int tiles = (M / 128) * (N / 128);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((gemm_kernel<MODE, VAR>), dim3(tiles), dim3(256), 0, 0, dA, dB, dC, M, N, K);
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_kernel<MODE, VAR>), dim3(tiles), dim3(256), 0, 0, dA, dB, dC, M, N, K);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  // numerics on a 64 x 128 block against fp64
  const int RM = 64, RN = 128;
  std::vector<float> hC((size_t)RM * N);
  CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  double max_err = 0, max_ref = 0, max_rel_sabs = 0, sum_sq = 0, sum_sq_ref = 0;
  for (int m = 0; m < RM; ++m)
    for (int n = 0; n < RN; ++n) {
      double s = 0, sa = 0;
      for (int k = 0; k < K; ++k) {
        const double p = (double)hA[(size_t)m * K + k] * (double)hB[(size_t)n * K + k];
        s += p; sa += fabs(p);
      }
      const double err = fabs((double)hC[(size_t)m * N + n] - s);
      max_err = fmax(max_err, err); max_ref = fmax(max_ref, fabs(s));
      max_rel_sabs = fmax(max_rel_sabs, err / sa);
      sum_sq += err * err; sum_sq_ref += s * s;
    }
  printf("%-10s M=%6d N=%5d K=%5d  %8.1f us  %7.1f TFLOP/s(fp32-equivalent)  max|err|=%.3e  rms_rel=%.3e  max err/sum|ab|=%.3e\n",
         name, M, N, K, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12, max_err, sqrt(sum_sq / sum_sq_ref), max_rel_sabs);
}

int main() {
  const int shapes[][3] = {{8192, 2048, 1024}, {16384, 4096, 2048}, {8192, 2048, 256}, {65536, 256, 320}};
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    srand(1);
    for (auto& v : hA) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : hB) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, hA.size() * 4));
    CK(hipMalloc(&dB, hB.size() * 4));
    CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    run<0, 0>("fp32 mfma", dA, dB, dC, M, N, K, hA, hB);
    run<3, 0>("bf16 x3", dA, dB, dC, M, N, K, hA, hB);
    run<6, 0>("bf16 x6 v0", dA, dB, dC, M, N, K, hA, hB);
    run<6, 1>("bf16 x6 v1", dA, dB, dC, M, N, K, hA, hB);
    run<6, 2>("bf16 x6 v2", dA, dB, dC, M, N, K, hA, hB);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  }
  return 0;
}
