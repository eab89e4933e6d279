// Probe: the adopted bf16x6 GEMM loop (128x128x32 tile, 256 threads, one LDS stage, 2 workgroups per CU) with the
// two bf16 MFMA shapes.  MI355X_MICROARCH.md (DVFS give-back, item 7): on random data a 16x16x32 loop delivers 1.12-1.15x
// the FLOP/s of a 32x32x16 loop at equal cycles per FLOP, because the chip holds a higher clock.
//   SHAPE 32: v_mfma_f32_32x32x16_bf16, LDS rows of 32 k (64 B) at an 80-byte pitch           (production layout)
//   SHAPE 16: v_mfma_f32_16x16x32_bf16, one ds_read_b128 = the 32 k of a row slot; pitch 96 B (conflict free)
//             or 64 B with the 16-byte chunks XOR-swizzled by (-(row >> 2)) & 3               (no padding)
// C[M][N] = A[M][K] * B[N][K]^T, fp32 in / out, exact 3-way bf16 split, 6 partial products.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/x6_shape_probe.cpp -o tools/probes/bin/x6_shape
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
__device__ __forceinline__ unsigned pack_hi(float e0, float e1) {
  return __builtin_amdgcn_perm(__float_as_uint(e1), __float_as_uint(e0), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float a) { return __uint_as_float(__float_as_uint(a) & 0xffff0000u); }

// SHAPE 32 / PITCH 80: production.  SHAPE 16: PITCH 96 (padded) or 64 (swizzled).
template <int SHAPE, int PITCH, int PRIO>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                      float* __restrict__ C, int M, int N, int K) {
  constexpr int BM = 128, BN = 128, BK = 32, NP = 3;
  constexpr int OP_BYTES = NP * BM * PITCH;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * OP_BYTES];
  unsigned char* As = smem;
  unsigned char* Bs = smem + OP_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nt = N / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
  // staging: 8 float4 per 32-k row, 32 rows per pass; a 16-lane ds_write_b64 group holds 2 rows whose banks must differ
  const int c4 = tid & 7;
  int row;
  if (PITCH == 80) row = (wave << 3) + (((lane >> 3) & 1) << 2) + (lane >> 4);              // rows 4 apart
  else if (PITCH == 96) { const int j = lane >> 3; row = (wave << 3) + ((j & 1) << 1) + ((j >> 1) & 1) + (j & 4); }   // 2 apart
  else row = (wave << 3) + (lane >> 3);                                                        // 64: 1 apart
  float4 ra[4], rb[4];
  auto load = [&](int ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ra[i] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + row + 32 * i) * K + ks * BK + c4 * 4);
      rb[i] = *reinterpret_cast<const float4*>(B + (size_t)(n0 + row + 32 * i) * K + ks * BK + c4 * 4);
    }
  };
  auto chunk_off = [&](int rr, int c8) {        // byte offset of the 8-byte unit c8 (0..7) of row rr
    if (PITCH == 64) return rr * 64 + ((((c8 >> 1) ^ ((-(rr >> 2)) & 3))) << 4) + ((c8 & 1) << 3);
    return rr * PITCH + c8 * 8;
  };
  auto store_op = [&](unsigned char* S, const float4* rv) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = row + 32 * i;
      float4 v = rv[i];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        uint2 w;
        w.x = pack_hi(v.x, v.y);
        w.y = pack_hi(v.z, v.w);
        *reinterpret_cast<uint2*>(S + p * BM * PITCH + chunk_off(rr, c4)) = w;
        if (p + 1 < NP) { v.x -= trunc_bf16(v.x); v.y -= trunc_bf16(v.y); v.z -= trunc_bf16(v.z); v.w -= trunc_bf16(v.w); }
      }
    }
  };
  const int nk = K / BK;
  if (SHAPE == 32) {
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    load(0);
    store_op(As, ra); store_op(Bs, rb);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
      const bool more = ks + 1 < nk;
      if (more) load(ks + 1);
      if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int g = 0; g < BK / 16; ++g) {
        bf16x8 fa[2][NP], fb[2][NP];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            fa[t][p] = *reinterpret_cast<const bf16x8*>(As + (p * BM + wm * 64 + t * 32 + r) * PITCH + g * 32 + h * 16);
            fb[t][p] = *reinterpret_cast<const bf16x8*>(Bs + (p * BM + wn * 64 + t * 32 + r) * PITCH + g * 32 + h * 16);
          }
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            f32x16 c = acc[tm][tn];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][2], fb[tn][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], fb[tn][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][1], fb[tn][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][1], fb[tn][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], fb[tn][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], fb[tn][0], c, 0, 0, 0);
            acc[tm][tn] = c;
          }
      }
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      __syncthreads();
      if (more) { store_op(As, ra); store_op(Bs, rb); }
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
  } else {
    const int r = lane & 15, q = lane >> 4;      // fragment row / k slot (8 k each)
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    load(0);
    store_op(As, ra); store_op(Bs, rb);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
      const bool more = ks + 1 < nk;
      if (more) load(ks + 1);
      if (PRIO) __builtin_amdgcn_s_setprio(1);
      // two halves of the wave tile's columns: 4 A fragments x 2 B fragments at a time keeps 18 fragment registers live
#pragma unroll
      for (int hn = 0; hn < 2; ++hn) {
        bf16x8 fa[4][NP], fb[2][NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int rr = wm * 64 + t * 16 + r;
            fa[t][p] = *reinterpret_cast<const bf16x8*>(As + p * BM * PITCH + (PITCH == 64 ? rr * 64 + ((q ^ ((-(rr >> 2)) & 3)) << 4) : rr * PITCH + q * 16));
          }
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int rr = wn * 64 + (hn * 2 + t) * 16 + r;
            fb[t][p] = *reinterpret_cast<const bf16x8*>(Bs + p * BM * PITCH + (PITCH == 64 ? rr * 64 + ((q ^ ((-(rr >> 2)) & 3)) << 4) : rr * PITCH + q * 16));
          }
        }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            f32x4 c = acc[tm][hn * 2 + tn];
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tm][2], fb[tn][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tm][0], fb[tn][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tm][1], fb[tn][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tm][1], fb[tn][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tm][0], fb[tn][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tm][0], fb[tn][0], c, 0, 0, 0);
            acc[tm][hn * 2 + tn] = c;
          }
      }
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      __syncthreads();
      if (more) { store_op(As, ra); store_op(Bs, rb); }
      __syncthreads();
    }
    // D of 16x16x32: lane (r = column, q): rows 4q .. 4q+3
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int m = m0 + wm * 64 + tm * 16 + 4 * q + e;
          const int n = n0 + wn * 64 + tn * 16 + r;
          C[(size_t)m * N + n] = acc[tm][tn][e];
        }
  }
}

// Ping-pong: ONE 512-thread workgroup per CU = two 4-wave groups, each with its own tile and its own LDS stage, running
// the same K loop half a period apart: while group 0 multiplies (MFMA + fragment reads) group 1 splits and stores its
// next K step (VALU + ds_write), then the roles swap; one workgroup barrier per phase.  Same work per wave as the
// adopted kernel; what changes is that the two waves of a SIMD are ALWAYS in complementary phases (two independent
// 256-thread workgroups drift in and out of lockstep).
template <int PRIO>
__global__ __launch_bounds__(512, 2) void gemm_pingpong_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                               float* __restrict__ C, int M, int N, int K) {
  constexpr int BM = 128, BN = 128, BK = 32, NP = 3, PITCH = 96;
  constexpr int OP_BYTES = NP * BM * PITCH;
  __shared__ __attribute__((aligned(16))) unsigned char smem_all[4 * OP_BYTES];
  const int tid5 = threadIdx.x, grp = tid5 >> 8, tid = tid5 & 255, lane = tid & 63, wave = tid >> 6;
  unsigned char* As = smem_all + grp * 2 * OP_BYTES;
  unsigned char* Bs = As + OP_BYTES;
  const int wm = wave >> 1, wn = wave & 1;
  const int nt = N / BN;
  const int tile = 2 * xcd_remap(blockIdx.x, gridDim.x) + grp;
  const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
  const int c4 = tid & 7;
  const int j = lane >> 3;
  const int row = (wave << 3) + ((j & 1) << 1) + ((j >> 1) & 1) + (j & 4);
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
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        uint2 w;
        w.x = pack_hi(v.x, v.y);
        w.y = pack_hi(v.z, v.w);
        *reinterpret_cast<uint2*>(S + p * BM * PITCH + rr * PITCH + c4 * 8) = w;
        if (p + 1 < NP) { v.x -= trunc_bf16(v.x); v.y -= trunc_bf16(v.y); v.z -= trunc_bf16(v.z); v.w -= trunc_bf16(v.w); }
      }
    }
  };
  const int r = lane & 15, q = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][jj][e] = 0.f;
  auto mma = [&]() {
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int hn = 0; hn < 2; ++hn) {
      bf16x8 fa[4][NP], fb[2][NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int t = 0; t < 4; ++t) fa[t][p] = *reinterpret_cast<const bf16x8*>(As + p * BM * PITCH + (wm * 64 + t * 16 + r) * PITCH + q * 16);
#pragma unroll
        for (int t = 0; t < 2; ++t) fb[t][p] = *reinterpret_cast<const bf16x8*>(Bs + p * BM * PITCH + (wn * 64 + (hn * 2 + t) * 16 + r) * PITCH + q * 16);
      }
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          f32x4 c = acc[tm][hn * 2 + tn];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tm][2], fb[tn][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tm][0], fb[tn][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tm][1], fb[tn][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tm][1], fb[tn][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tm][0], fb[tn][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[tm][0], fb[tn][0], c, 0, 0, 0);
          acc[tm][hn * 2 + tn] = c;
        }
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
  };
  const int nk = K / BK;
  // group 0: [load k+1, mma k] [store k+1] ...; group 1 runs the same sequence one phase later
  load(0);
  if (grp == 0) { store_op(As, ra); store_op(Bs, rb); }
  __syncthreads();
  for (int ph = 0; ph < 2 * nk + 1; ++ph) {
    const int t = ph - grp;                   // this group's own phase counter
    if (t >= 0 && t < 2 * nk) {
      const int ks = t >> 1;
      if ((t & 1) == 0) {                     // multiply step ks; next step's loads in flight behind it
        if (ks + 1 < nk) load(ks + 1);
        mma();
      } else if (ks + 1 < nk) {               // split + store step ks + 1
        store_op(As, ra); store_op(Bs, rb);
      }
    } else if (t == -1) {                     // group 1's first phase: store its step 0
      store_op(As, ra); store_op(Bs, rb);
    }
    __syncthreads();
  }
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + wm * 64 + tm * 16 + 4 * q + e;
        const int n = n0 + wn * 64 + tn * 16 + r;
        C[(size_t)m * N + n] = acc[tm][tn][e];
      }
}

template <int SHAPE, int PITCH, int PRIO>
static void run(const char* name, const float* dA, const float* dB, float* dC, int M, int N, int K, const std::vector<float>& hA,
                const std::vector<float>& hB) {
  const int tiles = (M / 128) * (N / 128);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto go = [&]() {
    if (SHAPE == 99) hipLaunchKernelGGL((gemm_pingpong_kernel<PRIO>), dim3(tiles / 2), dim3(512), 0, 0, dA, dB, dC, M, N, K);
    else hipLaunchKernelGGL((gemm_kernel<SHAPE == 99 ? 16 : SHAPE, SHAPE == 99 ? 96 : PITCH, PRIO>), dim3(tiles), dim3(256), 0, 0, dA, dB, dC, M, N, K);
  };
  for (int i = 0; i < 3; ++i) go();
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) go();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const int RM = 64, RN = 128;
  std::vector<float> hC((size_t)RM * N);
  CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  double max_rel_sabs = 0, sum_sq = 0, sum_sq_ref = 0;
  for (int m = 0; m < RM; ++m)
    for (int n = 0; n < RN; ++n) {
      double s = 0, sa = 0;
      for (int k = 0; k < K; ++k) {
        const double p = (double)hA[(size_t)m * K + k] * (double)hB[(size_t)n * K + k];
        s += p; sa += fabs(p);
      }
      const double err = fabs((double)hC[(size_t)m * N + n] - s);
      max_rel_sabs = fmax(max_rel_sabs, err / sa);
      sum_sq += err * err; sum_sq_ref += s * s;
    }
  printf("%-22s M=%6d N=%5d K=%5d  %8.1f us  %7.1f TFLOP/s  rms_rel=%.3e  max err/sum|ab|=%.3e\n", name, M, N, K, ms * 1e3,
         2.0 * M * N * K / (ms * 1e-3) / 1e12, sqrt(sum_sq / sum_sq_ref), max_rel_sabs);
}

int main() {
  const int shapes[][3] = {{16384, 4096, 2048}, {8192, 2048, 1024}, {8192, 2048, 256}, {65536, 256, 320}};
  for (int rep = 0; rep < 2; ++rep)
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
    run<32, 80, 0>("32x32x16 pitch80", dA, dB, dC, M, N, K, hA, hB);
    run<16, 96, 0>("16x16x32 pitch96", dA, dB, dC, M, N, K, hA, hB);
    run<16, 96, 1>("16x16x32 pitch96 prio", dA, dB, dC, M, N, K, hA, hB);
    run<99, 96, 0>("pingpong 2x4 waves", dA, dB, dC, M, N, K, hA, hB);
    run<99, 96, 1>("pingpong 2x4 waves prio", dA, dB, dC, M, N, K, hA, hB);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  }
  return 0;
}
