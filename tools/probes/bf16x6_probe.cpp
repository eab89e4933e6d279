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
#include <cstring>
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

__device__ const unsigned short* g_Bsplit;
__device__ const unsigned short* g_Asplit;      // VAR 4: B pre-split into 3 bf16 planes [3][N][K]
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
  uint2 rbs[4][3], ras[4][3];
  const unsigned short* Bsp = g_Bsplit;
  const unsigned short* Asp = g_Asplit;
  auto load = [&](int ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (VAR == 5) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
          ras[i][p] = *reinterpret_cast<const uint2*>(Asp + ((size_t)p * M + m0 + row + 32 * i) * K + ks * BK + c4 * 4);
      } else {
        ra[i] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + row + 32 * i) * K + ks * BK + c4 * 4);
      }
      if (VAR >= 4) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
          rbs[i][p] = *reinterpret_cast<const uint2*>(Bsp + ((size_t)p * N + n0 + row + 32 * i) * K + ks * BK + c4 * 4);
      } else {
        rb[i] = *reinterpret_cast<const float4*>(B + (size_t)(n0 + row + 32 * i) * K + ks * BK + c4 * 4);
      }
    }
  };
  auto store_presplit_a = [&](unsigned char* S) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(S + (p * BM + row + 32 * i) * ROWB + c4 * 8) = ras[i][p];
  };
  auto store_presplit = [&](unsigned char* S) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(S + (p * BM + row + 32 * i) * ROWB + c4 * 8) = rbs[i][p];
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
          if (p + 1 < NP && VAR != 6) {
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
  if (VAR == 5) store_presplit_a(As); else store_op(As, ra);
  if (VAR >= 4) store_presplit(Bs); else store_op(Bs, rb);
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const bool more = ks + 1 < nk;
    if (more && VAR != 7) load(ks + 1);
    if (VAR == 7) { for (int i = 0; i < 4; ++i) { asm volatile("" : "+v"(ra[i].x), "+v"(rb[i].x)); } }
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
    if (MODE != 0 && (VAR == 1 || VAR == 2) && more) {
      if (VAR == 1) split_op(pa, ra);
      split_op(pb, rb);
    }
    __syncthreads();
    if (more) {
      if (MODE != 0 && (VAR == 1 || VAR == 2)) { write_op(As, pa); write_op(Bs, pb); }
      else {
        if (VAR == 5) store_presplit_a(As); else store_op(As, ra);
        if (VAR >= 4) store_presplit(Bs); else store_op(Bs, rb);
      }
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


// Wave-specialised variant: 512 threads = 4 consumer waves (MFMA only) + 4 producer waves (global -> split -> LDS),
// two LDS stages, one barrier per K step, one workgroup per CU.
__global__ __launch_bounds__(512, 1) void gemm_ws_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                         float* __restrict__ C, int M, int N, int K) {
  constexpr int BM = 128, BN = 128, BK = 32, NP = 3, ROWB = 80;
  constexpr int OP_BYTES = NP * BM * ROWB, STAGE = 2 * OP_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  const int cw = wave & 3, wm = cw >> 1, wn = cw & 1, r = lane & 31, h = lane >> 5;
  const int nt = N / BN;
  const int ptid = tid & 255;
  const int c4 = ptid & 7, row = (((ptid >> 6) << 3) + (((lane >> 3) & 1) << 2) + (lane >> 4));
  const int nk = K / BK;
  const int tiles = (M / BM) * nt;
  f32x16 acc[2][2];
  float4 ra[4], rb[4];
  for (int tile = xcd_remap(blockIdx.x, gridDim.x); tile < tiles; tile += gridDim.x) {
    const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
    auto load = [&](int ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + row + 32 * i) * K + ks * BK + c4 * 4);
        rb[i] = *reinterpret_cast<const float4*>(B + (size_t)(n0 + row + 32 * i) * K + ks * BK + c4 * 4);
      }
    };
    auto store = [&](int buf) {
      unsigned char* As = smem + buf * STAGE;
      unsigned char* Bs = As + OP_BYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float4 v = ra[i], w = rb[i];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          uint2 a2, b2;
          a2.x = pack_hi(v.x, v.y); a2.y = pack_hi(v.z, v.w);
          b2.x = pack_hi(w.x, w.y); b2.y = pack_hi(w.z, w.w);
          *reinterpret_cast<uint2*>(As + (p * BM + row + 32 * i) * ROWB + c4 * 8) = a2;
          *reinterpret_cast<uint2*>(Bs + (p * BM + row + 32 * i) * ROWB + c4 * 8) = b2;
          if (p + 1 < NP) {
            v.x -= trunc_bf16(v.x); v.y -= trunc_bf16(v.y); v.z -= trunc_bf16(v.z); v.w -= trunc_bf16(v.w);
            w.x -= trunc_bf16(w.x); w.y -= trunc_bf16(w.y); w.z -= trunc_bf16(w.z); w.w -= trunc_bf16(w.w);
          }
        }
      }
    };
    if (!producer) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
    if (producer) {
      load(0);
      store(0);
      if (nk > 1) load(1);
    }
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
      const int buf = ks & 1;
      if (producer) {
        if (ks + 1 < nk) {
          store(buf ^ 1);
          if (ks + 2 < nk) load(ks + 2);
        }
      } else {
        const unsigned char* As = smem + buf * STAGE;
        const unsigned char* Bs = As + OP_BYTES;
#pragma unroll
        for (int g = 0; g < BK / 16; ++g) {
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
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][2], fb[tn][0], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], fb[tn][2], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][1], fb[tn][1], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][1], fb[tn][0], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], fb[tn][1], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], fb[tn][0], c, 0, 0, 0);
              acc[tm][tn] = c;
            }
        }
      }
      __syncthreads();
    }
    if (!producer) {
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
  }
}

// In-wave pipelined variant: 512 threads = 8 waves, every wave does BOTH jobs -- MFMAs on LDS stage `buf` interleaved with
// the split + LDS store of the next K step into stage `buf ^ 1` -- one barrier per K step, one workgroup per CU.
//   TILE_M = 128: waves 2(M) x 4(N), 64 x 32 each, BK = 32;   TILE_M = 256: waves 4(M) x 2(N), 64 x 64 each, BK = 16.
template <int TILE_M>
__global__ __launch_bounds__(512, 1) void gemm_pipe_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                           float* __restrict__ C, int M, int N, int K) {
  constexpr int BM = TILE_M, BN = 128, BK = TILE_M == 128 ? 32 : 16, NP = 3, ROWB = BK * 2 + 16;
  constexpr int A_BYTES = NP * BM * ROWB, B_BYTES = NP * BN * ROWB, STAGE = A_BYTES + B_BYTES;
  constexpr int TN = TILE_M == 128 ? 1 : 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = TILE_M == 128 ? wave >> 2 : wave >> 1, wn = TILE_M == 128 ? wave & 3 : wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int nt = N / BN, tiles = (M / BM) * nt, nk = K / BK;
  // staging: rows of BK floats = BK/4 float4 per row
  constexpr int F4R = BK / 4, ROWS_PP = 512 / F4R;          // float4 per row, rows per pass
  constexpr int APASS = BM / ROWS_PP, BPASS = BN / ROWS_PP;
  const int c4 = tid % F4R, row = tid / F4R;
  f32x16 acc[2][TN];
  float4 ra[APASS], rb[BPASS];
  for (int tile = xcd_remap(blockIdx.x, gridDim.x); tile < tiles; tile += gridDim.x) {
    const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
    auto load = [&](int ks) {
#pragma unroll
      for (int i = 0; i < APASS; ++i) ra[i] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + row + ROWS_PP * i) * K + ks * BK + c4 * 4);
#pragma unroll
      for (int i = 0; i < BPASS; ++i) rb[i] = *reinterpret_cast<const float4*>(B + (size_t)(n0 + row + ROWS_PP * i) * K + ks * BK + c4 * 4);
    };
    auto put = [&](unsigned char* S, int rows, int rr, float4 v) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        uint2 w;
        w.x = pack_hi(v.x, v.y); w.y = pack_hi(v.z, v.w);
        *reinterpret_cast<uint2*>(S + (p * rows + rr) * ROWB + c4 * 8) = w;
        if (p + 1 < NP) { v.x -= trunc_bf16(v.x); v.y -= trunc_bf16(v.y); v.z -= trunc_bf16(v.z); v.w -= trunc_bf16(v.w); }
      }
    };
    auto store_a = [&](int buf) {
#pragma unroll
      for (int i = 0; i < APASS; ++i) put(smem + buf * STAGE, BM, row + ROWS_PP * i, ra[i]);
    };
    auto store_b = [&](int buf) {
#pragma unroll
      for (int i = 0; i < BPASS; ++i) put(smem + buf * STAGE + A_BYTES, BN, row + ROWS_PP * i, rb[i]);
    };
    auto mma = [&](int buf, int g) {
      const unsigned char* As = smem + buf * STAGE;
      const unsigned char* Bs = As + A_BYTES;
      bf16x8 fa[2][NP], fb[TN][NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int t = 0; t < 2; ++t) fa[t][p] = *reinterpret_cast<const bf16x8*>(As + (p * BM + wm * 64 + t * 32 + r) * ROWB + g * 32 + h * 16);
#pragma unroll
        for (int t = 0; t < TN; ++t) fb[t][p] = *reinterpret_cast<const bf16x8*>(Bs + (p * BN + wn * 32 * TN + t * 32 + r) * ROWB + g * 32 + h * 16);
      }
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          f32x16 c = acc[tm][tn];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][2], fb[tn][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], fb[tn][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][1], fb[tn][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][1], fb[tn][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], fb[tn][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], fb[tn][0], c, 0, 0, 0);
          acc[tm][tn] = c;
        }
    };
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    load(0);
    store_a(0); store_b(0);
    if (nk > 1) load(1);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
      const int buf = ks & 1;
      const bool more = ks + 1 < nk;
      if (BK == 32) {
        mma(buf, 0);
        if (more) store_a(buf ^ 1);
        mma(buf, 1);
        if (more) store_b(buf ^ 1);
      } else {
        mma(buf, 0);
        if (more) { store_a(buf ^ 1); store_b(buf ^ 1); }
      }
      if (ks + 2 < nk) load(ks + 2);
      __syncthreads();
    }
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm * 64 + tm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const int n = n0 + wn * 32 * TN + tn * 32 + r;
          C[(size_t)m * N + n] = acc[tm][tn][e];
        }
    __syncthreads();
  }
}

// 2-deep register prefetch on the adopted structure (256 threads, one LDS stage, 2 workgroups per CU): the loads of K step
// ks + 2 are issued before the MFMAs of step ks, so a load has two MFMA phases to land.
__global__ __launch_bounds__(256, 2) void gemm_pf2_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          float* __restrict__ C, int M, int N, int K) {
  constexpr int BM = 128, BN = 128, BK = 32, NP = 3, ROWB = 80, OP_BYTES = NP * BM * ROWB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * OP_BYTES];
  unsigned char* As = smem;
  unsigned char* Bs = smem + OP_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int nt = N / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
  const int c4 = tid & 7, row = (wave << 3) + (((lane >> 3) & 1) << 2) + (lane >> 4);
  float4 ra0[4], rb0[4], ra1[4], rb1[4];
  auto load = [&](float4* qa, float4* qb, int ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      qa[i] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + row + 32 * i) * K + ks * BK + c4 * 4);
      qb[i] = *reinterpret_cast<const float4*>(B + (size_t)(n0 + row + 32 * i) * K + ks * BK + c4 * 4);
    }
  };
  auto put = [&](unsigned char* S, const float4* rv) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = rv[i];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        uint2 w;
        w.x = pack_hi(v.x, v.y); w.y = pack_hi(v.z, v.w);
        *reinterpret_cast<uint2*>(S + (p * BM + row + 32 * i) * ROWB + c4 * 8) = w;
        if (p + 1 < NP) { v.x -= trunc_bf16(v.x); v.y -= trunc_bf16(v.y); v.z -= trunc_bf16(v.z); v.w -= trunc_bf16(v.w); }
      }
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  auto mma = [&]() {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
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
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][2], fb[tn][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], fb[tn][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][1], fb[tn][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][1], fb[tn][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], fb[tn][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], fb[tn][0], c, 0, 0, 0);
          acc[tm][tn] = c;
        }
    }
  };
  const int nk = K / BK;                      // even
  load(ra0, rb0, 0);
  put(As, ra0); put(Bs, rb0);
  load(ra0, rb0, 1);
  __syncthreads();
  for (int ks = 0; ks < nk; ks += 2) {
    if (ks + 2 < nk) load(ra1, rb1, ks + 2);
    mma();
    __syncthreads();
    put(As, ra0); put(Bs, rb0);               // step ks + 1
    __syncthreads();
    if (ks + 3 < nk) load(ra0, rb0, ks + 3);
    mma();
    __syncthreads();
    if (ks + 2 < nk) { put(As, ra1); put(Bs, rb1); }
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
  auto go = [&]() {
    if (VAR == 9) hipLaunchKernelGGL(gemm_ws_kernel, dim3(tiles < 256 ? tiles : 256), dim3(512), 0, 0, dA, dB, dC, M, N, K);
    else if (VAR == 12) hipLaunchKernelGGL(gemm_pf2_kernel, dim3(tiles), dim3(256), 0, 0, dA, dB, dC, M, N, K);
    else if (VAR == 10) hipLaunchKernelGGL((gemm_pipe_kernel<128>), dim3(tiles < 256 ? tiles : 256), dim3(512), 0, 0, dA, dB, dC, M, N, K);
    else if (VAR == 11) hipLaunchKernelGGL((gemm_pipe_kernel<256>), dim3(tiles / 2 < 256 ? tiles / 2 : 256), dim3(512), 0, 0, dA, dB, dC, M, N, K);
    else hipLaunchKernelGGL((gemm_kernel<MODE, VAR >= 9 ? 0 : VAR>), dim3(tiles), dim3(256), 0, 0, dA, dB, dC, M, N, K);
  };
  for (int i = 0; i < 2; ++i) go();
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) go();
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
    run<6, 9>("bf16 x6 ws", dA, dB, dC, M, N, K, hA, hB);
    run<6, 12>("x6 prefetch-2", dA, dB, dC, M, N, K, hA, hB);
    run<6, 10>("x6 pipe 128x128", dA, dB, dC, M, N, K, hA, hB);
    run<6, 11>("x6 pipe 256x128", dA, dB, dC, M, N, K, hA, hB);
    {
      auto mk = [&](const std::vector<float>& h, size_t n) {
        std::vector<unsigned short> sp(3 * n);
        for (size_t i = 0; i < n; ++i) {
          float v = h[i];
          for (int p = 0; p < 3; ++p) {
            unsigned u; memcpy(&u, &v, 4); u &= 0xffff0000u;
            sp[(size_t)p * n + i] = (unsigned short)(u >> 16);
            float t; memcpy(&t, &u, 4); v -= t;
          }
        }
        unsigned short* d;
        CK(hipMalloc(&d, sp.size() * 2));
        CK(hipMemcpy(d, sp.data(), sp.size() * 2, hipMemcpyHostToDevice));
        return d;
      };
      unsigned short* dS = mk(hB, (size_t)N * K);
      unsigned short* dT = mk(hA, (size_t)M * K);
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_Bsplit), &dS, sizeof(dS)));
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_Asplit), &dT, sizeof(dT)));
      run<6, 6>("x6 no-resid  ", dA, dB, dC, M, N, K, hA, hB);
      run<6, 7>("x6 no-gloads ", dA, dB, dC, M, N, K, hA, hB);
      run<6, 4>("x6 B-presplit", dA, dB, dC, M, N, K, hA, hB);
      run<6, 5>("x6 AB-presplt", dA, dB, dC, M, N, K, hA, hB);
      CK(hipFree(dS)); CK(hipFree(dT));
    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  }
  return 0;
}
