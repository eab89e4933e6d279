// Probe (round 4): the f16x3 GEMM loop with its operands staged by LDS-DMA (global_load_lds_dwordx4: raw fp32
// straight into LDS, no VGPR round trip, no ds_write, one barrier per K step) and the 2-way fp16 split done per wave AFTER the
// fragment reads -- against the adopted structure (global -> VGPR -> split -> ds_write_b64 -> barrier -> ds_read_b128 fp16
// fragments -> MFMA -> barrier; conv_kernels.hip conv_xs_body / tools/probes/f16x3_probe.cpp gemm_kernel<2>).
// Round 1 rejected LDS-DMA staging on the fp32 MFMA; in the f16x3 regime the load -> split -> ds_write -> barrier chain is
// the measured bound (profiles/r03_h3_phases_probe.txt: MFMA only 526 -> + split / LDS writes 426 -> + loads from L2 331).
//
//   C[M][N] = A[M][K] * B[N][K]^T, fp32 in / out, f16x3 products (h0a*h0b + h0a*h1b + h1a*h0b on v_mfma_f32_16x16x32_f16).
//   Tile 128 x 128 x 32, 256 threads = 2 x 2 waves of 64 x 64.  LDS stage = A 128 rows x 128 B + B 128 rows x 128 B = 32 KB of
//   raw fp32.  A wave's DMA instruction moves 1 KB = 8 rows x 128 B (lane l -> row l / 8, 16-byte granule l % 8); the LDS
//   image is lane-linear, so the bank swizzle is applied to the per-lane SOURCE address: LDS granule g of row r holds k granule
//   g ^ ((r >> 1) & 7) -- the 16 rows of a fragment read (ds_read_b128, lane -> row l % 16) then cover 16 distinct 16-byte
//   slots of the 256-byte bank row.  A lane's 8 k values of a 16x16x32 fragment = 2 granules = 2 ds_read_b128.
//   STAGES 2: 64 KB, two workgroups per CU, vmcnt(0) before the barrier.  STAGES 3 / 4: 96 / 128 KB, one workgroup per CU,
//   counted vmcnt leaves STAGES - 2 K steps of DMA in flight across the raw s_barrier.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/ldsdma_probe.cpp -o tools/probes/bin/ldsdma
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}
__device__ __forceinline__ unsigned pack_f16(float e0, float e1) {
  f32x2 v = {e0, e1};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ f32x2 unpack_f16(unsigned w) { return __builtin_convertvector(__builtin_bit_cast(f16x2, w), f32x2); }
// 8 consecutive-k fp32 values (two float4) -> the two fp16 pieces as MFMA operands
__device__ __forceinline__ void split8(const float4& lo, const float4& hi, float s, uint4& h0, uint4& h1) {
  const float v[8] = {lo.x * s, lo.y * s, lo.z * s, lo.w * s, hi.x * s, hi.y * s, hi.z * s, hi.w * s};
  unsigned a[4], b[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    a[e] = pack_f16(v[2 * e], v[2 * e + 1]);
    const f32x2 u = unpack_f16(a[e]);
    b[e] = pack_f16(v[2 * e] - u.x, v[2 * e + 1] - u.y);
  }
  h0 = make_uint4(a[0], a[1], a[2], a[3]);
  h1 = make_uint4(b[0], b[1], b[2], b[3]);
}
#define MH(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)

// One DMA as inline assembly: the compiler must not know that it writes LDS -- for the builtin it inserts s_waitcnt vmcnt(0)
// in front of the next ds_read of ANY stage (it cannot prove that the stage being filled is not the one being read), which
// turns the prefetch into a synchronous load.  M0 = LDS byte address of the wave's 1 KB destination (lane l lands at + 16 l).
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) const char*)p);
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- LDS-DMA staging ------------------------------------------------------------------------------------------------------
template <int STAGES, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_dma_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                            float* __restrict__ C, int M, int N, int K, float sa, float sb) {
  constexpr int BM = 128, BN = 128, BK = 32, ROWB = BK * 4;          // 128-byte rows
  constexpr int OP = BM * ROWB, STAGE = 2 * OP;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[STAGES * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nt = N / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
  // DMA roles: instruction j (0..3) of this wave covers rows 32 * j + 8 * wave .. + 7 of each operand; lane -> (row, granule)
  const int drow = lane >> 3, dg = lane & 7;
  const float* asrc[4];
  const float* bsrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 32 * j + 8 * wave + drow;
    const int gk = dg ^ ((row >> 1) & 7);                               // the k granule this LDS slot holds
    asrc[j] = A + (size_t)(m0 + row) * K + gk * 4;
    bsrc[j] = B + (size_t)(n0 + row) * K + gk * 4;
  }
  const unsigned lds0 = lds_addr(smem);
  auto issue = [&](int ks, int buf) {
    const unsigned sA = lds0 + buf * STAGE, sB = sA + OP;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned chunk = (unsigned)((32 * j + 8 * wave) * ROWB);     // wave-uniform 1 KB destination
      glds16(asrc[j] + ks * BK, __builtin_amdgcn_readfirstlane(sA + chunk));
      glds16(bsrc[j] + ks * BK, __builtin_amdgcn_readfirstlane(sB + chunk));
    }
  };
  const int nk = K / BK;
  const int r = lane & 15, q = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][jj][e] = 0.f;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) issue(s, s);
  // fragment read offsets: row R, granules (2q) ^ f(R) and (2q + 1) ^ f(R)
  int aoff[4][2], boff[4][2];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ra_ = wm * 64 + t * 16 + r, rb_ = wn * 64 + t * 16 + r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      aoff[t][h] = ra_ * ROWB + (((2 * q + h) ^ ((ra_ >> 1) & 7)) << 4);
      boff[t][h] = rb_ * ROWB + (((2 * q + h) ^ ((rb_ >> 1) & 7)) << 4);
    }
  }
  for (int ks = 0; ks < nk; ++ks) {
    // this wave's DMAs of stage ks have landed (the STAGES - 2 younger stages may still be in flight) ...
    if (ks + STAGES - 2 < nk) wait_vm<8 * (STAGES - 2)>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();            // ... and so have everyone else's; every wave is done reading stage ks - 1
    asm volatile("" ::: "memory");
    if (ks + STAGES - 1 < nk) issue(ks + STAGES - 1, (ks + STAGES - 1) % STAGES);
    const unsigned char* sA = smem + (ks % STAGES) * STAGE;
    const unsigned char* sB = sA + OP;
    uint4 a0[4], a1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float4 lo = *reinterpret_cast<const float4*>(sA + aoff[t][0]);
      const float4 hi = *reinterpret_cast<const float4*>(sA + aoff[t][1]);
      split8(lo, hi, sa, a0[t], a1[t]);
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
      const float4 lo = *reinterpret_cast<const float4*>(sB + boff[tn][0]);
      const float4 hi = *reinterpret_cast<const float4*>(sB + boff[tn][1]);
      uint4 b0, b1;
      split8(lo, hi, sb, b0, b1);
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) {
        f32x4 c = acc[tm][tn];
        MH(a1[tm], b0, c); MH(a0[tm], b1, c); MH(a0[tm], b0, c);
        acc[tm][tn] = c;
      }
    }
    __builtin_amdgcn_s_setprio(0);
  }
  const float inv = 1.f / (sa * sb);
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + wm * 64 + tm * 16 + 4 * q + e;
        const int n = n0 + wn * 64 + tn * 16 + r;
        C[(size_t)m * N + n] = acc[tm][tn][e] * inv;
      }
}

// ---- the adopted structure (tools/probes/f16x3_probe.cpp gemm_kernel<2>): register staging, split before the LDS write ----
__global__ __launch_bounds__(256, 2) void gemm_reg_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                          float* __restrict__ C, int M, int N, int K, float sa, float sb) {
  constexpr int BM = 128, BN = 128, BK = 32, PITCH = 96, NP = 2;
  constexpr int OP_BYTES = NP * BM * PITCH;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * OP_BYTES];
  unsigned char* As = smem;
  unsigned char* Bs = smem + OP_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nt = N / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
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
  auto store_op = [&](unsigned char* S, const float4* rv, float s) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = row + 32 * i;
      float4 v = rv[i];
      v.x *= s; v.y *= s; v.z *= s; v.w *= s;
      uint2 w0, w1;
      w0.x = pack_f16(v.x, v.y); w0.y = pack_f16(v.z, v.w);
      const f32x2 b0 = unpack_f16(w0.x), b1 = unpack_f16(w0.y);
      w1.x = pack_f16(v.x - b0.x, v.y - b0.y); w1.y = pack_f16(v.z - b1.x, v.w - b1.y);
      *reinterpret_cast<uint2*>(S + rr * PITCH + c4 * 8) = w0;
      *reinterpret_cast<uint2*>(S + BM * PITCH + rr * PITCH + c4 * 8) = w1;
    }
  };
  const int nk = K / BK;
  const int r = lane & 15, q = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][jj][e] = 0.f;
  load(0);
  store_op(As, ra, sa); store_op(Bs, rb, sb);
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const bool more = ks + 1 < nk;
    if (more) load(ks + 1);
    __builtin_amdgcn_s_setprio(1);
    uint4 fa[4][NP], fb[4][NP];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t][p] = *reinterpret_cast<const uint4*>(As + p * BM * PITCH + (wm * 64 + t * 16 + r) * PITCH + q * 16);
        fb[t][p] = *reinterpret_cast<const uint4*>(Bs + p * BM * PITCH + (wn * 64 + t * 16 + r) * PITCH + q * 16);
      }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        f32x4 c = acc[tm][tn];
        MH(fa[tm][1], fb[tn][0], c); MH(fa[tm][0], fb[tn][1], c); MH(fa[tm][0], fb[tn][0], c);
        acc[tm][tn] = c;
      }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    if (more) { store_op(As, ra, sa); store_op(Bs, rb, sb); }
    __syncthreads();
  }
  const float inv = 1.f / (sa * sb);
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + wm * 64 + tm * 16 + 4 * q + e;
        const int n = n0 + wn * 64 + tn * 16 + r;
        C[(size_t)m * N + n] = acc[tm][tn][e] * inv;
      }
}


// ---- PRE-SPLIT operands: the two fp16 pieces already lie in global memory ([piece][row][K] halves), LDS-DMA staging, no VALU
// work in the K loop at all.  LDS stage = per operand 2 pieces x 128 rows x 64 B = 16 KB.  A DMA instruction moves 16 rows x
// 64 B of one piece (lane l -> row l / 4, LDS granule l % 4); LDS granule g' of row r holds k granule g' ^ ((r >> 2) & 3): the
// 16 rows of a fragment read (same k granule) then cover 16 distinct 16-byte slots of a 256-byte bank row.
__device__ __forceinline__ void glds16h(const _Float16* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int STAGES, int OCC>
__global__ __launch_bounds__(256, OCC) void gemm_pre_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ B,
                                                            float* __restrict__ C, int M, int N, int K, float inv) {
  constexpr int BM = 128, BN = 128, BK = 32, ROWB = BK * 2;          // 64-byte rows per piece
  constexpr int PIECE = BM * ROWB, OP = 2 * PIECE, STAGE = 2 * OP;    // 8 KB, 16 KB, 32 KB
  __shared__ __attribute__((aligned(1024))) unsigned char smem[STAGES * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nt = N / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
  // DMA roles: instruction j (0..3) of this wave: piece j & 1, rows 64 * (j >> 1) + 16 * wave .. + 15 of each operand
  const int drow = lane >> 2, dg = lane & 3;
  const _Float16* asrc[4];
  const _Float16* bsrc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 64 * (j >> 1) + 16 * wave + drow;
    const int gk = dg ^ ((row >> 2) & 3);
    asrc[j] = A + (size_t)(j & 1) * M * K + (size_t)(m0 + row) * K + gk * 8;
    bsrc[j] = B + (size_t)(j & 1) * N * K + (size_t)(n0 + row) * K + gk * 8;
  }
  const unsigned lds0 = lds_addr(smem);
  auto issue = [&](int ks, int buf) {
    const unsigned sA = lds0 + buf * STAGE, sB = sA + OP;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned chunk = (unsigned)((j & 1) * PIECE + (64 * (j >> 1) + 16 * wave) * ROWB);
      glds16h(asrc[j] + ks * BK, __builtin_amdgcn_readfirstlane(sA + chunk));
      glds16h(bsrc[j] + ks * BK, __builtin_amdgcn_readfirstlane(sB + chunk));
    }
  };
  const int nk = K / BK;
  const int r = lane & 15, q = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][jj][e] = 0.f;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < nk) issue(s, s);
  int aoff[4], boff[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ra_ = wm * 64 + t * 16 + r, rb_ = wn * 64 + t * 16 + r;
    aoff[t] = ra_ * ROWB + ((q ^ ((ra_ >> 2) & 3)) << 4);
    boff[t] = rb_ * ROWB + ((q ^ ((rb_ >> 2) & 3)) << 4);
  }
  for (int ks = 0; ks < nk; ++ks) {
    if (ks + STAGES - 2 < nk) wait_vm<8 * (STAGES - 2)>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ks + STAGES - 1 < nk) issue(ks + STAGES - 1, (ks + STAGES - 1) % STAGES);
    const unsigned char* sA = smem + (ks % STAGES) * STAGE;
    const unsigned char* sB = sA + OP;
    uint4 fa[4][2], fb[4][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t][p] = *reinterpret_cast<const uint4*>(sA + p * PIECE + aoff[t]);
        fb[t][p] = *reinterpret_cast<const uint4*>(sB + p * PIECE + boff[t]);
      }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        f32x4 c = acc[tm][tn];
        MH(fa[tm][1], fb[tn][0], c); MH(fa[tm][0], fb[tn][1], c); MH(fa[tm][0], fb[tn][0], c);
        acc[tm][tn] = c;
      }
    __builtin_amdgcn_s_setprio(0);
  }
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = m0 + wm * 64 + tm * 16 + 4 * q + e;
        const int n = n0 + wn * 64 + tn * 16 + r;
        C[(size_t)m * N + n] = acc[tm][tn][e] * inv;
      }
}
// pre-split operands, register staging (global -> VGPR -> ds_write_b128 -> barrier): what the split itself costs
__global__ __launch_bounds__(256, 2) void gemm_pre_reg_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ B,
                                                              float* __restrict__ C, int M, int N, int K, float inv) {
  constexpr int BM = 128, BN = 128, BK = 32, PITCH = 96, NP = 2;
  constexpr int OP_BYTES = NP * BM * PITCH;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * OP_BYTES];
  unsigned char* As = smem;
  unsigned char* Bs = smem + OP_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nt = N / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
  // a thread moves 16 bytes (8 k) of 2 rows x 2 pieces per operand: granule tid & 3, row (tid >> 2) + 64 i
  const int g = tid & 3, row = tid >> 2;
  uint4 ra[2][2], rb[2][2];
  auto load = [&](int ks) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ra[p][i] = *reinterpret_cast<const uint4*>(A + (size_t)p * M * K + (size_t)(m0 + row + 64 * i) * K + ks * BK + g * 8);
        rb[p][i] = *reinterpret_cast<const uint4*>(B + (size_t)p * N * K + (size_t)(n0 + row + 64 * i) * K + ks * BK + g * 8);
      }
  };
  auto store = [&]() {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        *reinterpret_cast<uint4*>(As + p * BM * PITCH + (row + 64 * i) * PITCH + g * 16) = ra[p][i];
        *reinterpret_cast<uint4*>(Bs + p * BM * PITCH + (row + 64 * i) * PITCH + g * 16) = rb[p][i];
      }
  };
  const int nk = K / BK;
  const int r = lane & 15, q = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][jj][e] = 0.f;
  load(0);
  store();
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const bool more = ks + 1 < nk;
    if (more) load(ks + 1);
    __builtin_amdgcn_s_setprio(1);
    uint4 fa[4][NP], fb[4][NP];
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[t][p] = *reinterpret_cast<const uint4*>(As + p * BM * PITCH + (wm * 64 + t * 16 + r) * PITCH + q * 16);
        fb[t][p] = *reinterpret_cast<const uint4*>(Bs + p * BM * PITCH + (wn * 64 + t * 16 + r) * PITCH + q * 16);
      }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        f32x4 c = acc[tm][tn];
        MH(fa[tm][1], fb[tn][0], c); MH(fa[tm][0], fb[tn][1], c); MH(fa[tm][0], fb[tn][0], c);
        acc[tm][tn] = c;
      }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    if (more) store();
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
        C[(size_t)m * N + n] = acc[tm][tn][e] * inv;
      }
}
static void host_split(const std::vector<float>& v, float s, std::vector<_Float16>& out) {
  const size_t n = v.size();
  out.resize(2 * n);
  for (size_t i = 0; i < n; ++i) {
    const float x = v[i] * s;
    const _Float16 h = (_Float16)x;
    out[i] = h;
    out[n + i] = (_Float16)(x - (float)h);
  }
}

static float pow2_scale(const std::vector<float>& v) {
  float mx = 0.f;
  for (float x : v) mx = std::fmax(mx, std::fabs(x));
  int e;
  std::frexp(mx, &e);                       // mx = f * 2^e, f in [0.5, 1)
  return std::ldexp(1.f, 15 - e);           // largest magnitude -> [2^14, 2^15)
}

template <typename F>
static void run(const char* name, F launch, float* dC, int M, int N, int K, const std::vector<float>& hA, const std::vector<float>& hB) {
  CK(hipMemset(dC, 0, (size_t)M * N * 4));
  launch();
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  std::vector<float> hC((size_t)M * N);
  CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0.0, sq = 0.0;
  long cnt = 0;
  for (int m = 0; m < M; m += 97)
    for (int n = 0; n < N; n += 61) {
      double ref = 0.0, sabs = 0.0;
      for (int k = 0; k < K; ++k) { const double p = (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k]; ref += p; sabs += std::fabs(p); }
      const double e = std::fabs(hC[(size_t)m * N + n] - ref) / sabs;
      worst = std::fmax(worst, e); sq += e * e; ++cnt;
    }
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  const int reps = 20;
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, a, b));
  ms /= reps;
  printf("  %-38s %8.1f us  %7.1f TFLOP/s   max err / sum|ab| %.2e  rms %.2e\n", name, 1e3 * ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12, worst,
         std::sqrt(sq / cnt));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}

int main() {
  const int shapes[][3] = {{16384, 4096, 2048}, {8192, 2048, 1024}, {8192, 2048, 256}, {65536, 256, 320}, {4864, 2048, 2304}};
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    std::mt19937 g(1);
    std::uniform_real_distribution<float> ud(-1.f, 1.f);
    for (auto& v : hA) v = ud(g);
    for (auto& v : hB) v = ud(g) * 0.05f;
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    const float sa = pow2_scale(hA), sb = pow2_scale(hB);
    const dim3 grid((M / 128) * (N / 128)), block(256);
    printf("M %d N %d K %d (%d workgroups)\n", M, N, K, grid.x);
    run("register staging (adopted)", [&] { hipLaunchKernelGGL(gemm_reg_kernel, grid, block, 0, 0, dA, dB, dC, M, N, K, sa, sb); }, dC, M, N, K, hA, hB);
    run("LDS-DMA 2 stages, 2 WG/CU", [&] { hipLaunchKernelGGL((gemm_dma_kernel<2, 2>), grid, block, 0, 0, dA, dB, dC, M, N, K, sa, sb); }, dC, M, N, K, hA, hB);
    run("LDS-DMA 3 stages, 1 WG/CU", [&] { hipLaunchKernelGGL((gemm_dma_kernel<3, 1>), grid, block, 0, 0, dA, dB, dC, M, N, K, sa, sb); }, dC, M, N, K, hA, hB);
    run("LDS-DMA 4 stages, 1 WG/CU", [&] { hipLaunchKernelGGL((gemm_dma_kernel<4, 1>), grid, block, 0, 0, dA, dB, dC, M, N, K, sa, sb); }, dC, M, N, K, hA, hB);
    {
      std::vector<_Float16> pA, pB;
      host_split(hA, sa, pA); host_split(hB, sb, pB);
      _Float16 *dpA, *dpB;
      CK(hipMalloc(&dpA, pA.size() * 2)); CK(hipMalloc(&dpB, pB.size() * 2));
      CK(hipMemcpy(dpA, pA.data(), pA.size() * 2, hipMemcpyHostToDevice));
      CK(hipMemcpy(dpB, pB.data(), pB.size() * 2, hipMemcpyHostToDevice));
      const float inv = 1.f / (sa * sb);
      run("pre-split, register staging", [&] { hipLaunchKernelGGL(gemm_pre_reg_kernel, grid, block, 0, 0, dpA, dpB, dC, M, N, K, inv); }, dC, M, N, K, hA, hB);
      run("pre-split, LDS-DMA 2 stages, 2 WG/CU", [&] { hipLaunchKernelGGL((gemm_pre_kernel<2, 2>), grid, block, 0, 0, dpA, dpB, dC, M, N, K, inv); }, dC, M, N, K, hA, hB);
      run("pre-split, LDS-DMA 3 stages, 1 WG/CU", [&] { hipLaunchKernelGGL((gemm_pre_kernel<3, 1>), grid, block, 0, 0, dpA, dpB, dC, M, N, K, inv); }, dC, M, N, K, hA, hB);
      run("pre-split, LDS-DMA 4 stages, 1 WG/CU", [&] { hipLaunchKernelGGL((gemm_pre_kernel<4, 1>), grid, block, 0, 0, dpA, dpB, dC, M, N, K, inv); }, dC, M, N, K, hA, hB);
      CK(hipFree(dpA)); CK(hipFree(dpB));
    }
    run("register staging (again)", [&] { hipLaunchKernelGGL(gemm_reg_kernel, grid, block, 0, 0, dA, dB, dC, M, N, K, sa, sb); }, dC, M, N, K, hA, hB);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  }
  return 0;
}
