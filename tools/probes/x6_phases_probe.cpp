// Probe: where does the bf16x6 K step lose time?  One 128x128x32 K step of the adopted kernel (4 waves, 64x64 per wave,
// 48 MFMA 32x32x16 per wave, 2 workgroups per CU), rebuilt phase by phase.  Results are garbage on purpose (no global
// operands); only the rate matters.  PH:
//   0  MFMAs only, fragments in registers, 6-chain per accumulator (the order of the adopted kernel)
//   1  MFMAs only, round-robin over the 4 accumulators inside each of the 6 terms
//   2  + the 24 ds_read_b128 fragment reads per K step (LDS never rewritten)
//   3  + the two barriers of a K step
//   4  + 24 ds_write_b64 of constant registers between the barriers (no split arithmetic)
//   5  + split arithmetic on register values (no global loads)
//   6  + 8 global dwordx4 loads per thread per K step (the full adopted loop)
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/x6_phases_probe.cpp -o /tmp/x6_phases
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__device__ __forceinline__ unsigned pack_hi(float e0, float e1) {
  return __builtin_amdgcn_perm(__float_as_uint(e1), __float_as_uint(e0), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float a) { return __uint_as_float(__float_as_uint(a) & 0xffff0000u); }

#ifndef RANDOM_DATA
#define RANDOM_DATA 0     // 1: operands with random mantissas and signs (switching power), 0: constants / zeros
#endif
template <int PH>
__global__ __launch_bounds__(256, 2) void phase_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                       int K, int nk) {
  constexpr int BM = 128, ROWB = 80, OP = 3 * BM * ROWB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * OP];
  unsigned char* As = smem;
  unsigned char* Bs = smem + OP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int c4 = tid & 7;
  const int row = (wave << 3) + (((lane >> 3) & 1) << 2) + (lane >> 4);
  for (int i = tid; i < 2 * OP / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = RANDOM_DATA ? (0x3c003c00u ^ ((i * 2654435761u) & 0x80ff80ffu)) : 0x3c003c00u + i;
  __syncthreads();
  const size_t a_base = (size_t)(blockIdx.x % 64) * BM * K, b_base = (size_t)(blockIdx.x % 16) * BM * K;
  float4 ra[4], rb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { ra[i] = make_float4(1.f + tid, 2.f, 3.f, 4.f + i); rb[i] = make_float4(.1f, .2f + tid, .3f, .4f + i); }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  bf16x8 fa[2][2][3], fb[2][2][3];            // [g][t][piece]
  auto read_frags = [&](int g) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fa[g][t][p] = *reinterpret_cast<const bf16x8*>(As + (p * BM + wm * 64 + t * 32 + r) * ROWB + g * 32 + h * 16);
        fb[g][t][p] = *reinterpret_cast<const bf16x8*>(Bs + (p * BM + wn * 64 + t * 32 + r) * ROWB + g * 32 + h * 16);
      }
  };
  read_frags(0);
  read_frags(1);
  auto store_op = [&](unsigned char* S, const float4* rv, bool split) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = rv[i];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        uint2 w;
        if (split) { w.x = pack_hi(v.x, v.y); w.y = pack_hi(v.z, v.w); }
        else { w.x = __float_as_uint(v.x) + p; w.y = __float_as_uint(v.y); }
        *reinterpret_cast<uint2*>(S + (p * BM + row + 32 * i) * ROWB + c4 * 8) = w;
        if (split && p < 2) { v.x -= trunc_bf16(v.x); v.y -= trunc_bf16(v.y); v.z -= trunc_bf16(v.z); v.w -= trunc_bf16(v.w); }
      }
    }
  };
  for (int ks = 0; ks < nk; ++ks) {
    if (PH >= 6) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const float4*>(A + a_base + (size_t)(row + 32 * i) * K + (ks % (K / 32)) * 32 + c4 * 4);
        rb[i] = *reinterpret_cast<const float4*>(B + b_base + (size_t)(row + 32 * i) * K + (ks % (K / 32)) * 32 + c4 * 4);
      }
    }
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      if (PH >= 2) { asm volatile("" ::: "memory"); read_frags(g); }
      if (PH == 1) {
#pragma unroll
        for (int term = 0; term < 6; ++term) {
          const int pa = term == 0 ? 2 : term == 1 ? 0 : term == 2 ? 1 : term == 3 ? 1 : 0;
          const int pb = term == 0 ? 0 : term == 1 ? 2 : term == 2 ? 1 : term == 3 ? 0 : term == 4 ? 1 : 0;
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
              acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][tm][pa], fb[g][tn][pb], acc[tm][tn], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
          for (int tn = 0; tn < 2; ++tn) {
            f32x16 c = acc[tm][tn];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][tm][2], fb[g][tn][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][tm][0], fb[g][tn][2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][tm][1], fb[g][tn][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][tm][1], fb[g][tn][0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][tm][0], fb[g][tn][1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][tm][0], fb[g][tn][0], c, 0, 0, 0);
            acc[tm][tn] = c;
          }
      }
    }
    if (PH >= 3) __syncthreads();
    if (PH >= 4) { store_op(As, ra, PH >= 5); store_op(Bs, rb, PH >= 5); }
    if (PH >= 3) __syncthreads();
    if (PH == 5) {          // keep the split inputs changing so the compiler cannot hoist the arithmetic
#pragma unroll
      for (int i = 0; i < 4; ++i) { ra[i].x += 1.f; rb[i].y += 1.f; asm volatile("" : "+v"(ra[i].z), "+v"(rb[i].w)); }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  C[(size_t)blockIdx.x * 256 + tid] = s;
}


// PH 7: the A operand never touches LDS -- every lane loads the 8 consecutive k of "its" row (two dwordx4) for each of its
// 2 row blocks x 2 k groups straight into registers and splits them into MFMA fragments there (the two waves that share
// the rows both do it); only B is staged through LDS (half the LDS writes and fragment reads, 1.5x the split arithmetic).
__global__ __launch_bounds__(256, 2) void adirect_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                         int K, int nk) {
  constexpr int BM = 128, ROWB = 80, OP = 3 * BM * ROWB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[OP];
  unsigned char* Bs = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int c4 = tid & 7;
  const int row = (wave << 3) + (((lane >> 3) & 1) << 2) + (lane >> 4);
  for (int i = tid; i < OP / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + i;
  __syncthreads();
  const size_t a_base = (size_t)(blockIdx.x % 64) * BM * K, b_base = (size_t)(blockIdx.x % 16) * BM * K;
  float4 ra[2][2][2], rb[4];            // A raw: [g][tm][half of the 8 k]
  bf16x8 fa[2][2][3];
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  auto load = [&](int ks) {
    const int k0 = (ks % (K / 32)) * 32;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int q = 0; q < 2; ++q)
          ra[g][tm][q] = *reinterpret_cast<const float4*>(A + a_base + (size_t)(wm * 64 + tm * 32 + r) * K + k0 + g * 16 + h * 8 + q * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      rb[i] = *reinterpret_cast<const float4*>(B + b_base + (size_t)(row + 32 * i) * K + k0 + c4 * 4);
  };
  auto split_a = [&]() {
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int tm = 0; tm < 2; ++tm) {
        float4 u = ra[g][tm][0], v = ra[g][tm][1];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          uint4 w;
          w.x = pack_hi(u.x, u.y); w.y = pack_hi(u.z, u.w); w.z = pack_hi(v.x, v.y); w.w = pack_hi(v.z, v.w);
          fa[g][tm][p] = *reinterpret_cast<bf16x8*>(&w);
          if (p < 2) {
            u.x -= trunc_bf16(u.x); u.y -= trunc_bf16(u.y); u.z -= trunc_bf16(u.z); u.w -= trunc_bf16(u.w);
            v.x -= trunc_bf16(v.x); v.y -= trunc_bf16(v.y); v.z -= trunc_bf16(v.z); v.w -= trunc_bf16(v.w);
          }
        }
      }
  };
  auto store_b = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = rb[i];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        uint2 w;
        w.x = pack_hi(v.x, v.y); w.y = pack_hi(v.z, v.w);
        *reinterpret_cast<uint2*>(Bs + (p * BM + row + 32 * i) * ROWB + c4 * 8) = w;
        if (p < 2) { v.x -= trunc_bf16(v.x); v.y -= trunc_bf16(v.y); v.z -= trunc_bf16(v.z); v.w -= trunc_bf16(v.w); }
      }
    }
  };
  load(0);
  split_a();
  store_b();
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    load(ks + 1);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      bf16x8 fb[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          fb[t][p] = *reinterpret_cast<const bf16x8*>(Bs + (p * BM + wn * 64 + t * 32 + r) * ROWB + g * 32 + h * 16);
#pragma unroll
      for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
          f32x16 c = acc[tm][tn];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][tm][2], fb[tn][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][tm][0], fb[tn][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][tm][1], fb[tn][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][tm][1], fb[tn][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][tm][0], fb[tn][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[g][tm][0], fb[tn][0], c, 0, 0, 0);
          acc[tm][tn] = c;
        }
    }
    __syncthreads();
    store_b();
    split_a();
    __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  C[(size_t)blockIdx.x * 256 + tid] = s;
}


// PH 8 / 9: the same work per 32 k, restructured as an in-wave software pipeline that keeps two workgroups per CU:
// K step = 16, two LDS stages of 30 KB each (61 KB per workgroup as before), ONE barrier per step.  While a wave issues
// the 24 MFMAs of step k from stage k & 1 it splits the registers of step k + 1 and writes them into the other stage; the
// global loads of step k + 2 are in flight.  PH 9 pins the interleaving with sched_group_barrier (1 MFMA : 4 VALU,
// a DS write after every second MFMA); PH 8 leaves the order to the compiler.
template <int PIN>
__global__ __launch_bounds__(256, 2) void pipe16_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                        int K, int nk) {
  constexpr int BM = 128, ROWB = 48, PL = BM * ROWB, OP = 3 * PL, STAGE = 2 * OP;        // 16 bf16 (32 B) + 16 B pad per row
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int c4 = tid & 3, row = tid >> 2;                  // 4 float4 per 16-k row, 64 rows per pass, 2 passes per operand
  for (int i = tid; i < 2 * STAGE / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = RANDOM_DATA ? (0x3c003c00u ^ ((i * 2654435761u) & 0x80ff80ffu)) : 0x3c003c00u + i;
  __syncthreads();
  const size_t a_base = (size_t)(blockIdx.x % 64) * BM * K, b_base = (size_t)(blockIdx.x % 16) * BM * K;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float4 rn[4], rf[4];                                    // registers of step k + 1 (being split) and k + 2 (in flight)
  auto load = [&](float4* rv, int ks) {
    const int k0 = (ks % (K / 16)) * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      rv[i] = *reinterpret_cast<const float4*>(A + a_base + (size_t)(row + 64 * i) * K + k0 + c4 * 4);
      rv[2 + i] = *reinterpret_cast<const float4*>(B + b_base + (size_t)(row + 64 * i) * K + k0 + c4 * 4);
    }
  };
  load(rn, 1);
  load(rf, 2);
  for (int ks = 0; ks < 2 * nk; ++ks) {
    unsigned char* cur = smem + (ks & 1) * STAGE;
    unsigned char* nxt = smem + ((ks & 1) ^ 1) * STAGE;
    bf16x8 fa[2][3], fb[2][3];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fa[t][p] = *reinterpret_cast<const bf16x8*>(cur + (p * BM + wm * 64 + t * 32 + r) * ROWB + h * 16);
        fb[t][p] = *reinterpret_cast<const bf16x8*>(cur + OP + (p * BM + wn * 64 + t * 32 + r) * ROWB + h * 16);
      }
    // split + store of step k + 1, written so that the scheduler can spread it between the MFMAs
    uint2 pc[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = rn[i];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        pc[i][p].x = pack_hi(v.x, v.y);
        pc[i][p].y = pack_hi(v.z, v.w);
        if (p < 2) { v.x -= trunc_bf16(v.x); v.y -= trunc_bf16(v.y); v.z -= trunc_bf16(v.z); v.w -= trunc_bf16(v.w); }
      }
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
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        *reinterpret_cast<uint2*>(nxt + (i >> 1) * OP + (p * BM + row + 64 * (i & 1)) * ROWB + c4 * 8) = pc[i][p];
    if (PIN) {
#pragma unroll
      for (int m = 0; m < 24; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // four VALU (the split is ~90 of them)
        if (m & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // a DS write
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) rn[i] = rf[i];
    load(rf, ks + 3);
    __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  C[(size_t)blockIdx.x * 256 + tid] = s;
}


// PH 10: one workgroup per CU, one wave per SIMD with the whole 512-entry register file: wave tile 128 x 128 (4 x 4 MFMA
// tiles, 256 accumulator registers), workgroup tile 256 x 256, K step 16 with two LDS stages (2 x 73.7 KB), one barrier
// per step, in-wave pipeline as PH 8.  Per multiply-accumulate this halves the global loads, the split arithmetic, the
// LDS writes and the fragment reads of the 128 x 128 workgroup tile.
template <int PIN>
__global__ __launch_bounds__(256, 1) void big_tile_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                          int K, int nk) {
  constexpr int BM = 256, ROWB = 48, PL = BM * ROWB, OP = 3 * PL, STAGE = 2 * OP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  const int c4 = tid & 3, row = tid >> 2;                  // 4 float4 per 16-k row, 64 rows per pass, 4 passes per operand
  for (int i = tid; i < 2 * STAGE / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = RANDOM_DATA ? (0x3c003c00u ^ ((i * 2654435761u) & 0x80ff80ffu)) : 0x3c003c00u + i;
  __syncthreads();
  const size_t a_base = (size_t)(blockIdx.x % 32) * BM * K, b_base = (size_t)(blockIdx.x % 8) * BM * K;
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float4 rn[8], rf[8];
  auto load = [&](float4* rv, int ks) {
    const int k0 = (ks % (K / 16)) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      rv[i] = *reinterpret_cast<const float4*>(A + a_base + (size_t)(row + 64 * i) * K + k0 + c4 * 4);
      rv[4 + i] = *reinterpret_cast<const float4*>(B + b_base + (size_t)(row + 64 * i) * K + k0 + c4 * 4);
    }
  };
  load(rn, 1);
  load(rf, 2);
  for (int ks = 0; ks < 2 * nk; ++ks) {
    unsigned char* cur = smem + (ks & 1) * STAGE;
    unsigned char* nxt = smem + ((ks & 1) ^ 1) * STAGE;
    bf16x8 fa[4][3], fb[4][3];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        fa[t][p] = *reinterpret_cast<const bf16x8*>(cur + (p * BM + wm * 128 + t * 32 + r) * ROWB + h * 16);
        fb[t][p] = *reinterpret_cast<const bf16x8*>(cur + OP + (p * BM + wn * 128 + t * 32 + r) * ROWB + h * 16);
      }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float4 v = rn[i];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        uint2 w;
        w.x = pack_hi(v.x, v.y);
        w.y = pack_hi(v.z, v.w);
        *reinterpret_cast<uint2*>(nxt + (i >> 2) * OP + (p * BM + row + 64 * (i & 3)) * ROWB + c4 * 8) = w;
        if (p < 2) { v.x -= trunc_bf16(v.x); v.y -= trunc_bf16(v.y); v.z -= trunc_bf16(v.z); v.w -= trunc_bf16(v.w); }
      }
    }
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < 4; ++tn) {
        f32x16 c = acc[tm][tn];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][2], fb[tn][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], fb[tn][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][1], fb[tn][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][1], fb[tn][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], fb[tn][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], fb[tn][0], c, 0, 0, 0);
        acc[tm][tn] = c;
      }
    if (PIN == 1) {
#pragma unroll
      for (int m = 0; m < 96; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        if ((m & 3) == 3) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    } else if (PIN == 2) {                                   // all fragment reads first, then 1 MFMA : 2 VALU, writes late
      __builtin_amdgcn_sched_group_barrier(0x100, 24, 0);
#pragma unroll
      for (int m = 0; m < 96; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        if (m >= 48 && (m & 1)) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) rn[i] = rf[i];
    load(rf, ks + 3);
    __syncthreads();
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  C[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int PH>
static int run(const char* name, const float* dA, const float* dB, float* dC, int K, int nk, int wgs) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto go = [&]() {
    if (PH == 7) hipLaunchKernelGGL(adirect_kernel, dim3(wgs), dim3(256), 0, 0, dA, dB, dC, K, nk);
    else if (PH == 8) hipLaunchKernelGGL((pipe16_kernel<0>), dim3(wgs), dim3(256), 0, 0, dA, dB, dC, K, nk);
    else if (PH == 9) hipLaunchKernelGGL((pipe16_kernel<1>), dim3(wgs), dim3(256), 0, 0, dA, dB, dC, K, nk);
    else if (PH == 10) hipLaunchKernelGGL((big_tile_kernel<1>), dim3(wgs / 4), dim3(256), 2 * 2 * 3 * 256 * 48, 0, dA, dB, dC, K, nk);
    else if (PH == 11) hipLaunchKernelGGL((big_tile_kernel<0>), dim3(wgs / 4), dim3(256), 2 * 2 * 3 * 256 * 48, 0, dA, dB, dC, K, nk);
    else if (PH == 12) hipLaunchKernelGGL((big_tile_kernel<2>), dim3(wgs / 4), dim3(256), 2 * 2 * 3 * 256 * 48, 0, dA, dB, dC, K, nk);
    else hipLaunchKernelGGL((phase_kernel<(PH >= 7 ? 0 : PH)>), dim3(wgs), dim3(256), 0, 0, dA, dB, dC, K, nk);
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
  const double flops = 2.0 * 128 * 128 * 32 * (double)nk * wgs;
  printf("%-44s wgs=%4d nk=%4d  %8.1f us  %7.1f TFLOP/s(fp32-equivalent)  %5.1f %% of 416.7\n", name, wgs, nk, ms * 1e3,
         flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 416.7 * 100);
  return 0;
}

int main() {
  CK(hipFuncSetAttribute((const void*)big_tile_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 3 * 256 * 48));
  CK(hipFuncSetAttribute((const void*)big_tile_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 3 * 256 * 48));
  CK(hipFuncSetAttribute((const void*)big_tile_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * 3 * 256 * 48));
  const int K = 1024;
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, (size_t)8192 * K * 4));
  CK(hipMalloc(&dB, (size_t)2048 * K * 4));
  CK(hipMalloc(&dC, (size_t)4096 * 256 * 4));
  if (RANDOM_DATA) {
    std::vector<float> h((size_t)8192 * K);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    CK(hipMemcpy(dA, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, h.data() + 12345, (size_t)2048 * K * 4, hipMemcpyHostToDevice));
  } else {
    CK(hipMemset(dA, 0, (size_t)8192 * K * 4));
    CK(hipMemset(dB, 0, (size_t)2048 * K * 4));
  }
  printf("RANDOM_DATA=%d\n", RANDOM_DATA);
  for (int wgs : {512, 1024}) {
    const int nk = 64;
    run<0>("0 MFMA only, 6-chain per accumulator", dA, dB, dC, K, nk, wgs);
    run<1>("1 MFMA only, round-robin accumulators", dA, dB, dC, K, nk, wgs);
    run<2>("2 + fragment reads (24 ds_read_b128)", dA, dB, dC, K, nk, wgs);
    run<3>("3 + two barriers", dA, dB, dC, K, nk, wgs);
    run<4>("4 + 24 ds_write_b64 (no split)", dA, dB, dC, K, nk, wgs);
    run<5>("5 + split arithmetic", dA, dB, dC, K, nk, wgs);
    run<6>("6 + global loads (full loop)", dA, dB, dC, K, nk, wgs);
    run<7>("7 full loop, A direct to registers", dA, dB, dC, K, nk, wgs);
    run<8>("8 K step 16, 2 LDS stages, 1 barrier, compiler order", dA, dB, dC, K, nk, wgs);
    run<9>("9 same, MFMA / VALU / DS-write interleave pinned", dA, dB, dC, K, nk, wgs);
    run<10>("10 256x256 tile, 1 wave / SIMD, 128x128 per wave", dA, dB, dC, K, nk, wgs);
    run<11>("11 same, compiler order", dA, dB, dC, K, nk, wgs);
    run<12>("12 same, reads first / writes late", dA, dB, dC, K, nk, wgs);
  }
  return 0;
}
