// Probe: what does the stream-K seam cost in this workload, as a separate fix-up launch (the adopted form) and folded
// into the conv launch?  C[M][N] = A[M][K] * B[N][K]^T with the bf16x6 loop of the engine (128x128x32 tiles, 256 threads,
// 2 workgroups per CU); the tiles x K-steps units are dealt to `nwg` workgroups in equal contiguous runs, so every
// workgroup ends with <= 2 partial tiles parked as 64 KB fp32 slabs.
//   V0  partial slabs + a second launch that sums them in workgroup order (what conv_kernels.hip does)
//   V1  in-launch: plain slab stores, agent-scope release fence, ticket; the LAST arriver of a tile acquires, sums the
//       slabs in workgroup order (same arithmetic as V0) and writes the tile
//   V2  as V1 with write-through (sc1) slab stores and sc1 slab loads instead of the two fences
// No workgroup ever waits for another (safe beside other launches); results are compared with V0 bit for bit.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/splitk_seam_probe.cpp -o tools/probes/bin/splitk_seam
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__device__ __forceinline__ unsigned pack_hi(float e0, float e1) {
  return __builtin_amdgcn_perm(__float_as_uint(e1), __float_as_uint(e0), 0x07060302u);
}
__device__ __forceinline__ float trunc_bf16(float a) { return __uint_as_float(__float_as_uint(a) & 0xffff0000u); }
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_sc1(float* p, float4 v) {
  const f4v w = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ float4 load_sc1(const float* p) {
  f4v w;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
  return make_float4(w.x, w.y, w.z, w.w);
}

struct Args {
  const float* A; const float* B; float* C; float* ws; int* cnt;
  int M, N, K, per, nwg;
};

// sum the slabs of `tile` in workgroup order, rows [r0, r1), and write C
template <int MODE>
__device__ __forceinline__ void combine(const Args& p, int tile, int r0, int r1, int tid) {
  constexpr int BM = 128, BN = 128;
  const int nt = p.N / BN, ksteps = p.K / 32;
  const long a = (long)tile * ksteps, b = a + ksteps;
  const int g0 = (int)(a / p.per), g1 = (int)((b - 1) / p.per);
  const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
  const int c4 = tid & 31, cr = tid >> 5;                 // 32 float4 per row, 8 rows per pass
  for (int row = r0 + cr; row < r1; row += 8) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = g0; g <= g1; ++g) {
      const int slot = ((long)g * p.per >= a) ? 0 : 1;
      const float* src = p.ws + ((size_t)g * 2 + slot) * (BM * BN) + row * BN + c4 * 4;
      const float4 t = MODE == 2 ? load_sc1(src) : *reinterpret_cast<const float4*>(src);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    *reinterpret_cast<float4*>(p.C + (size_t)(m0 + row) * p.N + n0 + c4 * 4) = s;
  }
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void streamk_kernel(const Args p) {
  constexpr int BM = 128, BN = 128, ROWB = 80, OP = 3 * BM * ROWB, LDC = BN + 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * OP];
  __shared__ int s_last;
  unsigned char* As = smem;
  unsigned char* Bs = smem + OP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5, c4 = tid & 7;
  const int row = (wave << 3) + (((lane >> 3) & 1) << 2) + (lane >> 4);
  const int ksteps = p.K / 32, nt = p.N / BN;
  const long U = (long)(p.M / BM) * nt * ksteps;
  const int bid = blockIdx.x;
  long u = (long)bid * p.per, u_end = u + p.per;
  if (u_end > U) u_end = U;
  const long u_begin = u;
  while (u < u_end) {
    const int tile = (int)(u / ksteps), ks_begin = (int)(u - (long)tile * ksteps);
    int ks_end = ksteps;
    if ((long)ks_end - ks_begin > u_end - u) ks_end = ks_begin + (int)(u_end - u);
    const int m0 = (tile / nt) * BM, n0 = (tile % nt) * BN;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float4 ra[4], rb[4];
    auto load = [&](int ks) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ra[i] = *reinterpret_cast<const float4*>(p.A + (size_t)(m0 + row + 32 * i) * p.K + ks * 32 + c4 * 4);
        rb[i] = *reinterpret_cast<const float4*>(p.B + (size_t)(n0 + row + 32 * i) * p.K + ks * 32 + c4 * 4);
      }
    };
    auto store = [&](unsigned char* S, const float4* rv) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float4 v = rv[i];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          uint2 w;
          w.x = pack_hi(v.x, v.y); w.y = pack_hi(v.z, v.w);
          *reinterpret_cast<uint2*>(S + (q * BM + row + 32 * i) * ROWB + c4 * 8) = w;
          if (q < 2) { v.x -= trunc_bf16(v.x); v.y -= trunc_bf16(v.y); v.z -= trunc_bf16(v.z); v.w -= trunc_bf16(v.w); }
        }
      }
    };
    load(ks_begin);
    __syncthreads();
    store(As, ra); store(Bs, rb);
    __syncthreads();
    for (int ks = ks_begin; ks < ks_end; ++ks) {
      const bool more = ks + 1 < ks_end;
      if (more) load(ks + 1);
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        bf16x8 fa[2][3], fb[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            fa[t][q] = *reinterpret_cast<const bf16x8*>(As + (q * BM + wm * 64 + t * 32 + r) * ROWB + g * 32 + h * 16);
            fb[t][q] = *reinterpret_cast<const bf16x8*>(Bs + (q * BM + wn * 64 + t * 32 + r) * ROWB + g * 32 + h * 16);
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
      __syncthreads();
      if (more) { store(As, ra); store(Bs, rb); }
      __syncthreads();
    }
    // epilogue through LDS (two passes of 64 rows)
    float* Cs = reinterpret_cast<float*>(smem);
    const bool full = ks_begin == 0 && ks_end == ksteps;
    float* slab = p.ws + ((size_t)bid * 2 + (u == u_begin ? 0 : 1)) * (BM * BN);
    const int e4 = tid & 31, er = tid >> 5;
#pragma unroll
    for (int ep = 0; ep < 2; ++ep) {
      if (ep) __syncthreads();
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          Cs[(wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * LDC + wn * 64 + tn * 32 + r] = acc[ep][tn][e];
      __syncthreads();
      for (int lr = er; lr < 64; lr += 8) {
        const int trow = ((lr >> 5) << 6) + (ep << 5) + (lr & 31);
        const float4 v = *reinterpret_cast<const float4*>(Cs + lr * LDC + e4 * 4);
        if (full) *reinterpret_cast<float4*>(p.C + (size_t)(m0 + trow) * p.N + n0 + e4 * 4) = v;
        else if (MODE == 2) store_sc1(slab + trow * BN + e4 * 4, v);
        else *reinterpret_cast<float4*>(slab + trow * BN + e4 * 4) = v;
      }
    }
    if (!full && MODE >= 1) {
      const long a = (long)tile * ksteps, b = a + ksteps;
      const int g0 = (int)(a / p.per), g1 = (int)((b - 1) / p.per);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        if (MODE == 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        const int prev = __hip_atomic_fetch_add(p.cnt + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == g1 - g0;
        if (s_last && MODE == 1) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      }
      __syncthreads();
      if (s_last) {
        combine<MODE>(p, tile, 0, BM, tid);
        if (tid == 0) p.cnt[tile] = 0;
      }
    }
    __syncthreads();
    u += ks_end - ks_begin;
  }
}

__global__ __launch_bounds__(256) void fixup_kernel(const Args p) {
  const int ksteps = p.K / 32;
  const int tile = blockIdx.x;
  const long a = (long)tile * ksteps, b = a + ksteps;
  if ((int)(a / p.per) == (int)((b - 1) / p.per) && a % p.per == 0 && b % p.per == 0) return;   // whole in one workgroup
  if ((int)(a / p.per) == (int)((b - 1) / p.per)) return;
  combine<0>(p, tile, blockIdx.y * 16, blockIdx.y * 16 + 16, threadIdx.x);
}

template <int MODE>
static int run(const char* name, Args p, int tiles, std::vector<float>* out, const std::vector<float>* ref) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto go = [&]() {
    hipLaunchKernelGGL((streamk_kernel<MODE>), dim3(p.nwg), dim3(256), 0, 0, p);
    if (MODE == 0) hipLaunchKernelGGL(fixup_kernel, dim3(tiles, 8), dim3(256), 0, 0, p);
  };
  CK(hipMemset(p.C, 0, (size_t)p.M * p.N * 4));
  for (int i = 0; i < 3; ++i) go();
  CK(hipDeviceSynchronize());
  const int reps = 40;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) go();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  out->resize((size_t)p.M * p.N);
  CK(hipMemcpy(out->data(), p.C, out->size() * 4, hipMemcpyDeviceToHost));
  long bad = 0;
  if (ref) for (size_t i = 0; i < out->size(); ++i) bad += (*out)[i] != (*ref)[i];
  printf("  %-58s %7.1f us per GEMM   %s\n", name, 1e3 * ms / reps, ref ? (bad ? "MISMATCH vs V0" : "bit-identical to V0") : "");
  if (bad) printf("     %ld differing elements\n", bad);
  return 0;
}

int main(int argc, char** argv) {
  const int wg_budget = argc > 1 ? atoi(argv[1]) : 512;      // workgroups the plan deals the units to
  const int shapes[][3] = {{4864, 256, 2304}, {4864, 256, 1024}, {1664, 256, 2304}, {4864, 1024, 512}, {19328, 128, 1152}};
  for (auto& s : shapes) {
    const int M = s[0], N = s[1], K = s[2];
    const int tiles = (M / 128) * (N / 128), ksteps = K / 32;
    const long U = (long)tiles * ksteps;
    int nwg = wg_budget;
    if (U / nwg < 3) nwg = (int)(U / 3);
    const int per = (int)((U + nwg - 1) / nwg);
    nwg = (int)((U + per - 1) / per);
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    srand(1);
    for (auto& v : hA) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : hB) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
    Args p;
    float *dA, *dB;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4));
    CK(hipMalloc(&p.C, (size_t)M * N * 4)); CK(hipMalloc(&p.ws, (size_t)nwg * 2 * 128 * 128 * 4)); CK(hipMalloc(&p.cnt, tiles * 4));
    CK(hipMemset(p.cnt, 0, tiles * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    p.A = dA; p.B = dB; p.M = M; p.N = N; p.K = K; p.per = per; p.nwg = nwg;
    printf("M=%d N=%d K=%d: %d tiles x %d K steps over %d workgroups (%d K steps each, %.1f contributors per tile)\n", M, N, K, tiles,
           ksteps, nwg, per, (double)nwg / tiles + 1);
    std::vector<float> v0, v1, v2;
    run<0>("V0 slabs + separate fix-up launch", p, tiles, &v0, nullptr);
    run<1>("V1 in-launch, release / acquire fences, last arriver sums", p, tiles, &v1, &v0);
    run<2>("V2 in-launch, sc1 slab stores and loads, last arriver sums", p, tiles, &v2, &v0);
    // sanity of V0 itself on one row against fp64
    double max_err = 0;
    for (int n = 0; n < N; n += 7) {
      double sref = 0;
      for (int k = 0; k < K; ++k) sref += (double)hA[(size_t)5 * K + k] * hB[(size_t)n * K + k];
      const double e = fabs(sref - v0[(size_t)5 * N + n]);
      if (e > max_err) max_err = e;
    }
    printf("  V0 row 5 vs fp64: max |err| %.2e\n", max_err);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(p.C)); CK(hipFree(p.ws)); CK(hipFree(p.cnt));
  }
  return 0;
}
