// Probe (round 4): a STREAMING kernel structure for the short-K 1x1 convolutions of layer1-3, whose launches are HBM-bound
// and run at 2-3.5 TB/s in the tiled implicit-GEMM kernel (a 128 x 128 tile with 2-8 K steps spends most of its time in the
// fixed cost per tile: prologue, first operand latency, two barriers per K step, the LDS-staged epilogue).
//   out[m][n] = relu(a[n] * sum_k X[m][k] W[n][k] + b[n] + res[m][n]),  mask8[m][n / 4] = bits(out > 0)
// Structure: one 1024-thread workgroup per CU keeps the WHOLE weight matrix of its column range in LDS, split once into the two
// fp16 pieces of the f16x3 product; each of its 16 waves then streams strips of 16 pixels on its own -- no barrier after the
// weight staging: X fragments straight from global into registers (a lane's 8 k values are 32 contiguous bytes), split per wave,
// D = W_frag * X_frag^T on v_mfma_f32_16x16x32_f16 so that a lane ends up with 4 CONSECUTIVE channels of one pixel (float4
// residual loads / output stores, one mask byte per lane), next strip's loads in flight behind the current strip's MFMAs.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/skinny_probe.cpp -o tools/probes/bin/skinny
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned pack_f16(float e0, float e1) {
  f32x2 v = {e0, e1};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}
__device__ __forceinline__ f32x2 unpack_f16(unsigned w) { return __builtin_convertvector(__builtin_bit_cast(f16x2, w), f32x2); }
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

// K: reduction channels (64 / 128 / 256), NC: output channels per workgroup (its LDS holds NC x K x 2 pieces x 2 bytes)
template <int K, int NC, int WAVES, int FHMAX = 8>
__global__ __launch_bounds__(WAVES * 64, 1) void skinny_fwd_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W,
                                                                   const float* __restrict__ sc, const float* __restrict__ bi,
                                                                   const float* __restrict__ res, float* __restrict__ out,
                                                                   unsigned char* __restrict__ mask8, int M, int N, float sx, float sw) {
  constexpr int KS = K / 32;                 // K steps of the 16x16x32 MFMA
  constexpr int PITCH = K * 2 + 16;          // bytes per weight row of one piece (16-byte pad: conflict-free fragment reads)
  constexpr int NF = NC / 16;                // 16-channel fragments
  constexpr int FH = NF < FHMAX ? NF : FHMAX;        // fragments per pass (4 accumulator registers each)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [2 pieces][NC rows][PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.y * NC;
  // ---- weights of this column range: fp32 -> the two fp16 pieces, once ----
  for (int i = tid; i < NC * (K / 8); i += WAVES * 64) {
    const int r = i / (K / 8), c8 = i % (K / 8);
    const float4 lo = *reinterpret_cast<const float4*>(W + (size_t)(n0 + r) * K + c8 * 8);
    const float4 hi = *reinterpret_cast<const float4*>(W + (size_t)(n0 + r) * K + c8 * 8 + 4);
    uint4 h0, h1;
    split8(lo, hi, sw, h0, h1);
    *reinterpret_cast<uint4*>(smem + r * PITCH + c8 * 16) = h0;
    *reinterpret_cast<uint4*>(smem + NC * PITCH + r * PITCH + c8 * 16) = h1;
  }
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  const float inv = 1.f / (sx * sw);
  // strips of 16 pixels, dealt round-robin over all waves of the grid
  const int nstrips = (M + 15) / 16;
  const int gw = blockIdx.x * WAVES + wave, gstride = gridDim.x * WAVES;
  float4 xr[KS][2];
  auto load_x = [&](int strip) {
    const int m = strip * 16 + fr;
    const float* p = X + (size_t)(m < M ? m : M - 1) * ldx + fq * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      xr[ks][0] = *reinterpret_cast<const float4*>(p + ks * 32);
      xr[ks][1] = *reinterpret_cast<const float4*>(p + ks * 32 + 4);
    }
  };
  int strip = gw;
  if (strip < nstrips) load_x(strip);
  for (; strip < nstrips; strip += gstride) {
    uint4 x0[KS], x1[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) split8(xr[ks][0], xr[ks][1], sx, x0[ks], x1[ks]);
    const int nxt = strip + gstride;
    if (nxt < nstrips) load_x(nxt);                       // next strip's activations in flight behind this strip's work
    const int m = strip * 16 + fr;
    const bool ok = m < M;
    const size_t row = (size_t)(ok ? m : M - 1);
#pragma unroll 1
    for (int half = 0; half < NF; half += FH) {            // FH fragments (<= 128 channels) at a time
      float4 rs[FH];
      if (res) {
#pragma unroll
        for (int f = 0; f < FH; ++f) rs[f] = *reinterpret_cast<const float4*>(res + row * N + n0 + (half + f) * 16 + 4 * fq);
      }
      f32x4 acc[FH];
#pragma unroll
      for (int f = 0; f < FH; ++f) {
        acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const unsigned char* wp = smem + ((half + f) * 16 + fr) * PITCH + ks * 64 + fq * 16;
          const uint4 w0 = *reinterpret_cast<const uint4*>(wp);
          const uint4 w1 = *reinterpret_cast<const uint4*>(wp + NC * PITCH);
          MH(w1, x0[ks], acc[f]); MH(w0, x1[ks], acc[f]); MH(w0, x0[ks], acc[f]);
        }
        __builtin_amdgcn_sched_barrier(0);        // keep the fragment reads of later fragments from being hoisted (register pressure)
      }
#pragma unroll
      for (int f = 0; f < FH; ++f) {
        const int n = n0 + (half + f) * 16 + 4 * fq;
        const float4 s4 = *reinterpret_cast<const float4*>(sc + n), b4 = *reinterpret_cast<const float4*>(bi + n);
        float4 v = make_float4(acc[f][0] * inv * s4.x + b4.x, acc[f][1] * inv * s4.y + b4.y, acc[f][2] * inv * s4.z + b4.z,
                               acc[f][3] * inv * s4.w + b4.w);
        if (res) { v.x += rs[f].x; v.y += rs[f].y; v.z += rs[f].z; v.w += rs[f].w; }
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        if (ok) {
          *reinterpret_cast<float4*>(out + row * N + n) = v;
          mask8[row * (N / 4) + (n >> 2)] = (unsigned char)((v.x > 0.f) | ((v.y > 0.f) << 1) | ((v.z > 0.f) << 2) | ((v.w > 0.f) << 3));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

static float pow2_scale(const std::vector<float>& v) {
  float mx = 0.f;
  for (float x : v) mx = std::fmax(mx, std::fabs(x));
  int e;
  std::frexp(mx, &e);
  return std::ldexp(1.f, 15 - e);
}

template <int K, int NC, int WAVES, int FHMAX = 8>
static void run(int M, int N, bool with_res, int wgs) {
  std::vector<float> hX((size_t)M * K), hW((size_t)N * K), hR((size_t)M * N), hs(N), hb(N);
  std::mt19937 g(1);
  std::uniform_real_distribution<float> ud(-1.f, 1.f);
  for (auto& v : hX) v = ud(g) > 0 ? ud(g) : 0.f;          // ReLU-like input
  for (auto& v : hW) v = ud(g) * 0.1f;
  for (auto& v : hR) v = ud(g);
  for (auto& v : hs) v = 1.f + 0.2f * ud(g);
  for (auto& v : hb) v = 0.05f * ud(g);
  float *dX, *dW, *dR, *dO, *ds, *db;
  unsigned char* dM;
  CK(hipMalloc(&dX, hX.size() * 4)); CK(hipMalloc(&dW, hW.size() * 4)); CK(hipMalloc(&dR, hR.size() * 4));
  CK(hipMalloc(&dO, hR.size() * 4)); CK(hipMalloc(&ds, N * 4)); CK(hipMalloc(&db, N * 4)); CK(hipMalloc(&dM, (size_t)M * N / 4));
  CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dR, hR.data(), hR.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(ds, hs.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice));
  const float sx = pow2_scale(hX), sw = pow2_scale(hW);
  constexpr int lds = 2 * NC * (K * 2 + 16);
  auto kern = skinny_fwd_kernel<K, NC, WAVES, FHMAX>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const dim3 grid(wgs, N / NC), block(WAVES * 64);
  auto launch = [&] { hipLaunchKernelGGL(kern, grid, block, lds, 0, dX, K, dW, ds, db, with_res ? dR : nullptr, dO, dM, M, N, sx, sw); };
  launch();
  CK(hipDeviceSynchronize()); CK(hipGetLastError());
  std::vector<float> hO((size_t)M * N);
  CK(hipMemcpy(hO.data(), dO, hO.size() * 4, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int m = 0; m < M; m += 211)
    for (int n = 0; n < N; n += 7) {
      double r = 0;
      for (int k = 0; k < K; ++k) r += (double)hX[(size_t)m * K + k] * hW[(size_t)n * K + k];
      r = r * hs[n] + hb[n] + (with_res ? hR[(size_t)m * N + n] : 0.0);
      r = r > 0 ? r : 0;
      worst = std::fmax(worst, std::fabs(hO[(size_t)m * N + n] - r));
    }
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipEventRecord(a));
  for (int i = 0; i < 20; ++i) launch();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
  const double bytes = 4.0 * ((double)M * K * (N / NC) + (double)M * N * (with_res ? 2 : 1)) + (double)M * N / 4;
  printf("  M %6d K %3d N %4d NC %3d waves %2d wgs %4d res %d : %7.1f us  %5.2f TB/s (X counted per column range)  max abs err %.2e\n", M, K, N, NC, WAVES,
         wgs, (int)with_res, 1e3 * ms, bytes / (ms * 1e-3) / 1e12, worst);
  CK(hipFree(dX)); CK(hipFree(dW)); CK(hipFree(dR)); CK(hipFree(dO)); CK(hipFree(ds)); CK(hipFree(db)); CK(hipFree(dM));
}

int main() {
  // layer1 conv3 (64 -> 256, M = 3 x 120 x 214): engine 60 us with the residual in the step, 43 us without in isolation
  for (int wgs : {256, 512}) {
    run<64, 256, 8>(77040, 256, true, wgs);
    run<64, 256, 8>(77040, 256, false, wgs);
    run<64, 256, 16, 4>(77040, 256, true, wgs);
    run<64, 128, 8>(77040, 256, true, wgs);
  }
  // layer2 conv3 (128 -> 512, M = 19260): engine 40-46 us with the residual
  for (int wgs : {128, 256}) {
    run<128, 256, 8>(19260, 512, true, wgs);
    run<128, 128, 8>(19260, 512, true, wgs);
  }
  // layer3 conv3 (256 -> 1024, M = 4860): engine 29 us
  run<256, 128, 8, 4>(4860, 1024, true, 32);
  // layer1 conv1 of blocks 1, 2 (256 -> 64): engine 32-37 us
  run<256, 64, 8, 4>(77040, 64, false, 256);
  return 0;
}
