// Probe: which CUs / XCDs does a CU-masked stream (hipExtStreamCreateWithCUMask) run workgroups on?  For each mask a grid
// of 2048 workgroups records HW_REG_XCC_ID and HW_REG_HW_ID; the histogram tells how mask bits map to XCDs.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/cu_mask_probe.cpp -o tools/probes/bin/cu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void who(unsigned* out, int spin) {
  unsigned xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;       // keep the workgroup resident for a while so the grid spreads
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = hw; }
  if (a == 12345.f) out[0] = 0;
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t s;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data());
  if (e != hipSuccess) { printf("%-28s create failed: %s\n", name, hipGetErrorString(e)); return; }
  const int n = 2048;
  unsigned* d;
  CK(hipMalloc(&d, n * 8));
  hipLaunchKernelGGL(who, dim3(n), dim3(256), 0, s, d, 20000);
  CK(hipStreamSynchronize(s));
  std::vector<unsigned> h(2 * n);
  CK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
  std::map<unsigned, std::set<unsigned>> cus;
  for (int i = 0; i < n; ++i) cus[h[2 * i]].insert((h[2 * i + 1] >> 8) & 0xffu);     // cu_id[11:8], sh_id[12], se_id[15:13]
  printf("%-28s XCDs used:", name);
  int tot = 0;
  for (auto& kv : cus) { printf(" %u(%zu CUs)", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
  printf("  total %d distinct (xcc, cu) pairs\n", tot);
  CK(hipFree(d));
  CK(hipStreamDestroy(s));
}

int main() {
  auto bits = [](std::initializer_list<std::pair<int, int>> ranges) {
    std::vector<uint32_t> m(8, 0);
    for (auto r : ranges) for (int b = r.first; b < r.second; ++b) m[b / 32] |= 1u << (b % 32);
    return m;
  };
  run("all 256", bits({{0, 256}}));
  run("bits 0..63", bits({{0, 64}}));
  run("bits 64..127", bits({{64, 128}}));
  run("bits 0..31", bits({{0, 32}}));
  run("bits 0..7", bits({{0, 8}}));
  { std::vector<uint32_t> m(8, 0); for (int b = 0; b < 256; b += 8) m[b / 32] |= 1u << (b % 32); run("every 8th bit (b%8==0)", m); }
  { std::vector<uint32_t> m(8, 0); for (int b = 0; b < 256; ++b) if (b % 8 < 2) m[b / 32] |= 1u << (b % 32); run("b%8 in {0,1}", m); }
  { std::vector<uint32_t> m(8, 0); for (int b = 0; b < 256; ++b) if (b % 4 == 0) m[b / 32] |= 1u << (b % 32); run("b%4 == 0", m); }
  return 0;
}
