// Stand-alone timing of sgd_update_all_kernel on a synthetic table (tuning aid, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -I e-osvos_amd/csrc tools/probes/upd_probe.cpp e-osvos_amd/csrc/misc_kernels.o -o /tmp/upd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "kernels.h"
using namespace eosvos;
int main(int argc, char** argv) {
  const int pad = argc > 1 ? atoi(argv[1]) : 0;       // extra floats between slabs
  const int mode = argc > 2 ? atoi(argv[2]) : 0;      // 0: update, 1: no update (lr null)
  struct L { int rows, rowlen, splits; };
  std::vector<L> layers = {{512, 4608, 7}, {2048, 512, 8}, {2048, 1024, 4}, {512, 2048, 8}, {512, 4608, 7},
                           {2048, 512, 8}, {512, 2048, 8}, {512, 4608, 7}, {2048, 512, 8}, {256, 2048, 15},
                           {256, 18432, 5}, {256, 18432, 5}, {256, 18432, 5}, {256, 1280, 24}, {256, 2736, 11},
                           {256, 2304, 14}};
  std::vector<UpdEntry> tab;
  long woff = 0, wsoff = 0; int blk = 0, lroff = 0;
  for (auto& l : layers) {
    UpdEntry u; u.w_off = woff; u.ws_off = wsoff; u.n = l.rows * l.rowlen; u.slab = u.n + pad; u.splits = l.splits;
    u.rowlen = l.rowlen; u.lr_off = lroff; u.norm_off = lroff; u.blk0 = blk;
    blk += (u.n + 1024 * UPD_CHUNKS - 1) / (1024 * UPD_CHUNKS);
    woff += u.n; wsoff += (long)u.slab * l.splits; lroff += l.rows; tab.push_back(u);
  }
  float *W, *ws, *na, *lr; UpdEntry* dt;
  hipMalloc(&W, woff * 4); hipMalloc(&ws, wsoff * 4); hipMalloc(&na, lroff * 4); hipMalloc(&lr, lroff * 4);
  hipMalloc(&dt, tab.size() * sizeof(UpdEntry));
  hipMemset(W, 0, woff * 4); hipMemset(ws, 0, wsoff * 4); hipMemset(na, 0, lroff * 4); hipMemset(lr, 0, lroff * 4);
  hipMemcpy(dt, tab.data(), tab.size() * sizeof(UpdEntry), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) launch_sgd_update_all(dt, (int)tab.size(), blk, W, ws, na, mode ? nullptr : lr, nullptr, nullptr, nullptr, 0);
  hipEventRecord(e0, 0);
  for (int it = 0; it < 10; ++it) launch_sgd_update_all(dt, (int)tab.size(), blk, W, ws, na, mode ? nullptr : lr, nullptr, nullptr, nullptr, 0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)wsoff * 4 + (mode ? 1.0 : 2.0) * woff * 4;
  printf("pad %d mode %d: %d WGs, %.1f us per launch, %.2f TB/s (%.0f MB)\n", pad, mode, blk, ms * 100, bytes / (ms / 10 * 1e-3) / 1e12, bytes / 1e6);
  return 0;
}
