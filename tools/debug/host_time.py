"""Host time of eosvos_finetune_step calls (no synchronisation inside the loop) against the device time of the same steps:
is the iteration launch-bound anywhere?   python tools/debug/host_time.py [B]"""
import sys
import time
import torch
sys.path.insert(0, '.')
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
eng = Engine('resnet50', 480, 854, max_batch=B)
eng.load_model_state(synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50'))
x, y = synthetic.synthetic_frames(B, 480, 854)
xg, yg = x.cuda(), y.cuda()
for _ in range(5):
    eng.finetune_step(xg, yg, sync_loss=False)
eng.synchronize()
N = 50
host = []
t0 = time.perf_counter()
for _ in range(N):
    a = time.perf_counter()
    eng.finetune_step(xg, yg, sync_loss=False)
    host.append(time.perf_counter() - a)
t1 = time.perf_counter()
eng.synchronize()
t2 = time.perf_counter()
host.sort()
print(f'B {B}: host per call median {host[N // 2] * 1e3:.2f} ms (min {host[0] * 1e3:.2f}, max {host[-1] * 1e3:.2f}); loop {1e3 * (t1 - t0) / N:.2f} ms/step, '
      f'after the final synchronize {1e3 * (t2 - t0) / N:.2f} ms/step; queue drained in {1e3 * (t2 - t1):.2f} ms')
eng.close()
