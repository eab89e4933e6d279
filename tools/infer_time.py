"""ms per frame of eosvos_infer (forward + sigmoid) at batch 1..3, 480x854."""
import sys, time, torch
sys.path.insert(0, '.')
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine
e = Engine('resnet50', 480, 854, max_batch=3)
e.load_model_state(synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50'))
x, _ = synthetic.synthetic_frames(3, 480, 854)
xg = x.cuda()
for b in (1, 2, 3):
    xb = xg[:b].contiguous()
    for _ in range(5): e.infer(xb)
    e.synchronize(); t0 = time.perf_counter()
    for _ in range(30): e.infer(xb)
    e.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f'batch {b}: {1e3 * dt:.2f} ms per call, {1e3 * dt / b:.2f} ms per frame')
