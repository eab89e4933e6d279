"""ms per frame of eosvos_infer (forward + sigmoid) at batch 1..3, 480x854."""
import sys, time, torch
sys.path.insert(0, '.')
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine
MB = int(sys.argv[1]) if len(sys.argv) > 1 else 3
e = Engine('resnet50', 480, 854, max_batch=MB)
e.load_model_state(synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50'))
x, y = synthetic.synthetic_frames(MB, 480, 854)
xg = x.cuda()
for b in range(1, MB + 1):
    xb = xg[:b].contiguous()
    for _ in range(5): e.infer(xb)
    e.synchronize(); t0 = time.perf_counter()
    for _ in range(30): e.infer(xb)
    e.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f'batch {b}: {1e3 * dt:.2f} ms per call, {1e3 * dt / b:.2f} ms per frame')

x3, y3 = xg[:3].contiguous(), y.cuda()[:3].contiguous()
for _ in range(3): e.finetune_step(x3, y3, sync_loss=False)
e.synchronize(); t0 = time.perf_counter()
for _ in range(20): e.finetune_step(x3, y3, sync_loss=False)
e.synchronize(); print(f'batch-3 fine-tune iteration on this engine (max_batch {MB}): {(time.perf_counter() - t0) * 50:.2f} ms')
