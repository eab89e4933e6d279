"""Wall time of the forward (+ loss) and of the backward + update of one batch-B iteration, separately (each phase bracketed by a
device synchronize), with and without the side stream:   python tools/debug/phase_time.py [B]"""
import os
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
for _ in range(3):
    eng.finetune_step(xg, yg, sync_loss=False)
eng.synchronize()
for side in (1, 0, 1, 0):
    eng.set_side_stream(bool(side)) if hasattr(eng, 'set_side_stream') else None
    for _ in range(3):
        eng.finetune_step(xg, yg, sync_loss=False)
    eng.synchronize()
    tf = tb = 0.0
    n = 30
    for _ in range(n):
        t0 = time.perf_counter()
        eng.forward(xg, want_logits=False)
        eng.loss_bce(yg)
        eng.synchronize()
        t1 = time.perf_counter()
        eng.backward_step()
        eng.synchronize()
        t2 = time.perf_counter()
        tf += t1 - t0
        tb += t2 - t1
    t0 = time.perf_counter()
    for _ in range(n):
        eng.finetune_step(xg, yg, sync_loss=False)
    eng.synchronize()
    ts = (time.perf_counter() - t0) / n
    print(f'B {B} side stream {side}: forward+loss {tf / n * 1e3:.2f} ms  backward+update {tb / n * 1e3:.2f} ms  sum {(tf + tb) / n * 1e3:.2f}  fused step {ts * 1e3:.2f} ms')
eng.close()
