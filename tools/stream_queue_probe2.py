"""Does the rate of N single-queue engines on consecutive new torch streams depend on what the process did before?
Rounds of: build N engines (no side stream) on new streams, time them, close them; optionally one null-stream engine WITH a
side stream is built (and closed) before each round, as an evaluation of the next model would.
usage: python tools/stream_queue_probe2.py N rounds with_null_engine(0/1)"""
import sys
import time

import torch

sys.path.insert(0, '.')
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine

N, R, with_null = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == '1'
B, steps = 3, 20
sd, lrs = synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50')
x, y = synthetic.synthetic_frames(B, 480, 854)
xg, yg = x.cuda(), y.cuda()
for r in range(R):
    null_eng = None
    if with_null:
        null_eng = Engine('resnet50', 480, 854, max_batch=B)
        null_eng.load_model_state(sd, lrs)
        null_eng.finetune_step(xg, yg)
    engs = []
    for i in range(N):
        with torch.cuda.stream(torch.cuda.Stream()):
            e = Engine('resnet50', 480, 854, max_batch=B, side_stream=False)
            e.load_model_state(sd, lrs)
            e.set_wg_budget(256)
        engs.append(e)

    def rounds(n):
        for _ in range(n):
            for e in engs:
                with torch.cuda.stream(e.stream):
                    e.finetune_step(xg, yg, sync_loss=False)
    rounds(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rounds(steps)
    torch.cuda.synchronize()
    print(f'round {r}: {N} engines (batch {B}), null-stream engine with side stream alive: {with_null}: {N * steps / (time.perf_counter() - t0):.1f} it/s', flush=True)
    for e in engs:
        e.close()
    if null_eng is not None:
        null_eng.close()
