"""Total fine-tune iterations/s of N engines side by side (one stream each), by workgroup budget.
usage: [EOSVOS_NO_SIDE_STREAM=1] python tools/inflight_sweep.py B N1,N2,.. budget1,budget2,.. [steps]"""
import sys
import time

import torch

sys.path.insert(0, '.')
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine

B = int(sys.argv[1])
Ns = [int(v) for v in sys.argv[2].split(',')]
budgets = [int(v) for v in sys.argv[3].split(',')]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
sd, lrs = synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50')
x, y = synthetic.synthetic_frames(B, 480, 854)
xg, yg = x.cuda(), y.cuda()
for N in Ns:
    engs = []
    for _ in range(N):
        with torch.cuda.stream(torch.cuda.Stream()):
            e = Engine('resnet50', 480, 854, max_batch=B)
            e.load_model_state(sd, lrs)
        engs.append(e)
    for budget in budgets:
        for e in engs:
            e.set_wg_budget(budget)

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
        dt = time.perf_counter() - t0
        print(f'batch {B}: {N} engines, wg_budget {budget}: {N * steps / dt:.1f} it/s', flush=True)
    for e in engs:
        e.close()
