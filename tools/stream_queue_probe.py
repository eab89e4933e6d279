"""How does the rate of N single-queue engines depend on the streams created before theirs?  (ROCm deals HIP streams onto
GPU_MAX_HW_QUEUES hardware queues: the first ones get a queue each, later ones the least-referenced queue.)
usage: python tools/stream_queue_probe.py N D [use_null]   -- D dummy streams first, then N engine streams"""
import sys
import time

import torch

sys.path.insert(0, '.')
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine

N, D = int(sys.argv[1]), int(sys.argv[2])
use_null = len(sys.argv) > 3 and sys.argv[3] == '1'
B, steps = 1, 30
sd, lrs = synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50')
x, y = synthetic.synthetic_frames(B, 480, 854)
xg, yg = x.cuda(), y.cuda()
dummies = [torch.cuda.Stream() for _ in range(D)]
for d in dummies:
    with torch.cuda.stream(d):
        torch.zeros(1, device='cuda')          # make sure the stream really exists on the device
engs = []
for i in range(N):
    st = torch.cuda.current_stream() if (use_null and i == 0) else torch.cuda.Stream()
    with torch.cuda.stream(st):
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
print(f'N={N} engines after D={D} dummy streams, first engine on the null stream: {use_null}: {N * steps / (time.perf_counter() - t0):.1f} it/s', flush=True)
