"""Do CU-partitioned streams (hipExtStreamCreateWithCUMask) beat free sharing for engines that run side by side?
N engines at batch B, each fine-tuning on its own stream; total iterations/s with
  (a) ordinary streams, every engine planning for wg_budget workgroups,
  (b) one CU partition per engine (bits [k*256/N, (k+1)*256/N) = the same CU rows on every XCD), planning for 2 x its CUs.
usage: python tools/cu_partition_probe.py N B [steps]"""
import ctypes
import sys
import time

import torch

sys.path.insert(0, '.')
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine

N, B = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
hip = ctypes.CDLL('libamdhip64.so')
sd, lrs = synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50')
x, y = synthetic.synthetic_frames(B, 480, 854)
xg, yg = x.cuda(), y.cuda()


def masked_stream(lo, hi):
    words = (ctypes.c_uint32 * 8)()
    for b in range(lo, hi):
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def run(streams, budget):
    engs = []
    for st in streams:
        with torch.cuda.stream(st):
            e = Engine('resnet50', 480, 854, max_batch=B)
            e.load_model_state(sd, lrs)
            e.set_wg_budget(budget)
        engs.append(e)
    torch.cuda.synchronize()

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
    for e in engs:
        e.close()
    return len(engs) * steps / dt


for budget in (0, 256, 128):
    print(f'{N} engines, batch {B}, ordinary streams, wg_budget {budget}: {run([torch.cuda.Stream() for _ in range(N)], budget):.1f} it/s', flush=True)
per = 256 // N
for budget in (2 * per, 4 * per if 4 * per <= 512 else 512):
    budget = max(64, budget // 64 * 64)
    print(f'{N} engines, batch {B}, one {per}-CU partition each, wg_budget {budget}: '
          f'{run([masked_stream(k * per, (k + 1) * per) for k in range(N)], budget):.1f} it/s', flush=True)
