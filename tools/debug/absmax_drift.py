"""How stale may an f16x3 operand scale be?  (VERDICT r04 #4a: producers that write pre-split fp16 pairs need the CONSUMER's scale
before the tensor exists, i.e. the previous iteration's absmax plus a safety exponent.)  Runs the 50 fine-tune iterations of
fixture G20's batch sequence (480x854, batch 3) and records, for every named activation / gradient tensor, the binary exponent of
its absmax after each iteration: the largest jump between consecutive iterations is the safety margin such a scheme needs.

    python tools/debug/absmax_drift.py [iterations]
"""
import math
import os
import sys

os.environ.setdefault('EOSVOS_MODE_GUARD', '0')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from eosvos_amd import synthetic  # noqa: E402
from eosvos_amd.engine import Engine  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 50
H, W, B = 480, 854, 3
eng = Engine('resnet50', H, W, max_batch=B)
eng.load_model_state(synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50'))
names = ['c1', 'p1', 'cat', 'proj', 'dcat', 'd1', 'd2', 'g_c1', 'g_p1', 'g_cat', 'g_proj', 'g_dcat', 'g_d1', 'g_d2']
for i in range(16):
    names += [f'blk{i}.t1', f'blk{i}.t2', f'blk{i}.out', f'blk{i}.g_t1', f'blk{i}.g_t2', f'blk{i}.g_out']
hist = {n: [] for n in names}
for it in range(T):
    x, y = synthetic.synthetic_frames(B, H, W, seed=21 + it)
    eng.finetune_step(x.cuda(), y.cuda())
    for n in names:
        a = float(eng.debug_tensor(n)[:B].abs().max())
        hist[n].append(math.frexp(a)[1] if a > 0 else None)
worst = {}
for n, e in hist.items():
    d = [abs(e[i + 1] - e[i]) for i in range(len(e) - 1) if e[i] is not None and e[i + 1] is not None]
    worst[n] = (max(d) if d else 0, max(v for v in e if v is not None) - min(v for v in e if v is not None))
act = {n: w for n, w in worst.items() if '.g_' not in n and not n.startswith('g_')}
grd = {n: w for n, w in worst.items() if n not in act}
for title, grp in (('activations', act), ('gradients', grd)):
    print(f'{title}: {len(grp)} tensors, largest exponent jump between consecutive iterations {max(w[0] for w in grp.values())}, '
          f'largest range over {T} iterations {max(w[1] for w in grp.values())}; tensors with a jump >= 2: '
          f'{[(n, w) for n, w in grp.items() if w[0] >= 2]}')
print('exponent history of the three widest-ranging gradient tensors:')
for n, w in sorted(grd.items(), key=lambda kv: -kv[1][1])[:3]:
    print(' ', n, hist[n])
eng.close()
