"""Per-layer time of every conv launch (fwd / dgrad / wgrad, fix-ups and Winograd transforms included, one stream) under
different workgroup budgets: how much would a per-layer choice gain over the fixed whole-chip plan?
    python tools/budget_sweep.py [batch]"""
import sys, torch
sys.path.insert(0, '.')
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
e = Engine('resnet50', 480, 854, max_batch=B)
e.load_model_state(synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50'))
x, y = synthetic.synthetic_frames(B, 480, 854)
e.finetune_step(x.cuda(), y.cuda())
budgets = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 448, 384, 320, 256]
from eosvos_amd.topology import conv_infos
n = len(conv_infos('resnet50'))
tot = {b: [0.0, 0.0, 0.0] for b in budgets}
best = [0.0, 0.0, 0.0]
rows = []
for ci in range(1, n):
    for kind in (0, 1, 2):
        ms = {}
        for b in budgets:
            e.set_wg_budget(b)
            try:
                ms[b] = e.bench_conv(ci, kind, B, reps=10)[0] * 1e3
            except Exception as exc:
                if not rows and not any(tot[0]):
                    print('skipping conv', ci, kind, '-', exc)
                ms = None
                break
        if ms is None:
            continue
        for b in budgets:
            tot[b][kind] += ms[b]
        bb = min(budgets, key=lambda b: ms[b])
        best[kind] += ms[bb]
        if ms[bb] < 0.93 * ms[0]:
            rows.append((ci, 'fwd dgrad wgrad'.split()[kind], {b: round(v, 1) for b, v in ms.items()}))
print(f'batch {B}: summed us per pass, by budget (0 = 512 workgroups)')
for b in budgets:
    print(f'  budget {b or 512:4d}: fwd {tot[b][0]:8.1f}  dgrad {tot[b][1]:8.1f}  wgrad {tot[b][2]:8.1f}  total {sum(tot[b]):8.1f}')
print(f'  best per layer: fwd {best[0]:8.1f}  dgrad {best[1]:8.1f}  wgrad {best[2]:8.1f}  total {sum(best):8.1f}  ({100 * (1 - sum(best) / sum(tot[0])):.1f} % below the fixed 512)')
print('layers where another budget is >= 7 % faster:')
for r in rows:
    print('  conv', r[0], r[1], r[2])
