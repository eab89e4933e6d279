"""Step time with and without Engine.autotune (per-launch workgroup budgets).  python tools/autotune_ab.py"""
import sys, time, torch
sys.path.insert(0, '.')
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine
for B in (1, 3):
    eng = Engine('resnet50', 480, 854, max_batch=B)
    eng.load_model_state(synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50'))
    x, y = synthetic.synthetic_frames(B, 480, 854); xg, yg = x.cuda(), y.cuda()
    def t():
        for _ in range(5): eng.finetune_step(xg, yg, sync_loss=False)
        eng.synchronize(); t0 = time.perf_counter()
        for _ in range(50): eng.finetune_step(xg, yg, sync_loss=False)
        eng.synchronize(); return (time.perf_counter() - t0) * 20
    a = t()
    t0 = time.perf_counter(); ch = eng.autotune(B); dt = time.perf_counter() - t0
    eng.reset()
    b = t()
    print('B', B, 'ms/step %.3f -> %.3f with %d overrides (tuning took %.1f s)' % (a, b, len(ch), dt), flush=True)
    print('   ', sorted(ch.items()))
    eng.close()
