"""The range guard's own sequence on ONE engine: forward in f16x3, then bf16x6, then f32, then f16x3 again -- every named
tensor of each pass against a FRESH engine's pass in the same mode (round-4 verdict item 1: the guard tripped at 1x97x163
with 3.2e-3 although fresh engines per mode agree to 1e-6)."""
import os
import sys

os.environ['EOSVOS_MODE_GUARD'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from eosvos_amd import engine as em, synthetic  # noqa: E402
from eosvos_amd.engine import Engine  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (97, 163)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
PRE = os.environ.get('REPRO_PRE', '')          # e.g. "2x480x854": build + run + close such an engine first (process history)
sd = synthetic.synthetic_state('resnet50')
lrs = synthetic.synthetic_lrs('resnet50')
x, y = synthetic.synthetic_frames(B, H, W, seed=5)
names = ['c1', 'p1'] + [f'blk{i}.{k}' for i in range(16) for k in ('t1', 't2', 'out')] + ['cat', 'proj', 'dcat', 'd1', 'd2', 'lowlog', 'logits']


def grab(e):
    return {n: e.debug_tensor(n).cpu()[:B] for n in names}


if PRE:
    for mode in ('f16x3', 'bf16x6'):
        em.set_matrix_mode(mode)
        b, h, w = [int(v) for v in PRE.split('x')]
        e = Engine('resnet50', h, w, max_batch=b, device='cuda:0')
        e.load_model_state(sd, lrs)
        e.forward(synthetic.synthetic_frames(b, h, w, seed=5)[0].cuda(), want_logits=False)
        e.close()
fresh = {}
for mode in ('f16x3', 'bf16x6', 'f32'):
    em.set_matrix_mode(mode)
    e = Engine('resnet50', H, W, max_batch=B, device='cuda:0')
    e.load_model_state(sd, lrs)
    e.forward(x.cuda())
    fresh[mode] = grab(e)
    e.close()
em.set_matrix_mode('f16x3')
e = Engine('resnet50', H, W, max_batch=B, device='cuda:0')
e.load_model_state(sd, lrs)
for k, mode in enumerate(('f16x3', 'bf16x6', 'f32', 'f16x3', 'bf16x6')):
    em.set_matrix_mode(mode)
    e.forward(x.cuda())
    got = grab(e)
    bad = []
    for n in names:
        s = float(fresh[mode][n].abs().max()) + 1e-30
        d = float((got[n] - fresh[mode][n]).abs().max()) / s
        if d > 0:
            bad.append((n, d))
    print(f'pass {k} ({mode}) on the live engine vs a fresh engine: {len(bad)} tensors differ; first: {bad[:6]}; logits {dict(bad).get("logits", 0.0):.3e}', flush=True)
e.close()
em.set_matrix_mode('f16x3')
