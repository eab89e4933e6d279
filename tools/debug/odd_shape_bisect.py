"""Bisect a matrix-mode discrepancy on one frame size: every named internal tensor of the forward pass in f16x3 / bf16x6 / f32
(range guard OFF) against each other.

    EOSVOS_MODE_GUARD=0 python tools/debug/odd_shape_bisect.py [H W [B]]

Round-4 verdict item 1: 1 x 97 x 163 tripped the guard (f16x3 vs bf16x6 logits 3.2e-3) on the benign synthetic state.
"""
import os
import sys

os.environ['EOSVOS_MODE_GUARD'] = '0'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from eosvos_amd import engine as em, synthetic, topology  # noqa: E402
from eosvos_amd.engine import Engine  # noqa: E402

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (97, 163)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
SEED = int(os.environ.get('BISECT_SEED', '5'))
sd = synthetic.synthetic_state('resnet50')
lrs = synthetic.synthetic_lrs('resnet50')
x, y = synthetic.synthetic_frames(B, H, W, seed=SEED)
nblk = 16
names = ['c1', 'p1'] + [f'blk{i}.{k}' for i in range(nblk) for k in ('t1', 't2', 'out')] + ['cat', 'proj', 'dcat', 'd1', 'd2', 'lowlog', 'logits']
res = {}
for mode in ('f16x3', 'bf16x6', 'f32'):
    em.set_matrix_mode(mode)
    e = Engine('resnet50', H, W, max_batch=B, device='cuda:0')
    e.load_model_state(sd, lrs)
    out = e.forward(x.cuda())
    d = {}
    for n in names:
        try:
            d[n] = e.debug_tensor(n).cpu()[:B]
        except Exception as ex:      # noqa: BLE001
            d[n] = None
    d['out'] = out.cpu()
    # one fine-tune step: loss + gradient norm (backward-side discrepancy)
    e.keep_grads(True)
    d['loss'] = e.finetune_step(x.cuda(), y.cuda())
    d['grads'] = e.get_grads().cpu()
    for n in ('g_c1', 'g_p1', 'g_dcat', 'g_cat'):
        d[n] = e.debug_tensor(n).cpu()[:B]
    res[mode] = d
    e.close()
em.set_matrix_mode('f16x3')
print(f'shape {B}x{H}x{W} seed {SEED}; guard log {em.GUARD_LOG}')
print('%-12s %12s %12s %12s   (max |a-b| / max|b|)' % ('tensor', 'h3-x6', 'h3-f32', 'x6-f32'))
for n in names + ['out', 'g_dcat', 'g_cat', 'g_p1', 'g_c1', 'grads']:
    a, b, c = res['f16x3'][n], res['bf16x6'][n], res['f32'][n]
    if a is None:
        continue
    s = float(c.abs().max()) + 1e-30
    print('%-12s %12.3e %12.3e %12.3e   scale %.3e shape %s' % (n, float((a - b).abs().max()) / s, float((a - c).abs().max()) / s,
                                                              float((b - c).abs().max()) / s, s, tuple(a.shape)))
print('loss', {m: res[m]['loss'] for m in res})
# where is the worst c1 / p1 element?
for n in ('c1', 'p1'):
    a, b = res['f16x3'][n], res['f32'][n]
    dd = (a - b).abs()
    i = int(dd.argmax())
    idx = [int(v) for v in torch.unravel_index(torch.tensor(i), dd.shape)]
    print(n, 'worst at (b,c,y,x) =', idx, 'h3 %.6f f32 %.6f' % (float(a.flatten()[i]), float(b.flatten()[i])),
          'count >1e-5:', int((dd > 1e-5).sum()))
