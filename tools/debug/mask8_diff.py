"""Which gradient tensors differ between the byte-mask and the fp32-mask data gradients (must be none)."""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from eosvos_amd import synthetic, topology
from eosvos_amd.engine import Engine
H, W, B = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (96, 160, 3)))
sd, lrs = synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50')
x, y = synthetic.synthetic_frames(B, H, W, seed=7)
tr = topology.trainable('resnet50')
offs = np.cumsum([0] + [int(np.prod(s)) for _, s in tr])
g = []
for env in (None, '1'):
    if env: os.environ['EOSVOS_TUNE_NO_MASK8'] = env
    else: os.environ.pop('EOSVOS_TUNE_NO_MASK8', None)
    e = Engine('resnet50', H, W, max_batch=B, side_stream=False)
    e.load_model_state(sd, lrs)
    e.keep_grads(True)
    e.finetune_step(x.cuda(), y.cuda())
    g.append(e.get_grads().cpu())
    names = ['g_p1', 'g_c1'] + [f'blk{i}.{k}' for i in range(16) for k in ('g_out', 'g_t2', 'g_t1')] + ['g_cat', 'g_proj', 'g_dcat', 'g_d1']
    g.append({n: e.debug_tensor(n).cpu() for n in names})
    e.close()
for i, (n, s) in enumerate(tr):
    d = float((g[0][offs[i]:offs[i+1]] - g[2][offs[i]:offs[i+1]]).abs().max())
    if d > 0: print('grad', n, d, float(g[2][offs[i]:offs[i+1]].abs().max()))
for n in g[1]:
    d = float((g[1][n] - g[3][n]).abs().max())
    if d > 0: print('tensor', n, d, float(g[3][n].abs().max()), 'differing elements', int((g[1][n] != g[3][n]).sum()), 'of', g[1][n].numel())
print('done')
