import sys, torch
sys.path.insert(0, '.')
from eosvos_amd import synthetic, engine as em
from eosvos_amd.engine import Engine
from oracle import deeplab
H, W = 96, 160
x, y = synthetic.synthetic_frames(1, H, W, seed=3)
state = {k: v.clone() for k, v in synthetic.synthetic_state().items()}
f = 2.0 ** 40
state['backbone.layer2.1.bn1.weight'][5] *= f
state['backbone.layer2.1.bn1.bias'][5] *= f
state['backbone.layer2.1.conv2.weight'][:, 5] /= f
ref = deeplab.forward(state, x)
outs = {}
for mode in ('f16x3', 'bf16x6', 'f32'):
    em.set_matrix_mode(mode)
    import os; os.environ['EOSVOS_MODE_GUARD'] = '0'
    e = Engine('resnet50', H, W, max_batch=1)
    e.load_model_state(state, synthetic.synthetic_lrs())
    o = e.forward(x.cuda()).cpu()
    t1 = e.debug_tensor('blk4.t1').cpu()
    t2 = e.debug_tensor('blk4.t2').cpu()
    print(mode, 'logit diff vs oracle', float((o - ref).abs().max()), 't1 ch5 max', float(t1[0, 5].max()), 'other max', float(t1[0, :5].max()), 't2 max', float(t2.abs().max()), flush=True)
    outs[mode] = (o, t2)
    e.close()
print('t2 diff f16x3 vs bf16x6', float((outs['f16x3'][1] - outs['bf16x6'][1]).abs().max()))
