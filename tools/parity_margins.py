"""Prints how far the full-size (480x854) results are from the reference fixtures (margins of the parity tests):
forward logits, mask bits, and the 10-step C1 fine-tune trajectory.  EOSVOS_LIB selects the build."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
g2 = np.load(os.path.join(G, 'g2_forward.npz')); g45 = np.load(os.path.join(G, 'g45_finetune.npz'))
eng = Engine('resnet50', 480, 854, max_batch=1)
eng.load_model_state(synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50'))
x, y = synthetic.synthetic_frames(1, 480, 854, seed=7)
xg, yg = x.cuda(), y.cuda()
out = eng.forward(xg).cpu()
print('forward: max |logit - ref| on the sampled grid %.3e (tol 1e-3)' % np.abs(out[0, 0, ::8, ::7].numpy() - g2['full_bn_logits_sub']).max())
mask = np.packbits((out >= 0).numpy().astype(np.uint8))
print('forward: differing mask bits %d (reference pixels within rounding of 0: %d)' % (int(np.unpackbits(mask ^ g2['full_bn_mask']).sum()), int(g2['full_bn_near_zero'][0])))
eng.reset()
losses = np.array([eng.finetune_step(xg, yg) for _ in range(10)])
rel = np.abs(losses - g45['c1_losses']) / np.abs(g45['c1_losses'])
print('C1 (T=10): max relative loss difference %.3e (tol 1e-3); per step' % rel.max(), np.array2string(rel, precision=1))
out = eng.forward(xg).cpu()
print('C1 final logits: max |diff| on the sampled grid %.3e (tol 2e-2)' % np.abs(out[0, 0, ::8, ::7].numpy() - g45['c1_final_logits_sub']).max())
mask = np.packbits((out >= 0).numpy().astype(np.uint8))
print('C1 final mask: differing bits %d' % int(np.unpackbits(mask ^ g45['c1_final_mask']).sum()))
