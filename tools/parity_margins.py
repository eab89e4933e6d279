"""Prints how far the GPU results are from the reference-generated fixtures (the margins the parity tests' tolerances
are derived from, ~10x these): full-size forward, C1 (T=10, B=1), C2 (T=3, B=3: the benchmarked configuration), the
reduced small-size trajectories and a K=5 meta task -- in the default matrix mode (bf16x6) and with the fp32 MFMA."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eosvos_amd import synthetic, topology
from eosvos_amd import engine as engine_mod
from eosvos_amd.engine import Engine
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
g2 = np.load(os.path.join(G, 'g2_forward.npz')); g45 = np.load(os.path.join(G, 'g45_finetune.npz'))
g15 = np.load(os.path.join(G, 'g15_c2_full_b3.npz')); g7 = np.load(os.path.join(G, 'g7_meta_task.npz'))
tr = topology.trainable('resnet50')
offs = np.cumsum([0] + [int(np.prod(s)) for _, s in tr])
sd, lrs = synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50')


def bits(a, b):
    return int(np.unpackbits(a ^ b).sum())


PREV = engine_mod.get_matrix_mode()
for mode in ('f16x3', 'bf16x6', 'f32'):
    engine_mod.set_matrix_mode(mode)
    print('== matrix mode', mode)
    eng = Engine('resnet50', 480, 854, max_batch=3)
    eng.load_model_state(sd, lrs)
    x, y = synthetic.synthetic_frames(1, 480, 854, seed=7)
    xg, yg = x.cuda(), y.cuda()
    out = eng.forward(xg).cpu()
    print('forward: max |logit - ref| on the sampled grid %.3e' % np.abs(out[0, 0, ::8, ::7].numpy() - g2['full_bn_logits_sub']).max())
    print('forward: differing mask bits %d (reference pixels within rounding of 0: %d)' % (bits(np.packbits((out >= 0).numpy().astype(np.uint8)), g2['full_bn_mask']), int(g2['full_bn_near_zero'][0])))
    eng.reset()
    losses = np.array([eng.finetune_step(xg, yg) for _ in range(10)])
    rel = np.abs(losses - g45['c1_losses']) / np.abs(g45['c1_losses'])
    print('C1 (T=10,B=1): max relative loss difference %.3e' % rel.max())
    out = eng.forward(xg).cpu()
    print('C1 final logits: max |diff| on the sampled grid %.3e; mask bits differing %d' % (
        np.abs(out[0, 0, ::8, ::7].numpy() - g45['c1_final_logits_sub']).max(), bits(np.packbits((out >= 0).numpy().astype(np.uint8)), g45['c1_final_mask'])))
    params = eng.get_params().cpu()
    print('C1 params: max relative L2-norm difference per tensor %.3e' % max(
        abs(float(params[offs[i]:offs[i + 1]].double().norm()) - g45['c1_param_fp'][i][1]) / g45['c1_param_fp'][i][1] for i in range(len(tr))))
    # C2 at full size, batch 3
    batches = [synthetic.synthetic_frames(3, 480, 854, seed=21 + it) for it in range(3)]
    eng.reset(); eng.keep_grads(True)
    losses = []
    for it, (xb, yb) in enumerate(batches):
        losses.append(eng.finetune_step(xb.cuda(), yb.cuda()))
        if it == 0:
            grads = eng.get_grads().cpu()
    eng.keep_grads(False)
    print('C2 (T=3,B=3): max relative loss difference %.3e' % (np.abs(np.array(losses) - g15['losses']) / np.abs(g15['losses'])).max())
    print('C2 first-step grads: max |diff|/max|ref| over the 5 full tensors %.3e; max relative L2 difference over all tensors %.3e' % (
        max(np.abs(grads[offs[i]:offs[i + 1]].view(*g15[f'grad_{i}'].shape).numpy() - g15[f'grad_{i}']).max() / np.abs(g15[f'grad_{i}']).max() for i in g15['small_ids']),
        max(abs(float(grads[offs[i]:offs[i + 1]].double().norm()) - g15['grad_fp'][i][1]) / (g15['grad_fp'][i][1] + 1e-12) for i in range(len(tr)))))
    params = eng.get_params().cpu()
    print('C2 params after T=3: max |diff|/max|ref| over the 5 full tensors %.3e' % max(
        np.abs(params[offs[i]:offs[i + 1]].view(*g15[f'param_{i}'].shape).numpy() - g15[f'param_{i}']).max() / np.abs(g15[f'param_{i}']).max() for i in g15['small_ids']))
    out = eng.forward(batches[0][0].cuda()).cpu()
    print('C2 final logits: max |diff| on the sampled grid %.3e; mask bits differing %d (near zero in ref: %d)' % (
        np.abs(out[:, 0, ::8, ::7].numpy() - g15['final_logits_sub']).max(), bits(np.packbits((out >= 0).numpy().astype(np.uint8)), g15['final_mask']), int(g15['final_near_zero'][0])))
    eng.close()
    # small size: reduced C2 + meta task K=5
    eng = Engine('resnet50', 96, 160, max_batch=3)
    eng.load_model_state(sd, lrs)
    sb = [synthetic.synthetic_frames(3, 96, 160, seed=7 + it) for it in range(5)]
    eng.reset(); eng.keep_grads(True)
    losses = []
    for it, (xb, yb) in enumerate(sb):
        losses.append(eng.finetune_step(xb.cuda(), yb.cuda()))
        if it == 0:
            grads = eng.get_grads().cpu()
    eng.keep_grads(False)
    print('small C2 (T=5,B=3): max relative loss difference %.3e; grads (5 full tensors) %.3e; grad L2 (all tensors) %.3e' % (
        (np.abs(np.array(losses) - g45['small_losses']) / np.abs(g45['small_losses'])).max(),
        max(np.abs(grads[offs[i]:offs[i + 1]].view(*g45[f'small_grad_{i}'].shape).numpy() - g45[f'small_grad_{i}']).max() / np.abs(g45[f'small_grad_{i}']).max() for i in g45['small_ids']),
        max(abs(float(grads[offs[i]:offs[i + 1]].double().norm()) - g45['small_grad_fp'][i][1]) / (g45['small_grad_fp'][i][1] + 1e-12) for i in range(len(tr)))))
    params = eng.get_params().cpu()
    print('small C2 params: %.3e; final logits max |diff| %.3e' % (
        max(np.abs(params[offs[i]:offs[i + 1]].view(*g45[f'small_param_{i}'].shape).numpy() - g45[f'small_param_{i}']).max() / np.abs(g45[f'small_param_{i}']).max() for i in g45['small_ids']),
        np.abs(eng.forward(sb[0][0].cuda()).cpu().numpy() - g45['small_final_logits']).max()))
    K = 5
    eng.load_model_state(sd, lrs)
    xk, yk = synthetic.synthetic_frames(1, 96, 160, seed=1000 + K)
    xm, ym = torch.flip(xk, dims=[3]).contiguous(), torch.flip(yk, dims=[3]).contiguous()
    eng.meta_task_begin()
    tl = [eng.finetune_step(xk.cuda(), yk.cuda(), accumulate=True) for _ in range(K)]
    flat = torch.zeros(eng.n_lr + eng.n_param, device='cuda')
    ml = eng.meta_grad(xm.cuda(), ym.cuda(), flat)
    flat = flat.cpu()
    ref = g7[f'k{K}_lr_grad']
    o2 = offs + eng.n_lr
    print('meta task K=5: train losses %.3e; meta loss %.3e; lr-grad max|diff|/max|ref| %.3e; init-grad L2 (all tensors) %.3e; init-grad last tensor %.3e' % (
        (np.abs(np.array(tl) - g7[f'k{K}_train_losses']) / np.abs(g7[f'k{K}_train_losses'])).max(), abs(ml - g7[f'k{K}_meta_loss'][0]) / abs(g7[f'k{K}_meta_loss'][0]),
        np.abs(flat[:eng.n_lr].numpy() - ref).max() / np.abs(ref).max(),
        max(abs(float(flat[o2[i]:o2[i + 1]].double().norm()) - g7[f'k{K}_init_grad_fp'][i][1]) / (g7[f'k{K}_init_grad_fp'][i][1] + 1e-12) for i in range(len(tr))),
        np.abs(flat[o2[-3]:o2[-2]].view(*g7[f'k{K}_init_grad_last'].shape).numpy() - g7[f'k{K}_init_grad_last']).max() / np.abs(g7[f'k{K}_init_grad_last']).max()))
    eng.close()
engine_mod.set_matrix_mode(PREV)
