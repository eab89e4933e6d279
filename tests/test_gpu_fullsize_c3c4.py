"""The benchmarked sizes of BASELINE configs[2] (e-OSVOS-OnA) and configs[3]/[4] (one meta task per rank) against fixtures
the REFERENCE produced at 480 x 854 (`tests/golden/make_g17.py`):

* G17: `src/util/evaluate.py::evaluate` run unmodified with the reference DeepLabV3Plus / MetaOptimizer on an 8-frame
  two-object sequence (batch 3, 4 iterations, online adaptation every 3 frames with 2 iterations, FIRST_STEP reset) ->
  the product's `finetune_object` loop on the HIP engine: batch composition of every iteration, train losses, logits of
  every predicted frame <= 1e-3 (north_star), masks and merged label maps bit-exact outside the recorded near-zero count.
* G7-full: one K = 5 meta task with the reference's autograd -> `eosvos_meta_grad`: losses, the 28 658 lr gradients
  <= 1e-3 of their max, per-tensor init-gradient L2.
"""
import os

import numpy as np
import pytest
import torch

from eosvos_amd import synthetic, topology

pytestmark = pytest.mark.gpu

FULL = (480, 854)
DEV = 'cuda:0'
BN_CFG = {'accum_stats': False, 'learn_weight': False, 'learn_bias': False}
MO_CFG = dict(init_lr=1e-3, learn_model_init=True, second_order_gradients=False, lr_hierarchy_level='NEURON',
              use_log_init_lr=False, max_lr=None)


def test_c3_online_adaptation_full_size_vs_reference_evaluate(golden_dir, monkeypatch):
    from eosvos_amd import config
    from eosvos_amd.engine import Engine
    from eosvos_amd.evaluate import finetune_object, merge_objects
    from eosvos_amd.helper_func import init_parent_model
    from eosvos_amd.meta_optim import MetaOptimizer
    g = np.load(os.path.join(golden_dir, 'g17_c3_full.npz'))
    seed, step, batch, eval_epochs, ona_epochs, n_frames, n_obj = [int(v) for v in g['scenario']]
    H, W = FULL
    base, gt = synthetic.synthetic_frames(1, H, W, seed=17, second_object=True)
    top = (torch.arange(H).view(-1, 1) < H // 2)
    objs = [(gt[0] * top).float(), (gt[0] * ~top).float()]
    seq = torch.cat([torch.roll(base, shifts=4 * i, dims=3) for i in range(n_frames)]).to(DEV)
    cfg = config.parse_cli(['with', 'DAVIS-2017', 'e-OSVOS-OnA', f'num_epochs.eval={eval_epochs}', f'eval_online_adapt.num_epochs={ona_epochs}',
                            f'eval_online_adapt.step={step}', 'data_cfg.random_train_transform=False', f'seed={seed}'])
    assert cfg['data_cfg']['batch_sizes']['train'] == batch and cfg['eval_online_adapt']['reset_model_mode'] == 'FIRST_STEP'
    model, _ = init_parent_model(architecture='DeepLabV3Plus', encoder='resnet50', train_encoder=True, batch_norm=BN_CFG)
    model.to(DEV)
    sd = synthetic.synthetic_state('resnet50')
    model.load_state_dict(sd)
    mo = MetaOptimizer(model, **MO_CFG)
    msd = {}
    for (n, _), lr in zip(topology.trainable('resnet50'), synthetic.synthetic_lrs('resnet50')):
        msd['log_init_lr_' + n.replace('.', '-')] = lr.clone()
    for n, _ in topology.trainable('resnet50'):
        msd['model_init_' + n.replace('.', '-')] = sd[n].clone()
    # record what the loop feeds the engine and what it predicts: batch size per training iteration, logits per frame
    batch_sizes, logits_seen = [], []
    real_call = type(model).__call__

    def call(self, inputs):
        batch_sizes.append(int(inputs.shape[0]))
        return real_call(self, inputs)
    monkeypatch.setattr(type(model), '__call__', call)
    real_infer = Engine.infer

    def infer(self, images):
        out = real_infer(self, images)
        logits_seen.extend(self.debug_tensor('logits')[:images.shape[0]].cpu())
        return out
    monkeypatch.setattr(Engine, 'infer', infer)
    probs, losses = [], []
    for o in range(n_obj):
        p, hist = finetune_object(model, mo, msd, seq, objs[o].to(DEV), cfg)
        probs.append(p)
        losses += [v for rnd in hist for v in rnd]
    # --- which frames / pseudo-labels entered every batch (evaluate.py:227-253): the batch sizes tell empty propagated frames
    assert batch_sizes == g['batch_sizes'].tolist()
    np.testing.assert_allclose(losses, g['train_losses'], rtol=2e-4)
    # --- logits of every predicted (object, frame), in the reference's order
    assert len(logits_seen) == len(g['infer_frame'])
    idx = torch.linspace(0, H * W - 1, g['logit_samples'].shape[1]).long()
    worst = 0.0
    for k, lg in enumerate(logits_seen):
        flat = lg.flatten()
        d = float(np.abs(flat[idx].numpy() - g['logit_samples'][k]).max())
        worst = max(worst, d)
        assert d <= 1e-3, (k, d)                                                   # north_star: logits within 1e-3
        l2 = float(flat.double().norm())
        assert abs(l2 - g['logit_fp'][k][1]) <= 1e-4 * g['logit_fp'][k][1], (k, l2, g['logit_fp'][k][1])
        bits = np.packbits((flat >= 0).numpy())
        ndiff = int(np.unpackbits(bits ^ g['mask_bits'][k]).sum())
        assert ndiff <= int(g['near_zero'][k]), (k, ndiff, int(g['near_zero'][k]))   # bit-exact outside |logit| < 1e-3
    # --- merged label maps = what evaluate() handed to imageio.imsave (evaluate.py:322-342)
    labels = merge_objects(model.engine, probs).cpu().numpy()
    assert labels.shape == g['labels'].shape
    budget = int(g['near_zero'].sum())
    assert int((labels != g['labels']).sum()) <= budget
    assert np.array_equal(labels[0], g['labels'][0])                              # the train frame: 2 * GT of both objects
    print('G17 worst sampled-logit difference %.3e, label pixels differing %d (near-zero budget %d)' % (
        worst, int((labels != g['labels']).sum()), budget))


def test_meta_task_full_size_vs_golden(golden_dir):
    """C4 / C5 per-rank work at the benchmarked size: K = 5 inner steps (batch 1) + meta frame at 480 x 854."""
    from eosvos_amd.engine import Engine
    g = np.load(os.path.join(golden_dir, 'g7_meta_task_full.npz'))
    K = int(g['K'][0])
    eng = Engine('resnet50', *FULL, max_batch=1, device=DEV)
    try:
        eng.load_model_state(synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50'))
        x, y = synthetic.synthetic_frames(1, *FULL, seed=1000)
        xm, ym = torch.flip(x, dims=[3]).contiguous(), torch.flip(y, dims=[3]).contiguous()
        xg, yg = x.to(DEV), y.to(DEV)
        eng.meta_task_begin()
        tl = [eng.finetune_step(xg, yg, accumulate=True) for _ in range(K)]
        flat = torch.zeros(eng.n_lr + eng.n_param, device=DEV)
        ml = eng.meta_grad(xm.to(DEV), ym.to(DEV), flat)
        np.testing.assert_allclose(tl, g['train_losses'], rtol=2e-5)
        assert abs(ml - g['meta_loss'][0]) <= 5e-5 * abs(g['meta_loss'][0])
        flat = flat.cpu()
        ref = g['lr_grads']
        lr_g = flat[:eng.n_lr].numpy()
        err = np.abs(lr_g - ref).max() / np.abs(ref).max()
        assert err <= 1e-3, err                                                   # lr gradients within 1e-3 of their max
        tr = topology.trainable('resnet50')
        offs = np.cumsum([0] + [int(np.prod(s)) for _, s in tr]) + eng.n_lr
        worst = 0.0
        for i, (n, s) in enumerate(tr):
            l2 = float(flat[offs[i]:offs[i + 1]].double().norm())
            r = float(g['init_grad_l2'][i])
            worst = max(worst, abs(l2 - r) / (r + 1e-30))
            assert abs(l2 - r) <= 3e-3 * r + 1e-9, (n, l2, r)                      # measured: worst tensor 8.1e-4 (profiles/r03_parity_margins.txt)
        print('G7-full: lr-grad max error / max %.3e, worst init-grad L2 deviation %.3e' % (err, worst))
    finally:
        eng.close()
