"""Pin the CPU oracle (oracle/) against the fixtures the UNMODIFIED reference produced
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from eosvos_amd import synthetic, topology
from oracle import deeplab, meta
from oracle import topology as otopo

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
SMALL = (96, 160)
FULL = (480, 854)


def fp(t):
    t = t.detach().double().flatten()
    idx = torch.linspace(0, t.numel() - 1, 16).long()
    return np.concatenate([[t.sum().item(), t.norm().item()], t[idx].numpy()])


def fp_close(a, b, rtol=2e-4, atol=1e-5):
    a, b = np.asarray(a), np.asarray(b)
    scale = max(abs(b[1]), 1e-12)          # l2 norm sets the scale of the sum and samples
    assert abs(a[1] - b[1]) <= rtol * scale + atol, (a[1], b[1])
    assert np.all(np.abs(a - b) <= rtol * scale * 4 + atol), np.abs(a - b).max()


def test_g1_layout(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'g1_layout.json')))
    for enc in ('resnet50', 'resnet101'):
        assert [k for k, _ in g[enc + '_bn_keys']] == otopo.state_dict_keys(enc, 'bn')
        assert [k for k, _ in g[enc + '_bn_keys']] == topology.model_state_keys(enc, 'bn')
        tr = g[enc + '_trainable']
        assert [n for n, _ in tr] == otopo.trainable_names(enc)
        assert [tuple(s) for _, s in tr] == otopo.trainable_shapes(enc)
        assert [(n, tuple(s)) for n, s in tr] == topology.trainable(enc)
        mk = g[enc + '_meta_keys']
        names = [n for n, _ in topology.trainable(enc)]
        exp = ['log_init_lr_' + n.replace('.', '-') for n in names] + \
              ['model_init_' + n.replace('.', '-') for n in names]
        assert [k for k, _ in mk] == exp
        lr_shapes = [tuple(s) for _, s in mk[:len(names)]]
        assert lr_shapes == [topology.neuron_lr_shape(s) for _, s in topology.trainable(enc)]
        convs = g[enc + '_convs']
        ours = otopo.conv_list(enc)
        assert len(convs) == len(ours)
        for r, c in zip(convs, ours):
            assert r == [c.name, c.cin, c.cout, c.k, c.stride, c.dil, c.pad, c.bias], (r, c)
    assert [k for k, _ in g['resnet50_gn_keys']] == otopo.state_dict_keys('resnet50', 'gn')
    assert len(g['resnet50_bn_keys']) == 374 and len(g['resnet50_gn_keys']) == 188
    assert g['train_without_dropout_all_eval'] is True
    n_par = sum(int(np.prod(s)) for _, s in g['resnet50_trainable'])
    assert n_par == 40289729 and len(g['resnet50_trainable']) == 64
    assert sum(int(np.prod(s)) for _, s in g['resnet50_meta_keys']) == 40318387


def test_g2_forward_small(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g2_forward.npz'))
    sd = synthetic.synthetic_state('resnet50')
    np.testing.assert_allclose(np.stack([fp(sd[n]) for n, _ in topology.trainable('resnet50')]),
                               g['weights_fp'], rtol=1e-6, atol=1e-7)
    x, _ = synthetic.synthetic_frames(2, *SMALL, seed=7)
    np.testing.assert_allclose(fp(x), g['small_bn_input_fp'], rtol=1e-6)
    for tag in ('bn', 'gn'):
        taps = {}
        with torch.no_grad():
            logits = deeplab.forward(sd, x, 'resnet50', tag, taps)
        ref = g[f'small_{tag}_logits']
        assert np.abs(logits.numpy() - ref).max() < 1e-4, np.abs(logits.numpy() - ref).max()
        assert np.array_equal(logits.numpy() >= 0, ref >= 0) or \
            ((logits.numpy() >= 0) != (ref >= 0)).sum() <= (np.abs(ref) < 1e-4).sum()
        for k in ('stem', 'layer1', 'layer2', 'layer3', 'layer4', 'aspp', 'low_logits'):
            fp_close(fp(taps[k]), g[f'small_{tag}_tap_{k}'])


def test_g2_forward_full(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g2_forward.npz'))
    sd = synthetic.synthetic_state('resnet50')
    x, _ = synthetic.synthetic_frames(1, *FULL, seed=7)
    np.testing.assert_allclose(fp(x), g['full_bn_input_fp'], rtol=1e-6)
    with torch.no_grad():
        logits = deeplab.forward(sd, x)
    sub = logits[0, 0, ::8, ::7].numpy()
    assert np.abs(sub - g['full_bn_logits_sub']).max() < 1e-4
    mask = np.packbits((logits >= 0).numpy().astype(np.uint8))
    diff = np.unpackbits(mask ^ g['full_bn_mask']).sum()
    assert diff <= int(g['full_bn_near_zero'][0]), diff
    fp_close(fp(logits), g['full_bn_logits_fp'])


def test_g3_loss(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g3_loss.npz'))
    lg, gt = torch.from_numpy(g['logits']), torch.from_numpy(g['gt'])
    assert abs(float(deeplab.bce_loss(lg, gt)) - g['mean'][0]) < 1e-6
    np.testing.assert_allclose(deeplab.bce_loss(lg, gt, False).numpy(), g['per_sample'], rtol=1e-6)
    for name, key in (('dice', 'dice'), ('cross_entropy_and_dice', 'ce_dice'), ('class_balanced_cross_entropy', 'cbce')):
        x = lg.clone().requires_grad_(True)
        l = deeplab.loss_fn(name, x, gt)
        assert abs(float(l) - g[key][0]) < 2e-6 * max(1.0, abs(g[key][0]))
        (d,) = torch.autograd.grad(l, x)
        np.testing.assert_allclose(d.numpy(), g[key + '_dlogits'], rtol=1e-4, atol=1e-9)


def test_g3b_per_sample_losses(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g3_loss.npz'))
    gb = np.load(os.path.join(golden_dir, 'g3b_per_sample_loss.npz'))
    lg, gt = torch.from_numpy(g['logits']), torch.from_numpy(g['gt'])
    for name in ('cross_entropy', 'dice', 'cross_entropy_and_dice', 'class_balanced_cross_entropy'):
        np.testing.assert_allclose(deeplab.loss_per_sample(name, lg, gt).reshape(-1).numpy(), gb['per_' + name], rtol=1e-6)


def _meta_inputs():
    sd = synthetic.synthetic_state('resnet50')
    lrs = synthetic.synthetic_lrs('resnet50')
    return sd, lrs


def test_g45_finetune_small(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g45_finetune.npz'))
    sd, lrs = _meta_inputs()
    names = otopo.trainable_names()
    batches = [synthetic.synthetic_frames(3, *SMALL, seed=7 + it) for it in range(5)]
    loss, grads, _ = meta.finetune_step(sd, lrs, *batches[0])
    assert abs(float(loss) - g['small_losses'][0]) < 1e-5
    for i in g['small_ids']:
        ref = g[f'small_grad_{i}']
        assert np.abs(grads[i].numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-7
    got = np.stack([fp(x) for x in grads])
    for a, b in zip(got, g['small_grad_fp']):
        fp_close(a, b, rtol=1e-3)
    losses, P = meta.finetune(sd, lrs, batches)
    np.testing.assert_allclose(losses, g['small_losses'], rtol=2e-4, atol=1e-5)
    for i in g['small_ids']:
        ref = g[f'small_param_{i}']
        assert np.abs(P[names[i]].numpy() - ref).max() <= 1e-3 * np.abs(ref).max()
    with torch.no_grad():
        lg = deeplab.forward(P, batches[0][0])
    assert np.abs(lg.numpy() - g['small_final_logits']).max() < 2e-3


def test_g45_c1_prefix(golden_dir):
    """First 2 of the 10 full-size C1 iterations (the GPU tests check all 10)."""
    g = np.load(os.path.join(golden_dir, 'g45_finetune.npz'))
    sd, lrs = _meta_inputs()
    x, y = synthetic.synthetic_frames(1, *FULL, seed=7)
    losses, _ = meta.finetune(sd, lrs, [(x, y)] * 2)
    np.testing.assert_allclose(losses, g['c1_losses'][:2], rtol=2e-4)


def test_g6_merge(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g6_merge.npz'))
    for c in range(g['probs'].shape[0]):
        lab = meta.merge_labels(torch.from_numpy(g['probs'][c]))
        assert np.array_equal(lab.numpy(), g['labels'][c])


@pytest.mark.parametrize('K', [2, 5])
def test_g7_meta_task(golden_dir, K):
    g = np.load(os.path.join(golden_dir, 'g7_meta_task.npz'))
    sd, lrs = _meta_inputs()
    x, y = synthetic.synthetic_frames(1, *SMALL, seed=1000 + K)
    xm, ym = torch.flip(x, dims=[3]), torch.flip(y, dims=[3])
    out = meta.meta_task(sd, lrs, [(x, y)] * K, (xm, ym))
    np.testing.assert_allclose(out['train_losses'], g[f'k{K}_train_losses'], rtol=2e-4)
    assert abs(out['meta_loss'] - g[f'k{K}_meta_loss'][0]) < 2e-4 * abs(g[f'k{K}_meta_loss'][0])
    lr_g = torch.cat([t.flatten() for t in out['g_lr']]).numpy()
    ref = g[f'k{K}_lr_grad']
    assert lr_g.shape == ref.shape == (28658,)
    assert np.abs(lr_g - ref).max() <= 2e-3 * np.abs(ref).max(), np.abs(lr_g - ref).max() / np.abs(ref).max()
    got = np.stack([fp(t) for t in out['g_init']])
    for a, b in zip(got, g[f'k{K}_init_grad_fp']):
        fp_close(a, b, rtol=2e-3)
    ref_last = g[f'k{K}_init_grad_last']
    assert np.abs(out['g_init'][-2].numpy() - ref_last).max() <= 2e-3 * np.abs(ref_last).max()


def test_g8_radam(golden_dir):
    g = np.load(os.path.join(golden_dir, 'g8_radam.npz'))
    ps = [torch.from_numpy(g[f'p0_{i}'].copy()) for i in range(3)]
    states = [dict() for _ in range(3)]
    for step in range(8):
        grads = [torch.from_numpy(g[f'g_{step}_{i}']) for i in range(3)]
        meta.outer_step(ps[:1], ps[1:], grads[:1], grads[1:], states, meta_batch_size=4,
                        grad_clip=0.1 if step >= 6 else None)
        for i in range(3):
            np.testing.assert_allclose(ps[i].numpy(), g[f'p_{step}_{i}'], rtol=1e-6, atol=1e-9)
    n5, _ = meta.radam_scalars(5)
    n6, _ = meta.radam_scalars(6)
    assert n5 < 5 <= n6          # the SGD->Adam switch happens at step 6


def test_g9_misc(golden_dir):
    g = json.load(open(os.path.join(golden_dir, 'g9_misc.json')))
    lrs = synthetic.synthetic_lrs('resnet50')
    np.testing.assert_allclose([float(l.mean()) for l in lrs], g['state_lr_loaded'], rtol=1e-6)
    np.testing.assert_allclose([float(l.mean()) for l in lrs], g['init_lr_loaded'], rtol=1e-6)
    assert g['init_lr_n'] == 64
    assert g['epoch_sampler'] == [[[0, 0, 0]], [[0, 1, 0, 1]]]


def test_g12_online_adapt_schedule():
    r = meta.online_adapt_schedule(num_frames=12, train_frame_id=0, step=5, train_batch_size=3)
    assert [(d['eval_min'], d['eval_max']) for d in r] == [(1, 6), (6, 11), (11, 12)]
    assert r[0]['propagate_frames'] == []
    assert r[1]['propagate_frames'] == [3, 2] and r[2]['propagate_frames'] == [8, 7]
    r = meta.online_adapt_schedule(num_frames=12, train_frame_id=0, step=0, train_batch_size=3)
    assert [(d['eval_min'], d['eval_max']) for d in r] == [(1, 12)]


HIER_CASES = [('SINGLE', False), ('TENSOR', False), ('TENSOR', True), ('NEURON', True), ('PARAM', False),
              ('PARAM', True), ('SINGLE', True)]


@pytest.mark.parametrize('level,use_log', HIER_CASES)
def test_g13_lr_hierarchy(golden_dir, level, use_log):
    """lr_hierarchy_level / use_log_init_lr (meta_optim.py:27-67,157-163,180-185) vs the reference's autograd."""
    g = np.load(os.path.join(golden_dir, 'g13_lr_hierarchy.npz'))
    tag = f'{level}_{int(use_log)}'
    sd = synthetic.synthetic_state('resnet50')
    store = synthetic.synthetic_lr_state('resnet50', level, use_log)
    x, y = synthetic.synthetic_frames(1, *SMALL, seed=1302)
    xm, ym = torch.flip(x, dims=[3]), torch.flip(y, dims=[3])
    out = meta.meta_task_hier(sd, store, level, use_log, [(x, y)] * 2, (xm, ym))
    np.testing.assert_allclose(out['train_losses'], g[tag + '_train_losses'], rtol=2e-4)
    assert abs(out['meta_loss'] - g[tag + '_meta_loss'][0]) < 2e-4 * abs(g[tag + '_meta_loss'][0])
    if level == 'PARAM':
        for a, b in zip(np.stack([fp(t) for t in out['g_lr']]), g[tag + '_lr_grad_fp']):
            fp_close(a, b, rtol=2e-3, atol=1e-4 * abs(b[1]) + 1e-12)
        for idx, key in ((-5, '_lr_grad_dec1'), (-2, '_lr_grad_last')):
            ref = g[tag + key]
            assert np.abs(out['g_lr'][idx].numpy() - ref).max() <= 2e-3 * np.abs(ref).max()
    else:
        got = out['g_lr']
        got = torch.cat([t.flatten() for t in got]).numpy() if isinstance(got, list) else got.flatten().numpy()
        ref = g[tag + '_lr_grad']
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 2e-3 * np.abs(ref).max(), np.abs(got - ref).max() / np.abs(ref).max()
    for a, b in zip(np.stack([fp(t) for t in out['g_init']]), g[tag + '_init_grad_fp']):
        fp_close(a, b, rtol=2e-3)


@pytest.mark.parametrize('tag,bptt,multi', [('trunc', 2, None), ('multi', 4, [0.1, 0.2, 0.3, 0.4]),
                                            ('both', 2, [0.1, 0.2, 0.3, 0.4])])
def test_g14_bptt_schedules(golden_dir, tag, bptt, multi):
    """Truncated and multi-step BPTT (meta_run.py:154-221) vs the reference's autograd (fixture G14)."""
    g = np.load(os.path.join(golden_dir, 'g14_bptt.npz'))
    sd, lrs = _meta_inputs()
    x, y = synthetic.synthetic_frames(1, *SMALL, seed=1404)
    xm, ym = torch.flip(x, dims=[3]), torch.flip(y, dims=[3])
    out = meta.meta_task_bptt(sd, lrs, [(x, y)] * 4, (xm, ym), bptt_epochs=bptt, multi_step_bptt_loss=multi)
    np.testing.assert_allclose(out['meta_losses'], g[tag + '_meta_losses'], rtol=2e-4)
    lr_g = torch.cat([t.flatten() for t in out['g_lr']]).numpy()
    ref = g[tag + '_lr_grad']
    assert np.abs(lr_g - ref).max() <= 2e-3 * np.abs(ref).max(), np.abs(lr_g - ref).max() / np.abs(ref).max()
    for a, b in zip(np.stack([fp(t) for t in out['g_init']]), g[tag + '_init_grad_fp']):
        fp_close(a, b, rtol=2e-3)
    ref_last = g[tag + '_init_grad_last']
    assert np.abs(out['g_init'][-2].numpy() - ref_last).max() <= 2e-3 * np.abs(ref_last).max()


def test_g19_heavy_tailed_state_meta_task(golden_dir):
    """The oracle on `synthetic.heavy_tailed_state` (BatchNorm statistics over 4-6 decades, near-dead channels) against
    the reference's autograd: K = 2 meta task at 96x160 of fixture G19."""
    g = np.load(os.path.join(golden_dir, 'g19_heavy_tailed.npz'))
    sd, lrs = synthetic.heavy_tailed_state(), synthetic.synthetic_lrs()
    x, y = synthetic.synthetic_frames(1, *SMALL, seed=1002)
    out = meta.meta_task(sd, lrs, [(x, y)] * 2, (torch.flip(x, dims=[3]), torch.flip(y, dims=[3])))
    np.testing.assert_allclose(out['train_losses'], g['meta_train_losses'], rtol=2e-4)
    assert abs(out['meta_loss'] - g['meta_loss'][0]) < 2e-4 * abs(g['meta_loss'][0])
    lr_g = torch.cat([t.flatten() for t in out['g_lr']]).numpy()
    assert np.abs(lr_g - g['meta_lr_grad']).max() <= 2e-3 * np.abs(g['meta_lr_grad']).max()
    for a, b in zip(np.stack([fp(t) for t in out['g_init']]), g['meta_init_grad_fp']):
        fp_close(a, b, rtol=2e-3)


def test_g19_heavy_tailed_state_c2_first_iteration(golden_dir):
    """Same state, the benchmarked shape (480x854, batch 3): loss and all 64 gradient norms of the first iteration."""
    g = np.load(os.path.join(golden_dir, 'g19_heavy_tailed.npz'))
    sd, lrs = synthetic.heavy_tailed_state(), synthetic.synthetic_lrs()
    x, y = synthetic.synthetic_frames(3, 480, 854, seed=21)
    loss, grads, _ = meta.finetune_step(sd, lrs, x, y)
    assert abs(float(loss) - g['losses'][0]) <= 2e-5 * g['losses'][0]
    for gr, ref in zip(grads, g['grad_fp']):
        assert abs(float(gr.double().norm()) - ref[1]) <= 1e-3 * ref[1] + 1e-12
