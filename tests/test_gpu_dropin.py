"""The reference-mirroring Python surface (networks / meta_optim / helper_func / evaluate /
train_meta) driven exactly like the reference's loops, checked against the golden fixtures and the
CPU oracle.  Needs an MI355X: pytest -m gpu."""
import os

import numpy as np
import pytest
import torch

from eosvos_amd import synthetic, topology

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
SMALL = (96, 160)
BN_CFG = {'accum_stats': False, 'learn_weight': False, 'learn_bias': False}
MO_CFG = dict(init_lr=1e-3, learn_model_init=True, second_order_gradients=False, lr_hierarchy_level='NEURON',
              use_log_init_lr=False, max_lr=None)


def _meta_state():
    sd = synthetic.synthetic_state('resnet50')
    lrs = synthetic.synthetic_lrs('resnet50')
    out = {}
    for (n, _), lr in zip(topology.trainable('resnet50'), lrs):
        out['log_init_lr_' + n.replace('.', '-')] = lr.clone()
    for n, _ in topology.trainable('resnet50'):
        out['model_init_' + n.replace('.', '-')] = sd[n].clone()
    return sd, out


@pytest.fixture(scope='module')
def model_and_optim():
    from eosvos_amd.helper_func import init_parent_model
    from eosvos_amd.meta_optim import MetaOptimizer
    model, parent_states = init_parent_model(architecture='DeepLabV3Plus', encoder='resnet50', train_encoder=True,
                                             decoder_norm_layer='BatchNorm2d', replace_batch_with_group_norms=False,
                                             batch_norm=BN_CFG, roi_pool_output_sizes=None,
                                             eval_augment_rpn_proposals_mode=None, box_nms_thresh=None,
                                             maskrcnn_loss=None)
    assert parent_states == {}
    sd, msd = _meta_state()
    model.load_state_dict(sd)
    mo = MetaOptimizer(model, **MO_CFG)
    return model, mo, msd


def test_reference_style_finetune_loop(model_and_optim, golden_dir):
    """The loop of evaluate.py:196-274, verbatim call sequence."""
    from eosvos_amd.helper_func import compute_loss
    model, mo, msd = model_and_optim
    g = np.load(os.path.join(golden_dir, 'g45_finetune.npz'))
    mo.load_state_dict(msd)
    mo.reset()
    mo.eval()
    model.train_without_dropout()
    losses = []
    for it in range(5):
        x, y = synthetic.synthetic_frames(3, *SMALL, seed=7 + it)
        outputs = model(x.to(DEV))
        train_loss = compute_loss('cross_entropy', outputs[-1], y.to(DEV))
        losses.append(train_loss.item())
        model.zero_grad()
        mo.set_train_loss(train_loss)
        mo.only_box_head = False
        mo.step(train_loss)
        mo.meta_model.detach_param_groups()
    np.testing.assert_allclose(losses, g['small_losses'], rtol=5e-4, atol=1e-5)
    sd = model.state_dict()
    assert list(sd) == topology.model_state_keys('resnet50') and len(sd) == 374
    tr = topology.trainable('resnet50')
    for i in g['small_ids']:
        ref = g[f'small_param_{i}']
        assert np.abs(sd[tr[i][0]].numpy() - ref).max() <= 2e-3 * np.abs(ref).max()
    model.eval()
    out = model(synthetic.synthetic_frames(3, *SMALL, seed=7)[0].to(DEV))[-1]
    assert np.abs(out.cpu().numpy() - g['small_final_logits']).max() < 5e-3
    per = compute_loss('cross_entropy', out, synthetic.synthetic_frames(3, *SMALL, seed=7)[1].to(DEV),
                       {'batch_average': False})
    assert per.shape == (3,)
    with pytest.raises(NotImplementedError):
        compute_loss('lovasz', out, out)            # unknown name, helper_func.py:55-56


def test_reference_style_meta_task(model_and_optim, golden_dir):
    """meta_run.py:121-214 call sequence (K=2), `.grad` of named_parameters() vs the golden."""
    from eosvos_amd.helper_func import compute_loss
    model, mo, msd = model_and_optim
    g = np.load(os.path.join(golden_dir, 'g7_meta_task.npz'))
    K = 2
    mo.load_state_dict(msd)
    mo.init_zero_grad()
    mo.zero_grad()
    mo.train()
    mo.reset()
    model.train_without_dropout()
    x, y = synthetic.synthetic_frames(1, *SMALL, seed=1000 + K)
    xg, yg = x.to(DEV), y.to(DEV)
    for _ in range(K):
        loss = compute_loss('cross_entropy', model(xg)[-1], yg)
        mo.set_train_loss(loss)
        mo.step(loss)
    meta_loss = mo.meta_backward(torch.flip(xg, dims=[3]), torch.flip(yg, dims=[3]))
    assert abs(meta_loss - g[f'k{K}_meta_loss'][0]) <= 5e-4 * abs(g[f'k{K}_meta_loss'][0])
    names = [n for n, _ in mo.named_parameters()]
    assert names == list(g[f'k{K}_names'])
    lr_g = torch.cat([p.grad.flatten() for n, p in mo.named_parameters() if n.startswith('log_init_lr_')]).cpu().numpy()
    ref = g[f'k{K}_lr_grad']
    assert np.abs(lr_g - ref).max() <= 5e-3 * np.abs(ref).max()
    mo.reset()
    mo.eval()


def test_online_adaptation_sequence(model_and_optim):
    """Config e-OSVOS-OnA on a short synthetic 2-object sequence: schedule, FIRST_STEP restore,
    pseudo-label batches and the merged label maps, against the same loop run on the CPU oracle."""
    from eosvos_amd import config
    from eosvos_amd.evaluate import evaluate_sequence, online_adapt_schedule
    from eosvos_amd.helper_func import set_random_seeds
    from oracle import augment as oaug
    from oracle import deeplab, meta
    model, mo, msd = model_and_optim
    cfg = config.parse_cli(['with', 'DAVIS-2017', 'e-OSVOS-OnA', 'num_epochs.eval=3', 'eval_online_adapt.num_epochs=2',
                            'eval_online_adapt.step=3'])
    H, W, N = SMALL[0], SMALL[1], 6
    frames, gt = synthetic.synthetic_frames(1, H, W, seed=3, second_object=True)
    seq = torch.cat([torch.roll(frames, shifts=4 * i, dims=3) for i in range(N)])
    rows = torch.arange(H).view(-1, 1)
    objs = [(gt[0] * (rows < H // 2)).float(), (gt[0] * (rows >= H // 2)).float()]
    assert [(r['eval_min'], r['eval_max'], r['propagate_frames']) for r in online_adapt_schedule(N, 0, 3, 3)] == \
        [(1, 4, []), (4, 6, [3, 2])]
    assert online_adapt_schedule(N, 0, 3, 3) == meta.online_adapt_schedule(N, 0, 3, 3)
    labels, probs, hist = evaluate_sequence(model, mo, msd, seq.to(DEV), objs, cfg)
    assert labels.shape == (N, H, W) and labels.dtype == torch.uint8
    assert [len(h) for h in hist[0]] == [3, 2]

    # the same procedure on the CPU oracle
    sd = synthetic.synthetic_state('resnet50')
    lrs = synthetic.synthetic_lrs('resnet50')
    o_probs = []
    for gto in objs:
        y0 = gto.view(1, 1, H, W)
        masks = torch.zeros(N, 1, H, W)
        masks[0] = 2 * y0[0]
        x0 = seq[0:1]
        # round 0: random_train_transform (e-OSVOS configs): 3 flip / scale / rotate warps per iteration, drawn
        # after set_random_seeds(seed + epoch + round) as evaluate.py:221-224 does
        batches = []
        for epoch in (1, 2, 3):
            set_random_seeds(cfg.get('seed', 1) + epoch)
            xs, ys = [], []
            for _ in range(3):
                ai, ag, _p = oaug.random_flip_scale_rotate(x0[0].permute(1, 2, 0).contiguous().numpy(), y0[0, 0].numpy())
                xs.append(torch.from_numpy(ai).permute(2, 0, 1))
                ys.append(torch.from_numpy(ag)[None])
            batches.append((torch.stack(xs).contiguous(), torch.stack(ys).contiguous()))
        _, P = meta.finetune(sd, lrs, batches)
        P_first = P
        with torch.no_grad():
            for f in range(1, 4):
                masks[f] = torch.sigmoid(deeplab.forward(P, seq[f:f + 1]))[0]
        xs, ys = [x0], [y0]
        for f in (3, 2):
            pg = masks[f:f + 1].ge(0.5).float()
            if pg.sum() != 0:
                xs.append(seq[f:f + 1]); ys.append(pg)
        _, P = meta.finetune(P_first, lrs, [(torch.cat(xs), torch.cat(ys))] * 2)
        with torch.no_grad():
            for f in range(4, 6):
                masks[f] = torch.sigmoid(deeplab.forward(P, seq[f:f + 1]))[0]
        o_probs.append(masks[:, 0])
    o_stack = torch.stack(o_probs, dim=1)
    o_labels = torch.stack([meta.merge_labels(o_stack[f]) for f in range(N)])
    # north_star: logits within 1e-3.  The loop hands out probabilities; logit(p) is compared where it is well
    # conditioned (|logit| < 6: one fp32 ulp of p moves it by < 3e-5) -- the frames after the train frame (which is
    # seeded with 2 * GT, not a probability) -- and the probabilities themselves everywhere.
    for o in range(2):
        p_gpu, p_ref = probs[o].cpu()[1:].double(), o_probs[o][1:].double()
        assert float((p_gpu - p_ref).abs().max()) < 2.5e-4
        lg_ref = torch.log(p_ref) - torch.log1p(-p_ref)
        ok = lg_ref.abs() < 6
        lg_gpu = torch.log(p_gpu[ok]) - torch.log1p(-p_gpu[ok])
        assert float((lg_gpu - lg_ref[ok]).abs().max()) < 1e-3, float((lg_gpu - lg_ref[ok]).abs().max())
    near = ((o_stack - 0.5).abs() < 2.5e-4).any(dim=1)
    assert bool((labels.cpu() == o_labels)[~near].all())
    assert int((labels.cpu() != o_labels).sum()) <= int(near.sum())
    assert set(labels.unique().tolist()) <= {0, 1, 2}


def test_objects_in_flight_equal_one_after_the_other(model_and_optim):
    """`evaluate.run_objects_in_flight`: the two objects of a sequence fine-tuned side by side (one spawned model /
    engine / stream each, both planning for half the chip) give bit for bit what the same objects give one after the
    other on one engine at that workgroup budget -- online adaptation, augmentation draws and snapshots included --
    and the whole-chip run differs from it by summation order only."""
    from eosvos_amd import config
    from eosvos_amd.evaluate import finetune_object, object_workers, run_objects_in_flight
    model, mo, msd = model_and_optim
    cfg = config.parse_cli(['with', 'DAVIS-2017', 'e-OSVOS-OnA', 'num_epochs.eval=3', 'eval_online_adapt.num_epochs=2',
                            'eval_online_adapt.step=3'])
    H, W, N = SMALL[0], SMALL[1], 6
    frames, gt = synthetic.synthetic_frames(1, H, W, seed=3, second_object=True)
    seq = torch.cat([torch.roll(frames, shifts=4 * i, dims=3) for i in range(N)]).to(DEV)
    rows = torch.arange(H).view(-1, 1)
    objs = [(gt[0] * (rows < H // 2)).float(), (gt[0] * (rows >= H // 2)).float()]
    workers = object_workers(model, mo, MO_CFG, 2)
    assert all(w.model is not model for w in workers) and workers[1].stream != workers[0].stream
    res = run_objects_in_flight(workers, msd, seq, objs, cfg)
    assert workers[1].model.engine is not workers[0].model.engine
    assert not workers[0].model.engine.set_side_stream(True)            # built for side-by-side work: one queue, no second stream
    model._ensure_engine(H, W, 3)
    assert model.engine.set_wg_budget(256) == 256
    model.set_side_stream(False)         # the workers' configuration: the split plan (= fp32 summation order) depends on it
    torch.cuda.synchronize()
    one = [finetune_object(model, mo, msd, seq, g, cfg) for g in objs]              # same budget, one engine
    for (p2, h2), (p1, h1) in zip(res, one):
        assert h2 == h1 and torch.equal(p2, p1)
    model.set_wg_budget(0)
    model.set_side_stream(True)
    whole = [finetune_object(model, mo, msd, seq, g, cfg) for g in objs]
    for (p2, _), (p0, _) in zip(res, whole):
        assert float((p2 - p0).abs().max()) < 1e-3
    for w in workers:
        w.model.engine.close()


def test_train_meta_entry_points(tmp_path):
    from eosvos_amd import train_meta
    from eosvos_amd.checkpoint import load_meta_checkpoint
    res = train_meta.main(['with', 'DAVIS-2017', 'e-OSVOS', 'num_epochs.eval=2', f'save_dir={tmp_path}', 'env_suffix=e'],
                          height=96, width=160, num_frames=3, data_root=str(tmp_path / 'no_data'))
    labels = res['val']['labels']['synthetic00']
    assert labels.shape == (3, 96, 160) and set(labels.unique().tolist()) <= {0, 1, 2}
    # prediction PNGs and eval checkpoints of the eval worker (evaluate.py:332-382)
    assert os.path.exists(os.path.join(str(tmp_path), 'e', 'best_eval_preds', 'synthetic', 'val', 'synthetic00', '00002.png'))
    assert os.path.exists(os.path.join(str(tmp_path), 'e', 'last_val_meta_iter.model'))
    # the validation child process is exercised on CPU (tests/test_multiprocess.py) and by tools/concurrent_eval_smoke.py;
    # here the pytest process has already initialised the GPU, so none is spawned from it
    mt = train_meta.main(['with', 'YouTube-VOS', 'meta_batch_size=2', 'num_epochs.train=2', f'save_dir={tmp_path}',
                          'env_suffix=t'], height=96, width=160, num_meta_iters=1, data_root=str(tmp_path / 'no_data'),
                         eval_cmd=False)
    sd, info = load_meta_checkpoint(os.path.join(str(tmp_path), 't', 'last_meta_iter.model'))
    assert info['meta_iter'] == 1 and len(sd) == 128
    assert list(sd)[0] == 'log_init_lr_backbone-conv1-weight' and list(sd)[64] == 'model_init_backbone-conv1-weight'
    assert mt.step == 1


@pytest.mark.parametrize('camel_size', [(96, 160), (80, 128)], ids=['one_size', 'two_sizes'])
def test_train_meta_on_dataset_files(tmp_path, capsys, camel_size):
    """`train_meta.main` in meta-train mode on a DAVIS-2017 tree on disk: `MetaTaskset` sampling, the one-iteration-ahead
    prefetch of decoding + colour jitter on a worker thread, device-side flip / scale-rotate, three meta-iterations of two
    tasks in flight.  `two_sizes`: the second video has another frame size than the trainer's engines (the reference feeds
    videos at their native sizes): its tasks run on pooled engines of that size (`MetaTrainer._engines_for`) and their frames
    are warped at their own size (`eosvos_warp_affine_hw`)."""
    import json
    from PIL import Image
    from eosvos_amd import train_meta
    root = tmp_path / 'data' / 'DAVIS-2017'
    rng = np.random.default_rng(0)
    for seq in ('bear', 'camel'):
        (root / 'JPEGImages' / '480p' / seq).mkdir(parents=True)
        (root / 'Annotations' / '480p' / seq).mkdir(parents=True)
        hh, ww = camel_size if seq == 'camel' else (96, 160)
        for f in range(5):
            Image.fromarray(rng.integers(0, 256, (hh, ww, 3), dtype=np.uint8)).save(root / 'JPEGImages' / '480p' / seq / f'{f:05d}.jpg')
            lab = np.zeros((hh, ww), np.uint8)
            lab[30:60, 50 + 3 * f:100 + 3 * f] = 1
            if seq == 'camel':
                lab[65:78, 20:60] = 2
            Image.fromarray(lab, mode='L').save(root / 'Annotations' / '480p' / seq / f'{f:05d}.png')
    (root / 'train_seqs.txt').write_text('bear\ncamel\n')
    capsys.readouterr()
    mt = train_meta.main(['with', 'DAVIS-2017', 'meta_batch_size=2', 'num_epochs.train=2', 'datasets.train.eval=False',
                          f'save_dir={tmp_path}', 'env_suffix=files'], height=96, width=160, num_meta_iters=3,
                         data_root=str(tmp_path / 'data'), eval_cmd=False)
    lines = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith('{') and '"meta_iter"' in l]
    assert [l['meta_iter'] for l in lines] == [1, 2, 3] and all(l['data'] == 'files' for l in lines)
    # 3 tasks (bear, camel x 2 objects) in sub-batches of 2: a pass is [2, 1], then the next pass starts
    assert [len(l['meta_losses']) for l in lines] == [2, 1, 2], lines
    assert all(np.isfinite(v) for l in lines for v in l['meta_losses'])
    assert mt.step == 3 and mt.skipped_tasks == 0 and len(mt.engines) == 2
    assert sorted(mt._pool) == ([] if camel_size == (96, 160) else [camel_size])
    mt._drop_pool()


@pytest.mark.parametrize('level,use_log', [('TENSOR', True), ('SINGLE', False)])
def test_reference_style_meta_task_other_levels(golden_dir, level, use_log):
    """The meta_run.py:121-214 call sequence with `lr_hierarchy_level` TENSOR / SINGLE and
    `use_log_init_lr` (meta_optim.py:27-42,157-163,180-185): `.grad` of `log_init_lr` vs fixture G13."""
    from eosvos_amd.helper_func import compute_loss, init_parent_model
    from eosvos_amd.meta_optim import MetaOptimizer
    g = np.load(os.path.join(golden_dir, 'g13_lr_hierarchy.npz'))
    tag = f'{level}_{int(use_log)}'
    model, _ = init_parent_model(architecture='DeepLabV3Plus', encoder='resnet50', train_encoder=True,
                                 decoder_norm_layer='BatchNorm2d', replace_batch_with_group_norms=False,
                                 batch_norm=BN_CFG, roi_pool_output_sizes=None, eval_augment_rpn_proposals_mode=None,
                                 box_nms_thresh=None, maskrcnn_loss=None)
    model.to(DEV)
    sd = synthetic.synthetic_state('resnet50')
    model.load_state_dict(sd)
    mo = MetaOptimizer(model, **dict(MO_CFG, lr_hierarchy_level=level, use_log_init_lr=use_log, max_lr=1.3e-3))
    msd = {'log_init_lr': synthetic.synthetic_lr_state('resnet50', level, use_log)}
    for n, _ in topology.trainable('resnet50'):
        msd['model_init_' + n.replace('.', '-')] = sd[n].clone()
    mo.load_state_dict(msd)
    mo.init_zero_grad()
    mo.zero_grad()
    mo.train()
    mo.reset()
    model.train_without_dropout()
    x, y = synthetic.synthetic_frames(1, *SMALL, seed=1302)
    xg, yg = x.to(DEV), y.to(DEV)
    losses = []
    for _ in range(2):
        loss = compute_loss('cross_entropy', model(xg)[-1], yg)
        losses.append(float(loss))
        mo.set_train_loss(loss)
        mo.step(loss)
    np.testing.assert_allclose(losses, g[tag + '_train_losses'], rtol=5e-4)
    meta_loss = mo.meta_backward(torch.flip(xg, dims=[3]), torch.flip(yg, dims=[3]))
    assert abs(meta_loss - g[tag + '_meta_loss'][0]) <= 5e-4 * abs(g[tag + '_meta_loss'][0])
    grad = dict(mo.named_parameters())['log_init_lr'].grad.flatten().cpu().numpy()
    ref = g[tag + '_lr_grad']
    assert grad.shape == ref.shape
    assert np.abs(grad - ref).max() <= 5e-3 * np.abs(ref).max()
    model.engine.close()


def test_per_sample_losses_vs_golden(golden_dir):
    """`compute_loss(loss_func, outputs, gts, {'batch_average': False})` for every loss (run_loader metrics,
    helper_func.py:131-137) vs the reference's values on fixed logits (fixture G3b)."""
    from eosvos_amd.engine import Engine
    g = np.load(os.path.join(golden_dir, 'g3_loss.npz'))
    gb = np.load(os.path.join(golden_dir, 'g3b_per_sample_loss.npz'))
    eng = Engine('resnet50', 96, 160, max_batch=1, device=DEV)
    lg, gt = torch.from_numpy(g['logits']).to(DEV), torch.from_numpy(g['gt']).to(DEV)
    for name in ('cross_entropy', 'dice', 'cross_entropy_and_dice', 'class_balanced_cross_entropy'):
        got = torch.cat([eng.loss_of(name, lg[b], gt[b]) for b in range(lg.shape[0])]).cpu().numpy()
        np.testing.assert_allclose(got, gb['per_' + name], rtol=2e-5)
    eng.close()


def test_run_frames_metrics(model_and_optim):
    """`run_loader` for the DeepLab branch (helper_func.py:131-142): per-frame loss of the configured loss, >= 0.5
    accuracy and probabilities, against the CPU oracle."""
    from eosvos_amd.helper_func import run_frames
    from oracle import deeplab
    model, mo, msd = model_and_optim
    mo.load_state_dict(msd)
    mo.reset()
    x, y = synthetic.synthetic_frames(3, *SMALL, seed=31)
    sd = synthetic.synthetic_state('resnet50')
    with torch.no_grad():
        ref_logits = deeplab.forward(sd, x)
    for loss_func in ('cross_entropy', 'dice'):
        losses, accs, probs = run_frames(model, x.to(DEV), y.to(DEV), loss_func=loss_func)
        ref = deeplab.loss_per_sample(loss_func, ref_logits, y).reshape(-1)
        np.testing.assert_allclose(losses.numpy(), ref.numpy(), rtol=2e-4)
        assert float((probs.cpu() - torch.sigmoid(ref_logits)).abs().max()) < 1e-4
        ref_acc = (torch.sigmoid(ref_logits).ge(0.5) == y.bool()).float().view(3, -1).mean(1)
        assert float((accs - ref_acc).abs().max()) < 2e-3
    none = run_frames(model, x.to(DEV))
    assert none[0] is None and none[2].shape == (3, 1, *SMALL)


@pytest.mark.parametrize('tag,bptt,multi', [('trunc', 2, None), ('multi', 4, [0.1, 0.2, 0.3, 0.4]), ('both', 2, [0.1, 0.2, 0.3, 0.4])])
def test_dropin_bptt_schedules_vs_golden(golden_dir, tag, bptt, multi):
    """The reference's own task loop (`meta_run.py:154-221`: `reset(keep_state=True)` between segments, per-step
    `multi_step_bptt_loss` weights) written against the drop-in MetaOptimizer, vs the reference autograd (fixture G14)."""
    from eosvos_amd.helper_func import compute_loss, init_parent_model
    from eosvos_amd.meta_optim import MetaOptimizer
    g = np.load(os.path.join(golden_dir, 'g14_bptt.npz'))
    model, _ = init_parent_model(architecture='DeepLabV3Plus', encoder='resnet50', train_encoder=True,
                                 decoder_norm_layer='BatchNorm2d', replace_batch_with_group_norms=False,
                                 batch_norm=BN_CFG, roi_pool_output_sizes=None, eval_augment_rpn_proposals_mode=None,
                                 box_nms_thresh=None, maskrcnn_loss=None)
    model.to(DEV)
    sd = synthetic.synthetic_state('resnet50')
    model.load_state_dict(sd)
    mo = MetaOptimizer(model, **MO_CFG)
    msd = {}
    for (n, _), lr in zip(topology.trainable('resnet50'), synthetic.synthetic_lrs('resnet50')):
        msd['log_init_lr_' + n.replace('.', '-')] = lr.clone()
    for n, _ in topology.trainable('resnet50'):
        msd['model_init_' + n.replace('.', '-')] = sd[n].clone()
    mo.init_zero_grad()
    mo.load_state_dict(msd)
    mo.reset()
    mo.zero_grad()
    mo.train()
    model.train_without_dropout()
    x, y = synthetic.synthetic_frames(1, *SMALL, seed=1404)
    xg, yg = x.to(DEV), y.to(DEV)
    xm, ym = torch.flip(xg, dims=[3]).contiguous(), torch.flip(yg, dims=[3]).contiguous()
    K, meta_losses = 4, []
    for epoch in range(1, K + 1):
        loss = compute_loss('cross_entropy', model(xg)[-1], yg)
        mo.set_train_loss(loss)
        mo.step(loss)
        if multi:
            meta_losses.append(mo.meta_backward(xm, ym, weight=multi[epoch - 1]))
        stop = epoch == K
        if not epoch % bptt or stop:
            if not multi:
                meta_losses.append(mo.meta_backward(xm, ym))
            if not stop:
                mo.reset(keep_state=True)
    np.testing.assert_allclose(meta_losses, g[tag + '_meta_losses'], rtol=3e-5)
    params = dict(mo.named_parameters())
    got = torch.cat([p.grad.flatten() for n, p in params.items() if n.startswith('log_init_lr_')]).cpu().numpy()
    ref = g[tag + '_lr_grad']
    assert np.abs(got - ref).max() <= 2.5e-3 * np.abs(ref).max(), np.abs(got - ref).max() / np.abs(ref).max()
    init_g = [p.grad for n, p in params.items() if n.startswith('model_init_')]
    for i, gi in enumerate(init_g):
        r = g[tag + '_init_grad_fp'][i][1]
        assert abs(float(gi.double().norm()) - r) <= 4e-4 * r + 1e-9, i
    last = init_g[-2].cpu().numpy()
    assert np.abs(last - g[tag + '_init_grad_last']).max() <= 6e-4 * np.abs(g[tag + '_init_grad_last']).max()
    model.engine.close()


class _MixedSizeSequences:
    """Reader with two sequences of DIFFERENT frame sizes (YouTube-VOS has many): `evaluate_dataset` must rebuild the engine
    between them although the first one left a first-step snapshot behind (online adaptation)."""
    test_mode = False
    multi_object = 'single_id'
    all_frames = False
    seqs_names = ['wide', 'small']
    SIZES = {'wide': (96, 160), 'small': (64, 96)}

    def _objects(self, seq):
        h, w = self.SIZES[seq]
        frames, gt = synthetic.synthetic_frames(1, h, w, seed=21 + len(seq), second_object=True)
        top = (torch.arange(h).view(-1, 1) < h // 2)
        return frames, [(gt[0] * top).float(), (gt[0] * ~top).float()]

    def sequence_tensors(self, seq, device='cpu', with_frame_ids=False):
        frames, gts = self._objects(seq)
        seq_frames = torch.cat([torch.roll(frames, shifts=4 * i, dims=3) for i in range(5)])
        out = (seq_frames.to(device), [g.to(device) for g in gts])
        return out + ([0, 0],) if with_frame_ids else out

    def frame_names(self, seq):
        return [f'{i:05d}' for i in range(5)]

    def label_maps(self, seq):
        _, gts = self._objects(seq)
        h, w = self.SIZES[seq]
        lab = torch.zeros(h, w, dtype=torch.uint8)
        for o, g in enumerate(gts):
            lab[g[0] > 0] = o + 1
        return np.stack([torch.roll(lab, shifts=4 * i, dims=1).numpy() for i in range(5)])


@pytest.mark.parametrize('in_flight', [1, 2])
def test_evaluate_dataset_with_mixed_frame_sizes(model_and_optim, tmp_path, in_flight):
    from eosvos_amd import config
    from eosvos_amd.evaluate import evaluate_dataset
    model, mo, msd = model_and_optim
    cfg = config.parse_cli(['with', 'DAVIS-2017', 'e-OSVOS-OnA', 'num_epochs.eval=2', 'eval_online_adapt.num_epochs=1',
                            'eval_online_adapt.step=2'])
    cfg['datasets']['val'] = {'name': 'synthetic', 'split': 'val', 'eval': True}
    res = evaluate_dataset(model, mo, msd, _MixedSizeSequences(), cfg, 'val', save_dir=str(tmp_path), objects_in_flight=in_flight)
    assert res['labels']['wide'].shape == (5, 96, 160) and res['labels']['small'].shape == (5, 64, 96)
    for seq in ('wide', 'small'):
        assert set(res['labels'][seq].unique().tolist()) <= {0, 1, 2}
        assert np.array_equal(res['labels'][seq][0].numpy(), _MixedSizeSequences().label_maps(seq)[0])     # train frame = GT
    assert len(res['J_seq']) == 2 and all(np.isfinite(j) for j in res['J_seq'])
    for w in getattr(model, '_object_workers', None) or []:
        if w.model.engine is not None:
            w.model.engine.close()
    model._object_workers = None


def test_cli_loads_the_parent_checkpoint_into_the_engine(tmp_path, capsys):
    """`parent_model.train.paths` (`src/train_meta.py:91-96`): a parent `.model` file with NON-synthetic BatchNorm statistics
    -> `train_meta.main` -> the folded norm scale the engine holds (`eosvos_set_norm`) is that file's."""
    import json
    from eosvos_amd import train_meta
    sd = synthetic.synthetic_state('resnet50')
    sd['backbone.bn1.running_var'] = sd['backbone.bn1.running_var'] * 3.0 + 0.5
    sd['backbone.bn1.running_mean'] = sd['backbone.bn1.running_mean'] + 0.25
    torch.save(sd, tmp_path / 'parent.model')
    mt = train_meta.main(['with', 'YouTube-VOS', 'meta_batch_size=1', 'num_epochs.train=1', f'save_dir={tmp_path}', 'env_suffix=par',
                          f'parent_model.train.paths=[{tmp_path / "parent.model"}]'], height=96, width=160, num_meta_iters=1,
                         data_root=str(tmp_path / 'no_data'), eval_cmd=False)
    out = [json.loads(l) for l in capsys.readouterr().out.splitlines() if l.startswith('{')]
    assert {'parent_state': 'file', 'dataset_key': 'train'} in out
    a = mt.eng.debug_tensor('norm_a').flatten()[:64].cpu()
    b = mt.eng.debug_tensor('norm_b').flatten()[:64].cpu()
    ref_a = sd['backbone.bn1.weight'] / torch.sqrt(sd['backbone.bn1.running_var'] + 1e-5)
    ref_b = sd['backbone.bn1.bias'] - sd['backbone.bn1.running_mean'] * ref_a
    assert torch.allclose(a, ref_a, rtol=1e-6, atol=1e-7) and torch.allclose(b, ref_b, rtol=1e-5, atol=1e-6)
    syn = synthetic.synthetic_state('resnet50')
    assert not torch.allclose(a, syn['backbone.bn1.weight'] / torch.sqrt(syn['backbone.bn1.running_var'] + 1e-5), rtol=1e-3)


def test_meta_batch_with_two_frame_sizes_does_not_depend_on_which_size_is_primary():
    """The reference feeds every video at its native size (no resize in its data layer; DAVIS 480p is 854 or 910 wide, YouTube-VOS
    mostly 1280 x 720), so the tasks of one meta-batch differ in frame size.  `MetaTrainer` keeps a pool of engines per (H, W) that
    read the first engine's learned state: three tasks of two sizes through a trainer whose own engines have the first size, and
    through one whose engines have the second -- meta losses, the accumulated meta-gradient and the state after two outer steps are
    bit-identical, with one engine per size and with two in flight."""
    from eosvos_amd import synthetic
    from eosvos_amd.engine import Engine
    from eosvos_amd.meta_run import MetaTrainer
    sd, lrs = synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50')
    sizes = [(96, 160), (97, 163)]
    tasks = []
    for t, hw in enumerate([sizes[0], sizes[1], sizes[1]]):
        x, y = synthetic.synthetic_frames(1, *hw, seed=3000 + t)
        x, y = x.to(DEV), y.to(DEV)
        tasks.append((x, y, torch.flip(x, dims=[3]).contiguous(), torch.flip(y, dims=[3]).contiguous()))
    for n_engines in (1, 2):
        out = []
        for primary in sizes:
            engines = []
            for _ in range(n_engines):
                with torch.cuda.stream(torch.cuda.Stream()):
                    engines.append(Engine('resnet50', *primary, max_batch=1, device=DEV, side_stream=False))
            mt = MetaTrainer(engines[0], meta_batch_size=3, extra_engines=engines[1:])
            mt.load_state(sd, lrs)
            losses = [mt.meta_iteration(tasks, inner_steps=2) for _ in range(2)]
            torch.cuda.synchronize()
            pooled = {k: len(v) for k, v in mt._pool.items()}
            assert list(pooled) == [s for s in sizes if s != primary] and all(v <= n_engines for v in pooled.values()), pooled
            out.append((losses, mt.state.clone().cpu(), mt.exp_avg.clone().cpu()))
            mt._drop_pool()
            for e in reversed(engines):
                e.close()
        assert out[0][0] == out[1][0], (n_engines, out[0][0], out[1][0])
        assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2]), n_engines
        assert all(np.isfinite(v) for it in out[0][0] for v in it)


def test_model_keeps_the_engine_of_the_frame_size_it_just_left(model_and_optim):
    """Videos come at their native sizes; building an engine costs ~0.5 s (`tools/engine_create_time.py`).  The model parks the
    engine of the size it leaves (`DeepLabV3Plus._park_engine`, 2 kept) and takes it back for the next sequence of that size --
    with the learned state re-uploaded: the same logits bit for bit, also after a fine-tune step in between on the other size."""
    from eosvos_amd.helper_func import compute_loss
    model, mo, msd = model_and_optim
    mo.load_state_dict(msd)
    mo.reset()
    mo.eval()
    model.train_without_dropout()
    xa, ya = synthetic.synthetic_frames(1, 96, 160, seed=61)
    xb, yb = synthetic.synthetic_frames(1, 80, 128, seed=62)
    xa, ya, xb, yb = xa.to(DEV), ya.to(DEV), xb.to(DEV), yb.to(DEV)
    la = model(xa)[-1].clone()
    ea = model.engine
    lb = model(xb)[-1].clone()
    eb = model.engine
    assert eb is not ea and model._engine_cache.get((96, 160)) is ea
    assert torch.equal(model(xa)[-1], la) and model.engine is ea and model._engine_cache.get((80, 128)) is eb
    assert torch.equal(model(xb)[-1], lb) and model.engine is eb
    # a step on one size travels to the other size's engine (the fine-tuned weights are carried, networks._ensure_engine)
    out = model(xb)
    loss = compute_loss('cross_entropy', out[-1], yb)
    model.zero_grad()
    mo.set_train_loss(loss)
    mo.step(loss)
    l2 = model(xa)[-1].clone()
    assert model.engine is ea and not torch.equal(l2, la)
    mo.reset()                                     # theta <- learned init: both sizes back at the first logits
    assert torch.equal(model(xa)[-1], la) and torch.equal(model(xb)[-1], lb)
    model.close_engines()
