"""Plain DeepLabV3 (`src/networks/deeplabv3.py:10-83`, `init_parent_model(architecture='DeepLabV3')`,
`helper_func.py:343-344`): output stride 8, DeepLabHead = ASPP[12, 24, 36] -> 3x3 conv + BN + ReLU -> 1x1 conv,
logits resized x8.  Fixture G18 = layout, forward and a 3-step fine-tune trajectory of the reference class
(`tests/golden/make_golden.py --only g18`, through the torchvision stand-in of `_refshim.py`)."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from eosvos_amd import synthetic, topology

ENC = 'deeplabv3_resnet50'
BN_CFG = {'accum_stats': False, 'learn_weight': False, 'learn_bias': False}
MO_CFG = dict(init_lr=1e-3, learn_model_init=True, second_order_gradients=False, lr_hierarchy_level='NEURON',
              use_log_init_lr=False, max_lr=None)
SMALL = (96, 160)
FULL = (480, 854)


def fp(t):
    t = t.detach().double().flatten().cpu()
    idx = torch.linspace(0, t.numel() - 1, 16).long()
    return np.concatenate([[t.sum().item(), t.norm().item()], t[idx].numpy()])


def test_layout_python_c_oracle_vs_reference(golden_dir):
    from eosvos_amd import _ffi
    from oracle import topology as otopo
    g = json.load(open(os.path.join(golden_dir, 'g18_deeplabv3_layout.json')))
    assert [k for k, _ in g['keys']] == topology.model_state_keys(ENC) == otopo.state_dict_keys(ENC)
    assert [(n, tuple(s)) for n, s in g['trainable']] == [(n, tuple(s)) for n, s in topology.trainable(ENC)]
    assert otopo.trainable_names(ENC) == [n for n, _ in g['trainable']]
    lib = _ffi.load()
    arch = topology.ARCH_ID[ENC]
    convs = topology.conv_infos(ENC)
    assert lib.eosvos_num_convs(arch) == len(convs) == len(g['convs']) == 61
    for i, (c, ref) in enumerate(zip(convs, g['convs'])):
        info = (ctypes.c_int64 * 9)()
        assert lib.eosvos_conv_info(arch, i, info) == 0
        assert tuple(info[:8]) == (c.cin, c.cout, c.k, c.stride, c.dil, c.pad, int(c.norm is not None), int(c.bias))
        assert [c.name, c.cin, c.cout, c.k, c.stride, c.dil, c.pad, c.bias] == ref, (c, ref)
    # output stride 8: no stride in layer3 / layer4, dilations 1,2,2.. / 2,4,4
    by = {c.name: c for c in convs}
    assert by['backbone.layer3.0.conv1'].stride == 1 and by['backbone.layer3.0.downsample.0'].stride == 1
    assert [by[f'backbone.layer3.{i}.conv2'].dil for i in range(6)] == [1, 2, 2, 2, 2, 2]
    assert [by[f'backbone.layer4.{i}.conv2'].dil for i in range(3)] == [2, 4, 4]
    assert lib.eosvos_param_count(arch) == sum(int(np.prod(s)) for _, s in topology.trainable(ENC))


def test_oracle_forward_and_finetune_vs_reference(golden_dir):
    from oracle import deeplab, meta
    g = np.load(os.path.join(golden_dir, 'g18_deeplabv3.npz'))
    sd = synthetic.synthetic_state(ENC)
    x, _ = synthetic.synthetic_frames(2, *SMALL, seed=11)
    with torch.no_grad():
        lg = deeplab.forward(sd, x, ENC)
    assert float(np.abs(lg.numpy() - g['small_logits']).max()) < 1e-4
    batches = [synthetic.synthetic_frames(2, *SMALL, seed=11 + it) for it in range(3)]
    losses, P = meta.finetune(sd, synthetic.synthetic_lrs(ENC), batches, encoder=ENC)
    np.testing.assert_allclose(losses, g['small_losses'], rtol=2e-5)
    with torch.no_grad():
        fl = deeplab.forward(P, batches[0][0], ENC)
    assert float(np.abs(fl.numpy() - g['small_final_logits']).max()) < 2e-4


def test_init_parent_model_architecture_deeplabv3():
    from eosvos_amd.helper_func import init_parent_model
    from eosvos_amd.networks import DeepLabV3
    model, ps = init_parent_model(architecture='DeepLabV3', encoder='resnet50', train_encoder=True, batch_norm=BN_CFG)
    assert isinstance(model, DeepLabV3) and model.encoder == ENC and ps == {}
    assert list(model.state_dict()) == topology.model_state_keys(ENC)
    with pytest.raises(NotImplementedError):
        init_parent_model(architecture='MaskRCNN', encoder='resnet50', train_encoder=True)


@pytest.mark.gpu
def test_gpu_forward_and_finetune_vs_reference(golden_dir):
    """Forward (96x160 elementwise, 480x854 fingerprint + mask bits) and the 3-step fine-tune trajectory (losses, first-step
    gradients per tensor, parameters after 3 steps, final logits) through the drop-in classes."""
    from eosvos_amd.helper_func import compute_loss, init_parent_model
    from eosvos_amd.meta_optim import MetaOptimizer
    g = np.load(os.path.join(golden_dir, 'g18_deeplabv3.npz'))
    dev = 'cuda:0'
    model, _ = init_parent_model(architecture='DeepLabV3', encoder='resnet50', train_encoder=True, batch_norm=BN_CFG)
    model.to(dev)
    sd = synthetic.synthetic_state(ENC)
    model.load_state_dict(sd)
    model.eval()
    x, _ = synthetic.synthetic_frames(2, *SMALL, seed=11)
    lg = model(x.to(dev))[-1].cpu()
    assert float(np.abs(lg.numpy() - g['small_logits']).max()) < 1e-4
    mo = MetaOptimizer(model, **MO_CFG)
    msd = {}
    for (n, _), lr in zip(topology.trainable(ENC), synthetic.synthetic_lrs(ENC)):
        msd['log_init_lr_' + n.replace('.', '-')] = lr.clone()
    for n, _ in topology.trainable(ENC):
        msd['model_init_' + n.replace('.', '-')] = sd[n].clone()
    mo.load_state_dict(msd)
    mo.reset()
    mo.eval()
    model.train_without_dropout()
    model._ensure_engine(*SMALL, 2).keep_grads(True)
    losses, grads = [], None
    for it in range(3):
        xb, yb = synthetic.synthetic_frames(2, *SMALL, seed=11 + it)
        loss = compute_loss('cross_entropy', model(xb.to(dev))[-1], yb.to(dev))
        losses.append(float(loss))
        model.zero_grad()
        mo.set_train_loss(loss)
        mo.step(loss)
        mo.meta_model.detach_param_groups()
        if grads is None:
            grads = model.engine.get_grads().cpu()
    np.testing.assert_allclose(losses, g['small_losses'], rtol=5e-5)
    tr = topology.trainable(ENC)
    offs = np.cumsum([0] + [int(np.prod(s)) for _, s in tr])
    names = [n for n, _ in tr]
    for i, (n, s) in enumerate(tr):
        l2 = float(grads[offs[i]:offs[i + 1]].double().norm())
        r = g['small_grad_fp'][i][1]
        assert abs(l2 - r) <= 1e-3 * r + 1e-9, (n, l2, r)
    for nm in ('classifier.1.weight', 'classifier.4.weight', 'classifier.4.bias', 'backbone.layer3.1.conv2.weight'):
        i = names.index(nm)
        mine = grads[offs[i]:offs[i + 1]].view(*tr[i][1])[::8].numpy()
        ref = g['small_grad_' + nm]
        assert np.abs(mine - ref).max() <= 1e-3 * np.abs(ref).max(), nm
    cur = model.state_dict()
    for i, (n, s) in enumerate(tr):
        r = g['small_param_fp'][i]
        assert abs(float(cur[n].double().norm()) - r[1]) <= 3e-6 * r[1] + 1e-9, n
    model.eval()
    fl = model(synthetic.synthetic_frames(2, *SMALL, seed=11)[0].to(dev))[-1].cpu()
    assert float(np.abs(fl.numpy() - g['small_final_logits']).max()) < 1e-3          # north_star
    # 480 x 854 forward (batch 1) against the reference's fingerprint, mask bits exact outside |logit| < 1e-3
    model.load_state_dict(sd)
    xf, _ = synthetic.synthetic_frames(1, *FULL, seed=7)
    lf = model(xf.to(dev))[-1].cpu()
    assert float(np.abs(lf[0, 0, ::8, ::7].numpy() - g['full_logits_sub']).max()) < 1e-4
    bits = np.packbits((lf >= 0).numpy().astype(np.uint8))
    assert int(np.unpackbits(bits ^ g['full_mask']).sum()) <= int(g['full_near_zero'][0])
    model.engine.close()
