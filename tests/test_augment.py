"""First-frame augmentation: the numpy restatement of cv2.warpAffine (oracle/augment.py,
parity UNPINNED: OpenCV is not available here) is cross-checked against an independent implementation of the
same bicubic kernel (torch grid_sample, A = -0.75) and against exact properties; the HIP kernel is compared
with the restatement through the C-ABI."""
import math
import os
import random

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import augment

SMALL = (96, 160)
FULL = (480, 854)
DEV = 'cuda:0'


def _smooth_image(H, W, C=3, seed=0):
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(1, C, H // 8 + 2, W // 8 + 2, generator=g)
    return F.interpolate(low, size=(H, W), mode='bicubic', align_corners=True)[0].clamp(0, 1).permute(1, 2, 0).contiguous().numpy()


def _label(H, W):
    gt = np.zeros((H, W), np.float32)
    gt[H // 4: H // 2, W // 3: 2 * W // 3] = 1
    return gt


def test_cubic_table_properties():
    tab = augment.cubic_table()
    assert tab.shape == (32, 4) and tab.dtype == np.float32
    np.testing.assert_allclose(tab.sum(1), 1.0, atol=1e-6)
    assert tuple(tab[0]) == (0.0, 1.0, 0.0, 0.0)
    np.testing.assert_allclose(tab[16], [-0.09375, 0.59375, 0.59375, -0.09375])       # Keys kernel, a = -0.75, x = .5
    np.testing.assert_allclose(tab[1:][:, ::-1], tab[1:][::-1], atol=1e-6)             # symmetry w(x) = w(1-x) reversed


def test_identity_and_pure_translation_are_exact():
    img, gt = _smooth_image(*SMALL), _label(*SMALL)
    H, W = SMALL
    M = augment.get_rotation_matrix_2d((W / 2, H / 2), 0.0, 1.0)
    assert np.array_equal(augment.warp_affine(img, M, 'cubic'), img)
    assert np.array_equal(augment.warp_affine(gt, M, 'nearest'), gt)
    Mt = np.array([[1, 0, 7], [0, 1, -3]], np.float64)      # dst(x, y) = src(x - 7, y + 3)
    out = augment.warp_affine(gt, Mt, 'nearest')
    ref = np.zeros_like(gt)
    ref[:H - 3, 7:] = gt[3:, :W - 7]
    assert np.array_equal(out, ref)


@pytest.mark.parametrize('rot,sc', [(17.0, 1.1), (-28.5, 0.8), (5.25, 1.25)])
def test_restatement_vs_independent_bicubic(rot, sc):
    """Same Keys kernel in torch.grid_sample: differences come only from cv2's 1/32-pixel coordinate grid."""
    H, W = SMALL
    img = _smooth_image(H, W)
    M = augment.get_rotation_matrix_2d((W / 2, H / 2), rot, sc)
    out = augment.warp_affine(img, M, 'cubic')
    A = np.vstack([M, [0, 0, 1]])
    Ai = np.linalg.inv(A)
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    u = Ai[0, 0] * xs + Ai[0, 1] * ys + Ai[0, 2]
    v = Ai[1, 0] * xs + Ai[1, 1] * ys + Ai[1, 2]
    grid = torch.from_numpy(np.stack([2 * u / (W - 1) - 1, 2 * v / (H - 1) - 1], -1)).float()[None]
    ref = F.grid_sample(torch.from_numpy(img).permute(2, 0, 1)[None], grid, mode='bicubic', padding_mode='zeros',
                        align_corners=True)[0].permute(1, 2, 0).numpy()
    inner = (u > 2) & (u < W - 3) & (v > 2) & (v < H - 3)
    assert inner.mean() > 0.3
    assert np.abs(out - ref)[inner].max() < 6e-3          # smooth image, <= 1/64 pixel coordinate rounding
    assert np.abs(out - ref)[inner].mean() < 1e-3
    lab = augment.warp_affine(_label(H, W), M, 'nearest')
    refl = F.grid_sample(torch.from_numpy(_label(H, W))[None, None], grid, mode='nearest', padding_mode='zeros',
                         align_corners=True)[0, 0].numpy()
    assert (lab != refl).mean() < 0.01                    # only pixels whose source falls on a rounding tie / edge


def test_reference_draw_order_and_retry():
    """flip draw first, then (rot, sc) pairs until the label keeps its object (custom_transforms.py:53-78,200)."""
    H, W = SMALL
    img, gt = _smooth_image(H, W), _label(H, W)
    random.seed(11)
    r = [random.random() for _ in range(3)]
    random.seed(11)
    _, aug, p = augment.random_flip_scale_rotate(img, gt)
    assert p['flip'] == (r[0] < 0.5) and p['tries'] == 1
    assert p['rot'] == 60 * r[1] - 30 and p['sc'] == 0.5 * r[2] - 0.25 + 1
    assert 0 < aug.sum() < aug.size
    tiny = np.zeros((H, W), np.float32)
    tiny[0, 0] = 1                                        # a corner pixel is rotated out of the canvas by most draws

    class Seq:
        def __init__(self, vals): self.vals = list(vals)
        def random(self): return self.vals.pop(0)
    _, aug, p = augment.random_flip_scale_rotate(img, tiny, rng=Seq([0.9, 0.95, 0.999, 0.5, 0.5]))
    assert p['tries'] == 2 and p['rot'] == 0.0 and p['sc'] == 1.0 and aug[0, 0] == 1


# ---- HIP kernel vs the restatement, through the C-ABI ------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('size', [SMALL, FULL])
def test_warp_affine_vs_restatement(size):
    from eosvos_amd.custom_transforms import INTER_CUBIC, INTER_NEAREST, warp_affine
    from eosvos_amd.engine import Engine
    H, W = size
    eng = Engine('resnet50', H, W, max_batch=1, device=DEV)
    img, gt = _smooth_image(H, W), _label(H, W)
    timg = torch.from_numpy(img).permute(2, 0, 1).contiguous().to(DEV)
    tgt = torch.from_numpy(gt)[None].contiguous().to(DEV)
    for flip, rot, sc in [(0, 0.0, 1.0), (0, 14.5, 1.15), (1, -29.0, 0.76), (1, 3.0, 1.25), (0, 30.0, 0.75)]:
        src_i = img[:, ::-1].copy() if flip else img
        src_g = gt[:, ::-1].copy() if flip else gt
        ref_g = augment.rot_and_sc(src_g, rot, sc, True)
        ref_i = augment.rot_and_sc(src_i, rot, sc, False)
        lab, nz = warp_affine(eng, tgt, flip, rot, sc, INTER_NEAREST, count_nonzero=True)
        assert np.array_equal(lab[0].cpu().numpy(), ref_g)                 # integer coordinates: bit-exact
        assert nz == int((ref_g != 0).sum())
        out, _ = warp_affine(eng, timg, flip, rot, sc, INTER_CUBIC)
        got = out.permute(1, 2, 0).cpu().numpy()
        assert np.abs(got - ref_i).max() <= 2e-6, np.abs(got - ref_i).max()   # same taps and weights; fma contraction only
    eng.close()


@pytest.mark.gpu
def test_warp_affine_of_a_frame_whose_size_is_not_the_engines():
    """`eosvos_warp_affine_hw` (round 5): the tasks of a meta-batch come at their videos' native sizes, so a frame is warped at ITS
    size whatever size the engine was built for -- equal to the restatement, and bit-identical to the warp through an engine of that
    size."""
    from eosvos_amd.custom_transforms import INTER_CUBIC, INTER_NEAREST, warp_affine
    from eosvos_amd.engine import Engine
    H, W = 97, 163
    other = Engine('resnet50', *SMALL, max_batch=1, device=DEV)
    own = Engine('resnet50', H, W, max_batch=1, device=DEV)
    img, gt = _smooth_image(H, W), _label(H, W)
    timg = torch.from_numpy(img).permute(2, 0, 1).contiguous().to(DEV)
    tgt = torch.from_numpy(gt)[None].contiguous().to(DEV)
    for flip, rot, sc in [(0, 14.5, 1.15), (1, -29.0, 0.76)]:
        src_i = img[:, ::-1].copy() if flip else img
        src_g = gt[:, ::-1].copy() if flip else gt
        lab, nz = warp_affine(other, tgt, flip, rot, sc, INTER_NEAREST, count_nonzero=True)
        ref_g = augment.rot_and_sc(src_g, rot, sc, True)
        assert np.array_equal(lab[0].cpu().numpy(), ref_g) and nz == int((ref_g != 0).sum())
        out, _ = warp_affine(other, timg, flip, rot, sc, INTER_CUBIC)
        assert np.abs(out.permute(1, 2, 0).cpu().numpy() - augment.rot_and_sc(src_i, rot, sc, False)).max() <= 2e-6
        assert torch.equal(out, warp_affine(own, timg, flip, rot, sc, INTER_CUBIC)[0])
    other.close()
    own.close()


@pytest.mark.gpu
def test_first_frame_augmenter_matches_reference_sequence():
    from eosvos_amd.custom_transforms import FirstFrameAugmenter
    from eosvos_amd.engine import Engine
    H, W = SMALL
    eng = Engine('resnet50', H, W, max_batch=3, device=DEV)
    img, gt = _smooth_image(H, W), _label(H, W)
    timg = torch.from_numpy(img).permute(2, 0, 1).contiguous().to(DEV)
    tgt = torch.from_numpy(gt)[None].contiguous().to(DEV)
    random.seed(123)
    images, labels, params = FirstFrameAugmenter(eng).batch(timg, tgt, 3)
    random.seed(123)
    for b in range(3):
        ri, rg, rp = augment.random_flip_scale_rotate(img, gt)
        assert rp == params[b]
        assert np.array_equal(labels[b, 0].cpu().numpy(), rg)
        assert np.abs(images[b].permute(1, 2, 0).cpu().numpy() - ri).max() <= 2e-6
    # the augmented batch feeds the fine-tune step directly
    eng.load_model_state(__import__('eosvos_amd.synthetic', fromlist=['x']).synthetic_state('resnet50'),
                         __import__('eosvos_amd.synthetic', fromlist=['x']).synthetic_lrs('resnet50'))
    loss = eng.finetune_step(images, labels)
    assert math.isfinite(loss)
    eng.close()


@pytest.mark.gpu
def test_meta_taskset_materialises_tasks_on_the_device(tmp_path):
    """`MetaTaskset.task_tensors`: train frame x batch + meta frames in HBM, ONE flip / scale-rotate per task applied to
    frames (bicubic) and labels (nearest) alike (`meta_tasksets.py:111-137`, deterministic transforms)."""
    import random
    import numpy as np
    from PIL import Image
    from eosvos_amd import config as config_mod
    from eosvos_amd.data import DAVIS
    from eosvos_amd.engine import Engine
    from eosvos_amd.meta_tasksets import MetaTaskset
    root = tmp_path / 'DAVIS-2017'
    rng = np.random.default_rng(0)
    (root / 'JPEGImages' / '480p' / 'bear').mkdir(parents=True)
    (root / 'Annotations' / '480p' / 'bear').mkdir(parents=True)
    for f in range(4):
        Image.fromarray(rng.integers(0, 256, (96, 160, 3), dtype=np.uint8)).save(root / 'JPEGImages' / '480p' / 'bear' / f'{f:05d}.jpg')
        lab = np.zeros((96, 160), np.uint8)
        lab[30:60, 50 + f:100 + f] = 1
        Image.fromarray(lab, mode='L').save(root / 'Annotations' / '480p' / 'bear' / f'{f:05d}.png')
    (root / 'train_seqs.txt').write_text('bear\n')
    ds = DAVIS('train_seqs', str(root), multi_object='single_id')
    cfg = config_mod.parse_cli(['with', 'DAVIS-2017'])
    eng = Engine('resnet50', 96, 160, max_batch=1, device='cuda:0')
    ts = MetaTaskset(ds, cfg['data_cfg'], random_frame_transform_per_task=True)
    torch.manual_seed(0)
    random.seed(0)
    item = ts[0]
    xt, yt, xm, ym = ts.task_tensors(item, eng, 'cuda:0')
    assert xt.shape == (1, 3, 96, 160) and yt.shape == (1, 1, 96, 160) and xm.shape == (1, 3, 96, 160) and xt.is_cuda
    assert set(yt.unique().tolist()) <= {0.0, 1.0} and 0 < float(yt.sum()) < yt.numel()          # the object survives the warp
    assert 0 < float(ym.sum()) < ym.numel() and float(xt.min()) >= -0.2 and float(xt.max()) <= 1.2
    tr = item['transform']
    assert item['train_frame'] in tr['rot_sc'] and all(0.75 <= sc <= 1.25 and -30 <= rot <= 30 for rot, sc in tr['rot_sc'].values())
    # without the per-task transform the tensors are the decoded frame / mask themselves
    ts0 = MetaTaskset(ds, cfg['data_cfg'], random_frame_transform_per_task=False)
    it0 = ts0[0]
    x0, y0, _, _ = ts0.task_tensors(it0, eng, 'cuda:0')
    ds.multi_object_id = 0
    img, lab = ds.make_img_label_pair(it0['train_frame'])
    assert torch.equal(x0[0].cpu(), torch.from_numpy(img.transpose(2, 0, 1))) and torch.equal(y0[0, 0].cpu(), torch.from_numpy(lab))
    eng.close()


@pytest.mark.gpu
def test_cubic_warp_is_stable_beside_another_engine():
    """Regression: with its taps paired into packed-fp32 instructions the bicubic warp lost one tap in lanes 48..63 of
    a wave whenever another engine's bf16x6 conv kernels shared the SIMD (20-30 % of 96x160 warps; first seen as
    run-to-run differences of objects fine-tuned side by side).  The warp of one engine must be bit-stable while a
    second engine, on its own stream, runs forward passes."""
    from eosvos_amd import synthetic
    from eosvos_amd.custom_transforms import INTER_CUBIC, warp_affine
    from eosvos_amd.engine import Engine
    H, W = SMALL
    sd, lrs = synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50')
    x, _ = synthetic.synthetic_frames(3, H, W, seed=5)
    xg = x.to(DEV)
    e0 = Engine('resnet50', H, W, max_batch=3, device=DEV)
    with torch.cuda.stream(torch.cuda.Stream()):
        e1 = Engine('resnet50', H, W, max_batch=1, device=DEV)
    e0.load_model_state(sd, lrs)
    src = xg[0].clone()
    torch.cuda.synchronize()
    with torch.cuda.stream(e1.stream):
        ref = warp_affine(e1, src, 0, 17.0, 1.1, INTER_CUBIC)[0].clone()
    torch.cuda.synchronize()
    ref_np = augment.rot_and_sc(src.permute(1, 2, 0).cpu().numpy(), 17.0, 1.1, False)
    assert np.abs(ref.permute(1, 2, 0).cpu().numpy() - ref_np).max() <= 2e-6
    wrong = 0
    for _ in range(150):
        e0.forward(xg, want_logits=False)                      # enqueued on e0's stream, not waited for
        with torch.cuda.stream(e1.stream):
            got = warp_affine(e1, src, 0, 17.0, 1.1, INTER_CUBIC)[0].clone()
        torch.cuda.synchronize()
        wrong += int(not torch.equal(got, ref))
    assert wrong == 0, f'{wrong} of 150 warps differ'
    e0.close()
    e1.close()
