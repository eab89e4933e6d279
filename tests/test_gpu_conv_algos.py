"""Op-level parity of every convolution path on the GPU, through the C-ABI (eosvos_test_conv_algo /
eosvos_test_conv_bwd_algo drive the production conv_fwd / conv_dgrad / conv_wgrad): implicit GEMM, Winograd
F(2x2,3x3), F(4x4,3x3) and their dilated sub-grid forms, forward + data gradient + weight gradient compared
ELEMENTWISE with an fp64 torch convolution of the same op (torchvision Bottleneck / ASPP / decoder convs,
SURVEY 2.2 K3/K4).  Also pins the accuracy claim of the bf16x6 matrix mode against the fp32 MFMA.
"""
import pytest
import torch
import torch.nn.functional as F

from eosvos_amd import engine as engine_mod
from eosvos_amd.engine import Engine

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def eng():
    e = Engine('resnet50', 96, 160, max_batch=1, device=DEV)
    yield e
    e.close()


def _maxrel(a, ref):
    a, ref = a.double().cpu(), ref.double()
    return float((a - ref).abs().max() / ref.abs().max())


def _rmsrel(a, ref):
    a, ref = a.double().cpu(), ref.double()
    return float(((a - ref).pow(2).mean() / ref.pow(2).mean()).sqrt())


def _case(case, seed):
    B, H, W, Ci, Co, d = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Ci, H, W, generator=g)
    x = x * (torch.rand(B, Ci, H, W, generator=g) > 0.3)          # a ReLU output: exact zeros, mask = x > 0
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    a = torch.rand(Co, generator=g) + 0.5
    b = torch.randn(Co, generator=g) * 0.1
    gy = torch.randn(B, Co, H, W, generator=g)
    return x, w, a, b, gy


def _reference(x, w, a, b, gy, d):
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    y = F.conv2d(xd, wd, None, 1, d, d)
    out = F.relu(y * a.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1))
    y.backward(gy.double() * a.double().view(1, -1, 1, 1))      # engine gradients are w.r.t. the post-norm, pre-ReLU output
    dx = xd.grad * (x > 0)                                       # ReLU mask of the conv input
    return out.detach(), dx, wd.grad


nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV)

# (B, H, W, Cin, Cout, dilation)  -- shapes of the network's Winograd layers and awkward borders
WINO_CASES = [
    ('wino_f4', (1, 120, 214, 304, 256, 1)),      # decoder.last_conv.0 at 480x854 (tail launch for the 304 input channels; data-gradient GEMMs streamed)
    ('wino_f4', (1, 120, 214, 256, 256, 1)),      # decoder.last_conv.1: all three plane GEMMs on the streaming kernel (128-channel ranges)
    ('wino_f4', (3, 30, 54, 256, 256, 1)),        # batch 3
    ('wino_f4', (1, 30, 54, 512, 512, 2)),        # layer4 conv2, d = 2, 4, 8: F(4,3) on the d*d sub-grids
    ('wino_f4', (1, 30, 54, 512, 512, 4)),
    ('wino_f4', (1, 30, 54, 512, 512, 8)),
    ('wino_f2', (1, 30, 54, 512, 512, 8)),
    ('wino_f2', (1, 30, 54, 2048, 256, 6)),       # ASPP d = 6
    ('wino_f4', (1, 97, 161, 64, 64, 1)),         # 97 = 24*4 + 1, 161 = 40*4 + 1: tiles straddle both borders
    ('wino_f2', (1, 97, 161, 64, 64, 1)),
    ('wino_f2', (2, 31, 55, 128, 64, 3)),         # odd sub-grids: (11,10,10) x (19,18,18)
    ('wino_f4', (1, 33, 57, 64, 128, 2)),
    ('wino_f2', (1, 5, 7, 64, 64, 1)),            # smaller than one F(4,3) tile row
    ('direct', (1, 30, 54, 2048, 256, 12)),       # tap tables (d = 12, 18 keep the implicit-GEMM kernel)
    ('direct', (1, 30, 54, 2048, 256, 18)),
    ('direct', (1, 120, 214, 304, 256, 1)),
]
# measured maxima (profiles/r02_conv_algo_margins.txt): direct 3e-7, F(2,3) 1e-6, F(4,3) 8e-6 of the output scale
TOL = {'direct': 3e-6, 'wino_f2': 6e-6, 'wino_f4': 2.5e-5}


@pytest.mark.parametrize('algo,case', WINO_CASES)
def test_conv_paths_elementwise_vs_fp64(eng, algo, case):
    x, w, a, b, gy = _case(case, seed=sum(case))
    out_ref, dx_ref, dw_ref = _reference(x, w, a, b, gy, case[5])
    d = case[5]
    out = eng.test_conv_algo(algo, nhwc(x), w.to(DEV), a.to(DEV), b.to(DEV), None, True, 1, d, d)
    dx, dw = eng.test_conv_bwd_algo(algo, nhwc(x), w.to(DEV), nhwc(gy), 1, d, d, scale=a.to(DEV), mask=nhwc(x))
    errs = (_maxrel(out.permute(0, 3, 1, 2), out_ref), _maxrel(dx.permute(0, 3, 1, 2), dx_ref), _maxrel(dw, dw_ref))
    print(f'MARGIN {algo} {case} fwd {errs[0]:.2e} dx {errs[1]:.2e} dw {errs[2]:.2e}')
    assert max(errs) <= TOL[algo], errs
    # ReLU mask of the data gradient is exact
    assert bool((dx.permute(0, 3, 1, 2).cpu()[x <= 0] == 0).all())


def test_forced_algo_rejects_ineligible_shapes(eng):
    x = torch.randn(1, 8, 8, 64, device=DEV)
    w = torch.randn(64, 64, 3, 3, device=DEV)
    from eosvos_amd._ffi import EosvosError
    with pytest.raises(EosvosError):
        eng.test_conv_algo('wino_f4', x, w, None, None, None, False, 2, 1, 1)          # stride 2 has no Winograd form
    w1 = torch.randn(64, 64, 1, 1, device=DEV)
    with pytest.raises(EosvosError):
        eng.test_conv_algo('wino_f2', x, w1, None, None, None, False, 1, 1, 0)


STRIDED_CASES = [          # (B, H, W, Cin, Cout, k, stride, dil, pad): the non-Winograd production paths
    (3, 60, 107, 512, 256, 1, 2, 1, 0),           # layer3.0.conv1: coarse-grid gradient scattered to the even pixels
    (1, 120, 214, 128, 128, 3, 2, 1, 1),          # layer2.0.conv2: parity-major data gradient
    (2, 30, 54, 1280, 256, 1, 1, 1, 0),           # ASPP projection
    (1, 120, 214, 256, 48, 1, 1, 1, 0),           # decoder.conv1 (N = 48)
    (3, 120, 214, 128, 128, 3, 2, 1, 1),          # layer2.0.conv2 at batch 3: 602 uneven tiles -> whole tiles, longest first
    (3, 30, 54, 256, 256, 3, 1, 1, 1),            # layer3 conv2 at batch 3: 76 tiles, K = 2304 -> chunk-major split-K (x6)
    (3, 30, 54, 2048, 512, 1, 1, 1, 0),           # layer4 conv1 at batch 3: 152 tiles, K = 2048 -> split-K (x3)
    # short-K 1x1 convs on the large maps: the streaming kernel (conv1x1_stream_kernel, f16x3 mode), forward = K Cin, data
    # gradient = K Cout
    (3, 120, 214, 64, 256, 1, 1, 1, 0),           # layer1 conv3 / downsample (fwd K 64, N 256; dgrad K 256, N 64)
    (3, 120, 214, 256, 64, 1, 1, 1, 0),           # layer1 conv1 of blocks 1, 2
    (1, 120, 214, 64, 64, 1, 1, 1, 0),            # layer1.0.conv1 at batch 1 (25 680 pixels: a ragged last strip)
    (3, 60, 107, 128, 512, 1, 1, 1, 0),           # layer2 conv3 (19 260 pixels, 4 column ranges; data gradient K = 512)
    (3, 120, 214, 256, 128, 1, 1, 1, 0),          # layer2.0.conv1
    (3, 60, 107, 512, 128, 1, 1, 1, 0),           # layer2 conv1 of blocks 1-3: K = 512 (K-outer form, 64-channel ranges)
    # layer1's 3x3 convs: conv3x3_stream_kernel (nine taps' weights resident in LDS, shifted activations from global memory)
    (3, 120, 214, 64, 64, 3, 1, 1, 1),
    (1, 120, 214, 64, 64, 3, 1, 1, 1),            # 25 680 pixels: a ragged last strip, image borders inside strips
]


@pytest.mark.parametrize('case', STRIDED_CASES)
def test_strided_and_narrow_paths_elementwise_vs_fp64(eng, case):
    B, H, W, Ci, Co, k, s, d, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y = F.conv2d(xd, wd, None, s, p, d)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy.double())
    out = eng.test_conv_algo('auto', nhwc(x), w.to(DEV), None, None, None, False, s, d, p)
    dx, dw = eng.test_conv_bwd_algo('auto', nhwc(x), w.to(DEV), nhwc(gy), s, d, p)
    errs = (_maxrel(out.permute(0, 3, 1, 2), y.detach()), _maxrel(dx.permute(0, 3, 1, 2), xd.grad), _maxrel(dw, wd.grad))
    print(f'MARGIN auto {case} fwd {errs[0]:.2e} dx {errs[1]:.2e} dw {errs[2]:.2e}')
    assert max(errs) <= 2e-6, errs


def test_streaming_1x1_kernel_fused_epilogue_vs_fp64(eng):
    """The streaming kernel's fused forward epilogue -- norm scale + shift, residual, ReLU -- on layer1 conv3's shape at
    batch 3 (77 040 pixels, 64 -> 256), elementwise against fp64; every matrix mode gives the same answer to rounding (the
    other two run the tiled kernel)."""
    B, H, W, Ci, Co = 3, 120, 214, 64, 256
    g = torch.Generator().manual_seed(11)
    x = torch.relu(torch.randn(B, Ci, H, W, generator=g))
    w = torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5
    a, b = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.1
    res = torch.randn(B, Co, H, W, generator=g)
    ref = F.relu(F.conv2d(x.double(), w.double()) * a.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1) + res.double())
    prev = engine_mod.get_matrix_mode()
    try:
        for mode in ('f16x3', 'bf16x6'):
            engine_mod.set_matrix_mode(mode)
            out = eng.test_conv_algo('direct', nhwc(x), w.to(DEV), a.to(DEV), b.to(DEV), nhwc(res), True, 1, 1, 0).permute(0, 3, 1, 2)
            err = _maxrel(out, ref)
            print(f'MARGIN streaming 1x1 fused epilogue {mode}: {err:.2e}')
            assert err <= 2e-6, (mode, err)
    finally:
        engine_mod.set_matrix_mode(prev)


@pytest.mark.parametrize('shape', [(2, 480, 854), (1, 97, 163), (3, 99, 165), (3, 96, 160)], ids=lambda s: 'x'.join(map(str, s)))
def test_stem_on_the_matrix_cores_vs_fp64(shape):
    """f16x3 mode: the 7x7 stride-2 stem runs on the fp16 matrix cores (`stem_fwd_h3_kernel`; 8-float slots of the padded
    NHWC3 frame, the weight panel's absmax taken in the kernel, the frame's in the layout pass).  Its output (conv + frozen
    norm + ReLU, the engine's `c1`) elementwise against fp64, on the full frame, an odd-sized one (slots run 2 floats past a
    padded row) and a small batch-3 one; the bf16x6 mode (VALU kernel) gives the same answer to rounding."""
    from eosvos_amd import synthetic
    B, H, W = shape
    sd = synthetic.synthetic_state('resnet50')
    lrs = synthetic.synthetic_lrs('resnet50')
    x, _ = synthetic.synthetic_frames(B, H, W, seed=5)
    w = sd['backbone.conv1.weight'].double()
    g, be = sd['backbone.bn1.weight'].double(), sd['backbone.bn1.bias'].double()
    mu, var = sd['backbone.bn1.running_mean'].double(), sd['backbone.bn1.running_var'].double()
    a = g / torch.sqrt(var + 1e-5)
    ref = F.relu(F.conv2d(x.double(), w, None, 2, 3) * a.view(1, -1, 1, 1) + (be - mu * a).view(1, -1, 1, 1))
    prev = engine_mod.get_matrix_mode()
    try:
        for mode in ('f16x3', 'bf16x6'):
            engine_mod.set_matrix_mode(mode)
            e = Engine('resnet50', H, W, max_batch=B, device=DEV)
            e.load_model_state(sd, lrs)
            e._verify_pending = False          # range guard off: the mode under test must produce this c1 itself
            assert e.matrix_mode == mode
            e.forward(x.to(DEV), want_logits=False)
            c1 = e.debug_tensor('c1').cpu()[:B]
            e.close()
            err = _maxrel(c1, ref)
            print(f'MARGIN stem {shape} {mode}: {err:.2e}')
            assert err <= 2e-6, (mode, err)
    finally:
        engine_mod.set_matrix_mode(prev)


@pytest.mark.parametrize('shape', [(2, 480, 854), (1, 97, 163), (3, 99, 165)], ids=lambda s: 'x'.join(map(str, s)))
def test_stem_weight_gradient_on_the_matrix_cores_vs_fp64(shape):
    """f16x3 mode: the stem's weight gradient (`stem_wgrad_h3_kernel`: 307 440 pixels at batch 3 reduced in 512 slabs) against
    an fp64 weight gradient of the same operands -- the engine's own d(loss)/d(stem output) and the frame --, elementwise; the
    bf16x6 mode (fp32-MFMA stem kernel) for comparison.  The parameter gradient the engine exports carries the frozen-norm
    scale of the stem's BatchNorm."""
    from eosvos_amd import synthetic
    B, H, W = shape
    sd = synthetic.synthetic_state('resnet50')
    lrs = synthetic.synthetic_lrs('resnet50')
    x, y = synthetic.synthetic_frames(B, H, W, seed=9)
    a = (sd['backbone.bn1.weight'].double() / torch.sqrt(sd['backbone.bn1.running_var'].double() + 1e-5)).view(1, -1, 1, 1)
    wshape = sd['backbone.conv1.weight'].shape
    prev = engine_mod.get_matrix_mode()
    try:
        for mode in ('f16x3', 'bf16x6'):
            engine_mod.set_matrix_mode(mode)
            e = Engine('resnet50', H, W, max_batch=B, device=DEV)
            e.load_model_state(sd, lrs)
            e._verify_pending = False          # range guard off (see above)
            e.keep_grads(True)
            e.finetune_step(x.to(DEV), y.to(DEV))
            assert e.matrix_mode == mode
            g = e.get_grads().cpu()[:wshape.numel()].view(wshape)          # conv 0 is the first tensor of the flat vector
            gc1 = e.debug_tensor('g_c1').cpu()[:B].double()
            e.close()
            ref = torch.nn.grad.conv2d_weight(x.double(), wshape, gc1 * a, stride=2, padding=3)
            err = _maxrel(g, ref)
            print(f'MARGIN stem wgrad {shape} {mode}: {err:.2e}')
            assert err <= 3e-6, (mode, err)
    finally:
        engine_mod.set_matrix_mode(prev)


def test_bf16x6_is_at_least_as_accurate_as_the_fp32_mfma(eng):
    """The default matrix mode splits every fp32 operand exactly into 3 bf16 pieces and sums the 6 leading partial
    products in fp32; its error against fp64 must not exceed the fp32 MFMA's (both ~1e-7 of sum|a*b|)."""
    g = torch.Generator().manual_seed(5)
    B, H, W, Ci, Co = 1, 30, 54, 2048, 512
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5
    ref = F.conv2d(x.double(), w.double())
    sabs = F.conv2d(x.double().abs(), w.double().abs())
    gy = torch.randn(B, Co, H, W, generator=g)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    F.conv2d(xd, wd).backward(gy.double())
    res = {}
    prev = engine_mod.get_matrix_mode()
    try:
        for mode in ('f32', 'bf16x6', 'f16x3'):
            engine_mod.set_matrix_mode(mode)
            assert engine_mod.get_matrix_mode() == mode
            out = eng.test_conv_algo('direct', nhwc(x), w.to(DEV), None, None, None, False, 1, 1, 0).permute(0, 3, 1, 2)
            dx, dw = eng.test_conv_bwd_algo('direct', nhwc(x), w.to(DEV), nhwc(gy), 1, 1, 0)
            res[mode] = (float(((out.double().cpu() - ref).abs() / sabs).max()), _rmsrel(out, ref),
                         _rmsrel(dx.permute(0, 3, 1, 2), xd.grad), _rmsrel(dw, wd.grad))
    finally:
        engine_mod.set_matrix_mode(prev)
    print('MARGIN matrix modes (max err/sum|ab|, rms rel fwd, dx, dw):', res)
    for mode in ('bf16x6', 'f16x3'):          # both split modes: no worse than the fp32 MFMA
        for i in range(4):
            assert res[mode][i] <= 1.25 * res['f32'][i] + 1e-9, res
        assert res[mode][0] < 3e-7 and res[mode][1] < 1e-6


def test_bf16x6_special_values_propagate(eng):
    """Non-finite and denormal operands in the split mode vs the fp32-MFMA mode (the NaN-skip rule of the meta loop,
    `src/util/meta_run.py:209-211`, relies on a diverged task staying visibly non-finite):
      NaN      -> NaN in both modes (every output the element reaches);
      +-inf    -> non-finite in both; the split mode yields NaN where the fp32 MFMA yields +-inf (inf - top16(inf) is NaN:
                  documented in include/eosvos.h), untouched outputs stay exact;
      denormal -> treated as (at most) zero by both: outputs agree to fp32 rounding of the normal terms;
      values near FLT_MAX split exactly (no spurious overflow of the pieces)."""
    g = torch.Generator().manual_seed(9)
    B, H, W, Ci, Co = 1, 8, 16, 64, 64
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5
    x[0, 3, 1, 2] = float('nan')                 # pixel (1,2): every output channel becomes NaN
    x[0, 5, 2, 3] = float('inf')                 # pixel (2,3): +-inf by the sign of w[:, 5]
    x[0, 7, 3, 4] = float('-inf')
    x[0, 9, 4, 5] = 1e-41                        # denormal activation
    w[11, 13, 0, 0] = 1e-40                      # denormal weight
    x[0, 15, 5, 6] = 3.0e38                      # near FLT_MAX, times a small weight: stays finite
    w[:, 15] = w[:, 15] * 1e-3
    res = {}
    prev = engine_mod.get_matrix_mode()
    try:
        for mode in ('f32', 'bf16x6'):
            engine_mod.set_matrix_mode(mode)
            res[mode] = eng.test_conv_algo('direct', nhwc(x), w.to(DEV), None, None, None, False, 1, 1, 0).permute(0, 3, 1, 2).cpu()
    finally:
        engine_mod.set_matrix_mode(prev)
    a, b = res['f32'], res['bf16x6']
    assert bool(torch.isnan(a[0, :, 1, 2]).all()) and bool(torch.isnan(b[0, :, 1, 2]).all())
    for (py, px) in ((2, 3), (3, 4)):
        assert bool(torch.isinf(a[0, :, py, px]).all())                           # fp32 MFMA: +-inf
        assert not bool(torch.isfinite(b[0, :, py, px]).any())                    # split mode: non-finite (NaN)
    clean = torch.ones(H, W, dtype=torch.bool)
    for (py, px) in ((1, 2), (2, 3), (3, 4)):
        clean[py, px] = False
    assert bool(torch.isfinite(a[0][:, clean]).all()) and bool(torch.isfinite(b[0][:, clean]).all())
    xz = x.clone()
    xz[0, 9, 4, 5] = 0.0
    wz = w.clone()
    wz[11, 13, 0, 0] = 0.0
    ref = F.conv2d(torch.nan_to_num(xz, nan=0.0, posinf=0.0, neginf=0.0).double(), wz.double())
    scale = float(ref[0][:, clean].abs().max())
    assert float((b[0][:, clean].double() - ref[0][:, clean]).abs().max()) <= 2e-6 * scale       # denormals contribute nothing
    assert float((a[0][:, clean].double() - ref[0][:, clean]).abs().max()) <= 2e-6 * scale
    assert abs(float(b[0, 0, 5, 6]) - float(ref[0, 0, 5, 6])) <= 2e-6 * abs(float(ref[0, 0, 5, 6]))   # 3e38 * 1e-3 * w: exact split


def test_f16x3_special_values_and_dynamic_range(eng):
    """The fp16 split mode scales every operand tensor by a power of two taken from its largest FINITE magnitude
    (include/eosvos.h):
      NaN / +-inf -> NaN in every output the element reaches; they do not set the scale, so every other output keeps
                     fp32 accuracy;
      elements down to 2^-16 of the tensor's largest magnitude keep fp32 relative accuracy; smaller ones lose it
      gradually (absolute error <= 2^-40 of that maximum per element) -- the error stays far below fp32 rounding of the
      outputs the large elements dominate."""
    g = torch.Generator().manual_seed(9)
    B, H, W, Ci, Co = 1, 8, 16, 64, 64
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 1, 1, generator=g) / Ci ** 0.5
    x[0, 3, 1, 2] = float('nan')
    x[0, 5, 2, 3] = float('inf')
    x[0, 7, 3, 4] = float('-inf')
    x[0, :, 5, 6] *= 2.0 ** -14                  # a pixel whose activations are 16 000 x smaller than the rest
    x[0, :, 6, 7] *= 2.0 ** -30                  # ... and one a billion times smaller
    prev = engine_mod.get_matrix_mode()
    try:
        engine_mod.set_matrix_mode('f16x3')
        out = eng.test_conv_algo('direct', nhwc(x), w.to(DEV), None, None, None, False, 1, 1, 0).permute(0, 3, 1, 2).cpu()
    finally:
        engine_mod.set_matrix_mode(prev)
    assert bool(torch.isnan(out[0, :, 1, 2]).all())
    for (py, px) in ((2, 3), (3, 4)):
        assert not bool(torch.isfinite(out[0, :, py, px]).any())
    clean = torch.ones(H, W, dtype=torch.bool)
    for (py, px) in ((1, 2), (2, 3), (3, 4)):
        clean[py, px] = False
    ref = F.conv2d(torch.nan_to_num(x, nan=0.0, posinf=0.0, neginf=0.0).double(), w.double())
    sabs = F.conv2d(torch.nan_to_num(x, nan=0.0, posinf=0.0, neginf=0.0).double().abs(), w.double().abs())
    assert bool(torch.isfinite(out[0][:, clean]).all())
    rel = ((out.double() - ref).abs() / sabs)[0]
    normal = clean.clone()
    normal[5, 6] = normal[6, 7] = False
    assert float(rel[:, normal].max()) < 3e-7, float(rel[:, normal].max())          # fp32 accuracy
    assert float(rel[:, 5, 6].max()) < 3e-7, float(rel[:, 5, 6].max())               # 2^-14 of the maximum: still fp32 accuracy
    # 2^-30 of the maximum: absolute error <= 2^-40 of the tensor's maximum per element, i.e. tiny against the output scale
    big = float(x[torch.isfinite(x)].abs().max())
    err = (out.double() - ref).abs()[0, :, 6, 7]
    assert float(err.max()) < 64 * 2.0 ** -40 * big * float(w.abs().max()), float(err.max())


@pytest.mark.parametrize('mode', ['f16x3', 'bf16x6', 'f32'])
def test_finetune_trajectory_in_every_matrix_mode(mode):
    """Three fine-tune iterations + inference at 96x160, batch 2, against the CPU oracle in each matrix mode (the other
    GPU tests run in the default mode only): losses to 1e-4 relative, probabilities to 1e-4 absolute."""
    from eosvos_amd import synthetic
    from oracle import meta
    H, W, B = 96, 160, 2
    sd = synthetic.synthetic_state('resnet50')
    lrs = synthetic.synthetic_lrs('resnet50')
    x, y = synthetic.synthetic_frames(B, H, W, seed=7)
    ref_losses, ref_state = meta.finetune(sd, lrs, [(x, y)] * 3)
    prev = engine_mod.get_matrix_mode()
    try:
        engine_mod.set_matrix_mode(mode)
        e = Engine('resnet50', H, W, max_batch=B, device=DEV)
        e.load_model_state(sd, lrs)
        losses = [e.finetune_step(x.to(DEV), y.to(DEV)) for _ in range(3)]
        params = e.get_params().cpu()
        e.close()
    finally:
        engine_mod.set_matrix_mode(prev)
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 1e-4 * abs(b), (mode, losses, ref_losses)
    assert bool(torch.isfinite(params).all())


@pytest.mark.parametrize('side_stream', [True, False], ids=['two_streams', 'one_queue'])
def test_matrix_mode_switch_on_a_live_engine(side_stream):
    """The matrix mode is process-wide and may change under an engine that has already run a backward pass (bench.py's
    A/B extras do exactly that).  The cached launch plans -- grouped weight-gradient tables (with or without absmax
    slots), slab counts of the update tables -- depend on the mode: bf16x6 -> f16x3 -> f32 -> f16x3 on ONE engine, every
    segment continues the oracle's trajectory (a stale plan sums the wrong number of slabs or dereferences a null slot)."""
    from eosvos_amd import synthetic
    from oracle import meta
    H, W, B = 96, 160, 2
    sd = synthetic.synthetic_state('resnet50')
    lrs = synthetic.synthetic_lrs('resnet50')
    x, y = synthetic.synthetic_frames(B, H, W, seed=7)
    order = ['bf16x6', 'f16x3', 'f32', 'f16x3']
    ref_losses, _ = meta.finetune(sd, lrs, [(x, y)] * (2 * len(order)))
    prev = engine_mod.get_matrix_mode()
    losses = []
    try:
        engine_mod.set_matrix_mode(order[0])
        e = Engine('resnet50', H, W, max_batch=B, device=DEV, side_stream=side_stream)
        e.load_model_state(sd, lrs)
        for mode in order:
            engine_mod.set_matrix_mode(mode)
            losses += [e.finetune_step(x.to(DEV), y.to(DEV)) for _ in range(2)]
        e.close()
    finally:
        engine_mod.set_matrix_mode(prev)
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 2e-4 * abs(b), (losses, ref_losses)


def test_f16x3_is_invariant_under_power_of_two_reparametrisation():
    """Scaling a conv's weights by 2^-30 and its frozen-norm scale by 2^30 leaves the network unchanged in exact arithmetic;
    in the f16x3 mode the per-tensor scales absorb the factor exactly (same fp16 pieces), so the logits are bit-identical
    -- tensors of any magnitude use the full precision of the split (bf16x6 / fp32 have this property trivially)."""
    from eosvos_amd import synthetic
    H, W = 96, 160
    sd = synthetic.synthetic_state('resnet50')
    lrs = synthetic.synthetic_lrs('resnet50')
    x, _ = synthetic.synthetic_frames(1, H, W, seed=3)
    sd2 = {k: v.clone() for k, v in sd.items()}
    for conv, bn, f in (('backbone.layer1.0.conv1', 'backbone.layer1.0.bn1', 2.0 ** -30), ('backbone.layer3.2.conv2', 'backbone.layer3.2.bn2', 2.0 ** 24)):
        assert conv + '.weight' in sd2 and bn + '.weight' in sd2, [k for k in sd2 if 'layer1.0' in k][:8]
        sd2[conv + '.weight'] *= f
        # y = gamma * (conv(x) - mean) / sqrt(var + eps) + beta: scale the mean by f and gamma by 1 / f (exact: powers of two)
        sd2[bn + '.running_mean'] *= f
        sd2[bn + '.weight'] /= f
    prev = engine_mod.get_matrix_mode()
    outs = []
    try:
        engine_mod.set_matrix_mode('f16x3')
        for state in (sd, sd2):
            e = Engine('resnet50', H, W, max_batch=1, device=DEV)
            e.load_model_state(state, lrs)
            outs.append(e.forward(x.to(DEV)).cpu())
            e.close()
    finally:
        engine_mod.set_matrix_mode(prev)
    assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())
