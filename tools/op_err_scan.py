"""Elementwise error (max |diff| / max |ref|, fp64 reference) of fwd / dx / dw for a list of conv shapes in both matrix modes."""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from eosvos_amd import engine as em
from eosvos_amd.engine import Engine
eng = Engine('resnet50', 96, 160, max_batch=1)
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
mr = lambda a, r: float((a.double().cpu() - r).abs().max() / r.abs().max())
CASES = [(3, 24, 40, 64, 64, 1, 1, 1, 0), (3, 24, 40, 64, 256, 1, 1, 1, 0), (3, 24, 40, 256, 64, 1, 1, 1, 0), (3, 24, 40, 64, 64, 3, 1, 1, 1),
         (3, 24, 40, 256, 128, 1, 1, 1, 0), (3, 24, 40, 128, 128, 3, 2, 1, 1), (3, 12, 20, 128, 512, 1, 1, 1, 0), (3, 12, 20, 512, 256, 1, 2, 1, 0),
         (3, 6, 10, 1024, 256, 1, 1, 1, 0), (3, 6, 10, 256, 256, 3, 1, 1, 1), (3, 6, 10, 512, 512, 3, 1, 2, 2), (3, 6, 10, 2048, 256, 3, 1, 6, 6),
         (3, 6, 10, 2048, 256, 3, 1, 18, 18), (3, 6, 10, 1280, 256, 1, 1, 1, 0), (3, 24, 40, 256, 48, 1, 1, 1, 0), (3, 24, 40, 304, 256, 3, 1, 1, 1),
         (1, 24, 40, 64, 64, 1, 1, 1, 0), (1, 6, 10, 2048, 512, 1, 1, 1, 0)]
for case in CASES:
    B, H, W, Ci, Co, k, s, d, p = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(B, Ci, H, W, generator=g); w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y = F.conv2d(xd, wd, None, s, p, d); gy = torch.randn(y.shape, generator=g); y.backward(gy.double())
    row = []
    for mode in ('bf16x6', 'f32'):
        em.set_matrix_mode(mode)
        out = eng.test_conv_algo('direct', nhwc(x), w.cuda(), None, None, None, False, s, d, p)
        dx, dw = eng.test_conv_bwd_algo('direct', nhwc(x), w.cuda(), nhwc(gy), s, d, p)
        row.append('%s fwd %.1e dx %.1e dw %.1e' % (mode, mr(out.permute(0, 3, 1, 2), y.detach()), mr(dx.permute(0, 3, 1, 2), xd.grad), mr(dw, wd.grad)))
    print(case, ' | '.join(row))
em.set_matrix_mode('bf16x6')
