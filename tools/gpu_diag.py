"""Stage-by-stage GPU diagnostic (development aid; the graded checks are tests/ -m gpu).

Compares the HIP path with the CPU oracle op by op and prints the error of every stage,
so one gpurun call localises a wrong kernel.
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from eosvos_amd import synthetic, topology  # noqa: E402
from eosvos_amd.engine import Engine  # noqa: E402
from oracle import deeplab, meta  # noqa: E402

dev = 'cuda:0'


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def conv_cases(eng):
    g = torch.Generator().manual_seed(0)
    cases = [
        # B, H, W, Cin, Cout, k, s, d, p
        (1, 12, 20, 64, 64, 1, 1, 1, 0),
        (2, 12, 20, 64, 128, 3, 1, 1, 1),
        (1, 13, 21, 128, 64, 3, 2, 1, 1),
        (1, 12, 21, 64, 256, 1, 2, 1, 0),
        (1, 10, 14, 256, 128, 3, 1, 2, 2),
        (1, 30, 54, 128, 64, 3, 1, 18, 18),
        (1, 9, 11, 304, 256, 3, 1, 1, 1),
        (2, 9, 11, 256, 48, 1, 1, 1, 0),
        (1, 16, 16, 2048, 256, 3, 1, 6, 6),
        (3, 24, 40, 64, 64, 3, 1, 1, 1),
    ]
    for (B, H, W, Ci, Co, k, s, d, p) in cases:
        x = torch.randn(B, Ci, H, W, generator=g)
        w = torch.randn(Co, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
        a = torch.rand(Co, generator=g) + 0.5
        b = torch.randn(Co, generator=g)
        y = F.conv2d(x, w, None, s, p, d)
        res = torch.randn_like(y)
        ref = F.relu(y * a.view(1, -1, 1, 1) + b.view(1, -1, 1, 1) + res)
        xg = x.permute(0, 2, 3, 1).contiguous().to(dev)
        out = eng.test_conv(xg, w.to(dev), a.to(dev), b.to(dev), res.permute(0, 2, 3, 1).contiguous().to(dev),
                            True, s, d, p)
        e_f = rel(out.permute(0, 3, 1, 2), ref)
        gy = torch.randn_like(y)
        xr = x.clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        F.conv2d(xr, wr, None, s, p, d).backward(gy)
        dx, dw = eng.test_conv_bwd(xg, w.to(dev), gy.permute(0, 2, 3, 1).contiguous().to(dev), s, d, p)
        print(f'conv B{B} {H}x{W} {Ci}->{Co} k{k} s{s} d{d} p{p}: fwd {e_f:.2e} dgrad '
              f'{rel(dx.permute(0, 3, 1, 2), xr.grad):.2e} wgrad {rel(dw, wr.grad):.2e}', flush=True)


def net_case(H, W, B, steps=3):
    sd = synthetic.synthetic_state('resnet50')
    lrs = synthetic.synthetic_lrs('resnet50')
    eng = Engine('resnet50', H, W, max_batch=B, device=dev)
    eng.load_model_state(sd, lrs)
    x, y = synthetic.synthetic_frames(B, H, W, seed=7)
    taps = {}
    t0 = time.time()
    with torch.no_grad():
        ref = deeplab.forward(sd, x, taps=taps)
    t_cpu = time.time() - t0
    xg, yg = x.to(dev), y.to(dev)
    out = eng.forward(xg)
    torch.cuda.synchronize()
    print(f'[{H}x{W} B{B}] logits rel err {rel(out, ref):.2e}  max abs {float((out.cpu() - ref).abs().max()):.2e} '
          f'(cpu fwd {t_cpu:.2f}s)', flush=True)
    for name, key in (('p1', 'stem'), ('blk2.out', 'layer1'), ('blk6.out', 'layer2'), ('blk12.out', 'layer3'),
                      ('blk15.out', 'layer4'), ('proj', 'aspp'), ('d2', 'dec'), ('lowlog', 'low_logits')):
        print(f'   tap {key:10s} rel {rel(eng.debug_tensor(name), taps[key]):.2e}', flush=True)
    # gradients
    eng.keep_grads(True)
    eng.reset()
    loss_ref, grads_ref, _ = meta.loss_and_grads(sd, x, y)
    eng.forward(xg, want_logits=False)
    loss = eng.loss_bce(yg)
    eng.backward_step()
    g = eng.get_grads().cpu()
    print(f'   loss {float(loss):.6f} ref {float(loss_ref):.6f}', flush=True)
    off = 0
    worst = 0
    for (n, shape), gr in zip(topology.trainable('resnet50'), grads_ref):
        k = gr.numel()
        e = rel(g[off:off + k].view(shape), gr)
        worst = max(worst, e)
        if e > 1e-3 or n in ('backbone.conv1.weight', 'decoder.last_conv.8.weight', 'decoder.last_conv.8.bias',
                             'classifier.0.convs.4.1.weight', 'backbone.layer2.0.downsample.0.weight'):
            print(f'   grad {n:45s} rel {e:.2e}', flush=True)
        off += k
    print(f'   worst grad rel err {worst:.2e}', flush=True)
    # trajectory
    eng.reset()
    losses = [eng.finetune_step(xg, yg) for _ in range(steps)]
    ref_losses, _ = meta.finetune(sd, lrs, [(x, y)] * steps)
    print('   losses hip', ['%.6f' % l for l in losses], flush=True)
    print('   losses ref', ['%.6f' % l for l in ref_losses], flush=True)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 5
    for _ in range(n):
        eng.finetune_step(xg, yg, sync_loss=False)
    eng.synchronize()
    print(f'   finetune step {1e3 * (time.time() - t0) / n:.2f} ms  (B={B})', flush=True)
    eng.close()


if __name__ == '__main__':
    print(torch.cuda.get_device_name(0), flush=True)
    eng = Engine('resnet50', 96, 160, max_batch=2, device=dev)
    conv_cases(eng)
    eng.close()
    net_case(96, 160, 2)
    if len(sys.argv) > 1 and sys.argv[1] == 'full':
        net_case(480, 854, 1, steps=2)
