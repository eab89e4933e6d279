"""Kernel times of the streaming kernels (forward, data gradient) on layer1 / layer2 shapes at batch 3 and on a 16 x 16 map
(= prologue only):  tools/debug/stream_time.sh"""
import sys
import torch
sys.path.insert(0, '.')
from eosvos_amd.engine import Engine
e = Engine('resnet50', 96, 160, max_batch=1)
CASES = [(64, 64, 3, 120, 214), (64, 256, 1, 120, 214), (256, 64, 1, 120, 214), (128, 512, 1, 60, 107), (512, 128, 1, 60, 107), (256, 128, 1, 120, 214)]
for (ci, co, k, H, W) in CASES:
    for (B, h, w_) in ((1, 16, 16), (3, H, W)):
        x = torch.randn(B, h, w_, ci, device='cuda')
        w = torch.randn(co, ci, k, k, device='cuda') / (ci * k * k) ** 0.5
        g = torch.randn(B, h, w_, co, device='cuda')
        for _ in range(4):
            e.test_conv_algo('direct', x, w, None, None, None, False, 1, 1, k // 2)
            e.test_conv_bwd_algo('direct', x, w, g, 1, 1, k // 2)
e.synchronize()
e.close()
