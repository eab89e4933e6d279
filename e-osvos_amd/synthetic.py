"""Seeded synthetic weights / frames (no network for checkpoints or datasets).

Recipe of SURVEY.md section 8(d): conv weights N(0, 2/fan) with fan = Cout*k*k for
backbone and ASPP and Cin*k*k for the decoder (the reference initialises its decoder
with kaiming_normal_, `src/networks/deeplabv3plus.py:95-98`); BN running_mean
~N(0,.1), running_var ~U(.8,1.2), gamma ~U(.8,1.2) (x0.3 for every bn3), beta
~N(0,.05); learned per-neuron lrs init_lr*(1+U(-.5,.5)) as `meta_optim.py:57-58`.
Everything is drawn from CPU generators so both boxes reproduce the same tensors.
"""
import math

import torch

from .topology import conv_infos, neuron_lr_shape, trainable


def synthetic_state(encoder='resnet50', seed=99, dtype=torch.float32):
    """Model state dict (reference key names, BN layout) with well-conditioned values."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for c in conv_infos(encoder):
        fan = (c.cin if c.name.startswith('decoder') or c.name in ('classifier.1', 'classifier.4') else c.cout) * c.k * c.k
        w = torch.randn(c.cout, c.cin, c.k, c.k, generator=g, dtype=dtype) * math.sqrt(2.0 / fan)
        sd[c.name + '.weight'] = w
        if c.bias:
            bound = 1.0 / math.sqrt(c.cin * c.k * c.k)
            sd[c.name + '.bias'] = (torch.rand(c.cout, generator=g, dtype=dtype) * 2 - 1) * bound
        if c.norm is not None:
            gamma = torch.rand(c.cout, generator=g, dtype=dtype) * 0.4 + 0.8
            if c.norm.endswith('bn3'):
                gamma = gamma * 0.3
            sd[c.norm + '.weight'] = gamma
            sd[c.norm + '.bias'] = torch.randn(c.cout, generator=g, dtype=dtype) * 0.05
            sd[c.norm + '.running_mean'] = torch.randn(c.cout, generator=g, dtype=dtype) * 0.1
            sd[c.norm + '.running_var'] = torch.rand(c.cout, generator=g, dtype=dtype) * 0.4 + 0.8
            sd[c.norm + '.num_batches_tracked'] = torch.zeros((), dtype=torch.long)
    return sd


def heavy_tailed_state(encoder='resnet50', seed=99, stat_seed=1234, sigma_var=2.3, sigma_gamma=1.6, dead_frac=0.05,
                       dead_scale=1e-5):
    """The seeded recipe above with the BatchNorm statistics of a TRAINED parent checkpoint's kind instead of the benign
    U(0.8, 1.2) ones: running_var and gamma log-normal over >= 3 decades each (sigma of ln: 2.3 / 1.6, i.e. +-3 sigma spans 6 /
    4 decades), `dead_frac` of every layer's channels near-dead (gamma x dead_scale).  Each layer's gamma is then
    rescaled by ONE factor so that the folded scale a = gamma / sqrt(var + eps) keeps the root-mean-square of the benign
    recipe's -- the network stays trainable, but a few channels carry most of each activation tensor and many sit
    4-6 decades below the tensor's maximum: the regime in which a per-tensor power-of-two scale (the f16x3 matrix mode)
    has the least headroom.  Fixture G19 pins the engine against the unmodified reference on this state."""
    sd = synthetic_state(encoder, seed)
    g = torch.Generator().manual_seed(stat_seed)
    for c in conv_infos(encoder):
        if c.norm is None:
            continue
        n = c.cout
        gamma0, var0 = sd[c.norm + '.weight'], sd[c.norm + '.running_var']
        rms0 = float((gamma0 / torch.sqrt(var0 + 1e-5)).pow(2).mean().sqrt())
        var = torch.exp(torch.randn(n, generator=g) * sigma_var)
        gamma = torch.exp(torch.randn(n, generator=g) * sigma_gamma)
        sign = torch.where(torch.rand(n, generator=g) < 0.1, -1.0, 1.0)           # trained gammas are occasionally negative
        dead = torch.rand(n, generator=g) < dead_frac
        gamma = torch.where(dead, gamma * dead_scale, gamma) * sign
        a = gamma / torch.sqrt(var + 1e-5)
        gamma = gamma * (rms0 / float(a.pow(2).mean().sqrt()))
        sd[c.norm + '.weight'] = gamma
        sd[c.norm + '.running_var'] = var
        # the running mean of a channel scales with its standard deviation
        sd[c.norm + '.running_mean'] = sd[c.norm + '.running_mean'] * torch.sqrt(var)
    return sd


def synthetic_lrs(encoder='resnet50', init_lr=1e-3, seed=1):
    """List of NEURON lr tensors aligned with `trainable(encoder)`."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _, shape in trainable(encoder):
        s = neuron_lr_shape(shape)
        out.append(init_lr * (1.0 + (torch.rand(s, generator=g) - 0.5)))
    return out


def synthetic_lr_state(encoder='resnet50', level='NEURON', use_log=False, init_lr=1e-3, seed=1):
    """Learned lr state in the reference's layout for `lr_hierarchy_level` (meta_optim.py:27-67):
    SINGLE -> tensor (1,1); TENSOR -> tensor (G,1); NEURON / PARAM -> list of per-tensor tensors.
    Values init_lr*(1+U(-.5,.5)) (SINGLE: 1.2*init_lr), log() applied for `use_log_init_lr`."""
    g = torch.Generator().manual_seed(seed)
    tr = trainable(encoder)
    if level == 'SINGLE':
        out = torch.full((1, 1), 1.2 * init_lr)
    elif level == 'TENSOR':
        out = init_lr * (1.0 + (torch.rand(len(tr), 1, generator=g) - 0.5))
    elif level == 'NEURON':
        out = [init_lr * (1.0 + (torch.rand(neuron_lr_shape(s), generator=g) - 0.5)) for _, s in tr]
    elif level == 'PARAM':
        out = [init_lr * (1.0 + (torch.rand(tuple(s), generator=g) - 0.5)) for _, s in tr]
    else:
        raise NotImplementedError(level)
    if use_log:
        out = [o.log() for o in out] if isinstance(out, list) else out.log()
    return out


def synthetic_frames(batch, height, width, seed=7, second_object=False):
    """image (B,3,H,W) in [0,1), mask (B,1,H,W) in {0,1}: one rectangle per frame
    (shifted per batch element so the frames differ)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(batch, 3, height, width, generator=g)
    gt = torch.zeros(batch, 1, height, width)
    for b in range(batch):
        dx = 4 * b
        gt[b, 0, height // 4: height // 2, width // 3 + dx: 2 * width // 3 + dx] = 1.0
        if second_object:
            gt[b, 0, 5 * height // 8: 7 * height // 8, width // 8: width // 4] = 1.0
    # make the object visible in the image so fine-tuning has signal
    img = (img * 0.6 + 0.4 * gt).clamp_(0, 1 - 1e-6)
    return img, gt
