"""Functional torch-CPU restatement of the reference DeepLabV3+ forward pass.

Follows `src/networks/deeplabv3plus.py`:
  * `_DeepLabV3Plus.forward` (`:32-53`): backbone -> classifier(ASPP) -> decoder ->
    bilinear resize to the input size with align_corners=False;
  * `Decoder.forward` (`:84-93`): 1x1 conv + norm + ReLU on the layer1 feature,
    bilinear upsample (align_corners=True) of the ASPP output, concat, last_conv;
  * `DeepLabV3Plus.train_without_dropout` (`:259-280`) with
    `batch_norm.accum_stats=False`: every BatchNorm runs in eval mode (running
    statistics, frozen affine), every Dropout is the identity -- that is the only
    mode the fine-tuning loop uses (`src/util/evaluate.py:213`), so it is the only
    mode restated here;
  * `replace_batch_with_group_norms` (`:180-191`): GroupNorm(16, C) sharing the
    (frozen) BN affine.
`forward` returns the logits tensor (the reference returns `[logits]`, `:301`).

Parameters are a plain ``dict`` keyed by the reference state-dict names.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import torch
import torch.nn.functional as F

from .topology import bottleneck_cfg, conv_list, is_v3

EPS = 1e-5


def _norm(P, x, prefix, norm):
    if norm == 'bn':
        return F.batch_norm(x, P[prefix + '.running_mean'], P[prefix + '.running_var'],
                            P[prefix + '.weight'], P[prefix + '.bias'], False, 0.0, EPS)
    if norm == 'gn':
        return F.group_norm(x, 16, P[prefix + '.weight'], P[prefix + '.bias'], EPS)
    raise NotImplementedError(norm)


def _conv(P, x, c):
    return F.conv2d(x, P[c.name + '.weight'], P.get(c.name + '.bias') if c.bias else None,
                    stride=c.stride, padding=c.pad, dilation=c.dil)


def forward(P, x, encoder='resnet50', norm='bn', taps=None):
    """x: (B,3,H,W) fp32 in [0,1] -> logits (B,1,H,W).  `taps`, if a dict, receives
    named intermediate activations (used for per-stage golden checksums)."""
    convs = {c.name: c for c in conv_list(encoder)}

    def cnr(x, name, relu=True):
        c = convs[name]
        y = _norm(P, _conv(P, x, c), c.norm, norm)
        return F.relu(y) if relu else y

    inp_hw = x.shape[-2:]
    x = cnr(x, 'backbone.conv1')
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    if taps is not None:
        taps['stem'] = x
    low = None
    for b in bottleneck_cfg(encoder):
        p = b['prefix']
        idt = x
        y = cnr(x, p + '.conv1')
        y = cnr(y, p + '.conv2')
        y = cnr(y, p + '.conv3', relu=False)
        if b['has_ds']:
            idt = cnr(x, p + '.downsample.0', relu=False)
        x = F.relu(y + idt)
        lname = p.rsplit('.', 1)[0]
        if taps is not None:
            taps[lname.replace('backbone.', '')] = x  # overwritten until the last block
        if lname == 'backbone.layer1':
            low = x
    # ASPP
    a = 'classifier.0'
    hw = x.shape[-2:]
    branches = [cnr(x, f'{a}.convs.{i}.0') for i in range(4)]
    g = F.adaptive_avg_pool2d(x, 1)
    g = cnr(g, a + '.convs.4.1')
    branches.append(F.interpolate(g, size=hw, mode='bilinear', align_corners=False))
    x = cnr(torch.cat(branches, dim=1), a + '.project.0')
    if taps is not None:
        taps['aspp'] = x
    if is_v3(encoder):          # DeepLabHead tail + _SimpleSegmentationModel.forward (networks/deeplabv3.py:13,64-83)
        x = cnr(x, 'classifier.1')
        x = _conv(P, x, convs['classifier.4'])
        if taps is not None:
            taps['low_logits'] = x
        return F.interpolate(x, size=inp_hw, mode='bilinear', align_corners=False)
    # decoder
    low = cnr(low, 'decoder.conv1')
    x = F.interpolate(x, size=low.shape[-2:], mode='bilinear', align_corners=True)
    x = torch.cat((x, low), dim=1)
    x = cnr(x, 'decoder.last_conv.0')
    x = cnr(x, 'decoder.last_conv.4')
    if taps is not None:
        taps['dec'] = x
    x = _conv(P, x, convs['decoder.last_conv.8'])
    if taps is not None:
        taps['low_logits'] = x
    return F.interpolate(x, size=inp_hw, mode='bilinear', align_corners=False)


def bce_loss(logits, gt, batch_average=True):
    """`compute_loss('cross_entropy', ...)`, `src/util/helper_func.py:32-40`."""
    if batch_average:
        return F.binary_cross_entropy_with_logits(logits, gt)
    l = F.binary_cross_entropy_with_logits(logits, gt, reduction='none')
    return l.view(l.shape[0], -1).mean(dim=1)


def dice_loss(logits, gt):
    """`dice_loss(output, label, batch_average=True)`, `src/networks/loss_dice.py:4-31`."""
    pred = torch.sigmoid(logits).reshape(-1)
    lab = gt.reshape(-1)
    return 1 - (2.0 * (pred * lab).sum() + 1.0) / (pred.sum() + lab.sum() + 1.0)


def loss_fn(name, logits, gt):
    """`compute_loss` names, `src/util/helper_func.py:28-56` (batch_average=True)."""
    if name == 'cross_entropy':
        return bce_loss(logits, gt)
    if name == 'dice':
        return dice_loss(logits, gt)
    if name == 'cross_entropy_and_dice':
        return bce_loss(logits, gt) - (1 - dice_loss(logits, gt)).log()
    if name == 'class_balanced_cross_entropy':
        return class_balanced_bce_loss(logits, gt)
    raise NotImplementedError(name)


def loss_per_sample(name, logits, gt):
    """`compute_loss(name, ..., {'batch_average': False})`: every loss of `helper_func.py:28-56` reduces per sample
    (`loss_dice.py:33-40`, `loss_ce.py:26-40`, BCE `helper_func.py:38-39`), i.e. equals the batch-average form on
    each sample alone."""
    return torch.stack([loss_fn(name, logits[b:b + 1], gt[b:b + 1]) for b in range(logits.shape[0])])


def class_balanced_bce_loss(logits, gt):
    """`class_balanced_cross_entropy_loss(output, label, size_average=True, batch_average=True)`,
    `src/networks/loss_ce.py:15-60`: positives weighted by the negative fraction and vice versa."""
    lab = (gt >= 0.5).float()
    n_pos, n_neg = lab.sum(), (1.0 - lab).sum()
    elem = F.binary_cross_entropy_with_logits(logits, lab, reduction='none')
    loss = (n_neg * (lab * elem).sum() + n_pos * ((1.0 - lab) * elem).sum()) / (n_pos + n_neg)
    return loss / gt.numel()
