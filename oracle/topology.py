"""Layer inventory of DeepLabV3+ on a ResNet-50/101 encoder (oracle side).

Follows the reference constructor `src/networks/deeplabv3plus.py:104-155`:
torchvision ``resnet50(replace_stride_with_dilation=[False, True, True])`` wrapped
by ``IntermediateLayerGetter`` (`:110-116`), then the surgery at `:135-142`
(layer3[0].conv1 and layer3[0].downsample stride 2, every layer3 conv2 dilation 1,
layer4[2].conv2 dilation 8) which yields output-stride 16 with layer4 conv2
dilations [2, 4, 8]; ASPP rates [6, 12, 18] (`:15-20`); decoder (`:56-101`).

The torchvision pieces (Bottleneck v1.5: 1x1 -> 3x3(stride, dilation) -> 1x1 x4,
downsample = conv1x1(stride) + BN when the stride or the width changes;
ASPP = 1x1 | 3x3 d r1 | 3x3 d r2 | 3x3 d r3 | image pooling -> concat -> 1x1)
are restated from the published torchvision 0.4 architecture.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
from collections import namedtuple

Conv = namedtuple('Conv', 'name cin cout k stride dil pad norm bias')
# name  : state-dict prefix of the conv module ("backbone.layer1.0.conv1")
# norm  : state-dict prefix of the BatchNorm/GroupNorm that follows, or None
# bias  : True only for decoder.last_conv.8

_BLOCKS = {'resnet50': (3, 4, 6, 3), 'resnet101': (3, 4, 23, 3)}


def is_v3(encoder):
    """'deeplabv3_<resnet>': plain DeepLabV3, `src/networks/deeplabv3.py:10-83` -- torchvision
    ``resnet(replace_stride_with_dilation=[False, True, True])`` untouched (output stride 8: layer3 dilations 1, 2, 2, ...,
    layer4 2, 4, 4 by the rule "the first block of a dilated layer keeps the previous dilation"), DeepLabHead =
    ASPP[12, 24, 36] -> Conv3x3(256) + BN + ReLU -> Conv1x1(num_classes); `_SimpleSegmentationModel.forward` resizes the
    logits to the input size (bilinear, align_corners=False)."""
    return encoder.startswith('deeplabv3_')


def bottleneck_cfg(encoder):
    """Per block: (prefix, inplanes, width, stride_conv1, stride_conv2, dil, has_ds, ds_stride)."""
    v3 = is_v3(encoder)
    blocks = _BLOCKS[encoder.split('_')[-1]]
    out = []
    inplanes = 64
    # (width, stride on conv1, stride on conv2, dilations per block)
    layer_cfg = [
        (64, 1, 1, [1] * blocks[0]),
        (128, 1, 2, [1] * blocks[1]),
        # reference surgery: stride 2 moved to conv1 + downsample, conv2 dilation 1
        (256, 2, 1, [1] * blocks[2]),
        # torchvision dilation rule (previous dilation for block 0) + reference
        # override of the last block: 2, 4, ..., 4, 8
        (512, 1, 1, [2] + [4] * (blocks[3] - 2) + [8]),
    ]
    if v3:          # no surgery: strides replaced by dilation in layer3 / layer4
        layer_cfg[2] = (256, 1, 1, [1] + [2] * (blocks[2] - 1))
        layer_cfg[3] = (512, 1, 1, [2] + [4] * (blocks[3] - 1))
    for li, (width, s1, s2, dils) in enumerate(layer_cfg, start=1):
        for bi, d in enumerate(dils):
            first = bi == 0
            has_ds = first  # stride != 1 or inplanes != width * 4 holds for every first block
            out.append(dict(
                prefix=f'backbone.layer{li}.{bi}', inplanes=inplanes, width=width,
                s1=s1 if first else 1, s2=s2 if first else 1, dil=d,
                has_ds=has_ds, ds_stride=(s1 * s2) if first else 1))
            inplanes = width * 4
    return out


def conv_list(encoder='resnet50'):
    """All convolutions in ``named_parameters()`` order of the reference model."""
    L = [Conv('backbone.conv1', 3, 64, 7, 2, 1, 3, 'backbone.bn1', False)]
    for b in bottleneck_cfg(encoder):
        p, w = b['prefix'], b['width']
        L.append(Conv(p + '.conv1', b['inplanes'], w, 1, b['s1'], 1, 0, p + '.bn1', False))
        L.append(Conv(p + '.conv2', w, w, 3, b['s2'], b['dil'], b['dil'], p + '.bn2', False))
        L.append(Conv(p + '.conv3', w, 4 * w, 1, 1, 1, 0, p + '.bn3', False))
        if b['has_ds']:
            L.append(Conv(p + '.downsample.0', b['inplanes'], 4 * w, 1, b['ds_stride'], 1, 0,
                          p + '.downsample.1', False))
    a = 'classifier.0'
    L.append(Conv(a + '.convs.0.0', 2048, 256, 1, 1, 1, 0, a + '.convs.0.1', False))
    for i, r in enumerate((12, 24, 36) if is_v3(encoder) else (6, 12, 18), start=1):
        L.append(Conv(f'{a}.convs.{i}.0', 2048, 256, 3, 1, r, r, f'{a}.convs.{i}.1', False))
    L.append(Conv(a + '.convs.4.1', 2048, 256, 1, 1, 1, 0, a + '.convs.4.2', False))
    L.append(Conv(a + '.project.0', 1280, 256, 1, 1, 1, 0, a + '.project.1', False))
    if is_v3(encoder):
        L.append(Conv('classifier.1', 256, 256, 3, 1, 1, 1, 'classifier.2', False))
        L.append(Conv('classifier.4', 256, 1, 1, 1, 1, 0, None, True))
        return L
    L.append(Conv('decoder.conv1', 256, 48, 1, 1, 1, 0, 'decoder.bn1', False))
    L.append(Conv('decoder.last_conv.0', 304, 256, 3, 1, 1, 1, 'decoder.last_conv.1', False))
    L.append(Conv('decoder.last_conv.4', 256, 256, 3, 1, 1, 1, 'decoder.last_conv.5', False))
    L.append(Conv('decoder.last_conv.8', 256, 1, 1, 1, 1, 0, None, True))
    return L


def trainable_names(encoder='resnet50'):
    """Names of the trainable tensors in reference order (conv weights + the one bias)."""
    names = []
    for c in conv_list(encoder):
        names.append(c.name + '.weight')
        if c.bias:
            names.append(c.name + '.bias')
    return names


def trainable_shapes(encoder='resnet50'):
    shapes = []
    for c in conv_list(encoder):
        shapes.append((c.cout, c.cin, c.k, c.k))
        if c.bias:
            shapes.append((c.cout,))
    return shapes


def state_dict_keys(encoder='resnet50', norm='bn'):
    """Full state-dict key order of the reference module (374 keys BN / 188 GN for R50)."""
    keys = []
    for c in conv_list(encoder):
        # the ASPP pooling conv sits at index 1 of its Sequential, everything else
        # registers conv before norm
        keys.append(c.name + '.weight')
        if c.bias:
            keys.append(c.name + '.bias')
        if c.norm is not None:
            keys += [c.norm + '.weight', c.norm + '.bias']
            if norm == 'bn':
                keys += [c.norm + '.running_mean', c.norm + '.running_var',
                         c.norm + '.num_batches_tracked']
    return keys
