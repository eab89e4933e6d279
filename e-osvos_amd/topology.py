"""Host-side description of the DeepLabV3+/ResNet graph the engine executes.

Mirrors what `src/networks/deeplabv3plus.py:104-155` of the reference builds on top
of torchvision's ResNet (output stride 16, layer4 dilations 2/4/8, ASPP 6/12/18):
it only yields *names and shapes* in the reference's `named_parameters()` /
`state_dict()` order so checkpoints stay layout-compatible.  The arithmetic lives in
`csrc/` -- the C library exports the same table (`eosvos_conv_info`) and
`tests/test_topology.py` checks the two agree.
"""
from collections import namedtuple

ConvInfo = namedtuple('ConvInfo', 'name cin cout k stride dil pad norm bias')

BLOCKS = {'resnet50': (3, 4, 6, 3), 'resnet101': (3, 4, 23, 3)}
# 'deeplabv3_<encoder>': plain DeepLabV3 (`src/networks/deeplabv3.py:10-83`): output stride 8 (torchvision dilation rule,
# no stride surgery), DeepLabHead = ASPP[12, 24, 36] -> 3x3 conv + BN + ReLU -> 1x1 conv, no decoder
ARCH_ID = {'resnet50': 50, 'resnet101': 101, 'deeplabv3_resnet50': 1050, 'deeplabv3_resnet101': 1101}


def is_v3(encoder):
    return encoder.startswith('deeplabv3_')


def conv_infos(encoder='resnet50'):
    v3 = is_v3(encoder)
    nb = BLOCKS[encoder.split('_')[-1]]
    out = [ConvInfo('backbone.conv1', 3, 64, 7, 2, 1, 3, 'backbone.bn1', False)]
    inpl = 64
    widths = (64, 128, 256, 512)
    for li in range(4):
        w = widths[li]
        for bi in range(nb[li]):
            p = f'backbone.layer{li + 1}.{bi}'
            first = bi == 0
            s1 = 2 if (li == 2 and first and not v3) else 1   # DeepLabV3+ surgery: layer3[0].conv1 stride 2
            s2 = 2 if (li == 1 and first) else 1          # layer2[0].conv2 stride 2
            if v3:                                        # torchvision rule: the first block keeps the previous dilation
                d = {2: 1 if bi == 0 else 2, 3: 2 if bi == 0 else 4}.get(li, 1)
            elif li == 3:
                d = 2 if bi == 0 else (8 if bi == nb[3] - 1 else 4)
            else:
                d = 1
            out.append(ConvInfo(p + '.conv1', inpl, w, 1, s1, 1, 0, p + '.bn1', False))
            out.append(ConvInfo(p + '.conv2', w, w, 3, s2, d, d, p + '.bn2', False))
            out.append(ConvInfo(p + '.conv3', w, 4 * w, 1, 1, 1, 0, p + '.bn3', False))
            if first:
                out.append(ConvInfo(p + '.downsample.0', inpl, 4 * w, 1, s1 * s2, 1, 0,
                                    p + '.downsample.1', False))
            inpl = 4 * w
    a = 'classifier.0'
    out.append(ConvInfo(a + '.convs.0.0', 2048, 256, 1, 1, 1, 0, a + '.convs.0.1', False))
    for i, r in ((1, 12), (2, 24), (3, 36)) if v3 else ((1, 6), (2, 12), (3, 18)):
        out.append(ConvInfo(f'{a}.convs.{i}.0', 2048, 256, 3, 1, r, r, f'{a}.convs.{i}.1', False))
    out.append(ConvInfo(a + '.convs.4.1', 2048, 256, 1, 1, 1, 0, a + '.convs.4.2', False))
    out.append(ConvInfo(a + '.project.0', 1280, 256, 1, 1, 1, 0, a + '.project.1', False))
    if v3:                                                # DeepLabHead = Sequential(ASPP, Conv3x3, BN, ReLU, Conv1x1)
        out.append(ConvInfo('classifier.1', 256, 256, 3, 1, 1, 1, 'classifier.2', False))
        out.append(ConvInfo('classifier.4', 256, 1, 1, 1, 1, 0, None, True))
        return out
    out.append(ConvInfo('decoder.conv1', 256, 48, 1, 1, 1, 0, 'decoder.bn1', False))
    out.append(ConvInfo('decoder.last_conv.0', 304, 256, 3, 1, 1, 1, 'decoder.last_conv.1', False))
    out.append(ConvInfo('decoder.last_conv.4', 256, 256, 3, 1, 1, 1, 'decoder.last_conv.5', False))
    out.append(ConvInfo('decoder.last_conv.8', 256, 1, 1, 1, 1, 0, None, True))
    return out


def trainable(encoder='resnet50'):
    """[(name, shape)] of trainable tensors in reference `named_parameters()` order."""
    out = []
    for c in conv_infos(encoder):
        out.append((c.name + '.weight', (c.cout, c.cin, c.k, c.k)))
        if c.bias:
            out.append((c.name + '.bias', (c.cout,)))
    return out


def norm_layers(encoder='resnet50'):
    """[(prefix, channels)] of the 62 (R50) norm layers in module order."""
    return [(c.norm, c.cout) for c in conv_infos(encoder) if c.norm is not None]


def neuron_lr_shape(shape):
    """Shape of the NEURON-level lr tensor, `meta_optim.py:54-56`."""
    return (shape[0],) + (1,) * (len(shape) - 1)


def model_state_keys(encoder='resnet50', norm='bn'):
    keys = []
    for c in conv_infos(encoder):
        keys.append(c.name + '.weight')
        if c.bias:
            keys.append(c.name + '.bias')
        if c.norm is not None:
            keys += [c.norm + '.weight', c.norm + '.bias']
            if norm == 'bn':
                keys += [c.norm + '.running_mean', c.norm + '.running_var',
                         c.norm + '.num_batches_tracked']
    return keys
