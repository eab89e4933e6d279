"""Stand-ins that let the UNMODIFIED reference modules import in this container.

Used only by `make_golden.py` (build container, /root/reference present).  The
reference depends on torchvision 0.4, cv2, imageio, davis, prettytable, sacred --
none installed here.  torchvision's ResNet / ASPP / IntermediateLayerGetter are
restated below with plain torch.nn from their published architecture (attribute and
state-dict names must match because the reference indexes
`.layer3[i].conv1/.conv2/.downsample[0]`); everything else is a permissive dummy so
`util.helper_func` imports.  Nothing here is reference code.
"""
import sys
import types
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation,
                               dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(out + idt)


class ResNet(nn.Module):
    def __init__(self, layers, replace_stride_with_dilation=None):
        super().__init__()
        rswd = replace_stride_with_dilation or [False, False, False]
        self.inplanes, self.dilation = 64, 1
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], 2, rswd[0])
        self.layer3 = self._make_layer(256, layers[2], 2, rswd[1])
        self.layer4 = self._make_layer(512, layers[3], 2, rswd[2])
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, 1000)

    def _make_layer(self, planes, blocks, stride=1, dilate=False):
        prev = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        ds = None
        if stride != 1 or self.inplanes != planes * 4:
            ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False),
                               nn.BatchNorm2d(planes * 4))
        L = [Bottleneck(self.inplanes, planes, stride, ds, prev)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            L.append(Bottleneck(self.inplanes, planes, dilation=self.dilation))
        return nn.Sequential(*L)


def resnet50(pretrained=False, **kw):
    return ResNet([3, 4, 6, 3], kw.get('replace_stride_with_dilation'))


def resnet101(pretrained=False, **kw):
    return ResNet([3, 4, 23, 3], kw.get('replace_stride_with_dilation'))


class IntermediateLayerGetter(nn.ModuleDict):
    def __init__(self, model, return_layers):
        rl = dict(return_layers)
        layers = OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            rl.pop(name, None)
            if not rl:
                break
        super().__init__(layers)
        self.return_layers = dict(return_layers)

    def forward(self, x):
        out = OrderedDict()
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


class ASPPPooling(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(nn.AdaptiveAvgPool2d(1), nn.Conv2d(cin, cout, 1, bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU())

    def forward(self, x):
        size = x.shape[-2:]
        for m in self:
            x = m(x)
        return F.interpolate(x, size=size, mode='bilinear', align_corners=False)


class ASPP(nn.Module):
    def __init__(self, in_channels, atrous_rates):
        super().__init__()
        oc = 256
        mods = [nn.Sequential(nn.Conv2d(in_channels, oc, 1, bias=False), nn.BatchNorm2d(oc), nn.ReLU())]
        for r in atrous_rates:
            mods.append(nn.Sequential(nn.Conv2d(in_channels, oc, 3, padding=r, dilation=r, bias=False),
                                      nn.BatchNorm2d(oc), nn.ReLU()))
        mods.append(ASPPPooling(in_channels, oc))
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(nn.Conv2d(5 * oc, oc, 1, bias=False), nn.BatchNorm2d(oc),
                                     nn.ReLU(), nn.Dropout(0.5))

    def forward(self, x):
        return self.project(torch.cat([c(x) for c in self.convs], dim=1))


class DeepLabHead(nn.Sequential):
    """torchvision 0.4 `segmentation/deeplabv3.py`: ASPP[12, 24, 36] -> Conv3x3 -> BN -> ReLU -> Conv1x1."""

    def __init__(self, in_channels, num_classes):
        super().__init__(ASPP(in_channels, [12, 24, 36]), nn.Conv2d(256, 256, 3, padding=1, bias=False), nn.BatchNorm2d(256),
                         nn.ReLU(), nn.Conv2d(256, num_classes, 1))


class SimpleSegmentationModel(nn.Module):
    """torchvision 0.4 `segmentation/_utils.py::_SimpleSegmentationModel` (base of DeepLabV3): backbone features 'out' ->
    classifier -> bilinear resize to the input size, align_corners=False; result dict {'out': logits}."""

    def __init__(self, backbone, classifier, aux_classifier=None):
        super().__init__()
        self.backbone, self.classifier, self.aux_classifier = backbone, classifier, aux_classifier

    def forward(self, x):
        input_shape = x.shape[-2:]
        features = self.backbone(x)
        result = OrderedDict()
        x = self.classifier(features['out'])
        result['out'] = F.interpolate(x, size=input_shape, mode='bilinear', align_corners=False)
        return result


class _Any:
    """Permissive dummy: any attribute / call / subscript yields another dummy."""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, n):
        if n.startswith('__'):
            raise AttributeError(n)
        return _Any()

    def __call__(self, *a, **k):
        return _Any()

    def __getitem__(self, k):
        return _Any()


def _dummy_getattr(n):
    if n.startswith('__'):
        raise AttributeError(n)
    return _Any()


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


def install(ref_src='/root/reference/src'):
    sys.dont_write_bytecode = True
    if ref_src not in sys.path:
        sys.path.insert(0, ref_src)
    res = _mod('torchvision.models.resnet', resnet50=resnet50, resnet101=resnet101,
               ResNet=ResNet, Bottleneck=Bottleneck)
    dl = _mod('torchvision.models.segmentation.deeplabv3', ASPP=ASPP,
              DeepLabV3=type('DeepLabV3', (SimpleSegmentationModel,), {}), DeepLabHead=DeepLabHead)
    seg = _mod('torchvision.models.segmentation', deeplabv3=dl)
    ut = _mod('torchvision.models.utils', load_state_dict_from_url=lambda *a, **k: {})
    _ut = _mod('torchvision.models._utils', IntermediateLayerGetter=IntermediateLayerGetter)
    dummy_nn = type('DummyModule', (nn.Module,), {})
    det_sub = {}
    for sub in ('backbone_utils', 'roi_heads', 'rpn', 'transform', 'faster_rcnn', 'mask_rcnn',
                'image_list', 'generalized_rcnn', '_utils'):
        det_sub[sub] = _mod('torchvision.models.detection.' + sub)
        det_sub[sub].__getattr__ = _dummy_getattr
    det = _mod('torchvision.models.detection', MaskRCNN=dummy_nn, **det_sub)
    det.__getattr__ = _dummy_getattr
    models = _mod('torchvision.models', resnet=res, segmentation=seg, utils=ut, _utils=_ut,
                  detection=det)
    ops_b = _mod('torchvision.ops.boxes')
    ops_b.__getattr__ = _dummy_getattr
    ops_m = _mod('torchvision.ops.misc', FrozenBatchNorm2d=dummy_nn)
    ops_m.__getattr__ = _dummy_getattr
    ops = _mod('torchvision.ops', boxes=ops_b, misc=ops_m)
    ops.__getattr__ = _dummy_getattr
    tr = _mod('torchvision.transforms', Compose=lambda ts: (lambda x: x))
    tr.__getattr__ = _dummy_getattr
    _mod('torchvision', models=models, ops=ops, transforms=tr)
    for name in ('cv2', 'imageio', 'davis', 'prettytable', 'matplotlib', 'matplotlib.pyplot',
                 'sacred', 'visdom', 'tensorboardX', 'PIL', 'PIL.Image', 'scipy.misc'):
        if name in sys.modules and name.split('.')[0] in ('PIL', 'matplotlib'):
            continue
        try:
            __import__(name)
        except Exception:
            m = _mod(name)
            m.__getattr__ = _dummy_getattr
