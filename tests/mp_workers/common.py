"""Shared set-up of the multi-process CPU workers (TEST INFRASTRUCTURE): repo on sys.path, the stand-in engine swapped in
for the HIP engine inside THIS process only, a small frame size."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS, os.path.join(TESTS, 'golden')):
    if p not in sys.path:
        sys.path.insert(0, p)

from fake_engine import FakeDeepLab, FakeEngine  # noqa: E402

H, W = 16, 24


def fake_init_parent_model(architecture='DeepLabV3Plus', encoder='resnet50', batch_norm=None, **_kw):
    assert architecture == 'DeepLabV3Plus'
    return FakeDeepLab(encoder, num_classes=1, batch_norm=batch_norm), {}
