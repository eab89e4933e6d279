"""A/B timing of representative layers (tuning aid).  EOSVOS_LIB selects the build.
    python tools/kernel_ab.py [rounds]
Layers: decoder 3x3 (60,61), ASPP d=6 (54), layer4 conv2 (44), layer4 1x1 (46,47), layer3 (25,27),
layer2 (12,14), layer1 (2,3)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eosvos_amd import synthetic  # noqa: E402
from eosvos_amd.engine import Engine  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 3
LAYERS = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else None
eng = Engine('resnet50', 480, 854, max_batch=BATCH)
eng.load_model_state(synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50'))
x, y = synthetic.synthetic_frames(BATCH, 480, 854)
eng.finetune_step(x.cuda(), y.cuda())          # real activations / gradients in the buffers
layers = LAYERS or [60, 61, 54, 44, 46, 47, 25, 27, 12, 14, 2, 3]
res = {}
tms = {}
for r in range(rounds):
    for ci in layers:
        for kind in (0, 1, 2):
            ms, tf = eng.bench_conv(ci, kind, BATCH, reps=10)
            res.setdefault((ci, kind), []).append(tf)
            tms.setdefault((ci, kind), []).append(ms)
print('lib', os.environ.get('EOSVOS_LIB', 'default'), 'probe %.1f' % eng.mfma_probe())
tot = 0
for ci in layers:
    print('conv %2d  fwd %6.1f  dgrad %6.1f  wgrad %6.1f  TF/s   us: %6.1f %6.1f %6.1f' % (ci, *[max(res[(ci, k)]) for k in (0, 1, 2)], *[1e3 * min(tms[(ci, k)]) for k in (0, 1, 2)]))
