"""Per-launch time of every conv (fwd / dgrad / wgrad incl. fix-ups and transforms, one stream) under the current
environment -> JSON on stdout: {"ci:kind": us}.   python tools/layer_times.py [batch]
Run it under different EOSVOS_TUNE_* settings and diff the outputs (tools/layer_times_diff.py)."""
import json, sys
sys.path.insert(0, '.')
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine
from eosvos_amd.topology import conv_infos
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
e = Engine('resnet50', 480, 854, max_batch=B)
e.load_model_state(synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50'))
x, y = synthetic.synthetic_frames(B, 480, 854)
e.finetune_step(x.cuda(), y.cuda())
infos = conv_infos('resnet50')
out = {}
for ci in range(1, len(infos)):
    for kind in (0, 1, 2):
        try:
            out[f'{ci}:{kind}'] = round(e.bench_conv(ci, kind, B, reps=10)[0] * 1e3, 2)
        except Exception:
            pass
print(json.dumps(out))
