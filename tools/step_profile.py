"""Runs a few fine-tune steps at a given batch (profiling target for rocprofv3 --kernel-trace)."""
import os
import sys

os.environ.setdefault('EOSVOS_MODE_GUARD', '0')      # profiling target: no range-guard forwards in the trace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eosvos_amd import synthetic  # noqa: E402
from eosvos_amd.engine import Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = Engine('resnet50', 480, 854, max_batch=B, norm=os.environ.get('EOSVOS_STEP_NORM', 'bn'))      # EOSVOS_STEP_NORM=gn: GroupNorm(16) mode
eng.load_model_state(synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50'))
x, y = synthetic.synthetic_frames(B, 480, 854)
xg, yg = x.cuda(), y.cuda()
for _ in range(6):
    eng.finetune_step(xg, yg, sync_loss=False)
eng.synchronize()
eng.close()
