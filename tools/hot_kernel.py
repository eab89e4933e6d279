"""Runs the bench's roofline kernel (decoder.last_conv.0 forward, batch 3) a few times -- the
target of the rocprofv3 --pmc passes that produce roofline.traffic (profiles/)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eosvos_amd import synthetic  # noqa: E402
from eosvos_amd.engine import Engine  # noqa: E402

eng = Engine('resnet50', 480, 854, max_batch=3)
eng.load_model_state(synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50'))
x, y = synthetic.synthetic_frames(3, 480, 854)
eng.finetune_step(x.cuda(), y.cuda())
ms, fl = eng.time_hot_kernel(3, reps=5)
print('hot kernel %.4f ms  %.1f TFLOP/s' % (ms, fl / ms / 1e9))
eng.close()
