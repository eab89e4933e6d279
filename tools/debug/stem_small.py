"""One fine-tune step at 97x163 (which stem kernels run there):  rocprofv3 --kernel-trace --stats -- python3 tools/debug/stem_small.py"""
import sys
sys.path.insert(0, '.')
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine
e = Engine('resnet50', 97, 163, max_batch=1)
e.load_model_state(synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50'))
x, y = synthetic.synthetic_frames(1, 97, 163, seed=9)
print(e.finetune_step(x.cuda(), y.cuda()))
e.close()
