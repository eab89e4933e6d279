import sys, torch
sys.path.insert(0, '.')
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine
mb, B, H, W = (int(v) for v in sys.argv[1:5])
eng = Engine('resnet50', H, W, max_batch=mb)
eng.load_model_state(synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50'))
x, y = synthetic.synthetic_frames(B, H, W)
for i in range(3):
    print('step', i, eng.finetune_step(x.cuda(), y.cuda()), flush=True)
eng.close()
