import sys, time, torch
sys.path.insert(0, '.')
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine
sd, lrs = synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50')
torch.cuda.init()
for (h, w, b) in [(480, 854, 8), (480, 910, 8), (480, 854, 8), (720, 1280, 3), (480, 854, 1)]:
    t0 = time.perf_counter(); e = Engine('resnet50', h, w, max_batch=b); torch.cuda.synchronize(); t1 = time.perf_counter()
    e.load_model_state(sd, lrs); torch.cuda.synchronize(); t2 = time.perf_counter()
    x, y = synthetic.synthetic_frames(1, h, w); xg, yg = x.cuda(), y.cuda()
    e._verify_pending = False
    e.finetune_step(xg, yg); t3 = time.perf_counter()
    e.finetune_step(xg, yg); t4 = time.perf_counter()
    e.close(); torch.cuda.synchronize(); t5 = time.perf_counter()
    print(f'{h}x{w} max_batch {b}: create {1e3*(t1-t0):.0f} ms, load state {1e3*(t2-t1):.0f} ms, first step {1e3*(t3-t2):.0f} ms, second {1e3*(t4-t3):.1f} ms, close {1e3*(t5-t4):.0f} ms')
