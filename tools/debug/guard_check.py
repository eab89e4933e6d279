"""EOSVOS_DEBUG_GUARD=1 python tools/debug/guard_check.py [H W B]: run the engine's entry points, then look for writes
outside the engine's buffers."""
import os, sys, random, torch
os.environ['EOSVOS_DEBUG_GUARD'] = '1'
sys.path.insert(0, '.')
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine
from eosvos_amd.custom_transforms import FirstFrameAugmenter
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (96, 160)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 3
sd, lrs = synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50')
e = Engine('resnet50', H, W, max_batch=B)
e.load_model_state(sd, lrs)
def chk(what):
    n = e.lib.eosvos_debug_check_guards(e.h)
    print(what, '-> overwritten guard words:', n, flush=True)
chk('after create / load')
for b in range(1, B + 1):
    x, y = synthetic.synthetic_frames(b, H, W, seed=5)
    xg, yg = x.cuda(), y.cuda()
    e.forward(xg); chk(f'forward B={b}')
    e.load_model_state(sd, lrs)
    e.finetune_step(xg, yg); chk(f'finetune_step B={b}')
    e.infer(xg[:1].contiguous()); chk(f'infer after B={b}')
random.seed(3)
im, lab, _ = FirstFrameAugmenter(e).batch(xg[0].contiguous(), yg[0].contiguous(), B); chk('augment')
e.meta_task_begin()
for _ in range(2):
    e.finetune_step(xg[:1].contiguous(), yg[:1].contiguous())
g = torch.zeros(e.n_lr_store + e.n_param, device='cuda')
e.meta_grad(xg[:1].contiguous(), yg[:1].contiguous(), g); chk('meta task')
for budget in (256, 128):
    e.set_wg_budget(budget); e.load_model_state(sd, lrs)
    e.finetune_step(xg, yg); chk(f'finetune_step budget {budget}')
