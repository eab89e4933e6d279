"""Bit-reproducibility of [K fine-tune steps at batch 3, then frame-by-frame inference] across repetitions in one process."""
import hashlib, sys, torch
sys.path.insert(0, '.')
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sd, lrs = synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50')
x, y = synthetic.synthetic_frames(3, 480, 854)
xg, yg = x.cuda(), y.cuda()
frames = torch.cat([torch.roll(xg[:1], 4 * i, dims=3) for i in range(12)])
eng = Engine('resnet50', 480, 854, max_batch=3)
eng.load_model_state(sd, lrs)
def h(t):
    return hashlib.md5(t.cpu().numpy().tobytes()).hexdigest()[:10]
for rep in range(4):
    eng.reset()
    for _ in range(K):
        eng.finetune_step(xg, yg, sync_loss=False)
    ph = h(eng.get_params())
    outs = [h(eng.infer(frames[i:i + 1].contiguous())) for i in range(12)]
    outs3 = [h(eng.infer(frames[i:i + 3].contiguous())) for i in range(0, 12, 3)]
    print('rep', rep, 'params', ph, 'infer b1', outs[:4], outs[-2:], 'infer b3', outs3[:2], flush=True)
