"""Is a fine-tune trajectory bit-reproducible while another engine runs beside it?"""
import sys, torch
sys.path.insert(0, '.')
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (96, 160)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 3
sd, lrs = synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50')
x, y = synthetic.synthetic_frames(B, H, W, seed=5)
xg, yg = x.cuda(), y.cuda()
engs = []
for i in range(2):
    with torch.cuda.stream(torch.cuda.Stream() if i else torch.cuda.current_stream()):
        e = Engine('resnet50', H, W, max_batch=B)
        e.load_model_state(sd, lrs)
    engs.append(e)
torch.cuda.synchronize()
def run(which, steps=int(sys.argv[4]) if len(sys.argv) > 4 else 4):
    for e in which:
        with torch.cuda.stream(e.stream):
            e.load_model_state(sd, lrs)
    torch.cuda.synchronize()
    losses = {id(e): [] for e in which}
    for _ in range(steps):
        for e in which:
            with torch.cuda.stream(e.stream):
                losses[id(e)].append(e.finetune_step(xg, yg, sync_loss=False))
    out = []
    for e in which:
        e.synchronize()
        out.append(e.get_params().clone())
    torch.cuda.synchronize()
    return out
for budget in (0, 256):
    for e in engs: e.set_wg_budget(budget)
    solo0 = run([engs[0]])[0]
    solo0b = run([engs[0]])[0]
    solo1 = run([engs[1]])[0]
    both = run(engs)
    both2 = run(engs)
    print('budget', budget, 'solo repeat equal', torch.equal(solo0, solo0b), '| engine1 solo == engine0 solo', torch.equal(solo0, solo1),
          '| concurrent == solo: e0', torch.equal(both[0], solo0), 'e1', torch.equal(both[1], solo1),
          '| concurrent repeat equal', torch.equal(both[0], both2[0]), torch.equal(both[1], both2[1]),
          '| max diff', float((both[0] - solo0).abs().max()), float((both[1] - solo1).abs().max()))
