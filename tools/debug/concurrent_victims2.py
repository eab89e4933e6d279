import sys, torch
sys.path.insert(0, '.')
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine
from eosvos_amd.custom_transforms import warp_affine, INTER_CUBIC
H, W, B = 96, 160, 3
sd, lrs = synthetic.synthetic_state('resnet50'), synthetic.synthetic_lrs('resnet50')
x, y = synthetic.synthetic_frames(B, H, W, seed=5)
xg = x.cuda(); x2 = xg.clone()
engs = []
for i in range(2):
    with torch.cuda.stream(torch.cuda.Stream() if i else torch.cuda.current_stream()):
        e = Engine('resnet50', H, W, max_batch=B)
        e.load_model_state(sd, lrs)
    engs.append(e)
torch.cuda.synchronize()
e0, e1 = engs
src = xg[0].clone()
with torch.cuda.stream(e1.stream):
    ref = warp_affine(e1, src, 0, 17.0, 1.1, INTER_CUBIC)[0].clone()
torch.cuda.synchronize()
from collections import Counter
lanes, bad = Counter(), 0
for rep in range(300):
    with torch.cuda.stream(e0.stream):
        e0.forward(x2, want_logits=False)
    with torch.cuda.stream(e1.stream):
        got = warp_affine(e1, src, 0, 17.0, 1.1, INTER_CUBIC)[0].clone()
    torch.cuda.synchronize()
    d = got != ref
    if bool(d.any()):
        bad += 1
        for c, yy, xx in torch.nonzero(d).tolist():
            lanes[(xx % 64) // 16] += 1
print('wrong runs', bad, 'of 300; wrong pixels by 16-lane group of their wave:', dict(lanes))
