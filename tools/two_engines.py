"""Aggregate fine-tune step rate of N engines (independent objects / meta tasks) sharing one GPU, each on
its own stream pair: fills the tails and small grids a single batch-1 chain leaves idle."""
import sys, time, torch
sys.path.insert(0, ".")
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
BUDGET = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # eosvos_set_wg_budget of every engine
for n in (1, 2, 3, 4):
    engs = []
    for i in range(n):
        with torch.cuda.stream(torch.cuda.Stream()):
            e = Engine("resnet50", 480, 854, max_batch=B)
            e.load_model_state(synthetic.synthetic_state("resnet50"), synthetic.synthetic_lrs("resnet50"))
            e.set_wg_budget(BUDGET)
        engs.append(e)
    def step(e):
        with torch.cuda.stream(e.stream):          # an engine is called on the stream it is bound to
            e.finetune_step(xg, yg, sync_loss=False)
    x, y = synthetic.synthetic_frames(B, 480, 854); xg, yg = x.cuda(), y.cuda()
    torch.cuda.synchronize()
    for _ in range(3):
        for e in engs: step(e)
    for e in engs: e.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        for e in engs: step(e)
    t1 = time.perf_counter()
    for e in engs: e.synchronize()
    dt = time.perf_counter() - t0
    print("B", B, "budget", BUDGET, "engines", n, "ms per step (aggregate) %.2f" % (dt * 100 / n), " enqueue ms/step %.2f" % ((t1 - t0) * 100 / n))
    for e in engs: e.close()
