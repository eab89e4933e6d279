import sys,time,torch
sys.path.insert(0,".")
from eosvos_amd import synthetic
from eosvos_amd.engine import Engine
for B in (1,3):
    eng=Engine("resnet50",480,854,max_batch=B); eng.load_model_state(synthetic.synthetic_state("resnet50"), synthetic.synthetic_lrs("resnet50"))
    x,y=synthetic.synthetic_frames(B,480,854); xg,yg=x.cuda(),y.cuda()
    for _ in range(3): eng.finetune_step(xg,yg,sync_loss=False)
    eng.synchronize(); t0=time.perf_counter()
    for _ in range(50): eng.finetune_step(xg,yg,sync_loss=False)
    eng.synchronize(); print("B",B,"ms/step %.2f"%((time.perf_counter()-t0)*20)); eng.close()
