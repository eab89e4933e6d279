import json, os, sys, time, torch
sys.path.insert(0, '.')
from eosvos_amd import config, data, synthetic
from eosvos_amd import evaluate as ev
from eosvos_amd.helper_func import init_parent_model
from eosvos_amd.meta_optim import MetaOptimizer
N, H, W = 40, 480, 854
cfg = config.parse_cli(['with', 'DAVIS-2017', 'e-OSVOS', 'e-OSVOS-OnA', 'num_epochs.eval=100', 'eval_online_adapt.num_epochs=10', 'eval_online_adapt.step=5'])
cfg['datasets']['val'] = dict(cfg['datasets'].get('val', {}), name='synthetic', split='val', eval=True)
ds = data.SyntheticSequences(1, N, H, W, seed=3)
for in_flight in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3").split(",")]:
    model, _ = init_parent_model(**dict(cfg['parent_model']))
    model.to('cuda:0'); model.max_batch = 3
    model.load_state_dict(synthetic.synthetic_state('resnet50'))
    torch.manual_seed(1); mo = MetaOptimizer(model, **cfg['meta_optim_cfg'])      # (the lr init draws from torch's global RNG)
    msd = mo.state_dict()
    ev.evaluate_dataset(model, mo, msd, data.SyntheticSequences(1, 4, H, W, seed=3), dict(cfg, num_epochs=dict(cfg['num_epochs'], eval=2)), 'val', objects_in_flight=in_flight)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = ev.evaluate_dataset(model, mo, msd, ds, cfg, 'val', objects_in_flight=in_flight)
    torch.cuda.synchronize(); print('in flight', in_flight, 'seconds', round(time.perf_counter() - t0, 3), flush=True)
    for w in getattr(model, '_object_workers', None) or []:
        if w.model.engine is not None: w.model.engine.close()
    if model.engine is not None: model.engine.close()
