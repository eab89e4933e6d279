"""Objects side by side == one after the other, bit for bit, at 480x854 (the GPU test checks 96x160)."""
import sys, torch
sys.path.insert(0, '.')
from eosvos_amd import config, data, synthetic
from eosvos_amd.evaluate import finetune_object, object_workers, run_objects_in_flight
from eosvos_amd.helper_func import init_parent_model
from eosvos_amd.meta_optim import MetaOptimizer
cfg = config.parse_cli(['with', 'DAVIS-2017', 'e-OSVOS-OnA', 'num_epochs.eval=6', 'eval_online_adapt.num_epochs=3', 'eval_online_adapt.step=4'])
model, _ = init_parent_model(**dict(cfg['parent_model'])); model.to('cuda:0'); model.max_batch = 8
model.load_state_dict(synthetic.synthetic_state(cfg['parent_model']['encoder']))
mo = MetaOptimizer(model, **cfg['meta_optim_cfg']); msd = mo.state_dict()
ds = data.SyntheticSequences(1, 10, 480, 854, seed=3)
frames, gts = ds.sequence_tensors(ds.seqs_names[0], 'cuda:0')
gts = gts + [torch.roll(gts[0], 40, dims=2)]                      # three objects
workers = object_workers(model, mo, cfg['meta_optim_cfg'], 3)
ok = True
for rep in range(3):
    res = run_objects_in_flight(workers, msd, frames, gts, cfg)
    torch.cuda.synchronize()
    model.set_wg_budget(256)
    one = [finetune_object(model, mo, msd, frames, g, cfg) for g in gts]
    eq = [h2 == h1 and torch.equal(p2, p1) for (p2, h2), (p1, h1) in zip(res, one)]
    print('rep', rep, 'objects equal:', eq, flush=True)
    ok = ok and all(eq)
print('OK' if ok else 'MISMATCH')
