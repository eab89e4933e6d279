import cProfile, pstats, sys, time, torch
sys.path.insert(0, '.')
from eosvos_amd import config, data, synthetic
from eosvos_amd import evaluate as ev
from eosvos_amd.helper_func import init_parent_model
from eosvos_amd.meta_optim import MetaOptimizer
H, W = 480, 854
cfg = config.parse_cli(['with', 'DAVIS-2017', 'e-OSVOS', 'num_epochs.eval=50'])
model, _ = init_parent_model(**dict(cfg['parent_model']))
model.to('cuda:0'); model.max_batch = 8
model.load_state_dict(synthetic.synthetic_state(cfg['parent_model']['encoder']))
mo = MetaOptimizer(model, **cfg['meta_optim_cfg']); msd = mo.state_dict()
ds = data.SyntheticSequences(1, 6, H, W, seed=3)
frames, gts = ds.sequence_tensors(ds.seqs_names[0], 'cuda:0')
ev.finetune_object(model, mo, msd, frames, gts[0], dict(cfg, num_epochs=dict(cfg['num_epochs'], eval=3)))
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
ev.finetune_object(model, mo, msd, frames, gts[0], cfg)
torch.cuda.synchronize(); pr.disable(); dt = time.perf_counter() - t0
print('50 iterations + 5 frames: %.1f ms, %.2f ms per iteration' % (1e3 * dt, 1e3 * dt / 50))
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
