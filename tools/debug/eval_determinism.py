"""evaluate_dataset twice per configuration: are the label maps identical run to run, and between frame-by-frame and
batched inference?  usage: python tools/debug/eval_determinism.py [random_train_transform 0/1]"""
import hashlib, sys, torch
sys.path.insert(0, '.')
from eosvos_amd import config, data, synthetic
from eosvos_amd import evaluate as ev
from eosvos_amd.helper_func import init_parent_model
from eosvos_amd.meta_optim import MetaOptimizer
aug = (sys.argv[1] if len(sys.argv) > 1 else '1') == '1'
H, W, N = 480, 854, 16
cfg = config.parse_cli(['with', 'DAVIS-2017', 'e-OSVOS', 'num_epochs.eval=12', f'data_cfg.random_train_transform={aug}'])
cfg['datasets']['val'] = dict(cfg['datasets'].get('val', {}), name='synthetic', split='val', eval=True)
ds = data.SyntheticSequences(1, N, H, W, seed=3)
for infer_batch in (1, 8, 1):
    ev.INFER_BATCH = infer_batch
    for rep in range(2):
        model, _ = init_parent_model(**dict(cfg['parent_model']))
        model.to('cuda:0'); model.max_batch = 3
        model.load_state_dict(synthetic.synthetic_state('resnet50'))
        torch.manual_seed(1); mo = MetaOptimizer(model, **cfg['meta_optim_cfg'])      # (the lr init draws from torch's global RNG)
        msd = mo.state_dict()
        res = ev.evaluate_dataset(model, mo, msd, ds, cfg, 'val', objects_in_flight=1)
        lab = res['labels']['synthetic00']
        print('augment', aug, 'infer batch', infer_batch, 'rep', rep, 'labels md5', hashlib.md5(lab.numpy().tobytes()).hexdigest()[:10],
              'J', round(res['mean_J'], 4), 'pixels per frame', [int((lab[f] > 0).sum()) for f in (1, 5, 15)], flush=True)
        model.engine.close()
