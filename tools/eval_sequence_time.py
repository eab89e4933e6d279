"""End-to-end time of the evaluation worker on one synthetic 480p sequence (two objects, 70 frames): e-OSVOS-50
(50 fine-tune iterations at batch 3 with the device-side first-frame augmentation, then inference of the other frames)
and e-OSVOS-100-OnA (100 + 10 iterations every 5 frames), with the objects one after the other / side by side and
frame-by-frame / batched inference.   python tools/eval_sequence_time.py [frames]"""
import json, os, sys, time, torch
sys.path.insert(0, '.')
from eosvos_amd import config, data, synthetic
from eosvos_amd import evaluate as ev
from eosvos_amd.helper_func import init_parent_model
from eosvos_amd.meta_optim import MetaOptimizer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 70
NSEQ = int(sys.argv[2]) if len(sys.argv) > 2 else 1
H, W = 480, 854
for name, extra in (('e-OSVOS-50', ['num_epochs.eval=50']),
                    ('e-OSVOS-100-OnA', ['e-OSVOS-OnA', 'num_epochs.eval=100', 'eval_online_adapt.num_epochs=10', 'eval_online_adapt.step=5'])):
    cfg = config.parse_cli(['with', 'DAVIS-2017', 'e-OSVOS'] + extra)
    cfg['datasets']['val'] = dict(cfg['datasets'].get('val', {}), name='synthetic', split='val', eval=True)
    ds = data.SyntheticSequences(NSEQ, N, H, W, seed=3)
    for in_flight, infer_batch in ((1, 1), (1, 8), (3, 8)):
        ev.INFER_BATCH = infer_batch
        model, _ = init_parent_model(**dict(cfg['parent_model']))
        model.to('cuda:0')
        model.max_batch = 3
        model.load_state_dict(synthetic.synthetic_state(cfg['parent_model']['encoder']))
        torch.manual_seed(1); mo = MetaOptimizer(model, **cfg['meta_optim_cfg'])      # (the lr init draws from torch's global RNG)
        msd = mo.state_dict()
        ev.evaluate_dataset(model, mo, msd, data.SyntheticSequences(1, 4, H, W, seed=3), dict(cfg, num_epochs=dict(cfg['num_epochs'], eval=2)),
                            'val', objects_in_flight=in_flight)                      # warm-up: engines, tap tables
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = ev.evaluate_dataset(model, mo, msd, ds, cfg, 'val', objects_in_flight=in_flight)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(json.dumps({'config': name, 'frames': N, 'objects': 2, 'objects_in_flight': in_flight, 'inference_batch': infer_batch,
                          'sequences': NSEQ, 'seconds_per_sequence': round(dt / NSEQ, 3), 'seconds_per_object': round(dt / NSEQ / 2, 3),
                          'ms_per_object_frame': round(1e3 * res['time_per_frame'], 2), 'mean_J': round(res['mean_J'], 4),
                          'phases_s': {k: round(v, 3) for k, v in res.get('phases', {}).items()}}), flush=True)
        for w in getattr(model, '_object_workers', None) or []:
            if w.model.engine is not None:
                w.model.engine.close()
        if model.engine is not None:
            model.engine.close()
