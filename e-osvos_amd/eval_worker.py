"""The concurrent validation process of meta-training (BASELINE configs[4]; `src/train_meta.py:132-186`,
`src/util/evaluate.py:34-40,361-382`).

The reference spawns one `evaluate()` process per dataset with `eval: True`; it polls a shared flag, deep-copies the
shared meta-optimizer state, evaluates, and writes `last_{key}_meta_iter.model` / `best_{key}_meta_iter.model`.  Here
it is an ordinary child process (started by `train_meta.main` BEFORE the parent touches the GPU, never by re-executing a
GPU-initialised process) that reads checkpoint SNAPSHOTS: rank 0 of the trainer atomically replaces
`{run_dir}/eval_snapshot.model` every `vis_interval` meta-iterations; this process picks up each new snapshot, runs
`evaluate.evaluate_dataset` for every eval dataset on its own engine / GPU (it may share a GPU with a meta rank: the
reference cannot express 8 GPUs + 1 eval rank because 8 % 7 != 0), appends one JSON line per
(meta_iter, dataset) to `{run_dir}/eval_log.jsonl`, and exits when `{run_dir}/eval_stop` exists and no newer snapshot is
pending.

    python -m eosvos_amd.eval_worker --run-dir models/run --config cfg.json --device cuda:7 [--data-root data]
"""
import argparse
import json
import os
import sys
import time

import torch


def _datasets(cfg, data_root, height, width, num_frames):
    """{key: reader} for every dataset with `eval: True` whose root exists; else one synthetic set."""
    from . import data
    out = {}
    for key, ds in cfg['datasets'].items():
        if not ds.get('eval') or isinstance(ds.get('name'), list) or ds.get('split') is None:
            continue
        if os.path.isdir(os.path.join(data_root, ds['name'])):
            out[key] = data.open_dataset(ds['name'], ds['split'], data_root, multi_object=cfg['data_cfg']['multi_object'],
                                         normalize=cfg['data_cfg']['normalize'],
                                         full_resolution=cfg['data_cfg']['full_resolution'])
    if not out:
        key = next((k for k, d in cfg['datasets'].items() if d.get('eval')), 'val')
        cfg['datasets'].setdefault(key, {'name': 'synthetic', 'split': 'val', 'eval': True})
        out[key] = data.SyntheticSequences(1, num_frames, height, width, seed=cfg['seed'])
    return out


def run(cfg, run_dir, device, data_root='data', height=480, width=854, num_frames=12, poll=0.25, once=False,
        init_parent_model=None, MetaOptimizer=None, log=print, parent_pid=None):
    from .checkpoint import load_meta_checkpoint
    from .evaluate import evaluate_dataset
    from .train_meta import load_parent_state
    if init_parent_model is None:
        from .helper_func import init_parent_model
    if MetaOptimizer is None:
        from .meta_optim import MetaOptimizer
    snap = os.path.join(run_dir, 'eval_snapshot.model')
    stop = os.path.join(run_dir, 'eval_stop')
    model, parent_states = init_parent_model(**cfg['parent_model'])
    model.to(device)
    model.max_batch = max(model.max_batch, cfg['data_cfg']['batch_sizes']['train'])
    enc = cfg['parent_model']['encoder']
    load_parent_state(model, parent_states, None, enc)                # shapes for MetaOptimizer; per-dataset state below
    meta_optim = MetaOptimizer(model, **cfg['meta_optim_cfg'])
    readers = _datasets(cfg, data_root, height, width, num_frames)
    best = {k: 0.0 for k in readers}
    seen, done = None, []
    while True:
        mtime = os.path.getmtime(snap) if os.path.exists(snap) else None
        if mtime is not None and mtime != seen:
            seen = mtime
            sd, info = load_meta_checkpoint(snap)
            for key, ds in readers.items():
                # the parent checkpoint of THIS dataset key (weights the learned init replaces + the frozen norm
                # statistics it does not), evaluate.py:46-50
                load_parent_state(model, parent_states, key, enc, log)
                res = evaluate_dataset(model, meta_optim, sd, ds, cfg, key, save_dir=run_dir, meta_iter=info['meta_iter'],
                                       meta_epoch=info['meta_epoch'], best_mean_J=best[key], device=device,
                                       vis_win_names=info.get('vis_win_names'))
                best[key] = res['best_mean_J']
                line = {'dataset': key, 'meta_iter': info['meta_iter'], 'mean_J': res['mean_J'], 'best_mean_J': best[key],
                        'J_seq': res['J_seq'], 'time_per_frame': res['time_per_frame']}
                with open(os.path.join(run_dir, 'eval_log.jsonl'), 'a') as f:
                    f.write(json.dumps(line) + '\n')
                log(json.dumps(dict(line, mode='concurrent_eval')))
                done.append(line)
            if once:
                return done
            continue
        if os.path.exists(stop):
            return done
        if parent_pid is not None and os.getppid() != parent_pid:     # the trainer died without writing eval_stop
            log(json.dumps({'mode': 'concurrent_eval', 'exit': 'parent process is gone'}))
            return done
        time.sleep(poll)


def main(argv=None, **hooks):
    ap = argparse.ArgumentParser()
    ap.add_argument('--run-dir', required=True)
    ap.add_argument('--config', required=True, help='JSON dump of the resolved configuration')
    ap.add_argument('--device', default='cuda:0')
    ap.add_argument('--data-root', default='data')
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=854)
    ap.add_argument('--num-frames', type=int, default=12)
    ap.add_argument('--once', action='store_true')
    ap.add_argument('--parent-pid', type=int, default=None, help='exit when this is no longer the parent process')
    a = ap.parse_args(argv)
    cfg = json.load(open(a.config))
    if a.device.startswith('cuda'):
        torch.cuda.set_device(torch.device(a.device))
    return run(cfg, a.run_dir, a.device, a.data_root, a.height, a.width, a.num_frames, once=a.once,
               parent_pid=a.parent_pid, **hooks)


if __name__ == '__main__':
    main()
    sys.exit(0)
