"""Entry point with the reference's command-line grammar (`src/train_meta.py:21-27,41-47`):

    python -m eosvos_amd.train_meta with DAVIS-2017 e-OSVOS num_epochs.eval=50           # evaluation
    python -m eosvos_amd.train_meta with DAVIS-2017 e-OSVOS-OnA num_epochs.eval=100      # + online adapt.
    torchrun --nproc-per-node 8 -m eosvos_amd.train_meta with YouTube-VOS meta_batch_size=8  # meta-train

What is kept from the reference: config keys / named configs (config.py), the MetaOptimizer state
layout and the checkpoint files `{save_dir}/{env_suffix}/last_meta_iter.model`
(`train_meta.py:277-286`), `last_{key}_meta_iter.model` / `best_{key}_meta_iter.model` and the prediction PNGs
`{save_dir}/{run}/best_eval_preds/{name}/{split}/{seq}/{frame}.png` of the eval workers (`evaluate.py:68-90,332-382`),
`meta_optim_model_file` warm start (`:101-103`), `resume_meta_run_epoch_mode: LAST` (`:70-77`), EVAL mode when
`num_meta_processes_per_gpu == 0` (`:148-153`), the outer step (`:361-373`), the concurrent validation process
(`:175-186`).  What is replaced: the spawn + shared-CPU-memory worker protocol (`:155-201`,
`meta_run.py:88-99,237-243`) -> one process per GPU under torch.distributed (RCCL), tasks sharded over ranks, one
all-reduce per meta-iteration; evaluation work items (sequence, object) sharded over ranks.

Data: `datasets.*` are read from `{data_root}/{name}` (DAVIS-2016 / DAVIS-2017 / YouTube-VOS layouts, data.py) when
that directory exists; otherwise the seeded synthetic sequences of synthetic.py (`data=synthetic` in the log line) --
no dataset is reachable from the build or bench boxes.
"""
import json
import os
import random
import signal
import subprocess
import sys
import time

import torch

from . import config as config_mod
from . import data as data_mod
from . import synthetic
from .checkpoint import checkpoint_names, load_meta_checkpoint, save_meta_checkpoint
from .evaluate import evaluate_dataset
from .helper_func import init_parent_model, set_random_seeds
from .meta_optim import MetaOptimizer
from .meta_run import MetaTrainer
from .meta_tasksets import ConcatTaskset, MetaTaskset, task_order


def _dist():
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world == 1:
        return None, 0, 1, 0
    import torch.distributed as dist
    local = int(os.environ.get('LOCAL_RANK', 0))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    backend = os.environ.get('EOSVOS_DIST_BACKEND', 'nccl')      # 'gloo' for the CPU tests of the host logic
    if backend == 'nccl':
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device(f'cuda:{local}'))
    else:
        dist.init_process_group(backend)
    return dist, dist.get_rank(), world, local


def _train_tasksets(cfg, data_root):
    """MetaTaskset(s) of `datasets.train` (`meta_run.py:41-66`), or None when no dataset root exists."""
    tr = cfg['datasets']['train']
    names = tr['name'] if isinstance(tr['name'], list) else [tr['name']]
    splits = tr['split'] if isinstance(tr['split'], list) else [tr['split']]
    if not all(os.path.isdir(os.path.join(data_root, n)) for n in names):
        return None
    dc = cfg['data_cfg']
    sets = [MetaTaskset(data_mod.open_dataset(n, s, data_root, multi_object=dc['multi_object'], normalize=dc['normalize'],
                                              full_resolution=dc['full_resolution']), dc,
                        cfg['random_frame_transform_per_task'], cfg['random_flip_label'], cfg['random_no_label'],
                        cfg['single_obj_seq_mode'], cfg['random_box_coord_perm'], cfg['random_frame_epsilon'],
                        cfg['random_object_id_sub_group']) for n, s in zip(names, splits)]
    return ConcatTaskset(sets)


def load_parent_state(model, parent_states, key, encoder, log=None):
    """`model.load_state_dict(parent_states[key]['states'][0])` when `parent_model.<key>.paths` names a parent
    checkpoint (`train_meta.py:91-96`, `evaluate.py:46-50`); only without one the seeded synthetic weights / norm
    statistics (no pretrained file is reachable offline).  Returns 'file' or 'synthetic'."""
    states = (parent_states.get(key) or {}).get('states') or []
    if states:
        if len(states) > 1:
            raise NotImplementedError('more than one parent state per dataset (train_meta.py:92-93)')
        model.load_state_dict(states[0])
        src = 'file'
    else:
        model.load_state_dict(synthetic.synthetic_state(getattr(model, 'encoder', encoder)))    # ('deeplabv3_' + encoder for DeepLabV3)
        src = 'synthetic'
    if log is not None:
        log(json.dumps({'parent_state': src, 'dataset_key': key}))
    return src


def resume_checkpoint_name(mode):
    """`resume_meta_run_epoch_mode` -> file name (`train_meta.py:70-77`)."""
    if mode == 'LAST':
        return 'last_meta_iter.model'
    if mode and 'BEST' in mode:
        return f"best_{mode.split('_')[1].lower()}_meta_iter.model"
    raise NotImplementedError(mode)


def _start_eval_process(cfg, run_dir, device, data_root, height, width, num_frames, eval_cmd):
    """Child process for the datasets with `eval: True` (train_meta.py:175-186).  Must run before this process
    initialises the GPU."""
    os.makedirs(run_dir, exist_ok=True)
    for f in ('eval_stop', 'eval_snapshot.model'):
        if os.path.exists(os.path.join(run_dir, f)):
            os.remove(os.path.join(run_dir, f))
    cfg_path = os.path.join(run_dir, 'config.json')
    json.dump(cfg, open(cfg_path, 'w'))
    cmd = list(eval_cmd) if eval_cmd else [sys.executable, '-m', 'eosvos_amd.eval_worker']
    cmd += ['--run-dir', run_dir, '--config', cfg_path, '--device', device, '--data-root', data_root,
            '--height', str(height), '--width', str(width), '--num-frames', str(num_frames), '--parent-pid', str(os.getpid())]
    return subprocess.Popen(cmd, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _stop_eval_process(eval_proc, run_dir, timeout=600):
    """Ask the validation child to finish (it exits after the snapshot it is working on); kill it if it does not."""
    if eval_proc is None:
        return
    try:
        open(os.path.join(run_dir, 'eval_stop'), 'w').close()
        eval_proc.wait(timeout=timeout)
    except Exception:
        eval_proc.terminate()
        try:
            eval_proc.wait(timeout=30)
        except Exception:
            eval_proc.kill()


def main(argv=None, height=480, width=854, num_frames=12, num_meta_iters=None, data_root='data', eval_cmd=None,
         device=None):
    """`num_meta_iters`: None = run until SIGINT / SIGTERM like the reference's `while True` (`train_meta.py:207`; the
    iteration in flight is finished and checkpointed), or EOSVOS_NUM_META_ITERS; a number for tests and benchmarks."""
    cfg = config_mod.parse_cli(sys.argv[1:] if argv is None else argv)
    if num_meta_iters is None and os.environ.get('EOSVOS_NUM_META_ITERS'):
        num_meta_iters = int(os.environ['EOSVOS_NUM_META_ITERS'])
    # frame size / sequence length of the SYNTHETIC data (no dataset root): for the command line, which has no such
    # arguments in the reference's grammar (`EOSVOS_SYNTHETIC_SIZE=96x160 EOSVOS_SYNTHETIC_FRAMES=4 python -m eosvos_amd.train_meta with ...`)
    if os.environ.get('EOSVOS_SYNTHETIC_SIZE'):
        height, width = (int(v) for v in os.environ['EOSVOS_SYNTHETIC_SIZE'].lower().split('x'))
    if os.environ.get('EOSVOS_SYNTHETIC_FRAMES'):
        num_frames = int(os.environ['EOSVOS_SYNTHETIC_FRAMES'])
    run = cfg['env_suffix'] or 'run'
    run_dir = os.path.join(cfg['save_dir'], run)
    meta_mode = cfg['num_meta_processes_per_gpu'] != 0
    world_env, rank_env = int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('RANK', 0))
    # configs[4]: the validation process runs beside the meta ranks.  It is started first, while this process has not
    # touched the GPU yet; by default it shares the last GPU of the node with that rank's meta process.
    eval_proc = None
    has_eval = cfg['eval_datasets'] and any(d.get('eval') for d in cfg['datasets'].values())
    if meta_mode and has_eval and rank_env == 0 and eval_cmd is not False:      # eval_cmd=False: no validation process
        n_local = int(os.environ.get('LOCAL_WORLD_SIZE', world_env))
        eval_dev = device or (f'cuda:{cfg["num_eval_gpus"] - 1}' if cfg['num_eval_gpus'] else f'cuda:{max(n_local - 1, 0)}')
        eval_proc = _start_eval_process(cfg, run_dir, eval_dev, data_root, height, width, num_frames, eval_cmd)

    try:
        return _run(cfg, run, run_dir, meta_mode, eval_proc, height, width, num_frames, num_meta_iters, data_root, device)
    finally:
        # whatever ended the run (normal exit, data error, NaN assert, dist failure, KeyboardInterrupt): the validation
        # child must not keep polling for snapshots on a GPU it shares with a meta rank
        _stop_eval_process(eval_proc, run_dir)


def _run(cfg, run, run_dir, meta_mode, eval_proc, height, width, num_frames, num_meta_iters, data_root, device):
    dist, rank, world, local = _dist()
    dev = device or f'cuda:{local}'
    ck_last = checkpoint_names(cfg['save_dir'], run)['last']
    log = (lambda line: print(line, flush=True)) if rank == 0 else None

    pm = dict(cfg['parent_model'])
    model, parent_states = init_parent_model(**pm)
    model.to(dev)
    model.max_batch = max(model.max_batch, *cfg['data_cfg']['batch_sizes'].values())
    load_parent_state(model, parent_states, 'train', pm['encoder'], log)      # train_meta.py:91-96
    set_random_seeds(cfg['seed'])      # every rank draws the SAME initial lrs (the reference builds them once, in main)
    meta_optim = MetaOptimizer(model, **cfg['meta_optim_cfg'])
    meta_iter = 0
    if cfg['meta_optim_model_file']:
        sd, _ = load_meta_checkpoint(cfg['meta_optim_model_file'])
        meta_optim.load_state_dict(sd)
    if cfg['resume_meta_run_epoch_mode'] is not None:                          # train_meta.py:70-77,103-104
        ck_resume = os.path.join(run_dir, resume_checkpoint_name(cfg['resume_meta_run_epoch_mode']))
        sd, info = load_meta_checkpoint(ck_resume)                             # a missing file is an error, as in the reference
        meta_optim.load_state_dict(sd)
        meta_iter = info['meta_iter'] or 0

    if not meta_mode:                                                       # EVAL modus (train_meta.py:148-153)
        results = {}
        t0 = time.time()
        readers = {}
        for key, ds in cfg['datasets'].items():
            if cfg['eval_datasets'] and ds.get('eval') and not isinstance(ds['name'], list) and ds.get('split') and \
                    os.path.isdir(os.path.join(data_root, ds['name'])):
                dc = cfg['data_cfg']
                readers[key] = data_mod.open_dataset(ds['name'], ds['split'], data_root, multi_object=dc['multi_object'],
                                                     normalize=dc['normalize'], full_resolution=dc['full_resolution'])
        data_tag = 'files' if readers else 'synthetic'
        if not readers:
            cfg['datasets']['val'] = dict(cfg['datasets'].get('val', {}), name='synthetic', split='val', eval=True)
            readers['val'] = data_mod.SyntheticSequences(1, num_frames, height, width, seed=cfg['seed'])
        for key, reader in readers.items():
            load_parent_state(model, parent_states, key, pm['encoder'], log)   # evaluate.py:46-50
            results[key] = evaluate_dataset(model, meta_optim, meta_optim.state_dict(), reader, cfg, key, save_dir=run_dir,
                                            meta_iter=meta_iter, meta_epoch=0, dist=dist if world > 1 else None, device=dev)
        if dev.startswith('cuda'):
            torch.cuda.synchronize()
        if rank == 0:
            print(json.dumps({'mode': 'eval', 'data': data_tag, 'seconds': time.time() - t0,
                              'datasets': {k: {'mean_J': r['mean_J'], 'J_seq': r['J_seq'], 'time_per_frame': r['time_per_frame'],
                                               'labels_present': sorted({int(v) for l in r['labels'].values() for v in l.unique().tolist()})}
                                           for k, r in results.items()}}))
        if dist is not None:
            dist.destroy_process_group()
        return results

    # meta-training: tasks sharded over ranks, all-reduce + fused RAdam
    if cfg['meta_batch_size'] % world:
        raise ValueError('meta_batch_size must be a multiple of the number of ranks (train_meta.py:150)')
    sub = cfg['meta_batch_size'] // world
    oc = cfg['meta_optim_optim_cfg']
    if sub > 1 and dev.startswith('cuda'):
        model.side_stream = False          # several tasks per rank run side by side: one queue per engine (eosvos_set_side_stream)
    eng = model._ensure_engine(height, width, max(cfg['data_cfg']['batch_sizes'].values()))
    # several tasks per rank: up to 4 of them in flight together, one engine each on its own stream (a batch-1 task alone
    # leaves CUs idle in its tails; measured 26.8 -> 34.2 tasks/s per GPU at 480x854)
    extra_engines = []
    if sub > 1 and dev.startswith('cuda'):
        from .engine import Engine
        for _ in range(min(sub, 4) - 1):
            with torch.cuda.stream(torch.cuda.Stream()):
                extra_engines.append(Engine(model.encoder, height, width, max(cfg['data_cfg']['batch_sizes'].values()), dev,
                                            norm=model.norm, side_stream=False))
    mt = MetaTrainer(eng, dist=dist, extra_engines=extra_engines, meta_batch_size=cfg['meta_batch_size'], model_init_lr=oc['model_init_lr'],
                     log_init_lr_lr=oc['log_init_lr_lr'], model_init_weight_decay=oc['model_init_weight_decay'],
                     grad_clip=oc['grad_clip'], max_lr=cfg['meta_optim_cfg']['max_lr'],
                     lr_hierarchy_level=cfg['meta_optim_cfg']['lr_hierarchy_level'],
                     use_log_init_lr=cfg['meta_optim_cfg']['use_log_init_lr'], loss_func=cfg['loss_func'],
                     learn_model_init=cfg['meta_optim_cfg']['learn_model_init'], freeze_encoder=oc['freeze_encoder'])
    mt.load_state(model.state_dict(), [p.data for n, p in meta_optim.named_parameters() if n.startswith('log_init_lr')])
    if dist is not None:               # one learned state everywhere (the reference workers read main's shared tensors)
        dist.broadcast(mt.state, src=0)
        mt._push_state()
    set_random_seeds(cfg['seed'] + rank)                                    # meta_run.py:30
    tasksets = _train_tasksets(cfg, data_root)
    order, epoch = [], 0
    pool = prefetched = None
    if tasksets is not None:
        from concurrent.futures import ThreadPoolExecutor
        # EOSVOS_META_PREFETCH=0: decode on the main thread at the start of every iteration (for A/B timing)
        pool = ThreadPoolExecutor(max_workers=2) if os.environ.get('EOSVOS_META_PREFETCH', '1') != '0' else None

        def draw_and_prefetch():
            """The next sub-batch of this rank: the items (every random draw, on this thread, in order) and the decoding
            + colour jitter of their frames, started on the pool so that they run beside the GPU's current iteration."""
            nonlocal order, epoch
            if not order:                                                   # one worker's shuffled DataLoader, meta_run.py:76-85
                order = task_order(len(tasksets), sub, cfg['seed'] + rank, epoch)
                epoch += 1
            out = []
            for idx in order.pop(0):
                ts, i = tasksets.locate(idx)
                item = ts[i]
                out.append((ts, item, pool.submit(ts.load_frames, item) if pool else None))
            return out
        prefetched = draw_and_prefetch()
    # `while True` of train_meta.py:207: runs until stopped.  SIGINT / SIGTERM finish the iteration in flight, write
    # last_meta_iter.model and leave; every rank agrees on the stop through the flag summed into the all-reduce below.
    stop = {'flag': False}
    handlers = {}
    if num_meta_iters is None:
        for sig in (signal.SIGINT, signal.SIGTERM):
            try:
                handlers[sig] = signal.signal(sig, lambda *_: stop.__setitem__('flag', True))
            except ValueError:                                  # not the main thread (tests)
                pass
    it = 0
    while num_meta_iters is None or it < num_meta_iters:
        last = num_meta_iters is not None and it + 1 == num_meta_iters
        tasks = []
        if tasksets is not None:
            current = prefetched
            tasks = [ts.task_tensors(item, eng, dev, host=fut.result() if fut else None) for ts, item, fut in current]
            if not last:
                prefetched = draw_and_prefetch()
        else:
            for t in range(rank, cfg['meta_batch_size'], world):           # synthetic: task t of the meta-batch
                x, y = synthetic.synthetic_frames(1, height, width, seed=1000 + t + cfg['meta_batch_size'] * (meta_iter + it))
                xg, yg = x.to(dev), y.to(dev)
                tasks.append((xg, yg, torch.flip(xg, dims=[3]).contiguous(), torch.flip(yg, dims=[3]).contiguous()))
        losses = mt.meta_iteration(tasks, inner_steps=cfg['num_epochs']['train'], bptt_epochs=cfg['bptt_epochs'],
                                   multi_step_bptt_loss=cfg['multi_step_bptt_loss'] or None)
        done = meta_iter + it + 1
        if dist is not None and handlers:                       # one decision for all ranks (only a signal can set the flag)
            flag = torch.tensor([1.0 if stop['flag'] else 0.0], device=mt.state.device)
            dist.all_reduce(flag)
            stop['flag'] = bool(flag.item() > 0)
        last = last or stop['flag']
        if rank == 0:
            print(json.dumps({'mode': 'meta', 'data': 'files' if tasksets is not None else 'synthetic', 'meta_iter': done,
                              'meta_losses': losses, 'skipped_tasks': mt.skipped_tasks}), flush=True)
            if done == 1 or done % cfg['vis_interval'] == 0 or last:            # train_meta.py:275-286 (+ the last one)
                save_meta_checkpoint(ck_last, mt.state_dict(), done, 0)
                if eval_proc is not None:                                   # snapshot for the validation process (atomic)
                    tmp = os.path.join(run_dir, 'eval_snapshot.tmp')
                    save_meta_checkpoint(tmp, mt.state_dict(), done, 0)
                    os.replace(tmp, os.path.join(run_dir, 'eval_snapshot.model'))
        it += 1
        if last:
            break
    for sig, h in handlers.items():
        signal.signal(sig, h)
    if pool is not None:
        pool.shutdown(wait=True)
    if dist is not None:
        dist.destroy_process_group()
    return mt


if __name__ == '__main__':
    main()
