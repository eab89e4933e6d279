"""Entry point with the reference's command-line grammar (`src/train_meta.py:21-27,41-47`):

    python -m eosvos_amd.train_meta with DAVIS-2017 e-OSVOS num_epochs.eval=50           # evaluation
    python -m eosvos_amd.train_meta with DAVIS-2017 e-OSVOS-OnA num_epochs.eval=100      # + online adapt.
    torchrun --nproc-per-node 8 -m eosvos_amd.train_meta with YouTube-VOS meta_batch_size=8  # meta-train

What is kept from the reference: config keys / named configs (config.py), the MetaOptimizer state
layout and the checkpoint files `{save_dir}/{env_suffix}/last_meta_iter.model`
(`train_meta.py:277-286`), `meta_optim_model_file` warm start (`:101-103`),
`resume_meta_run_epoch_mode: LAST` (`:70-77`), EVAL mode when `num_meta_processes_per_gpu == 0`
(`:148-153`), the outer step (`:361-373`).  What is replaced: the spawn + shared-CPU-memory
worker protocol (`:155-201`, `meta_run.py:88-99,237-243`) -> one process per GPU under
torch.distributed (RCCL), tasks sharded over ranks, one all-reduce per meta-iteration.
The dataset layer (DAVIS / YouTube-VOS loaders, augmentation) is the next scope row
(SURVEY.md 8f.2): until it lands this entry point runs on the seeded synthetic sequences of
synthetic.py (`data=synthetic` in the log line).
"""
import json
import os
import sys
import time

import torch

from . import config as config_mod
from . import synthetic
from .checkpoint import checkpoint_names, load_meta_checkpoint, save_meta_checkpoint
from .evaluate import evaluate_sequence
from .helper_func import init_parent_model
from .meta_optim import MetaOptimizer
from .meta_run import MetaTrainer, shard_tasks


def _dist():
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world == 1:
        return None, 0, 1, 0
    import torch.distributed as dist
    local = int(os.environ.get('LOCAL_RANK', 0))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    torch.cuda.set_device(local)
    dist.init_process_group('nccl', device_id=torch.device(f'cuda:{local}'))
    return dist, dist.get_rank(), world, local


def main(argv=None, height=480, width=854, num_frames=12, num_meta_iters=2):
    cfg = config_mod.parse_cli(sys.argv[1:] if argv is None else argv)
    dist, rank, world, local = _dist()
    dev = f'cuda:{local}'
    run = cfg['env_suffix'] or 'run'
    ck_last = checkpoint_names(cfg['save_dir'], run)['last']

    pm = dict(cfg['parent_model'])
    model, _ = init_parent_model(**pm)
    model.to(dev)
    model.load_state_dict(synthetic.synthetic_state(pm['encoder']))       # no pretrained weights offline
    meta_optim = MetaOptimizer(model, **cfg['meta_optim_cfg'])
    meta_iter = 0
    if cfg['meta_optim_model_file']:
        sd, _ = load_meta_checkpoint(cfg['meta_optim_model_file'])
        meta_optim.load_state_dict(sd)
    if cfg['resume_meta_run_epoch_mode'] == 'LAST' and os.path.exists(ck_last):
        sd, info = load_meta_checkpoint(ck_last)
        meta_optim.load_state_dict(sd)
        meta_iter = info['meta_iter'] or 0

    if cfg['num_meta_processes_per_gpu'] == 0:                              # EVAL modus
        frames, gt = synthetic.synthetic_frames(1, height, width, seed=cfg['seed'], second_object=True)
        seq = torch.cat([torch.roll(frames, shifts=4 * i, dims=3) for i in range(num_frames)]).to(dev)
        objs = [(gt[0] * (torch.arange(height).view(-1, 1) < height // 2)).float(),
                (gt[0] * (torch.arange(height).view(-1, 1) >= height // 2)).float()]
        t0 = time.time()
        labels, _, hist = evaluate_sequence(model, meta_optim, meta_optim.state_dict(), seq, objs, cfg)
        torch.cuda.synchronize()
        dt = time.time() - t0
        if rank == 0:
            print(json.dumps({'mode': 'eval', 'data': 'synthetic', 'frames': num_frames, 'objects': len(objs),
                              'seconds_per_frame': dt / num_frames, 'final_train_loss': [h[0][-1] for h in hist],
                              'labels_present': sorted(int(v) for v in labels.unique().tolist())}))
        return labels

    # meta-training: tasks sharded over ranks, all-reduce + fused RAdam
    oc = cfg['meta_optim_optim_cfg']
    x0, _ = synthetic.synthetic_frames(1, height, width, seed=1)
    eng = model._ensure_engine(height, width, cfg['data_cfg']['batch_sizes']['train'])
    mt = MetaTrainer(eng, dist=dist, meta_batch_size=cfg['meta_batch_size'], model_init_lr=oc['model_init_lr'],
                     log_init_lr_lr=oc['log_init_lr_lr'], model_init_weight_decay=oc['model_init_weight_decay'],
                     grad_clip=oc['grad_clip'], max_lr=cfg['meta_optim_cfg']['max_lr'],
                     lr_hierarchy_level=cfg['meta_optim_cfg']['lr_hierarchy_level'],
                     use_log_init_lr=cfg['meta_optim_cfg']['use_log_init_lr'], loss_func=cfg['loss_func'],
                     learn_model_init=cfg['meta_optim_cfg']['learn_model_init'], freeze_encoder=oc['freeze_encoder'])
    mt.load_state(model.state_dict(), [p.data for n, p in meta_optim.named_parameters() if n.startswith('log_init_lr')])
    for it in range(num_meta_iters):
        tasks = []
        for t in shard_tasks(cfg['meta_batch_size'], rank, world):
            x, y = synthetic.synthetic_frames(1, height, width, seed=1000 + t + cfg['meta_batch_size'] * (meta_iter + it))
            xg, yg = x.to(dev), y.to(dev)
            tasks.append((xg, yg, torch.flip(xg, dims=[3]).contiguous(), torch.flip(yg, dims=[3]).contiguous()))
        losses = mt.meta_iteration(tasks, inner_steps=cfg['num_epochs']['train'], bptt_epochs=cfg['bptt_epochs'],
                                   multi_step_bptt_loss=cfg['multi_step_bptt_loss'] or None)
        if rank == 0:
            print(json.dumps({'mode': 'meta', 'meta_iter': meta_iter + it + 1, 'meta_losses': losses,
                              'skipped_tasks': mt.skipped_tasks}))
            if (meta_iter + it + 1) % cfg['vis_interval'] == 0 or it == num_meta_iters - 1:
                save_meta_checkpoint(ck_last, mt.state_dict(), meta_iter + it + 1, 0)
    if dist is not None:
        dist.destroy_process_group()
    return mt


if __name__ == '__main__':
    main()
