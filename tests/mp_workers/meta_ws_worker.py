"""One rank of `MetaTrainer.meta_iteration` on CPU (gloo) with the stand-in engine: several tasks per rank on several
engines (the concurrent-task path), a NaN task on one rank, a rank with a short task list (the tail of a YouTube-VOS
pass).  usage: meta_ws_worker.py OUT_PREFIX META_BATCH_SIZE"""
import os
import sys

import torch
import torch.distributed as dist

import common
from eosvos_amd import synthetic
from eosvos_amd.meta_run import MetaTrainer, shard_tasks

out, mbs = sys.argv[1], int(sys.argv[2])
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
if world > 1:
    dist.init_process_group('gloo')
sub = max(1, mbs // world)
engines = [common.FakeEngine('resnet50', common.H, common.W, 1) for _ in range(min(sub, 2))]
mt = MetaTrainer(engines[0], dist=dist if world > 1 else None, meta_batch_size=mbs, extra_engines=engines[1:])
sd = synthetic.synthetic_state('resnet50')
mt.load_state(sd, synthetic.synthetic_lrs('resnet50'))
collectives = 0
if world > 1:
    real = dist.all_reduce

    def counted(t, *a, **k):
        global collectives
        collectives += 1
        return real(t, *a, **k)
    dist.all_reduce = counted


def task(t, it, nan=False):
    x, y = synthetic.synthetic_frames(1, common.H, common.W, seed=1000 + t + 100 * it)
    if nan:
        x = x * float('nan')
    return (x, y, torch.flip(x, dims=[3]).contiguous(), torch.flip(y, dims=[3]).contiguous())


losses = []
for it in range(4):
    ids = shard_tasks(mbs, rank, world)
    if it == 3:                                   # the last pass of an epoch is short: the highest task ids do not exist
        ids = [t for t in ids if t < mbs - 3]
    tasks = [task(t, it, nan=(it == 1 and t == 2)) for t in ids]          # task 2 of iteration 1 diverges (meta_run.py:209-211)
    losses.append(mt.meta_iteration(tasks, inner_steps=2))
torch.save({'state': mt.state.clone(), 'step': mt.step, 'skipped': mt.skipped_tasks, 'collectives': collectives, 'losses': losses,
            'engines': len(mt.engines)}, f'{out}.{rank}')
if world > 1:
    dist.destroy_process_group()
