"""TEST INFRASTRUCTURE.  One rank, backend nccl (= RCCL), world_size 1, on cuda:0: `MetaTrainer.meta_iteration` through the
real `torch.distributed.all_reduce` -- RCCL initialisation, device binding and the stream ordering between the engines'
streams, torch's current stream and RCCL's run on hardware (a node's N > 1 path differs only in the ring).
usage: nccl_ws1_worker.py OUT"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29591')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from eosvos_amd import synthetic  # noqa: E402
from eosvos_amd.engine import Engine  # noqa: E402
from eosvos_amd.meta_run import MetaTrainer  # noqa: E402

H, W = 96, 160
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda:0'))
sd = synthetic.synthetic_state('resnet50')
lrs = synthetic.synthetic_lrs('resnet50')
calls = {'n': 0, 'mode_flags': 0}
real_all_reduce = dist.all_reduce


def counted(t, *a, **k):
    assert t.is_cuda
    if t.numel() == 1:                     # the collective matrix-mode verdict of the first meta-iteration after load_state
        calls['mode_flags'] += 1
        return real_all_reduce(t, *a, **k)
    calls['n'] += 1
    assert t.numel() == 40318387
    return real_all_reduce(t, *a, **k)


dist.all_reduce = counted
out = {}
for tag, tpr in (('one', 1), ('two', 2)):
    states = []
    for d in (dist, None):
        engines = []
        for _ in range(tpr):
            with torch.cuda.stream(torch.cuda.Stream()):
                engines.append(Engine('resnet50', H, W, max_batch=1, device='cuda:0', side_stream=False))
        mt = MetaTrainer(engines[0], dist=d, meta_batch_size=tpr, extra_engines=engines[1:])
        mt.load_state(sd, lrs)
        tasks = []
        for t in range(tpr):
            x, y = synthetic.synthetic_frames(1, H, W, seed=1000 + t)
            x, y = x.cuda(), y.cuda()
            tasks.append((x, y, torch.flip(x, dims=[3]).contiguous(), torch.flip(y, dims=[3]).contiguous()))
        losses = []
        s0 = mt.state.clone()
        for _ in range(2):
            losses.append(mt.meta_iteration(tasks, inner_steps=2))
        torch.cuda.synchronize()
        states.append((mt.state.clone().cpu(), losses, mt.step, float((mt.state - s0).abs().max())))
        for e in reversed(engines):        # the extra engines alias the first one's learned state: it goes last
            e.close()
    out[tag] = {'equal': bool(torch.equal(states[0][0], states[1][0])), 'losses': states[0][1], 'losses_ref': states[1][1],
                'step': states[0][2], 'finite': bool(torch.isfinite(states[0][0]).all()),
                'moved': states[0][3]}
out['all_reduce_calls'] = calls['n']
out['mode_flag_calls'] = calls['mode_flags']
out['backend'] = dist.get_backend()
dist.destroy_process_group()
torch.save(out, sys.argv[1])
