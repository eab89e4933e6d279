"""The library's own RCCL collective (`eosvos_comm_*`, `eosvos_allreduce_sum`) at world size 1 on the GPU: communicator made
through the C-ABI (no torch.distributed), `MetaTrainer(comm=...)` for two meta-iterations (one task, and two in flight) --
bit-identical to the trainer without a collective (the sum over one rank is the identity), i.e. the all-reduce is ordered
after the tasks' gradient accumulation and before the outer step.  Writes the result to argv[1]."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
import torch  # noqa: E402

from eosvos_amd import synthetic  # noqa: E402
from eosvos_amd.engine import Engine, RcclComm  # noqa: E402
from eosvos_amd.meta_run import MetaTrainer  # noqa: E402

H, W = 96, 160
torch.cuda.set_device(0)
sd = synthetic.synthetic_state('resnet50')
lrs = synthetic.synthetic_lrs('resnet50')
comm = RcclComm(1, 0, RcclComm.unique_id(), device=0)
# the raw collective: in place, asynchronous on the engine's stream
e0 = Engine('resnet50', H, W, max_batch=1, device='cuda:0')
v = torch.arange(1000, device='cuda', dtype=torch.float32)
e0.allreduce_sum(v, comm)
e0.synchronize()
out = {'identity': bool(torch.equal(v.cpu(), torch.arange(1000, dtype=torch.float32)))}
e0.close()
for tag, tpr in (('one', 1), ('two', 2)):
    states = []
    for c in (comm, None):
        engines = []
        for _ in range(tpr):
            with torch.cuda.stream(torch.cuda.Stream()):
                engines.append(Engine('resnet50', H, W, max_batch=1, device='cuda:0', side_stream=False))
        mt = MetaTrainer(engines[0], meta_batch_size=tpr, extra_engines=engines[1:], comm=c)
        mt.load_state(sd, lrs)
        tasks = []
        for t in range(tpr):
            x, y = synthetic.synthetic_frames(1, H, W, seed=1000 + t)
            x, y = x.cuda(), y.cuda()
            tasks.append((x, y, torch.flip(x, dims=[3]).contiguous(), torch.flip(y, dims=[3]).contiguous()))
        s0 = mt.state.clone()
        losses = [mt.meta_iteration(tasks, inner_steps=2) for _ in range(2)]
        torch.cuda.synchronize()
        states.append((mt.state.clone().cpu(), losses, float((mt.state - s0).abs().max())))
        # close the SOURCE of the aliased state first on purpose: the library un-aliases the others (they keep a valid copy)
        engines[0].close()
        for e in engines[1:]:
            with torch.cuda.stream(e.stream):
                e.reset()
                p = e.get_params()
                e.synchronize()
                assert bool(torch.equal(p, mt.state[mt.n_lr:]))        # its own copy of the learned init the source held
            e.close()
    out[tag] = {'equal': bool(torch.equal(states[0][0], states[1][0])), 'losses': states[0][1], 'losses_ref': states[1][1],
                'finite': bool(torch.isfinite(states[0][0]).all()), 'moved': states[0][2]}
comm.close()
torch.save(out, sys.argv[1])
