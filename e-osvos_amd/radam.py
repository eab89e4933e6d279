"""`RAdam` with the reference's call surface (`src/util/radam.py:5-94`) on the fused HIP kernel.

The reference builds one param group per tensor (`src/train_meta.py:110-127`) and calls
`step()` / `zero_grad()` (`:368-369`).  Here every tensor's moments live on the GPU and the
update is `eosvos_radam_step` (N_sma / step_size computed on the host exactly as
`radam.py:62-79`).  `MetaTrainer` (meta_run.py) is the fast path -- it applies the same kernel to
the whole flat state in two launches; this class exists so loops written against the reference's
optimizer API keep working.
"""
import torch


class RAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, degenerated_to_sgd=True,
                 engine=None):
        if engine is None:
            raise RuntimeError('RAdam needs engine= (eosvos_amd.engine.Engine): the step runs on the GPU')
        if not degenerated_to_sgd:
            raise NotImplementedError('degenerated_to_sgd=False')
        if isinstance(params, (list, tuple)) and params and isinstance(params[0], dict):
            groups = [dict(g) for g in params]
        else:
            groups = [{'params': list(params)}]
        for g in groups:
            g.setdefault('lr', lr)
            g.setdefault('betas', betas)
            g.setdefault('eps', eps)
            g.setdefault('weight_decay', weight_decay)
        self.param_groups = groups
        self.engine = engine
        self.state = {}

    def zero_grad(self):
        for g in self.param_groups:
            for p in g['params']:
                if getattr(p, 'grad', None) is not None:
                    p.grad.zero_()

    def step(self, grad_scale=1.0, grad_clip=0.0):
        """All tensors' kernels are enqueued on the engine's stream first, then ONE synchronisation, then the
        write-back (not one stream sync per tensor: the reference builds 128 single-tensor groups)."""
        dev = self.engine.device
        done = []
        with torch.cuda.stream(self.engine.stream):
            for g in self.param_groups:
                for p in g['params']:
                    if getattr(p, 'grad', None) is None:
                        continue
                    st = self.state.setdefault(id(p), {})
                    if not st:
                        st['step'] = 0
                        st['param'] = p.data.detach().to(dev, torch.float32).contiguous().clone()
                        st['exp_avg'] = torch.zeros_like(st['param'])
                        st['exp_avg_sq'] = torch.zeros_like(st['param'])
                    else:
                        st['param'].copy_(p.data)
                    st['step'] += 1
                    st['grad'] = p.grad.detach().to(dev, torch.float32).contiguous()     # kept alive until the sync
                    self.engine.radam_step(st['param'], st['grad'], st['exp_avg'], st['exp_avg_sq'], g['lr'],
                                           g['weight_decay'], st['step'], grad_scale=grad_scale, grad_clip=grad_clip,
                                           betas=g['betas'], eps=g['eps'])
                    done.append((p, st))
            self.engine.synchronize()
            for p, st in done:
                p.data.copy_(st['param'])
                st.pop('grad', None)
