"""Drop-in for `src/meta_optim/meta_optim.py::MetaOptimizer` (+ the `MetaModel` helper of
`src/meta_optim/meta_model.py`) on the MI355X engine.

Reference semantics kept:
  * learned per-neuron lr tensors `log_init_lr_*` of shape (Cout,1,1,1) / (1,) initialised to
    init_lr*(1+U(-.5,.5)) (`meta_optim.py:46-67`), learned init `model_init_*` aliasing the
    model's parameters (`:71-78`); `state_dict()` = 128 tensors for ResNet-50 in that order;
  * `reset()` re-points the model at the learned init and the state lr at the learned lr
    (`:144-163`) -> `eosvos_reset`; `reset(keep_state=True)` detaches (`:145-151`) -> no-op for
    first-order gradients;
  * `set_train_loss` / `step(loss)`: autograd.grad + theta <- theta - lr (.) grad
    (`:165-214`, `meta_model.py:78-80`) -> `eosvos_backward_step` on the gradient the fused
    BCE kernel left behind; in `.train()` mode the step also accumulates sum_k g_k for the
    closed-form meta-gradient;
  * `meta_backward(meta_loss)` stands in for `bptt_loss.backward()` (`meta_run.py:214`): fills
    `.grad` of `named_parameters()` (log_init_lr_* then model_init_*).
  * `lr_hierarchy_level` SINGLE / TENSOR / NEURON / PARAM and `use_log_init_lr`
    (`meta_optim.py:27-67,157-163,180-185`): the learned state keeps the reference's tensors
    (`log_init_lr` (1,1) / (G,1), or one `log_init_lr_<name>` per tensor); the engine expands it to the
    effective lr of the step and returns d/d(state) in the same layout (`eosvos_set_lr_state`).
Unsupported reference options raise NotImplementedError exactly like the reference does for
unknown hierarchy levels (`meta_optim.py:68-69`).
"""
import math
from collections import OrderedDict

import torch

from .networks import _Param
from .topology import neuron_lr_shape


class _MetaModelShim:
    """`meta_optim.meta_model` of the reference; only the members the loops touch."""

    def __init__(self, model):
        self.model = model

    def detach_param_groups(self):      # meta_model.py:62-65 -- the engine keeps no autograd graph
        pass

    @property
    def num_param_groups(self):
        return len(self.model._names)


class MetaOptimizer:
    def __init__(self, model, init_lr, learn_model_init, second_order_gradients, lr_hierarchy_level,
                 use_log_init_lr, max_lr):
        if lr_hierarchy_level not in ('SINGLE', 'TENSOR', 'NEURON', 'PARAM'):
            raise NotImplementedError                   # meta_optim.py:68-69
        if second_order_gradients:
            # the reference cannot run this on the DeepLab path either: step() calls
            # model.named_parameters_without_second_order_derivate() (meta_optim.py:195), which only the Mask R-CNN
            # model defines (mask_rcnn.py:536-543) -- DeepLabV3Plus raises AttributeError there
            raise NotImplementedError('second_order_gradients=True needs a double backward through every kernel (and is '
                                      'not reachable for DeepLabV3Plus in the reference: meta_optim.py:195)')
        self._max_lr = max_lr
        self._learn_model_init = bool(learn_model_init)
        self._lr_hierarchy_level = lr_hierarchy_level
        self._use_log_init_lr = bool(use_log_init_lr)
        self.training = True
        self.only_box_head = False          # evaluate.py:270 sets it; no-op for DeepLab (meta_model.py:73-76)
        self.model = model
        self.meta_model = _MetaModelShim(model)
        self.state = {'num_steps': 0}
        self._train_loss = None
        names, shapes = model._names, model._shapes
        self._lr_views = OrderedDict()
        lvl = lr_hierarchy_level
        if lvl == 'SINGLE':                               # meta_optim.py:27-31
            self._lr_flat = torch.ones(1).mul(init_lr)
            self._lr_views['log_init_lr'] = self._lr_flat.view(1, 1)
        elif lvl == 'TENSOR':                             # meta_optim.py:33-42
            self._lr_flat = torch.ones(len(names)).mul(init_lr)
            self._lr_flat += torch.rand_like(self._lr_flat).sub(0.5) * init_lr
            self._lr_views['log_init_lr'] = self._lr_flat.view(len(names), 1)
        else:                                             # meta_optim.py:44-66
            lshapes = [tuple(s) if lvl == 'PARAM' else neuron_lr_shape(s) for s in shapes]
            self._lr_flat = torch.zeros(sum(math.prod(ls) for ls in lshapes))
            off = 0
            for n, ls in zip(names, lshapes):
                k = math.prod(ls)
                v = self._lr_flat[off:off + k].view(ls)
                v.copy_(init_lr * (1.0 + (torch.rand(ls) - 0.5)))          # meta_optim.py:57-58
                self._lr_views['log_init_lr_' + n.replace('.', '-')] = v
                off += k
        if self._use_log_init_lr:
            self._lr_flat.log_()
        self._init_views = OrderedDict(('model_init_' + n.replace('.', '-'), model._views[n]) for n in names)
        self._params = OrderedDict()
        # learn_model_init False: the init tensors are not registered parameters (meta_optim.py:76-78): they keep
        # the model's weights, receive no meta-gradient and do not appear in state_dict()
        learned = list(self._lr_views.items()) + (list(self._init_views.items()) if self._learn_model_init else [])
        for k, v in learned:
            self._params[k] = _Param(k, v, True)
        self._grad_flat = None
        model._lr_flat = self._lr_flat
        model._lr_mode = (lvl, self._use_log_init_lr)
        model._dirty = True

    # ---- nn.Module-like surface ---------------------------------------------------------------
    def train(self, mode=True):
        self.training = mode
        eng = getattr(self.model, 'engine', None)
        if not mode and eng is not None and getattr(eng, 'in_meta_task', False) and getattr(eng, 'steps_since_reset', 1) == 0:
            # `reset()` then `eval()` (evaluate.py:196-198): the task that reset() opened in training mode will never
            # accumulate -- it is an evaluation fine-tune, and the engine may be rebuilt for another frame size during it
            eng.in_meta_task = False
        if not mode and getattr(self.model, '_pending_task_begin', False):
            self.model._pending_task_eval = True         # same, the engine not built yet (networks._ensure_engine)
        return self

    def eval(self):
        return self.train(False)

    def to(self, device):
        self.model.to(device)
        return self

    def share_memory(self):
        return self

    def named_parameters(self):
        return iter(self._params.items())

    def parameters(self):
        return iter(self._params.values())

    def state_dict(self):
        out = OrderedDict()
        for k, p in self._params.items():
            out[k] = p.data.clone()
        return out

    def load_state_dict(self, sd):
        missing = [k for k in self._params if k not in sd]
        if missing:
            raise KeyError(f'missing keys: {missing[:4]}...')
        # The evaluation loop reloads the SAME learned state for every object and every adaptation round
        # (evaluate.py:196,200): when neither the source tensors nor this optimizer's own storage changed since the last
        # load (same storages, same in-place version counters) the 161 MB copy and the engine upload it triggers are skipped.
        stamp = [(sd[k].data_ptr(), sd[k]._version, tuple(sd[k].shape)) for k in self._params]
        own = (self._lr_flat._version, self.model._flat._version, self.model._lr_flat is self._lr_flat)
        if getattr(self, '_loaded_stamp', None) == (stamp, own) and not self.model._dirty:
            return
        for k, p in self._params.items():
            p.data.copy_(sd[k].reshape(p.data.shape))
        self.model._lr_flat = self._lr_flat
        self.model._dirty = True
        self._loaded_stamp = (stamp, (self._lr_flat._version, self.model._flat._version, True))
        self._loaded_refs = [sd[k] for k in self._params]       # keeps the storages alive: an address cannot be reused by another tensor

    def zero_grad(self):
        if self._grad_flat is not None:
            self._grad_flat.zero_()

    def init_zero_grad(self):
        eng = self.model.engine
        dev = eng.device if eng is not None else self.model.device
        n = self._lr_flat.numel() + (self.model._flat.numel() if self._learn_model_init else 0)
        self._grad_flat = torch.zeros(n, device=dev)
        off = 0
        for p in self._params.values():
            k = p.data.numel()
            p.grad = self._grad_flat[off:off + k].view(p.data.shape)
            off += k

    @property
    def init_lr(self):                                    # meta_optim.py:84-96
        f = (lambda t: t.exp()) if self._use_log_init_lr else (lambda t: t)
        if self._lr_hierarchy_level in ('SINGLE', 'TENSOR'):
            return f(self._lr_views['log_init_lr']).clone()
        return torch.tensor([float(f(v).mean()) for v in self._lr_views.values()])

    @property
    def state_lr(self):                                   # meta_optim.py:98-110 (state = learned lr here)
        if self._lr_hierarchy_level == 'SINGLE':
            return self.init_lr.repeat(self.meta_model.num_param_groups, 1)     # `_init_state`, :158-160
        return self.init_lr

    def clamp_init_lr(self):                              # meta_optim.py:116-133
        if self._use_log_init_lr:
            self._lr_flat.clamp_(-33, None if self._max_lr is None else math.log(self._max_lr))
        else:
            self._lr_flat.clamp_(0, self._max_lr)
        self.model._lr_flat = self._lr_flat
        self.model._dirty = True

    # ---- inner loop ------------------------------------------------------------------------------
    def reset(self, keep_state=False):
        if keep_state:
            self._first_segment = False      # detached: nothing after this reaches log_init_lr_* / model_init_*
            return
        self._first_segment = True
        m = self.model
        if m.engine is not None:
            if m._dirty:
                m.push_state()
            if self.training:
                m.engine.meta_task_begin()          # theta <- init, sum_k g_k <- 0
            else:
                m.engine.reset()
        else:
            m._pending_task_begin = self.training   # engine is created at the first forward
            m._pending_task_eval = False
        self.state['num_steps'] = 0

    def set_train_loss(self, train_loss):
        self._train_loss = train_loss

    def step(self, train_loss):
        eng = getattr(train_loss, '_eosvos_engine', None)
        if eng is None:
            raise RuntimeError('step() needs the loss returned by eosvos_amd.helper_func.compute_loss')
        eng.backward_step(accumulate=self.training and getattr(self, '_first_segment', True))
        self.state['num_steps'] += 1

    def meta_backward(self, meta_inputs, meta_gts, loss_func='cross_entropy', weight=1.0):
        """`(weight * meta_loss).backward()` for one meta frame batch (`meta_run.py:155-214`): returns the meta loss
        (float) and ADDS `weight` x the task's meta-gradient into `.grad` of named_parameters().  `weight` is the
        per-step factor of `multi_step_bptt_loss`.  After `reset(keep_state=True)` the graph to the learned tensors is
        cut, so only the loss value is computed."""
        if self._grad_flat is None:
            self.init_zero_grad()
        eng = self.model._ensure_engine(meta_inputs.shape[2], meta_inputs.shape[3], meta_inputs.shape[0])
        eng.set_loss(loss_func)
        if not getattr(self, '_first_segment', True):
            eng.forward(meta_inputs.contiguous(), want_logits=False)
            return float(eng.loss(loss_func, meta_gts.contiguous().float()))
        task = torch.zeros(self._lr_flat.numel() + self.model._flat.numel(), device=self._grad_flat.device)
        loss = eng.meta_grad(meta_inputs.contiguous(), meta_gts.contiguous(), task, weight=weight,
                             init_grad=self._learn_model_init)
        if not math.isnan(loss):
            n = self._grad_flat.numel()             # learn_model_init False: only the lr slice is a Parameter
            self._grad_flat.add_(task[:n])
        return loss
