"""TEST INFRASTRUCTURE: a CPU stand-in for `eosvos_amd.engine.Engine` with the same method surface and a
2-parameter "network", so the HOST logic above the C-ABI (evaluation / online-adaptation loops, meta-train task
loop, outer step, sharding, checkpoints, entry points) can be exercised without a GPU and in multi-process gloo
tests.  It is never importable from the product (`tests/` only) and computes nothing the product ships:
    logits = theta[0] * mean_c(image) + theta[-1]           (first weight and the last bias of the parameter vector)
Loss BCE-with-logits (mean), SGD with the effective per-neuron lr, closed-form first-order meta-gradient like the
engine (`csrc/engine.cpp eosvos_meta_grad_ex`), RAdam / clamp through `oracle.meta`.
"""
import math

import torch
import torch.nn.functional as F

from eosvos_amd import networks
from eosvos_amd.topology import trainable

from oracle import meta as oracle_meta


class FakeEngine:
    instances = []

    def __init__(self, encoder='resnet50', height=480, width=854, max_batch=3, device='cpu', norm='bn'):
        self.encoder, self.norm = encoder, norm
        self.device = torch.device('cpu')
        self.height, self.width, self.max_batch = height, width, max_batch
        tr = trainable(encoder)
        self.shapes = [tuple(s) for _, s in tr]
        self.n_param = sum(math.prod(s) for s in self.shapes)
        self.n_lr = sum(s[0] for s in self.shapes)
        self.n_norm = 0
        self.lr_level, self.lr_log, self.n_lr_store = 'NEURON', False, self.n_lr
        self.init = torch.zeros(self.n_param)
        self.theta = torch.zeros(self.n_param)
        self.snap = torch.zeros(self.n_param)
        self.lr0, self.lrb = 0.0, 0.0                  # effective lr of theta[0] / theta[-1]
        self.gsum = torch.zeros(2)
        self.loss_name = 'cross_entropy'
        self.log = []
        self.infer_fn = None                            # optional (engine, images) -> probs hook of a test
        self._logits = self._x = self._dl = None
        self.last_masks = None
        self.norm_args = None
        FakeEngine.instances.append(self)

    # ---- state ----
    def close(self):
        pass

    def synchronize(self):
        pass

    def set_init(self, flat):
        self.init = flat.detach().float().cpu().clone()
        self.theta = self.init.clone()

    def set_norm(self, gamma, beta, mean, var, eps=1e-5):
        self.norm_args = tuple(t.detach().float().cpu().clone() for t in (gamma, beta, mean, var))

    def lr_store_count(self, level):
        return {'NEURON': self.n_lr, 'TENSOR': len(self.shapes), 'SINGLE': 1, 'PARAM': self.n_param}[level]

    def set_lr_state(self, level, use_log, flat):
        flat = flat.detach().float().cpu()
        self.lr_level, self.lr_log, self.n_lr_store = level, bool(use_log), flat.numel()
        f = (lambda v: math.exp(v)) if use_log else (lambda v: v)
        self.lr0, self.lrb = f(float(flat[0])), f(float(flat[-1]))

    def set_lr(self, flat):
        self.set_lr_state('NEURON', False, flat)

    def set_loss(self, name):
        if name not in ('cross_entropy', 'dice', 'cross_entropy_and_dice', 'class_balanced_cross_entropy'):
            raise NotImplementedError(name)
        self.loss_name = name

    def load_model_state(self, state_dict, lrs=None):
        names = [n for n, _ in trainable(self.encoder)]
        self.set_init(torch.cat([state_dict[n].reshape(-1).float() for n in names]))
        if lrs is not None:
            self.set_lr(torch.cat([l.reshape(-1).float() for l in lrs]))

    def reset(self):
        self.log.append('reset')
        self.theta = self.init.clone()

    def get_params(self):
        return self.theta.clone()

    def set_params(self, flat):
        self.theta = flat.detach().float().cpu().clone()

    def snapshot(self):
        self.log.append('snapshot')
        self.snap = self.theta.clone()

    def restore(self):
        self.log.append('restore')
        self.theta = self.snap.clone()

    # ---- hot loop ----
    def _net(self, images):
        return images.float().mean(dim=1, keepdim=True) * self.theta[0] + self.theta[-1]

    def forward(self, images, want_logits=True):
        self._x = images.float()
        self._logits = self._net(images)
        return self._logits.clone() if want_logits else None

    def loss(self, kind, masks):
        self.last_masks = masks.clone()
        lg = self._logits.clone().requires_grad_(True)
        l = F.binary_cross_entropy_with_logits(lg, masks.float())
        (self._dl,) = torch.autograd.grad(l, lg)
        return l.detach().view(1)

    def loss_bce(self, masks):
        return self.loss('cross_entropy', masks)

    def loss_of(self, kind, logits, masks):
        return F.binary_cross_entropy_with_logits(logits.float(), masks.float()).view(1)

    def _grads(self):
        g0 = float((self._dl * self._x.mean(dim=1, keepdim=True)).sum())
        gb = float(self._dl.sum())
        return torch.tensor([g0, gb])

    def backward_step(self, accumulate=False):
        g = self._grads()
        if accumulate:
            self.gsum += g
        self.theta[0] -= self.lr0 * g[0]
        self.theta[-1] -= self.lrb * g[1]
        self.log.append('step')

    def finetune_step(self, images, masks, accumulate=False, sync_loss=True):
        self.forward(images, want_logits=False)
        l = self.loss(self.loss_name, masks)
        self.backward_step(accumulate)
        return float(l) if sync_loss else None

    def infer(self, images):
        if self.infer_fn is not None:               # prescribed per-frame probabilities (the replay test)
            return torch.cat([self.infer_fn(self, images[b:b + 1]) for b in range(images.shape[0])])
        return torch.sigmoid(self._net(images))

    def merge_labels(self, probs):
        return oracle_meta.merge_labels(probs.float()).to(torch.uint8)

    # ---- meta-training ----
    def meta_task_begin(self):
        self.log.append('task_begin')
        self.gsum = torch.zeros(2)
        self.theta = self.init.clone()

    def meta_grad(self, images, masks, flat_meta_grad, weight=1.0, init_grad=True, new_segment=False, sync=True):
        self.forward(images, want_logits=False)
        l = self.loss(self.loss_name, masks)
        G = self._grads()
        n = self.n_lr_store
        # d/d lr = -sum_k g_k * G per neuron (first and last neuron only in this 2-parameter stand-in)
        dl = -self.gsum * G
        if self.lr_log:
            dl = dl * torch.tensor([self.lr0, self.lrb])
        flat_meta_grad[0] += weight * dl[0]
        flat_meta_grad[n - 1] += weight * dl[1]
        if init_grad:
            flat_meta_grad[n] += weight * G[0]
            flat_meta_grad[-1] += weight * G[1]
        if new_segment:
            self.gsum = torch.zeros(2)
        return float(l) if sync else l.detach().view(1)

    def radam_step(self, param, grad, exp_avg, exp_avg_sq, lr, weight_decay, step, grad_scale=1.0, grad_clip=0.0,
                   betas=(0.9, 0.999), eps=1e-8):
        g = grad * grad_scale
        if grad_clip and grad_clip > 0:
            g = g.clamp(-grad_clip, grad_clip)
        state = {'step': step - 1, 'exp_avg': exp_avg, 'exp_avg_sq': exp_avg_sq}
        oracle_meta.radam_step(param, g, state, lr, weight_decay, betas, eps)

    def clamp(self, param, lo, hi):
        param.clamp_(lo, hi)

    # ---- measurement hooks of bench.py ----
    def profile_launches(self, on=True):
        self._prof = {'stand_in_kernel': [0, 0.0, 0.0]} if on else None

    def profile_read(self):
        return {'stand_in_kernel': (6, 1.0, 6e9)}

    def mfma_probe(self, iters=0):
        return 0.0


class FakeDeepLab(networks.DeepLabV3Plus):
    """The product's DeepLabV3Plus drop-in class with its engine replaced by the stand-in (CPU tensors accepted)."""

    def __init__(self, *a, **k):
        k.setdefault('device', 'cpu')
        super().__init__(*a, **k)
        self.call_log = []

    def _ensure_engine(self, height, width, batch):
        e = self.engine
        if e is None or e.height != height or e.width != width or batch > e.max_batch:
            self.engine = FakeEngine(self.encoder, height, width, max(batch, self.max_batch))
            self._dirty = True
        if self._dirty:
            self.push_state()
        if getattr(self, '_pending_task_begin', False):
            self.engine.meta_task_begin()
            self._pending_task_begin = False
        return self.engine

    def __call__(self, inputs):
        b, _, h, w = inputs.shape
        e = self._ensure_engine(h, w, b)
        self.call_log.append([round(float(v) * 100) for v in inputs[:, 0, 0, 0]])
        logits = e.forward(inputs.contiguous().float())
        logits._eosvos_engine = e
        return [logits]
