"""torch-CPU restatement of the meta-optimizer side of the hot path.

  * `finetune_step`  : `MetaOptimizer.step` (`src/meta_optim/meta_optim.py:177-214`)
                       + `MetaModel.apply_param_groups_step` (`meta_model.py:78-80`)
                       + `detach_param_groups` (`meta_model.py:62-65`), NEURON lrs
                       (`meta_optim.py:46-67`): theta <- theta - lr[cout] * grad.
  * `finetune`       : the evaluation fine-tune loop, `src/util/evaluate.py:220-274`
                       (early stopping off, `cfgs/meta.yaml:97-99`).
  * `meta_task`      : one task of `src/util/meta_run.py:109-238` with first-order
                       gradients (`second_order_gradients: False`, `cfgs/meta.yaml:40`)
                       and `bptt_epochs == num_epochs.train`; uses the closed form of
                       SURVEY.md section 3.3 (theta_K = theta_0 - lr * sum_k g_k):
                         d/d init = G,  d/d lr[c] = -sum_{cin,kh,kw}(sum_k g_k * G).
  * `meta_task_hier` : the same for `lr_hierarchy_level` SINGLE / TENSOR / NEURON / PARAM and
                       `use_log_init_lr` (`meta_optim.py:27-67,157-163,180-185`).
  * `meta_task_bptt` : truncated / multi-step BPTT schedules (`meta_run.py:154-221`).
  * `radam_step`     : `RAdam.step`, `src/util/radam.py:28-94`, per-tensor groups as
                       built at `src/train_meta.py:110-127`.
  * `outer_step`     : average / clip / RAdam / clamp, `src/train_meta.py:361-373` and
                       `MetaOptimizer.clamp_init_lr` (`meta_optim.py:116-133`).
  * `merge_labels`   : multi-object merge, `src/util/evaluate.py:322-326`.
  * `online_adapt_schedule` : frame-range / propagated-frame index logic of
                       `src/util/evaluate.py:140-193,227-253`.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
"""
import math

import torch

from . import deeplab
from .topology import trainable_names


def loss_and_grads(P, x, y, encoder='resnet50', norm='bn', loss_name='cross_entropy'):
    names = trainable_names(encoder)
    Q = dict(P)
    leaves = []
    for n in names:
        t = P[n].detach().clone().requires_grad_(True)
        Q[n] = t
        leaves.append(t)
    logits = deeplab.forward(Q, x, encoder, norm)
    loss = deeplab.loss_fn(loss_name, logits, y)
    grads = torch.autograd.grad(loss, leaves)
    return loss.detach(), list(grads), logits.detach()


def finetune_step(P, lrs, x, y, encoder='resnet50', norm='bn', loss_name='cross_entropy'):
    """One inner step.  Returns (loss, grads, new P).  `lrs`: list aligned with
    trainable_names(), each broadcastable against its tensor ((Cout,1,1,1) or (1,))."""
    names = trainable_names(encoder)
    loss, grads, _ = loss_and_grads(P, x, y, encoder, norm, loss_name)
    Pn = dict(P)
    for n, lr, g in zip(names, lrs, grads):
        Pn[n] = P[n] - g * lr
    return loss, grads, Pn


def finetune(P, lrs, batches, encoder='resnet50', norm='bn'):
    """`batches`: iterable of (x, y).  Returns (loss list, final P)."""
    losses = []
    for x, y in batches:
        loss, _, P = finetune_step(P, lrs, x, y, encoder, norm)
        losses.append(float(loss))
    return losses, P


def effective_lrs(store, level, use_log, n_tensors):
    """The lr each tensor's step multiplies with: `state['log_lr']` of `_init_state`
    (`meta_optim.py:157-163`: SINGLE repeats the one value per param group) after the
    optional exp() of `step` (`:180-185`).  Iterating a (G,1) tensor yields (1,) rows."""
    if level == 'SINGLE':
        rows = list(store.repeat(n_tensors, 1))
    elif level == 'TENSOR':
        rows = list(store)
    elif level in ('NEURON', 'PARAM'):
        rows = list(store)
    else:
        raise NotImplementedError(level)
    return [r.exp() if use_log else r for r in rows]


def meta_task_hier(P0, store, level, use_log, train_batches, meta_batch, encoder='resnet50', norm='bn',
                   loss_name='cross_entropy'):
    """`meta_task` for any `lr_hierarchy_level` / `use_log_init_lr`: returns g_lr in the
    layout of the stored state.  First-order closed form: with lr_eff = f(store),
    d/d lr_eff = -(sum_k g_k) * G elementwise, summed over the elements that share one
    stored value (autograd of broadcasting / `repeat`), times lr_eff for log storage."""
    names = trainable_names(encoder)
    lrs = effective_lrs(store, level, use_log, len(names))
    res = meta_task(P0, lrs, train_batches, meta_batch, encoder, norm, _keep_elem=True, loss_name=loss_name)
    elem = res.pop('g_lr_elem')
    per_tensor = []
    for e, lr in zip(elem, lrs):
        if use_log:
            e = e * lr
        if level == 'PARAM':
            per_tensor.append(e)
        elif level == 'NEURON':
            per_tensor.append(e.sum(dim=tuple(range(1, e.dim())), keepdim=True) if e.dim() > 1 else e)
        else:
            per_tensor.append(e.sum().reshape(1))
    if level == 'TENSOR':
        res['g_lr'] = torch.stack(per_tensor)                   # (G,1)
    elif level == 'SINGLE':
        res['g_lr'] = torch.stack(per_tensor).sum().reshape(1, 1)
    else:
        res['g_lr'] = per_tensor
    return res


def meta_task(P0, lrs, train_batches, meta_batch, encoder='resnet50', norm='bn', _keep_elem=False,
              loss_name='cross_entropy'):
    """K inner steps + one meta frame.  Returns dict(meta_loss, train_losses,
    g_init (list), g_lr (list shaped like lrs))."""
    names = trainable_names(encoder)
    P = P0
    gsum = None
    train_losses = []
    for x, y in train_batches:
        loss, grads, P = finetune_step(P, lrs, x, y, encoder, norm, loss_name)
        train_losses.append(float(loss))
        gsum = [g.clone() for g in grads] if gsum is None else [a + g for a, g in zip(gsum, grads)]
    xm, ym = meta_batch
    meta_loss, G, _ = loss_and_grads(P, xm, ym, encoder, norm, loss_name)
    if _keep_elem:
        return dict(meta_loss=float(meta_loss), train_losses=train_losses, g_init=G,
                    g_lr_elem=[-(s * g) for s, g in zip(gsum, G)])
    g_lr = []
    for n, s, g, lr in zip(names, gsum, G, lrs):
        prod = -(s * g)
        if prod.dim() > 1:
            prod = prod.sum(dim=tuple(range(1, prod.dim())), keepdim=True)
        g_lr.append(prod.reshape(lr.shape))
    return dict(meta_loss=float(meta_loss), train_losses=train_losses, g_init=G, g_lr=g_lr)


def meta_task_bptt(P0, lrs, train_batches, meta_batch, bptt_epochs=None, multi_step_bptt_loss=None,
                   encoder='resnet50', norm='bn', loss_name='cross_entropy'):
    """The BPTT schedules of `src/util/meta_run.py:154-221` with first-order gradients: the meta loss is
    back-propagated every `bptt_epochs` steps (and at the last step) and the state is then detached
    (`meta_optim.reset(keep_state=True)`, `meta_optim.py:145-151`): that detaches the parameters
    (`meta_model.py:62-65`) AND the state lr (`state['log_lr'] = [l.detach() ...]`), so only the FIRST segment,
    theta_e = init - lr * sum_{k<=e} g_k, reaches the learned init and the learned lr; the later segments'
    backward passes leave no gradient on the meta-optimizer (their losses are still evaluated, NaN-checked and
    reported).  `multi_step_bptt_loss` weights a meta loss evaluated after every step (`:154-177`).
    Returns dict(meta_losses, g_init, g_lr)."""
    names = trainable_names(encoder)
    K = len(train_batches)
    bptt = bptt_epochs or K
    xm, ym = meta_batch
    P = P0
    gsum = None
    g_init = [torch.zeros_like(P0[n]) for n in names]
    g_lr = [torch.zeros_like(lr) for lr in lrs]
    meta_losses = []
    first_segment = True

    def add(weight):
        ml, G, _ = loss_and_grads(P, xm, ym, encoder, norm, loss_name)
        meta_losses.append(float(ml))
        if not first_segment:
            return
        for i, (s_, g, lr) in enumerate(zip(gsum, G, lrs)):
            prod = -(s_ * g) * weight
            if prod.dim() > 1:
                prod = prod.sum(dim=tuple(range(1, prod.dim())), keepdim=True)
            g_lr[i] += prod.reshape(lr.shape)
            g_init[i] += weight * g

    for epoch, (x, y) in enumerate(train_batches, start=1):
        _, grads, P = finetune_step(P, lrs, x, y, encoder, norm, loss_name)
        gsum = [g.clone() for g in grads] if gsum is None else [a + g for a, g in zip(gsum, grads)]
        boundary = epoch % bptt == 0 or epoch == K
        if multi_step_bptt_loss:
            add(multi_step_bptt_loss[epoch - 1])
        elif boundary:
            add(1.0)
        if boundary:
            first_segment = False
            gsum = None
    return dict(meta_losses=meta_losses, g_init=g_init, g_lr=g_lr)


def radam_scalars(step, beta1=0.9, beta2=0.999):
    """(N_sma, step_size) of `radam.py:62-79` (degenerated_to_sgd=True)."""
    beta2_t = beta2 ** step
    n_sma_max = 2 / (1 - beta2) - 1
    n_sma = n_sma_max - 2 * step * beta2_t / (1 - beta2_t)
    if n_sma >= 5:
        step_size = math.sqrt((1 - beta2_t) * (n_sma - 4) / (n_sma_max - 4) * (n_sma - 2) / n_sma
                              * n_sma_max / (n_sma_max - 2)) / (1 - beta1 ** step)
    else:
        step_size = 1.0 / (1 - beta1 ** step)
    return n_sma, step_size


def radam_step(p, grad, state, lr, weight_decay, betas=(0.9, 0.999), eps=1e-8):
    """In-place on `p`; `state` = dict(step, exp_avg, exp_avg_sq) (created on first use)."""
    if not state:
        state['step'] = 0
        state['exp_avg'] = torch.zeros_like(p)
        state['exp_avg_sq'] = torch.zeros_like(p)
    beta1, beta2 = betas
    state['exp_avg_sq'].mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    state['exp_avg'].mul_(beta1).add_(grad, alpha=1 - beta1)
    state['step'] += 1
    n_sma, step_size = radam_scalars(state['step'], beta1, beta2)
    if weight_decay != 0:
        p.add_(p, alpha=-weight_decay * lr)
    if n_sma >= 5:
        denom = state['exp_avg_sq'].sqrt().add_(eps)
        p.addcdiv_(state['exp_avg'], denom, value=-step_size * lr)
    else:
        p.add_(state['exp_avg'], alpha=-step_size * lr)
    return p


def outer_step(lr_params, init_params, lr_grads, init_grads, states, meta_batch_size,
               model_init_lr=1e-5, log_init_lr_lr=1e-5, model_init_weight_decay=1e-3,
               grad_clip=None, max_lr=None):
    """`states`: list of per-tensor dicts, lr tensors first then init tensors (the
    `named_parameters()` order of MetaOptimizer, `meta_optim.py:65-66,78`)."""
    k = 0
    for p, g in zip(lr_params, lr_grads):
        g = g / meta_batch_size
        if grad_clip is not None:
            g = g.clamp(-grad_clip, grad_clip)
        radam_step(p, g, states[k], log_init_lr_lr, 0.0)
        k += 1
    for p, g in zip(init_params, init_grads):
        g = g / meta_batch_size
        if grad_clip is not None:
            g = g.clamp(-grad_clip, grad_clip)
        radam_step(p, g, states[k], model_init_lr, model_init_weight_decay)
        k += 1
    for p in lr_params:  # clamp_init_lr, use_log_init_lr False: [0, max_lr]
        p.clamp_(0, max_lr)


def merge_labels(probs):
    """probs: (n_obj, H, W) -> (H, W) uint8 labels.  `evaluate.py:322-326`."""
    bg = probs.max(dim=0)[0].lt(0.5)
    lab = probs.argmax(dim=0) + 1
    lab[bg] = 0
    return lab.to(torch.uint8)


def online_adapt_schedule(num_frames, train_frame_id, step, train_batch_size):
    """Rounds of the evaluation loop.  Each round: dict(eval_min, eval_max,
    propagate_frames) -- `propagate_frames` are the earlier frames whose predicted
    masks (if non-empty) join the first frame in the adaptation batch."""
    rounds = []
    if step:
        meta_iter = range(train_frame_id + 1, num_frames, step)
        s = step
    else:
        meta_iter = [None]
        s = num_frames
    eval_max = None
    for r, _ in enumerate(meta_iter):
        if r == 0:
            eval_min = train_frame_id + 1
            eval_max = eval_min
            prop = []
        else:
            eval_min = eval_max
            n_prop = min(step, train_batch_size)
            start = step - n_prop + 1
            prop = [eval_min - j for j in range(start, step)]
        eval_max = min(eval_max + s, num_frames)
        rounds.append(dict(eval_min=eval_min, eval_max=eval_max, propagate_frames=prop))
        if eval_max == num_frames:
            break
    return rounds
