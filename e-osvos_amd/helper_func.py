"""Mirror of the hot-path helpers of `src/util/helper_func.py` on the MI355X engine.

  compute_loss        `helper_func.py:28-56`   (cross_entropy = fused BCE-with-logits kernel)
  init_parent_model   `helper_func.py:339-385` (DeepLabV3Plus only)
  run_frames          `helper_func.py:67-159`  (`run_loader` for the DeepLab branch `:131-142`:
                                                sigmoid, per-sample BCE, >=0.5 accuracy)
  early_stopping      `helper_func.py:388-397`
  EpochSampler        `helper_func.py:521-545`
  set_random_seeds    `helper_func.py:515-518`
`dice` (`networks/loss_dice.py:4-40`), `cross_entropy_and_dice` (`:45-54`) and
`class_balanced_cross_entropy` (`networks/loss_ce.py:15-60`) run as fused HIP kernels too; unknown
names raise NotImplementedError as in the reference (`:55-56`).
"""
import random

import numpy as np
import torch

from .networks import DeepLabV3, DeepLabV3Plus


def compute_loss(loss_func, outputs, gts, loss_kwargs=None):
    """`compute_loss(loss_func, outputs, gts, loss_kwargs=None)`, helper_func.py:28-56.  The returned
    0-dim loss carries the engine handle so `MetaOptimizer.step(loss)` can run the backward."""
    loss_kwargs = loss_kwargs or {}
    if loss_func not in ('cross_entropy', 'dice', 'cross_entropy_and_dice', 'class_balanced_cross_entropy'):
        raise NotImplementedError(f"loss_func='{loss_func}'")
    if loss_func == 'class_balanced_cross_entropy' and not loss_kwargs.get('size_average', True):
        raise NotImplementedError('class_balanced_cross_entropy with size_average=False')
    eng = getattr(outputs, '_eosvos_engine', None)
    if eng is None:
        raise RuntimeError('compute_loss needs logits produced by eosvos_amd.networks.DeepLabV3Plus '
                           '(there is no CPU/eager path)')
    gts = gts.contiguous().float()
    if loss_kwargs.get('batch_average', True):
        loss = eng.loss(loss_func, gts).view(())      # also leaves dL/dlogits in the engine
        loss._eosvos_engine = eng
        return loss
    # per-sample values (run_loader metrics, helper_func.py:131-137): every loss evaluated sample by sample
    return torch.cat([eng.loss_of(loss_func, outputs[b], gts[b]) for b in range(outputs.shape[0])])


def init_parent_model(architecture, encoder, train_encoder, decoder_norm_layer=None,
                      replace_batch_with_group_norms=False, batch_norm=None, roi_pool_output_sizes=None,
                      eval_augment_rpn_proposals_mode=None, box_nms_thresh=None, maskrcnn_loss=None, **datasets):
    """Same signature as the reference; returns (model, parent_states)."""
    if architecture == 'DeepLabV3':                         # helper_func.py:343-344
        model = DeepLabV3(encoder, num_classes=1, batch_norm=batch_norm, train_encoder=train_encoder)
    elif architecture == 'DeepLabV3Plus':
        model = DeepLabV3Plus(encoder, num_classes=1, batch_norm=batch_norm, train_encoder=train_encoder,
                              replace_batch_with_group_norms=replace_batch_with_group_norms)
    else:
        raise NotImplementedError(f"architecture='{architecture}': the MI355X engine implements DeepLabV3Plus and DeepLabV3")
    parent_states = {}
    for k, v in datasets.items():
        parent_states[k] = {
            'states': [torch.load(p, map_location='cpu', weights_only=False) for p in v.get('paths', [])],
            'splits': [np.loadtxt(p, dtype=str).tolist() for p in v.get('val_split_files', [])
                       if isinstance(p, str) and __import__('os').path.exists(p)],
        }
    return model, parent_states


def run_frames(model, frames, gts=None, loss_func='cross_entropy'):
    """Inference over frames (N,3,H,W) one at a time (batch 1, `test` batch size of the configs):
    returns (`loss_func` per frame or None, acc per frame or None, probs (N,1,H,W)) -- `run_loader`,
    helper_func.py:131-142, evaluates the configured loss with `batch_average: False`."""
    model.eval()
    probs, losses, accs = [], [], []
    for i in range(frames.shape[0]):
        x = frames[i:i + 1].contiguous()
        eng = model._ensure_engine(x.shape[2], x.shape[3], 1)
        p = eng.infer(x)
        probs.append(p)
        if gts is not None:
            logits = eng.debug_tensor('logits')[:1]
            losses.append(eng.loss_of(loss_func, logits, gts[i:i + 1].contiguous()))
            pred = p.ge(0.5)
            accs.append(pred.eq(gts[i:i + 1].bool()).float().mean().view(1))
    probs = torch.cat(probs)
    if gts is None:
        return None, None, probs
    return torch.cat(losses).cpu(), torch.cat(accs).cpu(), probs


def early_stopping(loss_hist, patience, min_loss_improv):
    if patience is None or len(loss_hist) <= patience:
        return False
    best = min(loss_hist)
    prev_best = min(loss_hist[:-patience])
    return not abs(best - prev_best) > min_loss_improv


def set_random_seeds(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


class EpochSampler:
    """Sample `num_epochs` passes over the dataset into ONE batch (`helper_func.py:521-545`)."""

    def __init__(self, dataset, shuffle, num_epochs, sampler=None):
        if shuffle and sampler is None:
            raise NotImplementedError('shuffle=True needs an explicit sampler here')
        self.sampler = sampler if sampler is not None else range(len(dataset))
        self.num_epochs = num_epochs

    def __iter__(self):
        batch = []
        for _ in range(self.num_epochs):
            batch.extend(self.sampler)
        yield batch

    def __len__(self):
        return 1
