"""Evaluation worker logic of `src/util/evaluate.py` for the DeepLab path, on tensors.

`evaluate_sequence` restates the intended semantics of `evaluate.py:132-326` (the reference's
own loop cannot run the DeepLab model at this commit):
  per object: fine-tune on the first frame for `num_epochs.eval` iterations (`:207-281`), predict
  the following frames (`:293-314`); with online adaptation every `step` frames re-fine-tune for
  `eval_online_adapt.num_epochs` iterations on [first frame + earlier frames with their own
  thresholded predictions as ground truth] (`:170-193,227-253`), restoring the round-0 weights
  first when `reset_model_mode == 'FIRST_STEP'` (`:200-205,283-287`);
  finally merge objects: background where max prob < 0.5, else argmax + 1 (`:322-326`), the
  train frame being seeded with 2*GT (`:167-168`).
Data loading is the caller's business; the first-frame augmentation of round 0
(`random_train_transform`, `evaluate.py:215-218`, `helper_func.py:255-261`) runs on the device
(`custom_transforms.FirstFrameAugmenter`) when `data_cfg.random_train_transform` is set, or through a
caller-supplied `augment(frame, gt, batch, seed)`; benchmarks use synthetic frames.
"""
import torch

from .helper_func import compute_loss, early_stopping, set_random_seeds


def online_adapt_schedule(num_frames, train_frame_id, step, train_batch_size):
    """Frame ranges and propagated frames per round (`evaluate.py:140-193,231-240`)."""
    rounds = []
    s = step if step else num_frames
    n_rounds = len(range(train_frame_id + 1, num_frames, step)) if step else 1
    eval_max = None
    for r in range(max(n_rounds, 1)):
        if r == 0:
            eval_min = train_frame_id + 1
            eval_max = eval_min
            prop = []
        else:
            eval_min = eval_max
            n_prop = min(step, train_batch_size)
            prop = [eval_min - j for j in range(step - n_prop + 1, step)]
        eval_max = min(eval_max + s, num_frames)
        rounds.append(dict(eval_min=eval_min, eval_max=eval_max, propagate_frames=prop))
        if eval_max == num_frames:
            break
    return rounds


def device_augment(model):
    """Round-0 batches = `batch` independent flip / scale / rotate warps of the first frame, drawn with the
    `random` state that `set_random_seeds(seed + epoch + round)` just set (evaluate.py:221-224)."""
    from .custom_transforms import FirstFrameAugmenter

    def fn(frame, gt, batch, seed):
        eng = model._ensure_engine(frame.shape[2], frame.shape[3], batch)
        images, labels, _ = FirstFrameAugmenter(eng).batch(frame[0].contiguous(), gt[0].contiguous(), batch)
        return images, labels
    return fn


def _repeat_batch(frame, gt, batch, seed):
    return frame.expand(batch, -1, -1, -1).contiguous(), gt.expand(batch, -1, -1, -1).contiguous()


def evaluate_sequence(model, meta_optim, meta_optim_state_dict, frames, object_gts, cfg, augment=None,
                      train_frame_id=0):
    """frames (N,3,H,W) on the GPU, object_gts: list of (1,H,W) binary masks of the train frame.
    cfg keys (names of cfgs/meta.yaml): num_epochs.eval, eval_online_adapt.{step,reset_model_mode,
    num_epochs,min_prop}, data_cfg.batch_sizes.train, seed, loss_func, train_early_stopping_cfg.
    Returns (labels (N,H,W) uint8, per-object probs list, train loss history per object)."""
    if augment is None and cfg['data_cfg'].get('random_train_transform'):
        augment = device_augment(model)
    augment = augment or _repeat_batch
    n = frames.shape[0]
    ona = cfg['eval_online_adapt']
    step = ona['step']
    bsz = cfg['data_cfg']['batch_sizes']['train']
    es = cfg.get('train_early_stopping_cfg', {'patience': None, 'min_loss_improv': 0.001})
    loss_func = cfg.get('loss_func', 'cross_entropy')
    probs_all, hist_all = [], []
    for gt in object_gts:
        gt = gt.to(frames.device).float().view(1, 1, *gt.shape[-2:])
        masks = torch.zeros(n, 1, *frames.shape[-2:], device=frames.device)
        masks[train_frame_id] = 2 * gt[0]                               # evaluate.py:167-168
        hist = []
        for r, rd in enumerate(online_adapt_schedule(n, train_frame_id, step, bsz)):
            if r == 0 or ona['reset_model_mode'] == 'FULL':
                meta_optim.load_state_dict(meta_optim_state_dict)
                meta_optim.reset()
                meta_optim.eval()
            elif ona['reset_model_mode'] == 'FIRST_STEP':
                meta_optim.load_state_dict(meta_optim_state_dict)
                if model._dirty:                    # new lrs: push them without touching theta twice
                    model.push_state()
                model.engine.restore()              # model.load_state_dict(model_state_dict_first_step)
                meta_optim.eval()
            num_epochs = cfg['num_epochs']['eval'] if r == 0 else ona['num_epochs']
            model.train_without_dropout()
            round_hist = []
            for epoch in range(1, num_epochs + 1):
                set_random_seeds(cfg.get('seed', 1) + epoch + r)
                if r == 0:
                    inputs, gts = augment(frames[train_frame_id:train_frame_id + 1], gt, bsz, cfg.get('seed', 1) + epoch)
                else:
                    inputs, gts = frames[train_frame_id:train_frame_id + 1], gt
                    for f in rd['propagate_frames']:
                        pg = masks[f:f + 1].ge(ona['min_prop']).float()
                        if pg.sum().item() != 0:                        # evaluate.py:239
                            inputs = torch.cat([inputs, frames[f:f + 1]])
                            gts = torch.cat([gts, pg])
                    inputs, gts = inputs.contiguous(), gts.contiguous()
                outputs = model(inputs)
                train_loss = compute_loss(loss_func, outputs[-1], gts)
                round_hist.append(train_loss.item())
                model.zero_grad()
                meta_optim.set_train_loss(train_loss)
                meta_optim.step(train_loss)
                meta_optim.meta_model.detach_param_groups()
                if early_stopping(round_hist, **es):
                    break
            hist.append(round_hist)
            if r == 0:
                model.engine.snapshot()             # model_state_dict_first_step (evaluate.py:283-287)
            model.eval()
            for f in range(rd['eval_min'], rd['eval_max']):
                masks[f] = model.engine.infer(frames[f:f + 1].contiguous())[0]
        probs_all.append(masks[:, 0])
        hist_all.append(hist)
    stack = torch.stack(probs_all, dim=1)                               # (N, n_obj, H, W)
    eng = model.engine
    labels = torch.stack([eng.merge_labels(stack[f].contiguous()) for f in range(n)])
    return labels, probs_all, hist_all
