"""Inputs shared by `make_g12.py` (reference run) and the replay test of the product's loop: tiny sequences whose
frames carry their index in the pixel values, per-object ground truth, and prescribed "network" probabilities
(including frames whose thresholded prediction is empty -- `evaluate.py:239` skips those propagated frames)."""
import torch

SCENARIOS = [
    dict(name='ona_first_step', seed=123, step=5, batch=3, reset_model_mode='FIRST_STEP', eval_epochs=3, ona_epochs=2,
         train_frame=0, seqs={'bear': dict(frames=12, objects=2), 'cows': dict(frames=7, objects=1)}),
    dict(name='ona_full_reset', seed=7, step=3, batch=2, reset_model_mode='FULL', eval_epochs=2, ona_epochs=1,
         train_frame=0, seqs={'dog': dict(frames=8, objects=1)}),
    dict(name='no_adaptation', seed=1, step=None, batch=3, reset_model_mode='FIRST_STEP', eval_epochs=2, ona_epochs=2,
         train_frame=0, seqs={'bear': dict(frames=5, objects=2)}),
]
# (sequence, object, frame) whose predicted probabilities stay below min_prop everywhere
EMPTY = {('bear', 1, 8), ('bear', 1, 9), ('bear', 0, 4), ('dog', 0, 2)}


def frame_image(i, hw):
    return torch.full((3, hw[0], hw[1]), i / 100.0)


def object_gt(seq, obj, hw):
    g = torch.zeros(1, hw[0], hw[1])
    if obj == 0:
        g[0, 1:4, 2:7] = 1.0
    else:
        g[0, 4:7, 6:11] = 1.0
    return g


def prob_map(seq, obj, f, hw):
    """Deterministic probabilities in (0, 1): a blob that drifts with the frame index; objects overlap on some frames so
    the arg-max merge (`evaluate.py:322-326`) has ties to break and background to find."""
    ys = torch.arange(hw[0]).view(-1, 1).float()
    xs = torch.arange(hw[1]).view(1, -1).float()
    cy, cx = (2.0 if obj == 0 else 5.0) + 0.25 * (f % 4), (4.0 if obj == 0 else 8.0) - 0.5 * (f % 3)
    p = torch.exp(-((ys - cy) ** 2 + (xs - cx) ** 2) / 6.0) * 0.95 + 0.01 * ((len(seq) + f) % 3)
    if (seq, obj, f) in EMPTY:
        p = p * 0.4
    return p.view(1, hw[0], hw[1])
