"""Checkpoint files layout-compatible with the reference.

`src/train_meta.py:277-286` writes `{save_dir}/{run}/last_meta_iter.model` =
torch.save({'meta_optim_state_dict', 'vis_win_names', 'meta_iter', 'meta_epoch'}); the eval
workers write `last_{key}_meta_iter.model` / `best_{key}_meta_iter.model` with the same dict
(`src/util/evaluate.py:361-382`).  RAdam state is not saved by the reference (`:281`).
"""
import os
from collections import OrderedDict

import torch


def save_meta_checkpoint(path, meta_optim_state_dict, meta_iter, meta_epoch, vis_win_names=None):
    sd = OrderedDict((k, v.detach().cpu().clone()) for k, v in meta_optim_state_dict.items())
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save({'meta_optim_state_dict': sd, 'vis_win_names': vis_win_names or {},
                'meta_iter': meta_iter, 'meta_epoch': meta_epoch}, path)


def load_meta_checkpoint(path):
    ck = torch.load(path, map_location='cpu', weights_only=False)
    return ck['meta_optim_state_dict'], {k: ck.get(k) for k in ('vis_win_names', 'meta_iter', 'meta_epoch')}


def checkpoint_names(save_dir, run, key=None):
    """File names the reference uses (train_meta.py:280, evaluate.py:364-382)."""
    base = os.path.join(save_dir, run)
    if key is None:
        return {'last': os.path.join(base, 'last_meta_iter.model')}
    return {'last': os.path.join(base, f'last_{key}_meta_iter.model'),
            'best': os.path.join(base, f'best_{key}_meta_iter.model')}
