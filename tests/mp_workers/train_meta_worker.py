"""One rank of `train_meta.main` in meta-train mode on CPU (gloo): usage train_meta_worker.py OUT_PREFIX SAVE_DIR"""
import os
import sys

import torch

import common
from eosvos_amd import train_meta

out, save_dir = sys.argv[1], sys.argv[2]
os.environ['EOSVOS_DIST_BACKEND'] = 'gloo'
train_meta.init_parent_model = common.fake_init_parent_model
eval_cmd = [sys.executable, os.path.join(common.HERE, 'eval_child.py')]
mt = train_meta.main(['with', 'YouTube-VOS', 'meta_batch_size=4', 'num_epochs.train=2', 'vis_interval=1', f'save_dir={save_dir}',
                      'env_suffix=mp', 'num_epochs.eval=2'], height=common.H, width=common.W, num_frames=4, num_meta_iters=2,
                     data_root=os.path.join(save_dir, 'no_data'), eval_cmd=eval_cmd, device='cpu')
torch.save({'state': mt.state.clone(), 'step': mt.step}, f'{out}.{os.environ.get("RANK", "0")}')
