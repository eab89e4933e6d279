"""`train_meta.main` with no iteration cap on CPU (stand-in engine): runs until SIGTERM.  usage: train_meta_forever.py SAVE_DIR"""
import os
import sys

import common
from eosvos_amd import train_meta

save_dir = sys.argv[1]
train_meta.init_parent_model = common.fake_init_parent_model
train_meta.main(['with', 'YouTube-VOS', 'meta_batch_size=1', 'num_epochs.train=1', 'vis_interval=1000', f'save_dir={save_dir}',
                 'env_suffix=forever'], height=common.H, width=common.W, num_meta_iters=None,
                data_root=os.path.join(save_dir, 'no_data'), eval_cmd=False, device='cpu')
