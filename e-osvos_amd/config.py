"""Sacred-compatible configuration surface of `src/train_meta.py` for the DeepLab hot path.

The reference wires `cfgs/meta.yaml` + `cfgs/torch.yaml` as base config and four named configs
(`train_meta.py:21-27`) and is driven as
    python src/train_meta.py with DAVIS-2017 e-OSVOS-OnA key=val ...
Sacred is not installed in the target image, so this module keeps the same *keys* (same names and
nesting as cfgs/meta.yaml), the same named configs and the
same `with <named...> <dotted.key=value...>` command-line grammar; an external YAML file in the
reference's format can be merged with `load_yaml`.  Values below are the reference defaults,
except that the hot-path model/loss are selected (`parent_model.architecture=DeepLabV3Plus`,
`loss_func=cross_entropy`, frozen BatchNorm) -- the reference's shipped defaults select
Mask R-CNN (`cfgs/meta.yaml:68-84`), which is out of scope.
"""
import copy

import yaml

BASE = {
    'seed': 1,
    'meta_batch_size': 4,
    'num_meta_processes_per_gpu': 1,
    'num_eval_gpus': None,
    'no_vis': True,
    'vis_interval': 10,
    'env_suffix': None,
    'save_dir': 'models',
    'resume_meta_run_epoch_mode': None,
    'increase_seed_per_meta_run': True,
    'random_frame_transform_per_task': True,
    'multi_step_bptt_loss': False,
    'random_frame_epsilon': None,
    'random_object_id_sub_group': False,
    'num_epochs': {'train': 5, 'eval': 10},
    'bptt_epochs': 5,
    'eval_online_adapt': {'step': 0, 'reset_model_mode': 'FIRST_STEP', 'num_epochs': 10, 'min_prop': 0.5},
    'meta_optim_model_file': None,
    'meta_optim_cfg': {'lr_hierarchy_level': 'NEURON', 'init_lr': 0.001, 'learn_model_init': True,
                       'second_order_gradients': False, 'use_log_init_lr': False, 'max_lr': None},
    'meta_optim_optim_cfg': {'model_init_lr': 0.00001, 'log_init_lr_lr': 0.00001, 'lr': 0.001,
                             'freeze_encoder': False, 'grad_clip': None, 'model_init_weight_decay': 0.001},
    'eval_datasets': True,
    'datasets': {'train': {'name': 'DAVIS-2016', 'split': 'train_seqs', 'eval': True},
                 'val': {'name': 'DAVIS-2016', 'split': 'val_seqs', 'eval': True},
                 'test': {'name': 'DAVIS-2016', 'split': None, 'eval': False}},
    'loss_func': 'cross_entropy',
    'parent_model': {'architecture': 'DeepLabV3Plus', 'train_encoder': True,
                     'batch_norm': {'accum_stats': False, 'learn_weight': False, 'learn_bias': False},
                     'replace_batch_with_group_norms': False, 'decoder_norm_layer': 'BatchNorm2d',
                     'eval_augment_rpn_proposals_mode': None, 'roi_pool_output_sizes': {'box': 7, 'mask': 28},
                     'maskrcnn_loss': None, 'box_nms_thresh': None, 'encoder': 'resnet50',
                     # per-dataset parent checkpoints (`init_parent_model(**datasets)`, helper_func.py:339-385)
                     'train': {'paths': [], 'val_split_files': ['data/DAVIS-2017/train_val_seqs.txt']},
                     'val': {'paths': [], 'val_split_files': ['data/DAVIS-2017/train_val_seqs.txt']},
                     'test': {'paths': [], 'val_split_files': ['data/DAVIS-2017/test-dev_seqs.txt']}},
    'train_early_stopping_cfg': {'patience': None, 'min_loss_improv': 0.001},
    'single_obj_seq_mode': 'KEEP',
    'random_flip_label': False,
    'random_no_label': False,
    'random_box_coord_perm': False,
    'data_cfg': {'multi_object': False, 'random_train_transform': False, 'num_workers': 0, 'pin_memory': False,
                 'normalize': False, 'full_resolution': False,
                 'frame_ids': {'train': 0, 'test': None, 'meta': None},
                 'batch_sizes': {'train': 1, 'test': 1, 'meta': 1},
                 'shuffles': {'train': True, 'test': False, 'meta': False},
                 'crop_sizes': {'train': None, 'test': None, 'meta': None}},
    'torch_cfg': {'print_config': False, 'device': None, 'deterministic': True, 'benchmark': False,
                  'vis': {'port': 8090, 'server': 'http://localhost'}},
}

# named configs of train_meta.py:24-27 (hot-path keys of cfgs/meta_davis-2017.yaml,
# meta_youtube-vos.yaml, eval_e-osvos.yaml, eval_e-osvos-OnA.yaml)
NAMED = {
    'DAVIS-2017': {'datasets': {'train': {'name': 'DAVIS-2017', 'split': 'train_seqs', 'eval': True},
                                'val': {'name': 'DAVIS-2017', 'split': 'val_seqs', 'eval': True},
                                'test': {'name': 'DAVIS-2017', 'split': 'test-dev_seqs', 'eval': False}},
                   'data_cfg': {'multi_object': 'single_id'}},
    'YouTube-VOS': {'datasets': {'train': {'name': ['YouTube-VOS', 'DAVIS-2017'],
                                           'split': ['train_dev_random_123_train_seqs', 'train_seqs'], 'eval': False},
                                 'val': {'name': 'YouTube-VOS', 'split': 'valid-all-frames_seqs', 'eval': False},
                                 'test': {'name': 'YouTube-VOS', 'split': None, 'eval': False},
                                 'train_dev_train_val': {'name': 'YouTube-VOS', 'split': 'train_dev_random_123_train_val_seqs',
                                                         'eval': False},
                                 'train_dev_val': {'name': 'YouTube-VOS', 'split': 'train_dev_random_123_val_seqs', 'eval': False},
                                 'val_davis16': {'name': 'DAVIS-2016', 'split': 'val_seqs', 'eval': False},
                                 'val_davis17': {'name': 'DAVIS-2017', 'split': 'val_seqs', 'eval': True}},
                    'parent_model': {'train': {'paths': [], 'val_split_files': []}, 'val': {'paths': [], 'val_split_files': []},
                                     'test': {'paths': [], 'val_split_files': []}},
                    'data_cfg': {'multi_object': 'single_id'}},
    # the iteration counts come from the command line as in the reference README
    # (`with DAVIS-2017 e-OSVOS num_epochs.eval=50`, `... e-OSVOS-OnA num_epochs.eval=100`)
    'e-OSVOS': {'no_vis': True, 'num_meta_processes_per_gpu': 0,
                'data_cfg': {'batch_sizes': {'train': 3}, 'random_train_transform': True}},
    'e-OSVOS-OnA': {'no_vis': True, 'num_meta_processes_per_gpu': 0,
                    'eval_online_adapt': {'step': 5, 'num_epochs': 10},
                    'data_cfg': {'batch_sizes': {'train': 3}, 'random_train_transform': True}},
}


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _set_dotted(cfg, key, value):
    parts = key.split('.')
    d = cfg
    for p in parts[:-1]:
        if p not in d or not isinstance(d[p], dict):
            raise KeyError(f'unknown config key: {key}')
        d = d[p]
    if parts[-1] not in d:
        raise KeyError(f'unknown config key: {key}')       # Sacred also rejects unknown keys
    d[parts[-1]] = value


def load_yaml(path, cfg=None):
    cfg = cfg if cfg is not None else copy.deepcopy(BASE)
    with open(path) as f:
        return _merge(cfg, yaml.safe_load(f) or {})


def parse_cli(argv):
    """`[with] <named config | key=value> ...` -> config dict (Sacred's CLI grammar, README.md:56-83)."""
    cfg = copy.deepcopy(BASE)
    args = list(argv)
    if args and args[0] == 'with':
        args = args[1:]
    updates = []
    for a in args:
        if '=' in a:
            k, v = a.split('=', 1)
            updates.append((k, yaml.safe_load(v)))
        elif a in NAMED:
            _merge(cfg, NAMED[a])
        elif a.endswith('.yaml'):
            load_yaml(a, cfg)
        else:
            raise KeyError(f'unknown named config: {a}')
    for k, v in updates:
        _set_dotted(cfg, k, v)
    unsupported(cfg)
    return cfg


def unsupported(cfg):
    """Options of cfgs/meta.yaml this build accepts as keys but does not implement must not be ignored silently."""
    crops = (cfg.get('data_cfg') or {}).get('crop_sizes') or {}
    if any(v is not None for v in crops.values()):
        raise NotImplementedError('data_cfg.crop_sizes (pad + random crop of every frame, data/vos_dataset.py:246-275) is not implemented: '
                                  'frames are fed at their native size')
