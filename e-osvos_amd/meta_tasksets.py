"""Meta-train task sampling, `src/meta_optim/meta_tasksets.py:55-155` for the DeepLab path.

One task = one (sequence, object) pair: a random train frame on which the object is visible, `batch_sizes.meta`
random meta frames (optionally within `random_frame_epsilon` of the train frame), and -- with
`random_frame_transform_per_task` -- ONE colour jitter / horizontal flip / scale-rotate drawn per task and applied
to its train and meta frames alike (`deterministic=True` transforms, `:115-137`).

The reference hands deep-copied DataLoaders to the worker; here an item is a small dict of indices and transform
parameters, and `task_tensors` materialises it straight into HBM: frames are decoded and colour-jittered on the host
(PIL, a handful of frames per task), the flip + scale/rotate warp runs on the device (`eosvos_warp_affine`, cubic for
frames / nearest for labels like `custom_transforms.py:42-52`).

Random draws follow the reference's generators: frame ids from the torch RNG (`vos_dataset.py:102-108`), transform
parameters from `random` (`custom_transforms.py:26-33,193-194`); the colour jitter restates torchvision 0.4's
`ColorJitter.get_params` + PIL `ImageEnhance` path (torchvision is not in the image: unpinned, like the cv2 warp).
Mask R-CNN-only options (`random_box_coord_perm`, `random_object_id_sub_group`, `single_obj_seq_mode` AUGMENT_*)
raise NotImplementedError.
"""
import copy
import random

import numpy as np
import torch
from PIL import Image, ImageEnhance

from .custom_transforms import INTER_CUBIC, INTER_NEAREST, RandomScaleNRotate, warp_affine


class ColorJitterParams:
    """torchvision 0.4 `ColorJitter.get_params(brightness, contrast, saturation, hue)`: one factor per property from
    `random.uniform`, applied in a `random.shuffle`d order (drawn once per task, `custom_transforms.py:131-137`)."""

    def __init__(self, brightness=.2, contrast=.2, saturation=.2, hue=.1, rng=random):
        ops = []
        if brightness:
            ops.append(('brightness', rng.uniform(max(0, 1 - brightness), 1 + brightness)))
        if contrast:
            ops.append(('contrast', rng.uniform(max(0, 1 - contrast), 1 + contrast)))
        if saturation:
            ops.append(('saturation', rng.uniform(max(0, 1 - saturation), 1 + saturation)))
        if hue:
            ops.append(('hue', rng.uniform(-hue, hue)))
        rng.shuffle(ops)
        self.ops = ops

    def __call__(self, img01):
        """(H,W,3) float in [0,1] -> same, through uint8 PIL like the reference (`custom_transforms.py:143-146`)."""
        im = Image.fromarray(np.uint8(img01 * 255))
        for name, f in self.ops:
            if name == 'brightness':
                im = ImageEnhance.Brightness(im).enhance(f)
            elif name == 'contrast':
                im = ImageEnhance.Contrast(im).enhance(f)
            elif name == 'saturation':
                im = ImageEnhance.Color(im).enhance(f)
            else:                                             # torchvision.transforms.functional.adjust_hue
                h, s, v = im.convert('HSV').split()
                nh = np.array(h, dtype=np.uint8)
                nh = (nh.astype(np.int32) + int(f * 255)).astype(np.uint8)     # uint8 wrap-around of the hue channel
                im = Image.merge('HSV', (Image.fromarray(nh, 'L'), s, v)).convert('RGB')
        return np.array(im, dtype=np.float32) / 255


class MetaTaskset:
    def __init__(self, dataset, data_cfg, random_frame_transform_per_task=True, random_flip_label=False,
                 random_no_label=False, single_obj_seq_mode='KEEP', random_box_coord_perm=False,
                 random_frame_epsilon=None, random_object_id_sub_group=False):
        if random_box_coord_perm or random_object_id_sub_group:
            raise NotImplementedError('box permutations / object sub-groups are Mask R-CNN options')
        if single_obj_seq_mode not in ('KEEP', 'IGNORE', 'ONLY'):
            raise NotImplementedError(f'single_obj_seq_mode={single_obj_seq_mode}')      # AUGMENT_* paste other sequences
        if random_frame_transform_per_task and data_cfg.get('random_train_transform'):
            raise NotImplementedError                                                    # meta_tasksets.py:138-140
        self.dataset, self.data_cfg = dataset, data_cfg
        self.random_frame_transform_per_task = random_frame_transform_per_task
        self.random_flip_label, self.random_no_label = random_flip_label, random_no_label
        self.random_frame_epsilon = random_frame_epsilon
        self.object_groups = []
        for seq in dataset.seqs_names:                                                  # meta_tasksets.py:37-50
            dataset.set_seq(seq)
            if dataset.num_objects == 1:
                if single_obj_seq_mode == 'IGNORE':
                    continue
            elif single_obj_seq_mode == 'ONLY':
                continue
            for i in range(dataset.num_object_groups):
                self.object_groups.append((seq, i))

    def __len__(self):
        return len(self.object_groups)

    def __getitem__(self, idx):
        seq, obj = self.object_groups[idx]
        ds = self.dataset
        ds.set_seq(seq)
        ds.multi_object_id = obj
        ds._label_id = None
        ds.random_frame_id_epsilon = None
        train_frame = ds.get_random_frame_id_with_label()                              # meta_tasksets.py:99
        if self.random_frame_epsilon is not None:
            ds.random_frame_id_epsilon = self.random_frame_epsilon
            ds.random_frame_id_anchor_frame = train_frame
        meta_frames = [ds.get_random_frame_id_with_label() for _ in range(self.data_cfg['batch_sizes']['meta'])]
        ds.random_frame_id_epsilon = None
        item = {'seq_name': seq, 'obj_id': obj, 'train_frame': train_frame, 'meta_frames': meta_frames,
                'box_coord_perm': None, 'transform': None, 'flip_label': False, 'no_label': False}
        if self.random_frame_transform_per_task:                                        # meta_tasksets.py:111-137
            color = ColorJitterParams(brightness=.2, contrast=.2, hue=.1, saturation=.2)
            flip = random.random() < 0.5                                                # RandomHorizontalFlip(deterministic=True)
            item['transform'] = {'color': color, 'flip': flip, 'snr': RandomScaleNRotate(rots=(-30, 30), scales=(.5, 1.0)),
                                 'rot_sc': {}}
        if self.random_flip_label:
            item['flip_label'] = bool(random.getrandbits(1))
        if self.random_no_label:
            item['no_label'] = bool(random.getrandbits(1))
        return item

    # ---- materialisation ------------------------------------------------------------------------------
    def load_frames(self, item):
        """The host half of `task_tensors`: decode every frame of the task and apply its colour jitter.  No random draw
        and no device work happen here, and the dataset is used through a private shallow copy, so this can run on a
        worker thread while the GPU is busy with the previous meta-iteration (`train_meta` prefetches one iteration
        ahead).  -> {frame_id: (image HxWx3 float32, label HxW float32)}."""
        ds = copy.copy(self.dataset)
        ds.set_seq(item['seq_name'])
        ds.multi_object_id = item['obj_id']
        ds._label_id = None
        out = {}
        for frame_id in [item['train_frame']] + list(item['meta_frames']):
            if frame_id in out:
                continue
            img, label = ds.make_img_label_pair(frame_id)
            if item['transform'] is not None:
                img = item['transform']['color'](img)
            out[frame_id] = (img, label)
        return out

    def _frame(self, item, frame_id, engine, device, host=None):
        tr = item['transform']
        if host is not None and frame_id in host:
            img, label = host[frame_id]
        else:
            ds = self.dataset
            ds.set_seq(item['seq_name'])
            ds.multi_object_id = item['obj_id']
            img, label = ds.make_img_label_pair(frame_id)
            if tr is not None:
                img = tr['color'](img)
        x = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).to(device)
        y = torch.from_numpy(label[None].copy()).to(device)
        if item['flip_label']:
            y = 1.0 - y
        if item['no_label']:
            y = torch.zeros_like(y)
        if tr is not None:
            # deterministic=True: the scale / rotation of a frame is drawn once per file name and the draw is repeated
            # until the warped label still shows the object (custom_transforms.py:56-75)
            key = frame_id
            has_object = bool((y != 0).any()) and bool((y == 0).any())
            while True:
                rot, sc = tr['rot_sc'].get(key) or tr['snr'].draw(random)
                lab, nz = warp_affine(engine, y.contiguous(), tr['flip'], rot, sc, INTER_NEAREST, count_nonzero=True)
                if not has_object or 0 < nz < y.numel():
                    tr['rot_sc'][key] = (rot, sc)
                    break
                tr['rot_sc'].pop(key, None)
            x, _ = warp_affine(engine, x.contiguous(), tr['flip'], rot, sc, INTER_CUBIC)
            y = lab
        return x, y

    def task_tensors(self, item, engine, device, host=None):
        """-> (x_train (B,3,H,W), y_train (B,1,H,W), x_meta (Bm,3,H,W), y_meta (Bm,1,H,W)) on `device`: the train frame
        repeated `batch_sizes.train` times (EpochSampler, `helper_func.py:521-545`) and the meta frames.  `host`: the
        result of `load_frames(item)` when it was prepared ahead of time."""
        bt = self.data_cfg['batch_sizes']['train']
        xt, yt = self._frame(item, item['train_frame'], engine, device, host)
        xs, ys = zip(*[self._frame(item, f, engine, device, host) for f in item['meta_frames']])
        return (xt.unsqueeze(0).expand(bt, -1, -1, -1).contiguous(), yt.unsqueeze(0).expand(bt, -1, -1, -1).contiguous(),
                torch.stack(xs).contiguous(), torch.stack(ys).contiguous())


class ConcatTaskset:
    """`torch.utils.data.ConcatDataset` of task sets (YouTube-VOS + DAVIS-2017, `meta_run.py:41-60`)."""

    def __init__(self, sets):
        self.sets = list(sets)
        self.offsets = np.cumsum([0] + [len(s) for s in self.sets])

    def __len__(self):
        return int(self.offsets[-1])

    def locate(self, idx):
        k = int(np.searchsorted(self.offsets, idx, side='right') - 1)
        return self.sets[k], idx - int(self.offsets[k])

    def __getitem__(self, idx):
        s, i = self.locate(idx)
        item = s[i]
        item['taskset'] = s
        return item


def task_order(n_tasks, sub_batch, seed, epoch):
    """`DataLoader(meta_task_set, shuffle=True, batch_size=sub_meta_batch_size)` of one worker (`meta_run.py:76-81`):
    a seeded permutation per pass over the task set, cut into sub-batches (the last one may be short)."""
    g = torch.Generator().manual_seed(seed + 7919 * epoch)
    perm = torch.randperm(n_tasks, generator=g).tolist()
    return [perm[i:i + sub_batch] for i in range(0, n_tasks, sub_batch)]
