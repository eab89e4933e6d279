"""Host-side feeder: DAVIS-layout sequence reader producing the tensor contract of SURVEY row A0.

Mirrors the parts of `src/data/vos_dataset.py` / `src/data/davis.py` the DeepLab loops need:
  * directory layout `root/JPEGImages/<res>/<seq>/*.jpg`, `root/Annotations/<res>/<seq>/*.png`, sequence lists
    `root/<seqs_key>.txt` (`davis.py:33-66`), sorted non-hidden files (`helpers.listdir_nohidden`);
  * `make_img_label_pair` (`vos_dataset.py:224-322`): RGB / 255 (mean subtraction only with `normalize`,
    `:273-276`), label = first channel of the PNG; single object: `label != 0` -> {0,1} (`:321-322`);
    `multi_object == 'single_id'`: the mask of object `multi_object_id` among the first frame's ids
    (`:289-320`); `num_objects` from the first annotation (`:52-70`);
  * `__getitem__` with a fixed `frame_id` (`:183-213`) and `test_mode` (label of frame 0 for every frame).
Decoding uses PIL (the reference uses cv2.imread(...)[..., ::-1] for JPEGs, which is the same RGB order).
`jaccard` replaces the `davis` package's J measure (`helper_func.py:444-458`): intersection over union of
binary masks, 1.0 when both are empty.  No GPU work happens here; `sequence_tensors(...)` hands device-ready
(N,3,H,W) / (1,H,W) tensors to `evaluate.evaluate_sequence`.
"""
import os
from collections import OrderedDict

import numpy as np
import torch
from PIL import Image


def listdir_nohidden(path):
    return [f for f in os.listdir(path) if not f.startswith('.')]


class DAVIS:
    mean_val = (104.00699, 116.66877, 122.67892)          # davis.py:20, subtracted from the RGB frame as is (:273-275)

    def __init__(self, seqs_key, root_dir, frame_id=None, transform=None, multi_object=False, normalize=False,
                 full_resolution=False):
        if multi_object not in (False, 'single_id', 'all'):
            raise NotImplementedError(multi_object)             # vos_dataset.py:292-293
        self.seqs_key, self.root_dir, self.frame_id = seqs_key, root_dir, frame_id
        self.transform, self.multi_object, self.normalize = transform, multi_object, normalize
        self.multi_object_id = None
        self.test_mode = 'test' in seqs_key                      # davis.py:26-27
        year = ''.join(ch for ch in os.path.basename(os.path.normpath(root_dir)) if ch.isdigit())
        self.year = int(year) if year else 2017
        res = '480p'
        if full_resolution:
            res = '1080p' if self.year == 2016 else 'Full-Resolution'
        seqs_file = os.path.join(root_dir, f'{seqs_key}.txt')
        keys = [l.strip() for l in open(seqs_file)] if os.path.exists(seqs_file) else [seqs_key]
        self.seqs = OrderedDict()
        for k in keys:
            d_img, d_lab = os.path.join(root_dir, 'JPEGImages', res, k), os.path.join(root_dir, 'Annotations', res, k)
            imgs = [os.path.join(d_img, f) for f in np.sort(listdir_nohidden(d_img))]
            labs = [os.path.join(d_lab, f) for f in np.sort(listdir_nohidden(d_lab))]
            if not self.test_mode:
                assert len(imgs) == len(labs), f'failure in: {k}'
            self.seqs[k] = {'imgs': imgs, 'labels': labs}
        self._num_objects = None
        self.seq_key = None
        self.imgs = [p for s in self.seqs.values() for p in s['imgs']]
        self.labels = [p for s in self.seqs.values() for p in s['labels']]
        if not os.path.exists(seqs_file):
            self.set_seq(seqs_key)

    @property
    def seqs_names(self):
        return list(self.seqs.keys())

    def set_seq(self, seq_name):                                 # vos_dataset.py:170-175
        self.imgs, self.labels = self.seqs[seq_name]['imgs'], self.seqs[seq_name]['labels']
        self.seq_key, self._num_objects = seq_name, None

    @property
    def num_objects(self):                                       # vos_dataset.py:52-70
        if self.seq_key is None:
            raise NotImplementedError
        if not self.multi_object:
            return 1
        if self._num_objects is None:
            label = np.atleast_3d(Image.open(self.labels[0]))[..., 0]
            self._num_objects = len([l for l in np.unique(label) if l != 0.0])
        return self._num_objects

    def make_img_label_pair(self, idx):                          # vos_dataset.py:224-322
        img = np.array(Image.open(self.imgs[idx]).convert('RGB'), dtype=np.float32)
        label = Image.open(self.labels[0] if self.test_mode else self.labels[idx])
        label = np.array(np.atleast_3d(label)[..., 0], dtype=np.float32)
        if self.normalize:
            img = img - np.array(self.mean_val, dtype=np.float32)
        img = img / 255.0
        if self.multi_object and self.num_objects > 1:
            if self.multi_object != 'single_id':
                raise NotImplementedError("multi_object='all' (object groups) is a Mask R-CNN path")
            unique_labels = [l for l in np.unique(label) if l != 0.0]
            if unique_labels:
                assert self.multi_object_id is not None and self.multi_object_id < self.num_objects
                oid = self.multi_object_id + 1.0
                label = (label == oid).astype(np.float32) if oid in unique_labels else np.zeros_like(label)
        else:
            label = np.where(label != 0.0, 1.0, 0.0).astype(np.float32)
        return img, label

    def __len__(self):
        return 1 if self.frame_id is not None else len(self.imgs)

    def __getitem__(self, idx):                                  # vos_dataset.py:183-213
        if self.frame_id is not None:
            idx = len(self.imgs) // 2 if self.frame_id == 'middle' else self.frame_id
        img, label = self.make_img_label_pair(idx)
        sample = {'image': img, 'gt': label, 'file_name': os.path.splitext(os.path.basename(self.imgs[idx]))[0]}
        if self.transform is not None:
            return self.transform(sample)
        return {'image': torch.from_numpy(img.transpose(2, 0, 1)), 'gt': torch.from_numpy(label[None]),     # ToTensor
                'file_name': sample['file_name']}

    def sequence_tensors(self, seq_name, device='cpu'):
        """(frames (N,3,H,W), [first-frame mask (1,H,W) per object]) of one sequence, ready for
        `evaluate.evaluate_sequence`."""
        self.set_seq(seq_name)
        frames = torch.stack([torch.from_numpy(self.make_img_label_pair(i)[0].transpose(2, 0, 1)) for i in range(len(self.imgs))])
        gts = []
        for o in range(self.num_objects):
            self.multi_object_id = o
            gts.append(torch.from_numpy(self.make_img_label_pair(0)[1][None]))
        self.multi_object_id = None
        return frames.to(device), [g.to(device) for g in gts]


    def frame_names(self, seq_name):
        """File stems of the frames of `seq_name` (prediction PNG names, `evaluate.py:333-340`)."""
        return [os.path.splitext(os.path.basename(p))[0] for p in self.seqs[seq_name]['imgs']]

    def label_maps(self, seq_name):
        """(N,H,W) integer ground-truth label maps of a sequence (object ids as annotated; binary in single-object
        mode) for the J measure."""
        out = []
        for p in self.seqs[seq_name]['labels']:
            lab = np.atleast_3d(Image.open(p))[..., 0]
            out.append(lab if self.multi_object else (lab != 0).astype(np.uint8))
        return np.stack(out)


def jaccard(pred, gt):
    """Region similarity J of two binary masks (the `davis` package's db_eval_iou): |A & B| / |A | B|, 1 if both empty."""
    pred, gt = np.asarray(pred).astype(bool), np.asarray(gt).astype(bool)
    union = np.logical_or(pred, gt).sum()
    return 1.0 if union == 0 else float(np.logical_and(pred, gt).sum()) / float(union)


def sequence_J(labels, gt_labels, num_objects):
    """Mean J over objects and over frames 1..N-2 (DAVIS evaluates all but the first and last frame)."""
    labels, gt_labels = np.asarray(labels), np.asarray(gt_labels)
    js = []
    for o in range(1, num_objects + 1):
        js.append(np.mean([jaccard(labels[f] == o, gt_labels[f] == o) for f in range(1, len(labels) - 1)]))
    return float(np.mean(js))
