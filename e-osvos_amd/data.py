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
        self._multi_object_id_to_label = None                    # YouTube-VOS: object ids are not 1..n (vos_dataset.py:313-315)
        self._label_id = None                                    # label file used for every frame (first annotated frame)
        self.random_frame_id_epsilon = None
        self.random_frame_id_anchor_frame = None
        self.all_frames = False
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

    # all per-sequence state lives in plain attributes: a shallow copy is an independent cursor over the same file lists,
    # which is how `evaluate_dataset` / `MetaTaskset.load_frames` read ahead on worker threads
    prefetchable = True

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
        return self.make_image(idx), self.make_label(idx)

    def make_image(self, idx):
        img = np.array(Image.open(self.imgs[idx]).convert('RGB'), dtype=np.float32)
        if self.normalize:
            img = img - np.array(self.mean_val, dtype=np.float32)
        return img / 255.0

    def make_label(self, idx):
        """The label half of `make_img_label_pair` (the frame-selection loop only needs this one: no JPEG decode)."""
        if self._label_id is not None:                           # vos_dataset.py:235-242
            label = Image.open(self.labels[self._label_id])
        else:
            label = Image.open(self.labels[0] if self.test_mode else self.labels[idx])
        label = np.array(np.atleast_3d(label)[..., 0], dtype=np.float32)
        if self.multi_object and self.num_objects > 1:
            if self.multi_object != 'single_id':
                raise NotImplementedError("multi_object='all' (object groups) is a Mask R-CNN path")
            unique_labels = [l for l in np.unique(label) if l != 0.0]
            if unique_labels:
                assert self.multi_object_id is not None and self.multi_object_id < self.num_objects
                oid = self.multi_object_id + 1.0
                if self._multi_object_id_to_label:
                    oid = float(self._multi_object_id_to_label[self.multi_object_id])
                label = (label == oid).astype(np.float32) if oid in unique_labels else np.zeros_like(label)
        else:
            label = np.where(label != 0.0, 1.0, 0.0).astype(np.float32)
        return label

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

    def sequence_tensors(self, seq_name, device='cpu', with_frame_ids=False):
        """(frames (N,3,H,W), [train-frame mask (1,H,W) per object]) of one sequence, ready for
        `evaluate.evaluate_sequence`; with `with_frame_ids` also the frame each object is first annotated in -- the frame
        it is fine-tuned on (`train_loader.dataset.set_gt_frame_id()` per object, `evaluate.py:132-137`): 0 for DAVIS,
        possibly later for YouTube-VOS objects (`youtube.py:131-143`)."""
        self.set_seq(seq_name)
        frames = torch.stack([torch.from_numpy(self.make_image(i).transpose(2, 0, 1)) for i in range(len(self.imgs))])
        keep = (self.frame_id, self._label_id)
        gts, fids = [], []
        for o in range(self.num_objects):
            self.multi_object_id = o
            self.set_gt_frame_id()
            fids.append(int(self.frame_id))
            gts.append(torch.from_numpy(self.make_label(self.frame_id)[None]))
        self.multi_object_id = None
        self.frame_id, self._label_id = keep
        if with_frame_ids:
            return frames.to(device), [g.to(device) for g in gts], fids
        return frames.to(device), [g.to(device) for g in gts]

    all_frames = False

    def has_label_file(self, seq_name, frame_name):
        """all-frames splits: is this frame one of the annotated ones (`evaluate.py:334-335`: the others get no PNG)."""
        return any(frame_name in l for l in self.seqs[seq_name]['labels'])


    # ---- frame selection of the meta-train tasks (vos_dataset.py:73-146) ------------------------------
    @property
    def num_object_groups(self):
        if self.multi_object == 'all':
            raise NotImplementedError("multi_object='all' (object groups) is a Mask R-CNN path")
        return self.num_objects

    num_objects_in_group = 1

    def set_gt_frame_id(self):                                   # davis.py: the first frame carries the annotation
        self.frame_id, self._label_id = 0, None

    def get_random_frame_id(self):                               # vos_dataset.py:102-108 (torch RNG, as the reference)
        n = len(self.imgs)
        if self.random_frame_id_epsilon is not None:
            lo = max(0, self.random_frame_id_anchor_frame - self.random_frame_id_epsilon)
            hi = min(self.random_frame_id_anchor_frame + self.random_frame_id_epsilon + 1, n)
            return torch.randint(lo, hi, (1,)).item()
        return torch.randint(n, (1,)).item()

    def has_frame_object(self, frame_id):                        # vos_dataset.py:118-122
        label = self.make_label(frame_id)
        return len([l for l in np.unique(label) if l != 0.0]) == self.num_objects_in_group

    def get_random_frame_id_with_label(self):                    # vos_dataset.py:124-142: redraw until the object is visible
        while True:
            f = self.get_random_frame_id()
            if self.has_frame_object(f):
                return f

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


class YouTube(DAVIS):
    """YouTube-VOS reader (`src/data/youtube.py`): `root/<split>/{JPEGImages,Annotations}/<seq>/`, `root/<split>/meta.json`
    (objects per video and the frames they are annotated in), sequence lists `root/<seqs_key>.txt` with
    split = seqs_key.split('_')[0]; object ids are the meta.json keys (not 1..n); valid / test splits are test-mode and
    an object may first appear after frame 0 (`get_gt_frame_id`, `:131-143`)."""

    def __init__(self, seqs_key, root_dir, frame_id=None, transform=None, multi_object=False, normalize=False,
                 full_resolution=False):
        if full_resolution:
            raise NotImplementedError                            # youtube.py:24-25
        if multi_object not in (False, 'single_id', 'all'):
            raise NotImplementedError(multi_object)
        import json
        self.seqs_key, self.root_dir, self.frame_id = seqs_key, root_dir, frame_id
        self.transform, self.multi_object, self.normalize = transform, multi_object, normalize
        self.multi_object_id = None
        self._multi_object_id_to_label = None
        self._label_id = None
        self.random_frame_id_epsilon = self.random_frame_id_anchor_frame = None
        self.year = 2017
        seqs_file = os.path.join(root_dir, f'{seqs_key}.txt')
        if not os.path.exists(seqs_file):
            raise NotImplementedError('YouTube-VOS needs a sequence list file')       # youtube.py:33-37
        keys = [l.strip() for l in open(seqs_file) if l.strip()]
        self._split = seqs_key.split('_')[0]
        seqs_dir = os.path.join(root_dir, self._split)
        self.test_mode = self._split in ('valid', 'test', 'valid-all-frames', 'test-all-frames')
        self.all_frames = 'all-frames' in self._split
        self._meta_data = json.load(open(os.path.join(seqs_dir, 'meta.json')))
        self.seqs = OrderedDict()
        for k in keys:
            d_img, d_lab = os.path.join(seqs_dir, 'JPEGImages', k), os.path.join(seqs_dir, 'Annotations', k)
            imgs = [os.path.join(d_img, f) for f in np.sort(listdir_nohidden(d_img))]
            labs = [os.path.join(d_lab, f) for f in np.sort(listdir_nohidden(d_lab))]
            if self.all_frames:                                  # "we never train on all frames", youtube.py:78-80
                labs = labs + [labs[0]] * (len(imgs) - len(labs))
            if not self.test_mode:
                assert len(imgs) == len(labs), f'{self._split} {k}'
            self.seqs[k] = {'imgs': imgs, 'labels': labs}
        self._num_objects = None
        self.seq_key = None
        self.imgs = [p for s_ in self.seqs.values() for p in s_['imgs']]
        self.labels = [p for s_ in self.seqs.values() for p in s_['labels']]

    @property
    def num_objects(self):                                       # youtube.py:112-122
        if self.seq_key is None:
            raise NotImplementedError
        if not self.multi_object:
            return 1
        return len(self._meta_data['videos'][self.seq_key]['objects'])

    def set_seq(self, seq_name):                                 # youtube.py:124-129
        super().set_seq(seq_name)
        self._multi_object_id_to_label = [int(k) for k in sorted(self._meta_data['videos'][seq_name]['objects'].keys())]

    def get_gt_frame_id(self, multi_object_id):                  # youtube.py:131-143
        info = [v for _, v in sorted(self._meta_data['videos'][self.seq_key]['objects'].items())]
        first = info[multi_object_id][0] if 'test' in self.seqs_key else info[multi_object_id]['frames'][0]
        frame_id = [p.find(first) != -1 for p in self.imgs].index(True)
        label_id = [p.find(first) != -1 for p in self.labels].index(True)
        return frame_id, label_id

    def set_gt_frame_id(self):                                   # youtube.py:183-184 (single_id)
        self.frame_id, self._label_id = self.get_gt_frame_id(self.multi_object_id or 0)

    def get_random_frame_id(self):                               # youtube.py:98-110: 5-frame annotation stride
        if self.random_frame_id_epsilon is not None and 'all-frames' not in self._split:
            assert self.random_frame_id_epsilon % 5 == 0
            eps = self.random_frame_id_epsilon // 5
            lo = max(0, self.random_frame_id_anchor_frame - eps)
            hi = min(self.random_frame_id_anchor_frame + eps + 1, len(self.imgs))
            return torch.randint(lo, hi, (1,)).item()
        return super().get_random_frame_id()


class SyntheticSequences:
    """Seeded stand-in sequences with the reader interface of this module (no dataset is reachable offline): per
    sequence a base frame rolled 4 px per frame, two disjoint rectangular objects."""
    test_mode = False
    multi_object = 'single_id'

    def __init__(self, n_seqs=1, num_frames=12, height=480, width=854, seed=1):
        from . import synthetic
        self._syn = synthetic
        self.n, self.h, self.w, self.seed = num_frames, height, width, seed
        self.seqs_names = [f'synthetic{i:02d}' for i in range(n_seqs)]

    def _objects(self, seq):
        frames, gt = self._syn.synthetic_frames(1, self.h, self.w, seed=self.seed + self.seqs_names.index(seq), second_object=True)
        top = (torch.arange(self.h).view(-1, 1) < self.h // 2)
        return frames, [(gt[0] * top).float(), (gt[0] * ~top).float()]

    all_frames = False

    def sequence_tensors(self, seq, device='cpu', with_frame_ids=False):
        frames, gts = self._objects(seq)
        frames = frames.to(device)                       # the rolls run where the sequence will live (a 70-frame 480p sequence is
        seq_frames = torch.cat([torch.roll(frames, shifts=4 * i, dims=3) for i in range(self.n)])     # 344 MB: 0.9 s on the host)
        if with_frame_ids:
            return seq_frames, [g.to(device) for g in gts], [0] * len(gts)
        return seq_frames, [g.to(device) for g in gts]

    def frame_names(self, seq):
        return [f'{i:05d}' for i in range(self.n)]

    def label_maps(self, seq):
        _, gts = self._objects(seq)
        lab = torch.zeros(self.h, self.w, dtype=torch.uint8)
        for o, g in enumerate(gts):
            lab[g[0] > 0] = o + 1
        return np.stack([torch.roll(lab, shifts=4 * i, dims=1).numpy() for i in range(self.n)])


def open_dataset(name, split, root='data', **kw):
    """`data_loaders` dataset choice (`helper_func.py:264-275`): DAVIS-2016 / DAVIS-2017 / YouTube-VOS under `root/<name>`."""
    if name in ('DAVIS-2016', 'DAVIS-2017'):
        return DAVIS(split, os.path.join(root, name), **kw)
    if name == 'YouTube-VOS':
        return YouTube(split, os.path.join(root, name), **kw)
    raise NotImplementedError(name)


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
