"""Device-side mirror of the first-frame augmentation of `src/data/custom_transforms.py`.

The reference re-runs `RandomHorizontalFlip` + `RandomScaleNRotate` (cv2 cubic / nearest warps on the host)
for every sample of every fine-tune iteration (`src/util/evaluate.py:224`, `helper_func.py:255-261`), which would
leave the engine's 20 ms iteration waiting for the host.  Here the frame and its label stay in HBM and each
sample is one `eosvos_warp_affine` launch per tensor; the random parameters are drawn on the host with the same
`random` calls in the same order as the reference (`custom_transforms.py:26-33,200`), so a seeded run picks the
same flips, angles and scales.

    aug = FirstFrameAugmenter(engine)
    images, gts = aug.batch(frame, label, batch_size)      # (B,3,H,W), (B,1,H,W) device tensors
"""
import ctypes
import random

import torch

from . import _ffi

INTER_NEAREST, INTER_CUBIC = 0, 2           # include/eosvos.h EOSVOS_INTER_*


def warp_affine(engine, src, flip, rot, sc, interp, out=None, count_nonzero=False):
    """One flip + scale/rotate warp of a (C,H,W) device tensor; returns (dst, nonzero or None)."""
    assert src.is_cuda and src.dtype == torch.float32 and src.is_contiguous() and src.dim() == 3
    # (any frame size: `eosvos_warp_affine_hw` -- the engine lends its stream and coefficient table only)
    dst = torch.empty_like(src) if out is None else out
    cnt = ctypes.c_int(0)
    _ffi.check(engine.lib.eosvos_warp_affine_hw(engine.h, ctypes.c_void_p(src.data_ptr()), src.shape[0], int(src.shape[1]), int(src.shape[2]),
                                                int(bool(flip)), float(rot), float(sc), interp, ctypes.c_void_p(dst.data_ptr()),
                                                ctypes.byref(cnt) if count_nonzero else None))
    return dst, (cnt.value if count_nonzero else None)


class RandomHorizontalFlip:
    """`custom_transforms.py:189-213` (non-deterministic mode): draws the flip decision."""

    def draw(self, rng=random):
        return rng.random() < 0.5


class RandomScaleNRotate:
    """`custom_transforms.py:9-92`, continuous ranges."""

    def __init__(self, rots=(-30, 30), scales=(.75, 1.25)):
        if not isinstance(rots, tuple) or not isinstance(scales, tuple):
            raise NotImplementedError('fixed lists of rotations / scales (custom_transforms.py:34-37)')
        self.rots, self.scales = rots, scales

    def draw(self, rng=random):
        rot = (self.rots[1] - self.rots[0]) * rng.random() - (self.rots[1] - self.rots[0]) / 2
        sc = (self.scales[1] - self.scales[0]) * rng.random() - (self.scales[1] - self.scales[0]) / 2 + 1
        return rot, sc


class FirstFrameAugmenter:
    """`data_loaders(random_train_transform=True)` for one (frame, label) pair, on the engine's GPU."""

    def __init__(self, engine, rots=(-30, 30), scales=(.75, 1.25), rng=random):
        self.engine, self.rng = engine, rng
        self.flip, self.snr = RandomHorizontalFlip(), RandomScaleNRotate(rots, scales)

    @staticmethod
    def has_object(label):
        """num_labels > 1 for a binary mask: both object and background pixels."""
        nz = int((label != 0).sum())
        return 0 < nz < label.numel()

    def sample(self, frame, label, out_image=None, out_label=None, has_object=None):
        """frame (3,H,W), label (1,H,W) device tensors -> (image, label, params)."""
        flip = self.flip.draw(self.rng)
        if has_object is None:
            has_object = self.has_object(label)
        total = label.numel()
        tries = 0
        while True:
            tries += 1
            rot, sc = self.snr.draw(self.rng)
            lab, nz = warp_affine(self.engine, label, flip, rot, sc, INTER_NEAREST, out=out_label, count_nonzero=True)
            if not has_object or 0 < nz < total:          # still_has_object, custom_transforms.py:66-70
                break
        img, _ = warp_affine(self.engine, frame, flip, rot, sc, INTER_CUBIC, out=out_image)
        return img, lab, dict(flip=flip, rot=rot, sc=sc, tries=tries)

    def batch(self, frame, label, batch_size):
        """`EpochSampler` batches the same first frame `batch_size` times (helper_func.py:521-545), each with
        its own random transform."""
        dev = frame.device
        images = torch.empty(batch_size, *frame.shape, device=dev)
        labels = torch.empty(batch_size, *label.shape, device=dev)
        has_object = self.has_object(label)                   # one host read for the batch, not two per sample
        params = [self.sample(frame, label, images[b], labels[b], has_object)[2] for b in range(batch_size)]
        return images, labels, params
