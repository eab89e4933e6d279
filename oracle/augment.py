"""numpy restatement of the reference's first-frame augmentation (TEST INFRASTRUCTURE ONLY).

Reference: `src/data/custom_transforms.py`
  * `RandomHorizontalFlip.__call__` (`:196-213`): `cv2.flip(tmp, flipCode=1)` with probability 0.5;
  * `RandomScaleNRotate` (`:9-92`): rot ~ U(-30, 30), scale ~ U(.75, 1.25) drawn with `random.random()`
    (`:26-33`), `cv2.getRotationMatrix2D((w/2, h/2), rot, sc)` + `cv2.warpAffine(tmp, M, (w, h), flags=...)`
    with INTER_NEAREST for the label and INTER_CUBIC for the frame (`:41-51`), redrawn until the warped label
    still holds every label value (`:53-78`);
  * applied in that order by `data_loaders` (`src/util/helper_func.py:255-261`).

OpenCV (opencv-python==4.1.0.25, `requirements.txt:63`) is a third-party dependency that is neither vendored in
the reference nor installed in this image, so **parity of this file is unpinned by a reference run**: it restates
the published algorithm of `modules/imgproc/src/imgwarp.cpp` (4.1): `getRotationMatrix2D` (float centre, double
matrix), `warpAffine` (matrix inversion; AB_BITS = 10 fixed-point source coordinates from the adelta/bdelta/X0/Y0
integer tables, round_delta = 512 for nearest and 16 for the 1/32-pixel interpolation grid),
`remapNearest` / `remapBicubic` for CV_32F with BORDER_CONSTANT 0 (interior pixels: four 4-term row expressions;
border pixels: constant if the centre tap is outside, else tap-by-tap accumulation of the inside taps) and the
bicubic table `interpolateCubic` (A = -0.75, float) whose 2-D weights are products cy[k1]*cx[k2].
"""
import math
import random

import numpy as np

AB_BITS = 10
AB_SCALE = 1 << AB_BITS
INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS


def get_rotation_matrix_2d(center, angle_deg, scale):
    cx, cy = float(np.float32(center[0])), float(np.float32(center[1]))
    ang = angle_deg * (math.pi / 180.0)
    alpha, beta = math.cos(ang) * scale, math.sin(ang) * scale
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy],
                     [-beta, alpha, beta * cx + (1 - alpha) * cy]], dtype=np.float64)


def cubic_table():
    """[32][4] float32 coefficients of `interpolateCubic(i/32)`; every operation rounded to float like the C code."""
    f = np.float32
    A = f(-0.75)
    tab = np.zeros((INTER_TAB_SIZE, 4), np.float32)
    for i in range(INTER_TAB_SIZE):
        x = f(i) * f(1.0 / 32)
        x1 = f(x + f(1))
        c0 = f(f(f(f(f(f(A * x1) - f(f(5) * A)) * x1) + f(f(8) * A)) * x1) - f(f(4) * A))
        c1 = f(f(f(f(f(f(A + f(2)) * x) - f(A + f(3))) * x) * x) + f(1))
        xm = f(f(1) - x)
        c2 = f(f(f(f(f(f(A + f(2)) * xm) - f(A + f(3))) * xm) * xm) + f(1))
        c3 = f(f(f(f(1) - c0) - c1) - c2)
        tab[i] = (c0, c1, c2, c3)
    return tab


def _tables(M, H, W, nearest):
    M = np.array(M, dtype=np.float64).reshape(6).copy()
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    rd = AB_SCALE // 2 if nearest else AB_SCALE // INTER_TAB_SIZE // 2
    xs, ys = np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64)
    adelta = np.rint(M[0] * xs * AB_SCALE).astype(np.int64)
    bdelta = np.rint(M[3] * xs * AB_SCALE).astype(np.int64)
    X0 = np.rint((M[1] * ys + M[2]) * AB_SCALE).astype(np.int64) + rd
    Y0 = np.rint((M[4] * ys + M[5]) * AB_SCALE).astype(np.int64) + rd
    return adelta, bdelta, X0, Y0


def warp_affine(src, M, interp):
    """src: (H, W) or (H, W, C) float32; `cv2.warpAffine(src, M, (W, H), flags=interp)`; interp 'nearest'|'cubic'."""
    src = np.asarray(src, np.float32)
    squeeze = src.ndim == 2
    if squeeze:
        src = src[:, :, None]
    H, W, C = src.shape
    adelta, bdelta, X0, Y0 = _tables(M, H, W, interp == 'nearest')
    Xf = X0[:, None] + adelta[None, :]
    Yf = Y0[:, None] + bdelta[None, :]
    out = np.zeros_like(src)
    if interp == 'nearest':
        sx, sy = Xf >> AB_BITS, Yf >> AB_BITS
        ok = (sx >= 0) & (sx < W) & (sy >= 0) & (sy < H)
        out[ok] = src[sy[ok], sx[ok]]
    elif interp == 'cubic':
        X, Y = Xf >> (AB_BITS - INTER_BITS), Yf >> (AB_BITS - INTER_BITS)
        sx, sy = (X >> INTER_BITS) - 1, (Y >> INTER_BITS) - 1
        tab = cubic_table()
        cx, cy = tab[X & (INTER_TAB_SIZE - 1)], tab[Y & (INTER_TAB_SIZE - 1)]      # (H, W, 4)
        interior = (sx >= 0) & (sx < max(W - 3, 0)) & (sy >= 0) & (sy < max(H - 3, 0))
        centre_out = (sx + 1 < 0) | (sx + 1 >= W) | (sy + 1 < 0) | (sy + 1 >= H)
        border = ~interior & ~centre_out
        f = np.float32
        # interior: row expressions, rows accumulated
        iy, ix = np.nonzero(interior)
        if iy.size:
            bx, by = sx[iy, ix], sy[iy, ix]
            total = None
            for i in range(4):
                row = None
                for j in range(4):
                    w = (cy[iy, ix, i] * cx[iy, ix, j]).astype(f)[:, None]
                    t = (src[by + i, bx + j] * w).astype(f)
                    row = t if row is None else (row + t).astype(f)
                total = row if total is None else (total + row).astype(f)
            out[iy, ix] = total
        iy, ix = np.nonzero(border)
        if iy.size:
            bx, by = sx[iy, ix], sy[iy, ix]
            total = np.zeros((iy.size, C), f)
            for i in range(4):
                for j in range(4):
                    yy, xx = by + i, bx + j
                    ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
                    w = (cy[iy, ix, i] * cx[iy, ix, j]).astype(f)[:, None]
                    v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
                    total = np.where(ok[:, None], (total + (v * w).astype(f)).astype(f), total)
            out[iy, ix] = total
    else:
        raise NotImplementedError(interp)
    return out[:, :, 0] if squeeze else out


def rot_and_sc(tmp, rot, sc, label):
    """`RandomScaleNRotate._rot_and_sc`, custom_transforms.py:41-51."""
    h, w = tmp.shape[:2]
    M = get_rotation_matrix_2d((w / 2, h / 2), rot, sc)
    return warp_affine(tmp, M, 'nearest' if label else 'cubic')


def random_flip_scale_rotate(image, gt, rots=(-30, 30), scales=(.75, 1.25), rng=random):
    """The train transform of `data_loaders` (helper_func.py:255-261) on one sample: returns
    (image, gt, dict(flip, rot, sc, tries)).  image (H,W,3) float32 in [0,1], gt (H,W) float32."""
    flip = rng.random() < 0.5                                         # custom_transforms.py:200
    if flip:
        image, gt = image[:, ::-1].copy(), gt[:, ::-1].copy()         # cv2.flip(.., 1)
    num_labels = len(np.unique(gt))
    tries = 0
    while True:
        tries += 1
        rot = (rots[1] - rots[0]) * rng.random() - (rots[1] - rots[0]) / 2            # :28-29
        sc = (scales[1] - scales[0]) * rng.random() - (scales[1] - scales[0]) / 2 + 1  # :31-32
        aug = rot_and_sc(gt, rot, sc, True)
        if not num_labels > 1 or len(np.unique(aug)) == num_labels:
            break
    return rot_and_sc(image, rot, sc, False), aug, dict(flip=flip, rot=rot, sc=sc, tries=tries)
