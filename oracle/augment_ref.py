"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's training-time input pipeline.

Reference: dataloaders/davis_2016.py:86-108 (``make_img_gt_pair``: BGR uint8 -> float32 minus mean, label / max)
and dataloaders/custom_transforms.py:21-52,87-121 (``RandomHorizontalFlip`` -> ``ScaleNRotate`` -> ``ToTensor``).
The geometry lives in a third-party dependency that is ABSENT here (OpenCV, no pinned version; README.md:21 names none):
``cv2.getRotationMatrix2D`` + ``cv2.warpAffine(INTER_CUBIC | INTER_NEAREST, BORDER_CONSTANT 0)``.  This file restates
OpenCV's published algorithm (modules/imgproc/src/imgwarp.cpp: WarpAffineInvoker, remapBicubic / remapNearest,
interpolateCubic with A = -0.75, 5-bit interpolation tables, 10-bit fixed-point coordinates) operation by operation in
float32 / float64 so that results can be compared bit for bit.  **Parity unpinned**: cv2 cannot be imported in this
container, so the restatement itself has not been checked against OpenCV outputs -- only its invariants (identity,
pure flips, integer translations, nearest vs. cubic on 0/1 masks) are pinned in tests/test_augment.py.
"""
import math

import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
AB_BITS = 10
AB_SCALE = 1 << AB_BITS
MEANVAL = (104.00699, 116.66877, 122.67892)          # train_online.py:36 / davis_2016.py default


def make_img_gt_pair(img_bgr_u8, label_u8, meanval=MEANVAL):
    """davis_2016.py:99-106"""
    img = np.array(img_bgr_u8, dtype=np.float32)
    img = np.subtract(img, np.array(meanval, dtype=np.float32))
    if label_u8 is None:
        gt = np.zeros(img.shape[:-1], dtype=np.uint8)
    else:
        gt = np.array(label_u8, dtype=np.float32)
        # numpy 1.x semantics (the reference's era): float32 array / python-or-float64 scalar stays float32
        gt = gt / np.float32(np.max([gt.max(), 1e-8]))
    return img, gt


def get_rotation_matrix_2d(center, angle, scale):
    """cv::getRotationMatrix2D (double precision)"""
    a = angle * math.pi / 180.0
    alpha, beta = math.cos(a) * scale, math.sin(a) * scale
    return np.array([[alpha, beta, (1 - alpha) * center[0] - beta * center[1]],
                     [-beta, alpha, beta * center[0] + (1 - alpha) * center[1]]], dtype=np.float64)


def invert_affine(M):
    """the in-place inversion cv::warpAffine applies when WARP_INVERSE_MAP is not set"""
    M = np.array(M, dtype=np.float64).copy().reshape(6)
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11; M[1] *= -D; M[3] *= -D; M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2] = b1; M[5] = b2
    return M


def _cv_round(v):
    return int(np.rint(v))            # lrint: round half to even


def _cubic_coeffs(frac_index):
    x = np.float32(frac_index) * np.float32(1.0 / INTER_TAB_SIZE)
    A = np.float32(-0.75)
    one, two, three, four, five, eight = (np.float32(v) for v in (1, 2, 3, 4, 5, 8))
    c0 = ((A * (x + one) - five * A) * (x + one) + eight * A) * (x + one) - four * A
    c1 = ((A + two) * x - (A + three)) * x * x + one
    c2 = ((A + two) * (one - x) - (A + three)) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    return np.array([c0, c1, c2, c3], dtype=np.float32)


def warp_affine(src, M, cubic):
    """cv2.warpAffine(src, M, (w, h), flags=INTER_CUBIC if cubic else INTER_NEAREST), BORDER_CONSTANT, borderValue 0.
    src: float32 [H,W] or [H,W,C]; M: 2x3 forward matrix (as getRotationMatrix2D returns)."""
    src = np.asarray(src, dtype=np.float32)
    squeeze = src.ndim == 2
    s = src[:, :, None] if squeeze else src
    h, w, cn = s.shape
    Mi = invert_affine(M)
    adelta = [_cv_round(Mi[0] * x * AB_SCALE) for x in range(w)]
    bdelta = [_cv_round(Mi[3] * x * AB_SCALE) for x in range(w)]
    round_delta = AB_SCALE // INTER_TAB_SIZE // 2 if cubic else AB_SCALE // 2
    dst = np.zeros_like(s)
    tabs = [_cubic_coeffs(i) for i in range(INTER_TAB_SIZE)]
    for y in range(h):
        X0 = _cv_round((Mi[1] * y + Mi[2]) * AB_SCALE) + round_delta
        Y0 = _cv_round((Mi[4] * y + Mi[5]) * AB_SCALE) + round_delta
        for x in range(w):
            if not cubic:
                X = int(np.clip((X0 + adelta[x]) >> AB_BITS, -32768, 32767))
                Y = int(np.clip((Y0 + bdelta[x]) >> AB_BITS, -32768, 32767))
                if 0 <= X < w and 0 <= Y < h:
                    dst[y, x] = s[Y, X]
                continue
            X = (X0 + adelta[x]) >> (AB_BITS - INTER_BITS)
            Y = (Y0 + bdelta[x]) >> (AB_BITS - INTER_BITS)
            sx = int(np.clip(X >> INTER_BITS, -32768, 32767)) - 1
            sy = int(np.clip(Y >> INTER_BITS, -32768, 32767)) - 1
            tx, ty = tabs[X & (INTER_TAB_SIZE - 1)], tabs[Y & (INTER_TAB_SIZE - 1)]
            wt = np.array([[ty[k1] * tx[k2] for k2 in range(4)] for k1 in range(4)], dtype=np.float32)
            if 0 <= sx < max(w - 3, 0) and 0 <= sy < max(h - 3, 0):
                for k in range(cn):
                    S = s[sy:sy + 4, sx:sx + 4, k]
                    acc = ((S[0, 0] * wt[0, 0] + S[0, 1] * wt[0, 1]) + S[0, 2] * wt[0, 2]) + S[0, 3] * wt[0, 3]
                    for r in range(1, 4):
                        acc = acc + (((S[r, 0] * wt[r, 0] + S[r, 1] * wt[r, 1]) + S[r, 2] * wt[r, 2]) + S[r, 3] * wt[r, 3])
                    dst[y, x, k] = acc
            elif sx >= w or sx + 4 <= 0 or sy >= h or sy + 4 <= 0:
                pass                                     # constant border value 0
            else:
                for k in range(cn):
                    acc = np.float32(0)
                    for i in range(4):
                        yi = sy + i
                        if not 0 <= yi < h:
                            continue
                        for j in range(4):
                            xj = sx + j
                            if 0 <= xj < w:
                                acc = acc + s[yi, xj, k] * wt[i, j]
                    dst[y, x, k] = acc
    return dst[:, :, 0] if squeeze else dst


def augment(img_bgr_u8, label_u8, flip, rot, sc, meanval=MEANVAL):
    """make_img_gt_pair -> [RandomHorizontalFlip] -> [ScaleNRotate] -> ToTensor with explicit parameters
    (flip: bool, rot/sc: None = no warp).  Returns (image float32 [3,H,W], gt float32 [1,H,W])."""
    img, gt = make_img_gt_pair(img_bgr_u8, label_u8, meanval)
    gt = np.asarray(gt, dtype=np.float32)
    if flip:
        img, gt = img[:, ::-1].copy(), gt[:, ::-1].copy()
    if rot is not None:
        h, w = img.shape[:2]
        M = get_rotation_matrix_2d((w / 2, h / 2), rot, sc)
        img = warp_affine(img, M, cubic=not bool(((img == 0) | (img == 1)).all()))
        gt = warp_affine(gt, M, cubic=not bool(((gt == 0) | (gt == 1)).all()))
    return img.transpose(2, 0, 1).copy(), gt[None].copy()
