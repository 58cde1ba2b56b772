#!/usr/bin/env python3
"""Pin the two host-adjacent restatements against the packages the reference really uses (VERDICT r02 "Missing #3").

This container has neither OpenCV nor scipy <= 1.1, so `oracle/augment_ref.py` (cv2.warpAffine / cv2.flip chain of the reference's
dataloaders/custom_transforms.py:21-52,87-121 + davis_2016.py:86-106) and the byte scaling of `scipy.misc.imsave`
(train_online.py:183-189) are restated from the published algorithms and marked "parity unpinned".  Run THIS script once on any box
that has them:

    pip install opencv-python "scipy<=1.1" pillow torch numpy          # (scipy.misc.imsave / bytescale were removed in scipy 1.2)
    python tools/make_cv2_goldens.py --reference /path/to/OSVOS-PyTorch

It drives the REAL reference classes (RandomHorizontalFlip, ScaleNRotate, ToTensor -- their random draws replaced by the values of each
case) on the seeded frames of tests/test_augment.py and writes tests/golden/augment_cv2.npz; and scipy.misc.bytescale / imsave on the
seeded logits of tests/test_gpu_ops.py::test_result_writer_bytes_png_and_jaccard -> tests/golden/bytescale.npz.  The tests
tests/test_augment.py::test_restatement_matches_the_cv2_goldens and tests/test_host_utils_cpu.py::test_bytescale_matches_the_scipy_golden
compare against those files and skip, with this instruction, while they are absent.  Commit the two .npz files with this script."""
import argparse
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

# (h, w, flip, rot, scale, soft label): the cases of tests/test_augment.py plus two at a DAVIS-like aspect
CASES = [(13, 17, False, None, None, False), (13, 17, True, None, None, False), (24, 31, False, 0.0, 1.0, False),
         (24, 31, True, 17.0, 1.1, False), (33, 40, False, -29.5, 0.77, False), (20, 27, True, 8.25, 1.24, True),
         (9, 6, False, 45.0, 0.8, False), (120, 214, True, -12.75, 0.93, False), (120, 214, False, 29.9, 1.249, False)]


def frame(h, w, seed, soft=False):
    """identical to tests/test_augment.py::_frame"""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if soft:
        lab = rng.integers(0, 256, (h, w), dtype=np.uint8)
    else:
        yy, xx = np.mgrid[0:h, 0:w]
        lab = ((((yy - h / 2) / (h / 3)) ** 2 + ((xx - w / 2) / (w / 4)) ** 2) < 1).astype(np.uint8) * 255
    return img, lab


def augment_goldens(ref):
    sys.path.insert(0, ref)
    import cv2  # noqa: F401  (the reference module needs it)
    from dataloaders import custom_transforms as tr
    meanval = (104.00699, 116.66877, 122.67892)                      # davis_2016.py:24
    out = {"cases": np.array([[h, w, int(f), np.nan if r is None else r, np.nan if s is None else s, int(soft)] for h, w, f, r, s, soft in CASES])}
    for ci, (h, w, flip, rot, sc, soft) in enumerate(CASES):
        img8, lab8 = frame(h, w, 7 + h, soft)
        # davis_2016.py:99-106 (make_img_gt_pair) on what cv2.imread would have returned
        img = np.subtract(np.array(img8, dtype=np.float32), np.array(meanval, dtype=np.float32))
        gt = np.array(lab8, dtype=np.float32)
        gt = gt / np.max([gt.max(), 1e-8])
        sample = {"image": img, "gt": gt}
        draws = []
        chain = []
        chain.append(tr.RandomHorizontalFlip())
        draws.append(0.25 if flip else 0.75)                          # random.random() < 0.5 flips
        if rot is not None:
            chain.append(tr.ScaleNRotate(rots=(-30, 30), scales=(.75, 1.25)))
            draws += [(rot + 30.0) / 60.0, (sc - 0.75) / 0.5]         # rot = 60 r - 30, sc = 0.5 r' - 0.25 + 1 (custom_transforms.py:25-29)
        chain.append(tr.ToTensor())
        it = iter(draws)
        real = tr.random.random
        tr.random.random = lambda: next(it)
        try:
            for t in chain:
                sample = t(sample)
        finally:
            tr.random.random = real
        out["image%d" % ci] = sample["image"].numpy()
        out["gt%d" % ci] = sample["gt"].numpy()
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "augment_cv2.npz"), **out)
    print("wrote tests/golden/augment_cv2.npz (%d cases, cv2 %s)" % (len(CASES), cv2.__version__))


def bytescale_goldens():
    import scipy
    import scipy.misc as sm
    import torch
    from PIL import Image
    g = torch.Generator().manual_seed(61)                              # tests/test_gpu_ops.py::test_result_writer_bytes_png_and_jaccard
    logits = (torch.randn(3, 1, 37, 53, generator=g) * 3 - 1)
    logits[2] = 0.25
    out = {"logits": logits.numpy()}
    for n in range(3):
        pred = np.squeeze(1 / (1 + np.exp(-logits[n].numpy().transpose(1, 2, 0))))       # train_online.py:184-186
        out["bytescale%d" % n] = sm.bytescale(pred)
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, "m.png")
            sm.imsave(p, pred)                                         # train_online.py:189
            out["imsave%d" % n] = np.array(Image.open(p))
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "bytescale.npz"), **out)
    print("wrote tests/golden/bytescale.npz (scipy %s)" % scipy.__version__)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference", help="checkout of kmaninis/OSVOS-PyTorch")
    ap.add_argument("--only", default="", choices=["", "augment", "bytescale"])
    a = ap.parse_args()
    if a.only in ("", "augment"):
        augment_goldens(a.reference)
    if a.only in ("", "bytescale"):
        bytescale_goldens()
