"""Training-time input pipeline on the device.

The reference builds ``Compose([RandomHorizontalFlip(), ScaleNRotate(rots=(-30, 30), scales=(.75, 1.25)), ToTensor()])``
over numpy arrays on DataLoader worker processes (train_online.py:92-97, custom_transforms.py) with OpenCV doing the
bicubic warp of a float32 frame -- several ms per frame per worker, which cannot feed a GPU that trains at > 100 frames/s.
``DeviceAugment`` takes the raw uint8 BGR frame (what ``cv2.imread`` returns) and the uint8 label, draws the same random
parameters in the same order from Python's ``random`` module, and produces the ``{'image': [3,H,W], 'gt': [1,H,W]}``
float32 CUDA tensors of the reference in one kernel launch (``osvos_augment_frame``).
"""
import ctypes as C
import math
import random

import torch

from ._lib import check, lib

MEANVAL = (104.00699, 116.66877, 122.67892)          # davis_2016.py default / train_online.py:36


def rotation_matrix_inverse(w, h, rot, sc):
    """getRotationMatrix2D((w/2, h/2), rot, sc) followed by the inversion cv::warpAffine applies: the 6 doubles of dst -> src."""
    a = rot * math.pi / 180.0
    alpha, beta = math.cos(a) * sc, math.sin(a) * sc
    cx, cy = w / 2, h / 2
    m = [alpha, beta, (1 - alpha) * cx - beta * cy, -beta, alpha, beta * cx + (1 - alpha) * cy]
    d = m[0] * m[4] - m[1] * m[3]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[4] * d, m[0] * d
    m[0] = a11; m[1] *= -d; m[3] *= -d; m[4] = a22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2] = b1; m[5] = b2
    return m


def augment_frame(img_bgr, label, flip=False, rot=None, sc=None, meanval=MEANVAL):
    """img_bgr: uint8 CUDA tensor [H,W,3]; label: uint8 CUDA tensor [H,W] or None; rot (degrees) / sc: None = no warp.
    Returns (image float32 [3,H,W], gt float32 [1,H,W])."""
    if not img_bgr.is_cuda or (label is not None and not label.is_cuda):
        raise RuntimeError("osvos_pytorch_amd.augment needs CUDA (ROCm) tensors; there is no CPU fallback")
    if img_bgr.dtype != torch.uint8 or img_bgr.dim() != 3 or img_bgr.shape[2] != 3:
        raise ValueError("img_bgr must be a uint8 [H,W,3] tensor")
    img_bgr = img_bgr.contiguous()
    h, w = int(img_bgr.shape[0]), int(img_bgr.shape[1])
    if label is not None:
        if label.dtype != torch.uint8 or tuple(label.shape) != (h, w):
            raise ValueError("label must be a uint8 [H,W] tensor matching the frame")
        label = label.contiguous()
    out_img = torch.empty((3, h, w), device=img_bgr.device, dtype=torch.float32)
    out_gt = torch.empty((1, h, w), device=img_bgr.device, dtype=torch.float32)
    scratch = torch.empty(2, device=img_bgr.device, dtype=torch.int32)
    mean = (C.c_float * 3)(*meanval)
    minv = None
    if rot is not None:
        minv = (C.c_double * 6)(*rotation_matrix_inverse(w, h, float(rot), float(sc)))
    vp = C.c_void_p
    check(lib().osvos_augment_frame(vp(img_bgr.data_ptr()), vp(label.data_ptr()) if label is not None else None, mean, int(bool(flip)), minv,
                                    vp(out_img.data_ptr()), vp(out_gt.data_ptr()), vp(scratch.data_ptr()), h, w,
                                    vp(torch.cuda.current_stream().cuda_stream)), "augment_frame")
    return out_img, out_gt


class DeviceAugment(object):
    """Drop-in for the reference's training transform chain.  Random draws, in the reference's order:
    ``random.random() < 0.5`` (flip, custom_transforms.py:92), then rotation and scale (custom_transforms.py:25-29)."""

    def __init__(self, rots=(-30, 30), scales=(.75, 1.25), flip=True, meanval=MEANVAL):
        if not (isinstance(rots, tuple) and isinstance(scales, tuple)):
            raise ValueError("DeviceAugment implements the continuous (tuple) ranges the reference's scripts use")
        self.rots, self.scales, self.flip, self.meanval = rots, scales, flip, meanval

    def draw(self):
        flip = self.flip and random.random() < 0.5
        rot = (self.rots[1] - self.rots[0]) * random.random() - (self.rots[1] - self.rots[0]) / 2
        sc = (self.scales[1] - self.scales[0]) * random.random() - (self.scales[1] - self.scales[0]) / 2 + 1
        return flip, rot, sc

    def __call__(self, img_bgr, label):
        flip, rot, sc = self.draw()
        image, gt = augment_frame(img_bgr, label, flip, rot, sc, self.meanval)
        return {'image': image, 'gt': gt}
