"""Drop-in for the reference ``layers/osvos_layers.py`` (same names, arguments and error
behaviour), with the loss computed by the HIP kernel behind ``osvos_cbce``.

Reference anchors: logit/sigmoid_np :11-16, class_balanced_cross_entropy_loss :19-48,
center_crop :51-56, upsample_filt :59-67, interp_surgery :72-85."""
from __future__ import division

import numpy as np
import torch

from ..autograd import CBCELossFunction


def logit(x):
    """numpy log-odds with the reference's 1e-8 guards (osvos_layers.py:11-12)."""
    return np.log(x / (1 - x + 1e-08) + 1e-08)


def sigmoid_np(x):
    return 1 / (1 + np.exp(-x))


def class_balanced_cross_entropy_loss(output, label, size_average=True, batch_average=True):
    """Class-balanced BCE with logits.  Positives are ``label >= 0.5``; the two class weights are
    counted over the whole tensor; ``size_average`` divides by numel, else ``batch_average`` by N.
    Returns a 0-dim CUDA tensor that supports ``.item()``, ``/=`` and ``.backward()``."""
    mode = 0 if size_average else (1 if batch_average else 2)
    return CBCELossFunction.apply(output, label, mode)


def center_crop(x, height, width):
    """Keep the central ``height x width`` window: floor(excess/2) rows/cols are dropped at the
    top/left and ceil(excess/2) at the bottom/right -- the pixels the reference keeps with its
    negative F.pad (osvos_layers.py:52-56).  (The network itself fuses this into the head kernel.)"""
    eh, ew = int(x.size(2)) - int(height), int(x.size(3)) - int(width)
    t, l = eh // 2, ew // 2
    return x[:, :, t:t + int(height), l:l + int(width)].clone()


def upsample_filt(size):
    """k x k bilinear interpolation kernel of the transposed convs."""
    factor = (size + 1) // 2
    center = factor - 1 if size % 2 == 1 else factor - 0.5
    ramp = 1 - np.abs(np.arange(size) - center) / factor
    return ramp[:, None] * ramp[None, :]


def interp_surgery(lay):
    """Write the bilinear filter on the channel diagonal of a transposed-conv weight.  Raises
    ValueError for non-square filters or in != out channels, like the reference."""
    m, k, h, w = lay.weight.data.size()
    if m != k:
        print('input + output channels need to be the same')
        raise ValueError
    if h != w:
        print('filters need to be square')
        raise ValueError
    filt = torch.from_numpy(upsample_filt(h)).to(lay.weight.dtype)
    with torch.no_grad():
        idx = torch.arange(m)
        lay.weight.data[idx, idx] = filt.to(lay.weight.device)
    return lay.weight.data
