"""Drop-in for the reference ``layers/osvos_layers.py`` (same names, arguments and error
behaviour), with the loss computed by the HIP kernel behind ``osvos_cbce``.

Reference anchors: logit/sigmoid_np :11-16, class_balanced_cross_entropy_loss :19-48,
center_crop :51-56, upsample_filt :59-67, interp_surgery :72-85."""
from __future__ import division

import numpy as np
import torch

from ..autograd import CBCELossFunction, cbce_step, cbce_step_multi


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


def class_balanced_cross_entropy_loss_step(output, label, size_average=True, batch_average=True, grad_scale=1.0, running=None, per_image=False,
                                           counts=None):
    """The loss as ONE micro-batch of the training loops uses it (train_online.py:127-141): ``(loss, grad)`` with ``grad`` already
    multiplied by the upstream gradient ``grad_scale`` (1 / nAveGrad ...) and ``running += loss`` done on the device; hand ``grad`` to
    ``torch.autograd.backward([output], [grad])``.  An extension next to the reference's function above, not a replacement.
    ``per_image=True``: the N images of ``output`` are N reference micro-batches of one image each (own class weights per image, the N losses
    summed) -- a whole accumulation window in one call.  ``counts=(n_pos, n_total, n_images)``: the tensors are a shard of a global batch with
    these counts (``parallel.global_class_counts``)."""
    mode = 0 if size_average else (1 if batch_average else 2)
    return cbce_step(output, label, mode, grad_scale, running, per_image=per_image, counts=counts)


def class_balanced_cross_entropy_loss_step_multi(outputs, label, size_average=True, batch_average=True, grad_scales=None, running=None, per_image=False,
                                                 counts=None):
    """``class_balanced_cross_entropy_loss_step`` for all heads of a micro-batch in one call (the parent loop: train_parent.py:143-147):
    ``(losses, grads)`` with ``losses`` a float32 tensor of the plain per-head losses."""
    mode = 0 if size_average else (1 if batch_average else 2)
    return cbce_step_multi(list(outputs), label, mode, list(grad_scales) if grad_scales is not None else [1.0] * len(outputs), running,
                           per_image=per_image, counts=counts)


def center_crop(x, height, width):
    """Central ``height x width`` window of ``x`` with the reference's rounding (osvos_layers.py:51-56): with
    ``c = (size_in - target) / -2`` the left/top side moves by ``ceil(c)`` and the right/bottom side by ``floor(c)``.
    Negative amounts crop (input larger than the target: floor(excess/2) rows/cols dropped at the top/left,
    ceil(excess/2) at the bottom/right); positive amounts zero-pad (input smaller than the target), exactly like the
    reference's ``F.pad`` with mixed-sign pads.  Returns a new tensor.  (The network itself fuses the crop into the
    head kernel; this helper exists for callers of the reference's ``layers.osvos_layers`` API.)"""
    height, width = int(height), int(width)
    eh, ew = int(x.size(2)) - height, int(x.size(3)) - width
    top, left = -(eh // 2), -(ew // 2)                 # ceil(c): > 0 pads, < 0 crops
    out = x.new_zeros((x.size(0), x.size(1), height, width))
    sy, sx = max(0, -top), max(0, -left)               # first source row / column kept
    dy, dx = max(0, top), max(0, left)                 # where it lands in the result
    nh, nw = min(int(x.size(2)) - sy, height - dy), min(int(x.size(3)) - sx, width - dx)
    if nh > 0 and nw > 0:
        out[:, :, dy:dy + nh, dx:dx + nw] = x[:, :, sy:sy + nh, sx:sx + nw]
    return out


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
