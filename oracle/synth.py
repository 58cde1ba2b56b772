"""ORACLE support (test infrastructure only): seeded synthetic frames, masks and weights.

No DAVIS data, parent checkpoint or VGG weights exist offline, so every parity test and the
benchmark run on synthetic inputs.  Everything here is generated with
numpy's PCG64 so the same seed gives the same bytes in the build container and on the GPU box.

Weight recipe: trunk / side_prep convs ~ N(0, sqrt(2 / (9*Cout))) (the rule the reference uses
for its VGG shell, vgg_osvos.py:171-173), biases 0, deconvs bilinear (osvos_layers.py:72-85).
The default reference init N(0, 0.001) (vgg_osvos.py:79) collapses logits to ~1e-10, which
would make parity vacuous; the heads are therefore *calibrated* on the synthetic frame so each
logit map has std ~3 and mean ~-1 (calibrate_heads).
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from .torch_ref import N_SCALES, bilinear_filter, state_dict_spec


def make_frame(n, h, w, seed=0):
    """Mean-subtracted-BGR-like frame [n,3,h,w] float32: low-frequency field * 50 + noise * 8."""
    rng = np.random.default_rng(1000 + seed)
    gh, gw = max(2, h // 16 + 2), max(2, w // 16 + 2)
    coarse = rng.standard_normal((n, 3, gh, gw))
    ys = np.linspace(0, gh - 1, h)
    xs = np.linspace(0, gw - 1, w)
    y0 = np.clip(np.floor(ys).astype(int), 0, gh - 2)
    x0 = np.clip(np.floor(xs).astype(int), 0, gw - 2)
    fy = (ys - y0)[None, None, :, None]
    fx = (xs - x0)[None, None, None, :]
    c00 = coarse[:, :, y0][:, :, :, x0]
    c01 = coarse[:, :, y0][:, :, :, x0 + 1]
    c10 = coarse[:, :, y0 + 1][:, :, :, x0]
    c11 = coarse[:, :, y0 + 1][:, :, :, x0 + 1]
    low = (c00 * (1 - fy) * (1 - fx) + c01 * (1 - fy) * fx + c10 * fy * (1 - fx) + c11 * fy * fx)
    out = low * 50.0 + rng.standard_normal((n, 3, h, w)) * 8.0
    return out.astype(np.float32)


def make_mask(n, h, w, seed=0):
    """Binary mask [n,1,h,w] float32 in {0,1}: one filled ellipse per image, ~20 % foreground."""
    rng = np.random.default_rng(2000 + seed)
    yy, xx = np.mgrid[0:h, 0:w]
    out = np.zeros((n, 1, h, w), np.float32)
    for i in range(n):
        cy = h * (0.35 + 0.3 * rng.random())
        cx = w * (0.35 + 0.3 * rng.random())
        ry = h * (0.2 + 0.1 * rng.random())
        rx = w * (0.2 + 0.1 * rng.random())
        out[i, 0] = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0).astype(np.float32)
    return out


def trainable_frame(n, h, w, seed=0):
    """(frame [n,3,h,w], mask [n,1,h,w]) in which the mask is LEARNABLE from the frame: the ellipse of make_mask carries its own colour
    and a finer texture on top of a damped low-frequency background -- the stand-in for a DAVIS object when a net has to be TRAINED on
    synthetic data (tests/trained_fixture.py: the parity fixture with realistic logit margins)."""
    rng = np.random.default_rng(5000 + seed)
    m = make_mask(n, h, w, seed)
    bg = make_frame(n, h, w, seed) * 0.6
    colour = np.array([55.0, -45.0, 35.0], np.float32).reshape(1, 3, 1, 1) * (0.8 + 0.4 * rng.random((n, 1, 1, 1))).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    tex = (12.0 * np.sin(yy / 2.5 + seed) * np.cos(xx / 3.5)).astype(np.float32)[None, None]
    x = bg * (1.0 - 0.5 * m) + m * (colour + tex) + rng.standard_normal((n, 3, h, w)).astype(np.float32) * 4.0
    return x.astype(np.float32), m


def make_weights(seed=1, bias_std=0.0):
    """OrderedDict of float32 arrays in the reference's state_dict order."""
    rng = np.random.default_rng(3000 + seed)
    out = OrderedDict()
    for key, shape in state_dict_spec():
        if key.startswith("upscale"):
            k = shape[-1]
            f = bilinear_filter(k).astype(np.float32)
            w = np.zeros(shape, np.float32)
            for c in range(shape[0]):
                w[c, c] = f
            out[key] = w
        elif key.endswith("weight"):
            cout, _, kh, kw = shape
            std = np.sqrt(2.0 / (kh * kw * cout))
            out[key] = (rng.standard_normal(shape) * std).astype(np.float32)
        else:
            out[key] = (rng.standard_normal(shape) * bias_std).astype(np.float32)
    return out


def calibrate_heads(weights, forward_fn, x, target_std=3.0, target_mean=-1.0):
    """Rescale score_dsn / fuse so every logit map of forward_fn(weights, x) has the target
    statistics.  forward_fn: (weights dict of np arrays, x np array) -> list of 5 np arrays."""
    w = OrderedDict((k, v.copy()) for k, v in weights.items())
    for i in range(N_SCALES):
        w["score_dsn.%d.bias" % i][:] = 0
    w["fuse.bias"][:] = 0
    outs = forward_fn(w, x)
    for i in range(N_SCALES):
        o = np.asarray(outs[i], np.float64)
        s = target_std / max(o.std(), 1e-30)
        w["score_dsn.%d.weight" % i] *= np.float32(s)
        w["score_dsn.%d.bias" % i][:] = np.float32(target_mean - o.mean() * s)
    o = np.asarray(outs[4], np.float64)
    s = target_std / max(o.std(), 1e-30)
    w["fuse.weight"] *= np.float32(s)
    w["fuse.bias"][:] = np.float32(target_mean - o.mean() * s)
    return w


def torch_forward_fn(dtype="float32"):
    """forward_fn for calibrate_heads backed by the torch-CPU oracle."""
    import torch
    from . import torch_ref
    td = getattr(torch, dtype)

    def fn(weights, x):
        with torch.no_grad():
            p = {k: torch.from_numpy(np.ascontiguousarray(v)).to(td) for k, v in weights.items()}
            outs = torch_ref.forward(p, torch.from_numpy(x).to(td))
        return [o.numpy() for o in outs]
    return fn


def calibrated_problem(n, h, w, seed=0, wseed=1):
    """(weights, frame, mask) with heads calibrated on that frame via the torch-CPU oracle."""
    x = make_frame(n, h, w, seed)
    m = make_mask(n, h, w, seed)
    wts = calibrate_heads(make_weights(wseed), torch_forward_fn(), x)
    return wts, x, m


# ---- synthetic stand-ins for the two pretrained-weight files the reference can start from (both absent offline) ----
# vgg_osvos.py:92-109 loads models/vgg_pytorch.pth (a torchvision VGG-16 state_dict) through its VGG shell;
# vgg_osvos.py:110-125 loads models/vgg_caffe.mat (Caffe export: weights[0][i] stored [kw,kh,cin,cout], biases[0][i] [cout,1]).
# The generators below write files of exactly those layouts from a seed, so the reference (tests/golden/make_golden.py) and
# the drop-in (tests/test_host_utils_cpu.py) can load the SAME bytes.
VGG16_FEATURE_CONVS = [0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28]      # conv indices inside torchvision's vgg16.features


def vgg_trunk_arrays(seed):
    """13 (weight [cout,cin,3,3], bias [cout]) float32 pairs, He-scaled, seeded."""
    from .torch_ref import STAGE_CHANNELS, STAGE_IN
    rng = np.random.default_rng(4000 + seed)
    out = []
    for si, chans in enumerate(STAGE_CHANNELS):
        cin = STAGE_IN[si]
        for cout in chans:
            w = (rng.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (9 * cout))).astype(np.float32)
            b = (rng.standard_normal((cout,)) * 0.1).astype(np.float32)
            out.append((w, b))
            cin = cout
    return out


def write_vgg_pytorch_pth(path, seed):
    """torchvision-layout VGG-16 state_dict: features.<i>.weight/bias for the 13 convs + the three classifier Linear
    layers (zeros; the reference's strict load_state_dict needs them present with the right shapes, ~550 MB on disk)."""
    import torch
    sd = OrderedDict()
    for fi, (w, b) in zip(VGG16_FEATURE_CONVS, vgg_trunk_arrays(seed)):
        sd["features.%d.weight" % fi] = torch.from_numpy(w)
        sd["features.%d.bias" % fi] = torch.from_numpy(b)
    for ci, (o, i) in zip((0, 3, 6), ((4096, 512 * 7 * 7), (4096, 4096), (1000, 4096))):
        sd["classifier.%d.weight" % ci] = torch.zeros(o, i)
        sd["classifier.%d.bias" % ci] = torch.zeros(o)
    torch.save(sd, path)


def write_vgg_caffe_mat(path, seed):
    """Caffe-export layout: 'weights' / 'biases' are 1 x 13 object arrays; weights[0][i] is the OIHW tensor transposed
    (all axes reversed -> [kw, kh, cin, cout]), biases[0][i] is [cout, 1]."""
    import scipy.io
    arrs = vgg_trunk_arrays(seed)
    ws = np.empty((1, len(arrs)), dtype=object)
    bs = np.empty((1, len(arrs)), dtype=object)
    for i, (w, b) in enumerate(arrs):
        ws[0, i] = np.ascontiguousarray(w.transpose())
        bs[0, i] = b[:, None].copy()
    scipy.io.savemat(path, {"weights": ws, "biases": bs})
