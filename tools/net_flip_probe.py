#!/usr/bin/env python3
"""ReLU decisions of the whole trunk per precision (checker-side probe, round 6): run the trained-like fixture's held-out frame through the network in every
fp32-family precision, read the 13 saved activations (osvos_net_ws_query) and count, per layer, the elements whose sign differs from the float64 oracle's
(vgg_osvos.py:136-145 restated in oracle/torch_ref.py).  A flipped element changes which gradient paths exist: this is what sets the gradient error floor."""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import trained_fixture as tf  # noqa: E402
from oracle import synth, torch_ref  # noqa: E402
from osvos_pytorch_amd import _lib  # noqa: E402

wts, frames, _ = tf.train_like()
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (240, 427)
x, m = synth.trainable_frame(1, H, W, seed=tf.RECIPE["frame_seed"] + 98)
p = {k: torch.as_tensor(v).double() for k, v in wts.items()}
truth, pre = [], []
cur = torch.from_numpy(x).double()
names = torch_ref.trunk_conv_names()
for si in range(5):
    if si > 0:
        cur = F.max_pool2d(cur, 2, 2, ceil_mode=True)
    for nm in names[si]:
        z = F.conv2d(cur, p[nm + ".weight"], p[nm + ".bias"], padding=1)
        pre.append(z)
        cur = F.relu(z)
        truth.append(cur)
lib = _lib.lib()
print("%-9s" % "layer" + "".join("%22s" % q for q in ("fp32", "fp32x3", "fp32x2", "fp32h2")) + "     (flips vs float64 | rel-L2 of the activation)")
rows = {}
for prec in ("fp32", "fp32x3", "fp32x2", "fp32h2"):
    net = tf.build(wts, prec)
    xg = torch.from_numpy(x).cuda().requires_grad_()
    outs = net.forward(xg)
    ws = outs[0].grad_fn.saved_tensors[0]
    for l in range(13):
        off, el, ch, hh, ww = C.c_size_t(), C.c_size_t(), C.c_int(), C.c_int(), C.c_int()
        _lib.check(lib.osvos_net_ws_query(1, H, W, net._runtime.dtype, l, C.byref(off), C.byref(el), C.byref(ch), C.byref(hh), C.byref(ww)))
        act = ws[off.value:off.value + 4 * el.value].view(torch.float32).view(1, hh.value, ww.value, ch.value).permute(0, 3, 1, 2).cpu().double()
        flips = (act > 0) != (truth[l] > 0)
        # how close to zero (in units of the layer's pre-activation std) were the flipped ones in truth?
        zs = float(pre[l].std())
        worst = float((pre[l].abs()[flips] / zs).max()) if flips.any() else 0.0
        rows.setdefault(l, []).append("%6d %8.1e %6.0e" % (int(flips.sum()), float((act - truth[l]).norm() / truth[l].norm()), worst))
for l in range(13):
    print("%-9s" % [n for st in names for n in st][l] + "".join("%22s" % r for r in rows[l]))
