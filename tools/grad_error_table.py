#!/usr/bin/env python3
"""Per-tensor gradient error of the precisions against the float64 oracle (checker-side tool: trains the trained-like fixture on the GPU, runs the oracle on
the host): for every parameter tensor rel-L2(grad - truth) under fp32 (exact MFMA), fp32x3, fp32x3b2 (two-piece backward) and fp32x2, on the fixture's
held-out frame at 240x427.  Answers: does the two-piece BACKWARD move any tensor's gradient error beyond what the fp32 arithmetics already show?"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import trained_fixture as tf  # noqa: E402
from oracle import synth, torch_ref  # noqa: E402
from layers.osvos_layers import class_balanced_cross_entropy_loss as cbce  # noqa: E402

wts, frames, _ = tf.train_like()
x, m = synth.trainable_frame(1, 240, 427, seed=tf.RECIPE["frame_seed"] + 98)
p = torch_ref.as_leaf_params(wts, dtype=torch.float64)
xi = torch.from_numpy(x).double().requires_grad_()
outs = torch_ref.forward(p, xi)
losses = [torch_ref.cbce_loss(o, torch.from_numpy(m).double(), size_average=False) for o in outs]
(0.5 * sum(losses[:-1]) + losses[-1]).backward()
truth = {k: v.grad.clone() for k, v in p.items() if v.grad is not None and not k.startswith("upscale")}
truth["input"] = xi.grad.clone()
table = {}
precs = ["fp32", "fp32x3", "fp32x3b2", "fp32x2", "fp32h2", "fp32x3h2"]
for prec in precs:
    net = tf.build(wts, prec)
    xg = torch.from_numpy(x).requires_grad_()
    o = net.forward(xg.cuda())
    gt = torch.from_numpy(m).cuda()
    ls = [cbce(t, gt, size_average=False) for t in o]
    (0.5 * sum(ls[:-1]) + ls[-1]).backward()
    g = {k: v.grad.cpu().double() for k, v in net.named_parameters() if v.grad is not None}
    g["input"] = xg.grad.double()
    table[prec] = {k: float((g[k] - t).norm() / (t.norm() + 1e-300)) for k, t in truth.items()}
print("%-22s %10s %10s %10s %10s %10s %10s | b2 / x3  h2 / x3" % ("tensor", *precs))
ratios, ratios_h = [], []
for k in truth:
    r = table["fp32x3b2"][k] / max(table["fp32x3"][k], 1e-30)
    rh = table["fp32h2"][k] / max(table["fp32x3"][k], 1e-30)
    ratios.append(r)
    ratios_h.append(rh)
    print("%-22s %s | %6.2f  %6.2f" % (k, " ".join("%10.2e" % table[q][k] for q in precs), r, rh))
for prec in precs:
    v = np.array(list(table[prec].values()))
    print("%-9s median %.2e  max %.2e  (%s)" % (prec, np.median(v), v.max(), max(table[prec], key=table[prec].get)))
print("fp32x3b2 / fp32x3 per-tensor error ratio: median %.2f, max %.2f, tensors above 2x: %d of %d" % (np.median(ratios), max(ratios), sum(r > 2 for r in ratios), len(ratios)))
print("fp32h2 / fp32x3 per-tensor error ratio: median %.2f, max %.2f, tensors above 2x: %d of %d" % (np.median(ratios_h), max(ratios_h), sum(r > 2 for r in ratios_h), len(ratios_h)))
