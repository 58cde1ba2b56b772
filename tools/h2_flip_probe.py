#!/usr/bin/env python3
"""Where do the arithmetics of the wide 3x3 convolutions put a pre-activation on the wrong side of zero?  (checker-side probe, round 6)
For a realistic operand pair -- the trained-like fixture's activations in front of conv2_1 / conv3_1 / conv4_1 (post-ReLU + pooled, strong per-channel DC) and
the fixture's weights -- run the exact fp32 MFMA kernel, f32x3 (three bf16 pieces), f32x2 (two) and h2 (two FP16 pieces under block exponents) WITHOUT ReLU and
compare with float64: rel-L2, the error in units of sum|a||w| (max, rms), and the number of outputs whose SIGN differs from float64's (a ReLU flip)."""
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import trained_fixture as tf  # noqa: E402
from oracle import synth, torch_ref  # noqa: E402
from osvos_pytorch_amd import ops  # noqa: E402

wts, frames, _ = tf.train_like()
x, m = synth.trainable_frame(1, 240, 427, seed=tf.RECIPE["frame_seed"] + 98)
p64 = torch_ref.as_leaf_params(wts, dtype=torch.float64)
acts = {}
h = torch.from_numpy(x).double()
# VGG trunk restated inline (vgg_osvos.py:136-145): record the input of the first convolution of stages 2..4
cfg = [[64, 64], [128, 128], [256, 256, 256], [512, 512, 512], [512, 512, 512]]
for s, stage in enumerate(cfg):
    if s > 0:
        h = F.max_pool2d(h, 2, 2, ceil_mode=True)
    for j in range(len(stage)):
        idx = (0 if s == 0 else 1) + 2 * j      # index inside the reference's nn.Sequential of the stage (pool first from stage 2 on)
        w = p64["stages.%d.%d.weight" % (s, idx)]; b = p64["stages.%d.%d.bias" % (s, idx)]
        if j == 0 and s in (1, 2, 3):
            acts[s] = (h.clone(), w.detach(), b.detach())
        h = F.relu(F.conv2d(h, w, b, padding=1))

def nhwc(t): return t.permute(0, 2, 3, 1).contiguous().float().cuda()
def nchw(t): return t.permute(0, 3, 1, 2).contiguous()
print("%-8s %-7s %10s %12s %12s %10s %10s" % ("layer", "kernel", "rel-L2", "max err/mag", "rms err/mag", "flips", "of"))
for s, (a, w, b) in acts.items():
    a32, w32, b32 = a.float(), w.float(), b.float()
    ref = F.conv2d(a32.double(), w32.double(), b32.double(), padding=1)
    mag = F.conv2d(a32.double().abs(), w32.double().abs(), b32.double().abs(), padding=1)
    cout = w.shape[0]
    xg = nhwc(a32)
    res = {"exact": nchw(ops.conv3x3(xg, ops.pack_fwd(w32.cuda()), b32.cuda(), cout)).cpu().double()}
    res["cpu-f32"] = F.conv2d(a32, w32, b32, padding=1).double()
    for name, pieces in (("x3", 3), ("x2", 2), ("h2", 22)):
        try:
            ops.set_x3_pieces(pieces)
            res[name] = nchw(ops.conv3x3_x3(xg, ops.pack_x3(w32.cuda()), b32.cuda(), cout)).cpu().double()
        finally:
            ops.set_x3_pieces(3)
    for name, y in res.items():
        e = (y - ref)
        print("stage%d   %-7s %10.2e %12.2e %12.2e %10d %10d" % (s + 1, name, float(e.norm() / ref.norm()), float((e.abs() / mag).max()), float(((e / mag) ** 2).mean() ** 0.5),
                                                       int(((y > 0) != (ref > 0)).sum()), ref.numel()))
    amax = a32.abs().amax(dim=(0, 2, 3))
    print("   activation max per channel: median %.3g, max %.3g; tensor mean %.3g;  weight |max| %.3g, rms %.3g" % (float(amax.median()), float(amax.max()), float(a32.mean()), float(w32.abs().max()), float(w32.pow(2).mean().sqrt())))
