#!/usr/bin/env python3
"""A/B of the wgrad kernel variants (OSVOS_WGRAD_VARIANT bits: 1 PIPE, 2 DBUF, 4 OCC2, 8 reference-order
slabs) and split targets (OSVOS_WGRAD_BLOCKS), interleaved rounds in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osvos_pytorch_amd import ops  # noqa: E402

layers = [("conv1_2", 480, 854, 64, 64), ("conv2_2", 240, 427, 128, 128), ("conv3_2", 120, 214, 256, 256),
          ("conv4_2", 60, 107, 512, 512), ("conv5_2", 30, 54, 512, 512), ("side1", 240, 427, 128, 16), ("side2", 120, 214, 256, 16), ("side3", 60, 107, 512, 16), ("side4", 30, 54, 512, 16), ("conv1_1", 480, 854, 3, 64)]
variants = [4, 5]
blocks = [512, 256]


def run(x, dy, cin, cout):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.conv3x3_wgrad(x, dy, cin, cout)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b)


print("layer     GF     | " + " ".join("v%-2d/b%-4d" % (v, b) for b in blocks for v in variants))
for name, h, w, cin, cout in layers:
    gf = 2.0 * h * w * cout * 9 * cin / 1e9
    x = torch.randn(1, h, w, (cin + 7) // 8 * 8, device="cuda")
    dy = torch.randn(1, h, w, cout, device="cuda")
    best = {}
    for rnd in range(4):
        for b in blocks:
            for v in variants:
                os.environ["OSVOS_WGRAD_VARIANT"] = str(v)
                os.environ["OSVOS_WGRAD_BLOCKS"] = str(b)
                t = run(x, dy, cin, cout)
                if rnd > 0:
                    best[(b, v)] = min(best.get((b, v), 1e9), t)
    print("%-9s %6.2f | %s" % (name, gf, " ".join("%-9.3f" % best[(b, v)] for b in blocks for v in variants)))
