#!/usr/bin/env python3
"""Tiny workload for rocprofv3 --pmc passes on the f32x3 kernels: conv3_2-shaped (120x214, 256->256, 30.29 GFLOP algorithmic) and
conv1_2-shaped (480x854, 64->64, 30.22 GFLOP) forward convolution, data gradient (with the ReLU mask) and weight gradient, 3 launches each,
next to the exact fp32 kernels on the conv3_2 shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osvos_pytorch_amd import ops, _lib  # noqa: E402

for (h, w, cin, cout) in [(120, 214, 256, 256), (480, 854, 64, 64)]:
    x = torch.randn(1, h, w, cin, device="cuda")
    dy = torch.randn(1, h, w, cout, device="cuda")
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wf, wd = ops.pack_fwd(wt), ops.pack_dgrad(wt)
    for _ in range(3):
        ops.conv3x3(x, wf, None, cout, relu=True, dtype=_lib.F32_X3)
        ops.conv3x3(dy, wd, None, cin, relu=False, mask=x, dtype=_lib.F32_X3)
        ops.conv3x3_wgrad(x, dy, cin, cout, dtype=_lib.F32_X3)
torch.cuda.synchronize()
