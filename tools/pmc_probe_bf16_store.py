#!/usr/bin/env python3
"""rocprofv3 --pmc workload: bf16 in / bf16 out convolutions (the network's bf16 mode) at batch 12, conv3_2 and conv4_2 shapes,
register-staged tile 8 and the persistent LDS-DMA kernel (tile 35), plus the bf16-input weight gradient."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osvos_pytorch_amd import _lib, ops  # noqa: E402
from osvos_pytorch_amd._lib import F32_BF16MFMA as DT  # noqa: E402

vp = C.c_void_p
for (h, w, c) in ((120, 214, 256), (60, 107, 512)):
    n = 12
    x = torch.randn(n, h, w, c, device="cuda").bfloat16()
    dy = torch.randn(n, h, w, c, device="cuda").bfloat16()
    wt = torch.randn(c, c, 3, 3, device="cuda") * 0.05
    wf = ops.pack_fwd(wt, DT)
    yb = torch.empty(n, h, w, c, device="cuda", dtype=torch.bfloat16)
    for tile in (8, 35):
        for _ in range(3):
            _lib.check(_lib.lib().osvos_conv3x3_bf16io(vp(x.data_ptr()), 1, vp(wf.data_ptr()), None, None, 0, None, vp(yb.data_ptr()), n, h, w, c, c, c, 1,
                                                      tile, vp(torch.cuda.current_stream().cuda_stream)), "conv")
    for _ in range(3):
        ops.conv3x3_wgrad_bf16act(x, dy, c, c)
torch.cuda.synchronize()
