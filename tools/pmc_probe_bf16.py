#!/usr/bin/env python3
"""rocprofv3 --pmc workload: bf16-MFMA conv3_2-shaped forward / data gradient / weight gradient at batch 12."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osvos_pytorch_amd import ops  # noqa: E402
from osvos_pytorch_amd._lib import F32_BF16MFMA as DT  # noqa: E402

n, h, w, cin, cout = 12, 120, 214, 256, 256
x = torch.randn(n, h, w, cin, device="cuda")
dy = torch.randn(n, h, w, cout, device="cuda")
wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
wf, wd = ops.pack_fwd(wt, DT), ops.pack_dgrad(wt, DT)
for _ in range(3):
    ops.conv3x3(x, wf, None, cout, relu=True, dtype=DT)
    ops.conv3x3(dy, wd, None, cin, relu=False, mask=x, dtype=DT)
    ops.conv3x3_wgrad(x, dy, cin, cout, dtype=DT)
torch.cuda.synchronize()
