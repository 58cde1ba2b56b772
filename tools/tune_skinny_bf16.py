#!/usr/bin/env python3
"""bf16-mode skinny convolutions at batch 12: the input gradient (64 -> 3) and the side-branch shapes, per tile."""
import ctypes as C
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from osvos_pytorch_amd import _lib, ops  # noqa: E402

vp = C.c_void_p
lib = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
st = lambda: vp(torch.cuda.current_stream().cuda_stream)  # noqa: E731


def timed(fn, reps=4):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best * 1e3


DT = _lib.F32_BF16MFMA
for name, h, w, cin, cout, ycs, want_bf16 in [("input gradient 64->3", 480, 854, 64, 3, 4, False), ("side1 dgrad 16->128", 240, 427, 16, 128, 128, True),
                                                ("side2 dgrad 16->256", 120, 214, 16, 256, 256, True), ("side3 dgrad 16->512", 60, 107, 16, 512, 512, True),
                                                ("side1 fwd 128->16", 240, 427, 128, 16, 16, False), ("side2 fwd 256->16", 120, 214, 256, 16, 16, False)]:
    x = torch.randn(n, h, w, cin, device="cuda").bfloat16()
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wpk = ops.pack_fwd(wt, DT)
    y = torch.empty(n, h, w, ycs, device="cuda", dtype=torch.float32) if not want_bf16 else None
    yb = torch.empty(n, h, w, ycs, device="cuda", dtype=torch.bfloat16) if want_bf16 else None
    res = []
    for tile in ([-1, 6, 106] if cout <= 32 else [-1, 1, 9, 8]):
        try:
            us = timed(lambda: _lib.check(lib.osvos_conv3x3_bf16io(vp(x.data_ptr()), 1, vp(wpk.data_ptr()), None, None, 0, vp(y.data_ptr()) if y is not None else None,
                                                                    vp(yb.data_ptr()) if yb is not None else None, n, h, w, cin, cout, ycs, 0, tile, st()), "conv"))
            res.append("tile %d: %.1f us" % (tile, us))
        except Exception as e:
            res.append("tile %d: %s" % (tile, str(e)[:40]))
    print("%-24s %s" % (name, "   ".join(res)))
