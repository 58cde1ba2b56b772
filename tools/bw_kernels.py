#!/usr/bin/env python3
"""Bandwidth kernels of the step at their network shapes: microseconds and the TB/s of their ALGORITHMIC bytes (every tensor read / written once),
against the 6.3 TB/s a float4 copy reaches on this chip (MI355X_MICROARCH.md).  usage: bw_kernels.py [batch]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from osvos_pytorch_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
STAGES = [(480, 854, 64), (240, 427, 128), (120, 214, 256), (60, 107, 512)]


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best * 1e3


def line(name, us, nbytes):
    print("%-34s %9.1f us  %8.1f MB  %6.2f TB/s" % (name, us, nbytes / 1e6, nbytes / us / 1e6))


tot = {}
for si, (h, w, c) in enumerate(STAGES):
    ho, wo = (h + 1) // 2, (w + 1) // 2
    x = torch.relu(torch.randn(n, h, w, c, device="cuda"))
    dy = torch.randn(n, ho, wo, c, device="cuda")
    ds = torch.randn(n, h, w, c, device="cuda") if si > 0 else None
    full, quarter = x.numel(), dy.numel()
    for tag, xb, dyb, dsb, esz, fwd, bwd in (("f32", x, dy, ds, 4, ops.maxpool2x2, ops.maxpool2x2_bwd),
                                              ("bf16", x.bfloat16(), dy.bfloat16(), ds.bfloat16() if ds is not None else None, 2,
                                               ops.maxpool2x2_bf16act, ops.maxpool2x2_bwd_bf16act)):
        us = timed(lambda: fwd(xb))
        line("pool fwd %s stage %d (%dx%dx%d)" % (tag, si, h, w, c), us, (full + quarter) * esz)
        tot[tag + " fwd"] = tot.get(tag + " fwd", 0) + us
        us = timed(lambda: bwd(xb, dyb, dsb))
        line("pool bwd %s stage %d%s" % (tag, si, " +side" if ds is not None else ""), us, (2 * full + quarter + (full if ds is not None else 0)) * esz)
        tot[tag + " bwd"] = tot.get(tag + " bwd", 0) + us
print("per step, batch %d: " % n + ", ".join("%s %.0f us" % kv for kv in tot.items()))
