#!/usr/bin/env python3
"""Op-level A/B of the stream-K form of the f32x3 convolution (conv3x3_f32x3.hip) against the plain grid (and, for the layers the network cuts
along K, against partial-sum launches + finalize kernel): per layer of the trunk at --height x --width x --batch, forward and data gradient.
ms per launch = best of --reps timings of 4 back-to-back launches.  Columns: plain | split-K auto | stream-K at each --grids value."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osvos_pytorch_amd import ops, _lib  # noqa: E402
from osvos_pytorch_amd._lib import F32_X3  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--width", type=int, default=854)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--grids", default="256,248,240,224")
ap.add_argument("--dgrad", type=int, default=1)
args = ap.parse_args()
grids = [int(g) for g in args.grids.split(",")]

chans = [[64, 64], [128, 128], [256, 256, 256], [512, 512, 512], [512, 512, 512]]
layers = []
h, w, cin = args.height, args.width, 3
for si, st in enumerate(chans):
    if si > 0:
        h, w = (h + 1) // 2, (w + 1) // 2
    for j, c in enumerate(st):
        if j != 2 and cin % 16 == 0:
            layers.append(("conv%d_%d" % (si + 1, j + 1), h, w, cin, c))
        cin = c


def timeit(fn, reps, inner=4):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(inner):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / inner)
    return best


n = args.batch
lib = _lib.lib()
print("stream-K A/B %dx%d batch %d (ms per launch; TF/s algorithmic)" % (args.width, args.height, n))
print("%-8s %-5s %-9s %-10s | %8s %8s | %s" % ("layer", "dir", "HxW", "Cin->Cout", "plain", "splitK", "  ".join("sk%-5d" % g for g in grids)))
tot = {"plain": 0.0, "best_plain": 0.0, **{g: 0.0 for g in grids}}
for name, h, w, cin, cout in layers:
    gf = 2.0 * n * h * w * cout * 9 * cin / 1e9
    for direction in (("fwd", "dgrad") if args.dgrad else ("fwd",)):
        kin, kout = (cin, cout) if direction == "fwd" else (cout, cin)
        x = torch.randn(n, h, w, kin, device="cuda").relu_()
        wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        wpk3 = ops.pack_x3(wt, dgrad=(direction == "dgrad"))
        plain = timeit(lambda: ops.conv3x3_x3(x, wpk3, None, kout, relu=True), args.reps)
        sk_ms = float("nan")
        if kin >= 256:
            wpk = ops.pack_fwd(wt) if direction == "fwd" else ops.pack_dgrad(wt)
            sk_ms = timeit(lambda: ops.conv3x3_splitk(x, wpk, None, kout, 0, relu=True, dtype=F32_X3), args.reps)
        row = []
        for g in grids:
            row.append(timeit(lambda: ops.conv3x3_x3_streamk(x, wpk3, None, kout, relu=True, grid=g), args.reps))
        bp = min(plain, sk_ms) if sk_ms == sk_ms else plain
        tot["plain"] += plain
        tot["best_plain"] += bp
        for g, r in zip(grids, row):
            tot[g] += r
        print("%-8s %-5s %4dx%-4d %4d->%-4d | %8.4f %8.4f | %s   (%.1f -> %.1f TF/s)"
              % (name, direction, h, w, kin, kout, plain, sk_ms, "  ".join("%7.4f" % r for r in row), gf / bp, gf / min(row)))
print("sums: plain %.4f  best of plain/split-K %.4f  " % (tot["plain"], tot["best_plain"]) + "  ".join("sk%d %.4f" % (g, tot[g]) for g in grids))
