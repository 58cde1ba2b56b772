#!/usr/bin/env python3
"""Per-layer sweep of the f32x3 convolution (conv3x3_f32x3.hip): tile configs x K splits, forward and data gradient,
next to the exact fp32 MFMA kernel's automatic choice.  Times are per launch from 4 back-to-back launches (warm clock);
TF/s are ALGORITHMIC fp32 FLOPs (2 N H W Cout 9 Cin) per second -- the f32x3 kernel executes 6x that on the bf16 pipe."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osvos_pytorch_amd import ops, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--width", type=int, default=854)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--tiles", default="")
ap.add_argument("--layers", default="")
ap.add_argument("--wgrad-only", action="store_true")
args = ap.parse_args()

chans = [[64, 64], [128, 128], [256, 256, 256], [512, 512, 512], [512, 512, 512]]
layers = []
h, w, cin = args.height, args.width, 3
for si, st in enumerate(chans):
    if si > 0:
        h, w = (h + 1) // 2, (w + 1) // 2
    for j, c in enumerate(st):
        if j != 2:      # conv3_3 == conv3_2 etc.
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


nt = _lib.lib().osvos_conv3x3_f32x3_tiles()
tiles = [int(t) for t in args.tiles.split(",")] if args.tiles else [200 + k for k in range(nt)]
n = args.batch
print("f32x3 sweep %dx%d batch %d; columns = tile id (200+k; +100 = XCD-local map); ms per launch" % (args.width, args.height, n))
tot_x, tot_e, tot_ps = 0.0, 0.0, 0.0
for name, h, w, cin, cout in layers:
    if args.wgrad_only or (args.layers and name not in args.layers.split(",")):
        continue
    gf = 2.0 * n * h * w * cout * 9 * cin / 1e9
    for direction in ("fwd", "dgrad"):
        kin, kout = (cin, cout) if direction == "fwd" else (cout, cin)
        if kin % 16 or kout < 32:
            continue
        x = torch.randn(n, h, w, kin, device="cuda")
        wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        wpk = ops.pack_fwd(wt) if direction == "fwd" else ops.pack_dgrad(wt)
        exact = timeit(lambda: ops.conv3x3(x, wpk, None, kout, relu=True), args.reps)
        auto = timeit(lambda: ops.conv3x3(x, wpk, None, kout, relu=True, dtype=_lib.F32_X3), args.reps)
        wpk3 = ops.pack_x3(wt, dgrad=(direction == "dgrad"))
        auto_ps = timeit(lambda: ops.conv3x3_x3(x, wpk3, None, kout, relu=True), args.reps)
        tot_ps += auto_ps
        best, bt = 1e9, None
        cells = []
        for t in tiles:
            row = []
            for ks in ([1] if kin < 256 else [1, 2, 4]):
                for mp in (0, 100):
                    try:
                        if ks == 1:
                            ms = timeit(lambda: ops.conv3x3(x, wpk, None, kout, relu=True, tile=t + mp), args.reps)
                        else:
                            ms = timeit(lambda: ops.conv3x3_splitk(x, wpk, None, kout, ks, relu=True, tile=t + mp), args.reps)
                    except RuntimeError:
                        ms = float("nan")
                    row.append(ms)
                    if ms == ms and ms < best:
                        best, bt = ms, (t + mp, ks)
            cells.append("%d:" % t + "/".join("%.3f" % r for r in row))
        tot_x += best
        tot_e += exact
        print("%-8s %-5s %4dx%-4d %4d->%-4d %6.2f GF | exact %.3f ms %6.1f TF/s | x3 auto %.3f, pre-split weights %.3f ms %6.1f TF/s | best tile %s ks %s: %.3f ms %6.1f TF/s (%.2fx)"
              % (name, direction, h, w, kin, kout, gf, exact, gf / exact, auto, auto_ps, gf / auto_ps, bt[0], bt[1], best, gf / best, exact / best))
        print("    [k1 map0/map1%s] " % ("/k2../k4.." if kin >= 256 else "") + "  ".join(cells))
print("sum over the listed layers: exact %.3f ms, f32x3 best listed tile %.3f ms, f32x3 automatic choice with pre-split weights %.3f ms" % (tot_e, tot_x, tot_ps))
print("weight gradient (fp32 x, dy -> fp32 dW, db; slabs + reduce included):")
tw_e, tw_x = 0.0, 0.0
for name, h, w, cin, cout in layers:
    if args.layers and name not in args.layers.split(","):
        continue
    if cin % 64 or cout % 64:
        continue
    gf = 2.0 * n * h * w * cout * 9 * cin / 1e9
    x = torch.randn(n, h, w, cin, device="cuda")
    dy = torch.randn(n, h, w, cout, device="cuda")
    e = timeit(lambda: ops.conv3x3_wgrad(x, dy, cin, cout), args.reps)
    t3 = timeit(lambda: ops.conv3x3_wgrad(x, dy, cin, cout, dtype=_lib.F32_X3), args.reps)
    tw_e += e
    tw_x += t3
    print("%-8s wgrad %4dx%-4d %4d->%-4d %6.2f GF | exact %.3f ms %6.1f TF/s | f32x3 %.3f ms %6.1f TF/s (%.2fx)" % (name, h, w, cin, cout, gf, e, gf / e, t3, gf / t3, e / t3))
print("weight gradient sum over the listed layers: exact %.3f ms, f32x3 %.3f ms" % (tw_e, tw_x))
