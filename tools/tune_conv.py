#!/usr/bin/env python3
"""Per-layer timing sweep of the conv3x3 tile configs and the wgrad kernel at a given resolution.
Writes a table to stdout (run on the GPU box; results steer pick_tile / the layer table)."""
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
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--tiles", default="")
ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
ap.add_argument("--noyb", type=int, default=0, help="with --xb: do not write the bf16 output copy")
ap.add_argument("--layers", default="", help="comma separated layer-name filter")
ap.add_argument("--xb", type=int, default=0, help="bf16 dtype only: 1 = bf16 activations in HBM (input read as bf16, bf16 copy of the output written)")
args = ap.parse_args()

chans = [[64, 64], [128, 128], [256, 256, 256], [512, 512, 512], [512, 512, 512]]
layers = []   # name, h, w, cin, cout
h, w, cin = args.height, args.width, 3
for si, st in enumerate(chans):
    if si > 0:
        h, w = (h + 1) // 2, (w + 1) // 2
    for j, c in enumerate(st):
        layers.append(("conv%d_%d" % (si + 1, j + 1), h, w, cin, c))
        cin = c
    if si > 0:
        layers.append(("side%d" % si, h, w, cin, 16))


def timeit(fn, reps):
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
    return best


import ctypes as C
_out = torch.empty(2048 * 256, device="cuda")
_it = 2000
def _peak():
    _lib.check(_lib.lib().osvos_debug_mfma_peak(C.c_void_p(_out.data_ptr()), 2048, _it, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
ms = timeit(_peak, 3)
print("fp32 MFMA-only probe: %.1f TFLOP/s (2048 WGs x 4 waves x %d x 4 MFMA 32x32x2)" % (2048 * 4 * _it * 4 * 2 * 32 * 32 * 2 / ms / 1e9, _it))
DT = _lib.F32 if args.dtype == "fp32" else _lib.F32_BF16MFMA
default_tiles = [2, 3, 5, 6, 9, 102, 103, 105, 106, 109] if args.dtype == "fp32" else [1, 3, 5, 8, 9, 10, 32, 35, 108, 110]
tiles = [int(t) for t in args.tiles.split(",")] if args.tiles else default_tiles
n = args.batch
print("layer            dir   HxW       Cin->Cout  GF    | " + " ".join("t%-6d" % t for t in tiles) + " | best  TF/s  auto")
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
for name, h, w, cin, cout in layers:
    if args.layers and name not in args.layers.split(","):
        continue
    gf = 2.0 * n * h * w * cout * 9 * cin / 1e9
    for direction in ("fwd", "dgrad"):
        kin, kout = (cin, cout) if direction == "fwd" else (cout, cin)
        kin_s = (kin + 7) // 8 * 8
        x = torch.randn(n, h, w, kin_s, device="cuda")
        wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        wpk = ops.pack_fwd(wt, DT) if direction == "fwd" else ops.pack_dgrad(wt, DT)
        ycs = kout if kout >= 8 else 4
        if args.xb:
            xb = x.bfloat16()
            yo = torch.empty(n, h, w, ycs, device="cuda")
            yb = torch.empty(n, h, w, ycs, device="cuda", dtype=torch.bfloat16) if (kout % 8 == 0 and not args.noyb) else None
            vp = C.c_void_p
            def run(t):
                # --xb 1 = what the network does in the bf16 mode: bf16 in, bf16-only out (fp32 out only where no bf16 copy is possible)
                _lib.check(_lib.lib().osvos_conv3x3_bf16io(vp((xb if args.xb == 1 else x).data_ptr()), int(args.xb == 1), vp(wpk.data_ptr()), None, None, 0,
                                                          vp(yo.data_ptr()) if (yb is None or args.xb != 1) else None,
                                                          vp(yb.data_ptr()) if yb is not None else None, n, h, w, kin_s, kout, ycs,
                                                          int(direction == "fwd"), t, vp(torch.cuda.current_stream().cuda_stream)), "conv")
        else:
            def run(t):
                ops.conv3x3(x, wpk, None, kout, relu=(direction == "fwd"), y_cs=ycs, tile=t, dtype=DT)
        res = []
        for t in tiles:
            try:
                ms = timeit(lambda: run(t), args.reps)
            except RuntimeError:
                ms = float("nan")
            res.append(ms)
        auto = timeit(lambda: run(-1), args.reps)
        best = min(r for r in res if r == r)
        bi = tiles[res.index(best)]
        tot[direction] += best
        print("%-16s %-5s %4dx%-4d %4d->%-4d %6.2f | %s | t%d %6.1f %.3f" % (
            name, direction, h, w, kin, kout, gf, " ".join("%-7.3f" % r for r in res), bi, gf / best, auto))
    x = torch.randn(n, h, w, (cin + 7) // 8 * 8, device="cuda")
    dy = torch.randn(n, h, w, cout, device="cuda")
    if args.xb == 1 and cin % 64 == 0 and cout % 64 == 0:
        xb16, dyb16 = x.bfloat16(), dy.bfloat16()
        ms32 = timeit(lambda: ops.conv3x3_wgrad(x, dy, cin, cout, dtype=DT), args.reps)
        ms = timeit(lambda: ops.conv3x3_wgrad_bf16act(xb16, dyb16, cin, cout), args.reps)
        print("%-16s wgrad fp32-in %.3f ms -> bf16-in:" % (name, ms32), end=" ")
    else:
        ms = timeit(lambda: ops.conv3x3_wgrad(x, dy, cin, cout, dtype=DT), args.reps)
    tot["wgrad"] += ms
    print("%-16s wgrad %4dx%-4d %4d->%-4d %6.2f | %.3f ms  %.1f TF/s" % (name, h, w, cin, cout, gf, ms, gf / ms))
print("sum of best: fwd %.3f ms, dgrad %.3f ms, wgrad %.3f ms" % (tot["fwd"], tot["dgrad"], tot["wgrad"]))
