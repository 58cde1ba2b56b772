#!/usr/bin/env python3
"""Per-layer timing of the P3 convolution tiles (conv3x3_p3.hip) and the P3 weight gradient next to the fp32-input f32x3 kernels they
replace, at the BASELINE resolution.  Run on the GPU box; the table steers pick_tile_p and the DESIGN numbers."""
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
ap.add_argument("--tiles", default="0,1,2,3,4,5,6,100,102,104,105")
ap.add_argument("--layers", default="")
ap.add_argument("--ksplits", default="", help="extra forced K splits to time on the Cin >= 256 layers, e.g. 2,4,8")
args = ap.parse_args()

chans = [[64, 64], [128, 128], [256, 256, 256], [512, 512, 512], [512, 512, 512]]
layers = []
h, w, cin = args.height, args.width, 3
for si, st in enumerate(chans):
    if si > 0:
        h, w = (h + 1) // 2, (w + 1) // 2
    for j, c in enumerate(st):
        if cin >= 16:
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


tiles = [int(t) for t in args.tiles.split(",")]
n = args.batch
print("layer     dir   HxW        Cin->Cout   GF   | x3(auto) | " + " ".join("q%-6d" % t for t in tiles) + " | best   TF/s(alg)  p3auto  ratio")
tot = {"x3": 0.0, "p3": 0.0, "wx3": 0.0, "wp3": 0.0}
for name, h, w, cin, cout in layers:
    if args.layers and name not in args.layers.split(","):
        continue
    gf = 2.0 * n * h * w * cout * 9 * cin / 1e9
    for direction in ("fwd", "dgrad"):
        kin, kout = (cin, cout) if direction == "fwd" else (cout, cin)
        if kin % 16:
            continue
        x = torch.relu(torch.randn(n, h, w, kin, device="cuda"))
        wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
        wpk3 = ops.pack_x3(wt, dgrad=(direction == "dgrad"))
        x3 = ops.f32_to_p3(x)
        skinny = kout < 32
        ycs = max(4, (kout + 3) // 4 * 4)
        y = torch.empty(n, h, w, ycs, device="cuda")
        y3 = None if skinny else torch.empty(n, 3, h, w, kout, device="cuda", dtype=torch.bfloat16)
        part = torch.empty(_lib.lib().osvos_conv3x3_p3_ws_bytes(n, h, w, max(kout, 4)), device="cuda", dtype=torch.uint8)
        import ctypes as C
        vp = C.c_void_p

        def run_p3(t, ks=0):
            _lib.check(_lib.lib().osvos_conv3x3_p3_abi(vp(x3.data_ptr()), vp(wpk3.data_ptr()), None, None, 0, 0, vp(y.data_ptr()) if skinny else None, ycs,
                                                      vp(y3.data_ptr()) if y3 is not None else None, kout, n, h, w, kin, kout, int(direction == "fwd"),
                                                      t, ks, vp(part.data_ptr()), vp(torch.cuda.current_stream().cuda_stream)), "conv_p3")

        def run_x3():
            ops.conv3x3_x3(x, wpk3, None, kout, relu=(direction == "fwd"), y_cs=ycs)
        ms_x3 = timeit(run_x3, args.reps)
        res = []
        tl = [7, 8, 107] if skinny else tiles
        for t in tl:
            try:
                res.append(timeit(lambda: run_p3(t), args.reps))
            except RuntimeError as e:
                res.append(float("nan"))
        auto = timeit(lambda: run_p3(-1), args.reps)
        best = min(r for r in res if r == r)
        bi = tl[res.index(best)]
        extra = ""
        if args.ksplits and kin >= 256:
            extra = "  ks: " + " ".join("%s=%.3f" % (k, timeit(lambda: run_p3(-1, int(k)), args.reps)) for k in args.ksplits.split(","))
        tot["x3"] += ms_x3
        tot["p3"] += min(best, auto)
        print("%-9s %-5s %4dx%-4d %4d->%-4d %6.2f | %-8.3f | %s | q%-3d %6.3f %6.1f  %6.3f  %.2fx%s" % (
            name, direction, h, w, kin, kout, gf, ms_x3, " ".join("%-7.3f" % r for r in res), bi, best, gf / best, auto, ms_x3 / min(best, auto), extra))
    if cin % 64 == 0 and cout % 64 == 0:
        x = torch.relu(torch.randn(n, h, w, cin, device="cuda"))
        dy = torch.randn(n, h, w, cout, device="cuda")
        x3, dy3 = ops.f32_to_p3(x), ops.f32_to_p3(dy)
        m0 = timeit(lambda: ops.conv3x3_wgrad(x, dy, cin, cout, dtype=_lib.F32_X3), args.reps)
        m1 = timeit(lambda: ops.conv3x3_wgrad_p3(x3, dy3, cin, cout), args.reps)
    elif cout == 16 and cin % 128 == 0:
        x = torch.relu(torch.randn(n, h, w, cin, device="cuda"))
        dy = torch.randn(n, h, w, cout, device="cuda")
        x3, dy3 = ops.f32_to_p3(x), ops.f32_to_p3(dy, cd=16)
        m0 = timeit(lambda: ops.conv3x3_wgrad(x, dy, cin, cout, dtype=_lib.F32), args.reps)
        m1 = timeit(lambda: ops.conv3x3_wgrad_p3(x3, dy3, cin, cout), args.reps)
    else:
        m0 = None
    if m0 is not None:
        tot["wx3"] += m0
        tot["wp3"] += m1
        print("%-9s wgrad %4dx%-4d %4d->%-4d %6.2f | fp32-in %.3f ms (%.1f TF/s)  P3-in %.3f ms (%.1f TF/s)  %.2fx" % (name, h, w, cin, cout, gf, m0, gf / m0, m1, gf / m1, m0 / m1))
print("sums: conv x3 %.3f ms -> p3 %.3f ms; wgrad fp32-in %.3f ms -> P3-in %.3f ms" % (tot["x3"], tot["p3"], tot["wx3"], tot["wp3"]))
