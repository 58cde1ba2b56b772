#!/usr/bin/env python3
"""Op-level timing of the Cin = 64 bf16-store convolutions AS THE NETWORK LAUNCHES THEM (osvos_conv3x3_bf16act_fused: bf16 in / out, sign bits,
fused pool + code bytes, one-bit ReLU mask on the data gradient) over tile ids -- the round-6 resident-filter persistent forms 36 / 37 against the
register-staged tiles 9 / 1.  usage: tune_p64.py [--batch 12] [--tiles 9,109,1,36,136,37,137] [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osvos_pytorch_amd import ops, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=12)
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--width", type=int, default=854)
ap.add_argument("--tiles", default="9,109,36,136,38,138")
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
tiles = [int(t) for t in args.tiles.split(",")]
n, H, W = args.batch, args.height, args.width


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[0], ts[len(ts) // 2]


cases = [("conv1_2 fwd +pool+bits", H, W, 64, 64, dict(relu=True, want_bits=False, want_pool=True), False),
         ("conv1_2 fwd inner(bits)", H, W, 64, 64, dict(relu=True, want_bits=True, want_pool=False), False),
         ("conv1_2 dgrad (bitmask)", H, W, 64, 64, dict(relu=False), True),
         ("conv2_1 fwd (bits)", (H + 1) // 2, (W + 1) // 2, 64, 128, dict(relu=True, want_bits=True, want_pool=False), False)]
print("batch %d; times in ms: best/median of %d" % (n, args.reps))
print("%-26s %8s | " % ("case", "GB alg") + " ".join("t%-11d" % t for t in tiles))
for name, h, w, cin, cout, kw, masked in cases:
    x = torch.randn(n, h, w, cin, device="cuda").bfloat16()
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wpk = ops.pack_fwd(wt, _lib.F32_BF16MFMA)
    b = torch.zeros(cout, device="cuda") if not masked else None
    mb = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, h, w, cout // 32), device="cuda", dtype=torch.int32) if masked else None
    gb = (n * h * w * (cin + cout) * 2 + (n * h * w * cout / 8 if (masked or kw.get("want_bits")) else 0) +
          (n * ((h + 1) // 2) * ((w + 1) // 2) * cout * 3 if kw.get("want_pool") else 0)) / 1e9
    # outputs allocated once per case, outside the timed call
    import ctypes as C
    y = torch.empty((n, h, w, cout), device="cuda", dtype=torch.bfloat16)
    bits = torch.empty((n, h, w, cout // 32), device="cuda", dtype=torch.int32) if kw.get("want_bits") else None
    pooled = torch.empty((n, (h + 1) // 2, (w + 1) // 2, cout), device="cuda", dtype=torch.bfloat16) if kw.get("want_pool") else None
    code = torch.empty((n, (h + 1) // 2, (w + 1) // 2, cout), device="cuda", dtype=torch.uint8) if kw.get("want_pool") else None
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    row = []
    for t in tiles:
        def fn():
            _lib.check(_lib.lib().osvos_conv3x3_bf16act_fused(P(x), P(wpk), P(b), P(mb), P(y), P(bits), P(pooled), P(code), n, h, w, cin, cout,
                                                              int(kw.get("relu", False)), t, st), "fused")
        try:
            best, med = timeit(fn, args.reps)
            row.append("%.3f/%.3f" % (best, med))
        except RuntimeError as e:
            row.append("err        ")
    print("%-26s %8.3f | " % (name, gb) + " ".join("%-12s" % r for r in row), flush=True)
