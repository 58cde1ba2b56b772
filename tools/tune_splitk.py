#!/usr/bin/env python3
"""standalone timing of the fp32 conv with K cut into parts (conv + finalize), deep layers at batch 1"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osvos_pytorch_amd import ops

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    return best

for name, h, w, cin, cout in [("conv3_2", 120, 214, 256, 256), ("conv4_2", 60, 107, 512, 512), ("conv5_2", 30, 54, 512, 512)]:
    x = torch.randn(1, h, w, cin, device="cuda"); wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    wpk = ops.pack_fwd(wt); b = torch.randn(cout, device="cuda")
    for tile in (3, 9, 5, 11):
        row = []
        for ks in (1, 2, 4, 8):
            row.append(timeit(lambda: ops.conv3x3_splitk(x, wpk, b, cout, ks, relu=True, tile=tile)))
        base = timeit(lambda: ops.conv3x3(x, wpk, b, cout, relu=True, tile=tile))
        print("%-8s tile %2d  no-ws %.3f | ksplit 1,2,4,8: %s" % (name, tile, base, "  ".join("%.3f" % r for r in row)))
