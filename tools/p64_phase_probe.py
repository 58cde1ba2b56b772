#!/usr/bin/env python3
"""Phase cycle split of conv3x3_bf16_p64_kernel (probe build: the library rebuilt with -DP64_PROF on the GPU box by tools/p64_phase_probe.sh):
per wave sums of s_memtime deltas -- 0 MFMA chunk, 1 epilogue piece, 2 s_waitcnt vmcnt, 3 barrier, 4 tile retire, 5 drain; 6 = kernel, 7 = tiles."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osvos_pytorch_amd import ops, _lib  # noqa: E402

n, H, W = 12, 480, 854
l = _lib.lib()
setp = l.osvos_debug_set_p64_prof
setp.argtypes = [C.c_void_p]
setp.restype = None
cases = [("conv1_2 fwd +pool", H, W, 64, dict(relu=True, want_pool=True), False), ("conv1_2 fwd bits", H, W, 64, dict(relu=True, want_bits=True), False),
         ("conv1_2 dgrad mask", H, W, 64, dict(relu=False), True), ("conv2_1 fwd bits", H // 2, (W + 1) // 2, 128, dict(relu=True, want_bits=True), False)]
names = ["mfma", "piece", "vmwait", "barrier", "retire", "drain", "advance"]
for name, h, w, cout, kw, masked in cases:
    x = torch.randn(n, h, w, 64, device="cuda").bfloat16()
    wpk = ops.pack_fwd(torch.randn(cout, 64, 3, 3, device="cuda") * 0.05, _lib.F32_BF16MFMA)
    b = torch.zeros(cout, device="cuda") if not masked else None
    mb = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, h, w, cout // 32), device="cuda", dtype=torch.int32) if masked else None
    prof = torch.zeros(256 * 8 * 10, device="cuda", dtype=torch.int64)
    for tile in (38, 138):
        for rep in range(3):
            prof.zero_()
            setp(C.c_void_p(prof.data_ptr()))
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.conv3x3_bf16act_fused(x, wpk, b, cout, mask_bits=mb, tile=tile, **kw)
            e.record()
            torch.cuda.synchronize()
        raw = prof.view(256, 8, 10).cpu()
        if name.startswith("conv1_2 dgrad") and tile == 38:
            print("   SIMD of waves 0..7 (HW_ID.SIMD_ID), workgroups 0, 1, 100:", [[int(raw[b, w, 9]) for w in range(8)] for b in (0, 1, 100)])
            for b in (0, 100):
                print("   wg %d per wave, cycles per chunk iteration [mfma piece vmwait barrier retire/4 drain advance]:" % b,
                      [[int(raw[b, w, k]) // (38 * 4) for k in range(7)] for w in range(8)])
        p = raw.double()
        tot = p[:, :, 7]
        line = "%-20s tile %3d  %.3f ms | kernel cycles/wave %.0f (s_memtime units), tiles/wg %.1f |" % (name, tile, a.elapsed_time(e), float(tot.mean()), float(p[:, :, 8].mean()))
        for g, gname in ((slice(0, 4), "grp0"), (slice(4, 8), "grp1")):
            line += " %s:" % gname + " ".join("%s %.0f%%" % (names[k], 100.0 * float(p[:, g, k].sum() / p[:, g, 7].sum())) for k in range(7))
        print(line, flush=True)
setp(None)
