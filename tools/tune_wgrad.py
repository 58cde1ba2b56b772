#!/usr/bin/env python3
"""A/B of the exact-fp32 wgrad kernel variants (OSVOS_WGRAD_VARIANT bits: 1 PIPE, 2 DBUF, 4 OCC2, 8 reference-order slabs) and split
targets (OSVOS_WGRAD_BLOCKS).  The library reads these knobs ONCE per process (OSVOS_ENV_INT caches them: no getenv per launch), so every
(variant, blocks) pair runs in its own subprocess; the parent collects the per-layer minima.  `--child` is the worker."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAYERS = [("conv1_2", 480, 854, 64, 64), ("conv2_2", 240, 427, 128, 128), ("conv3_2", 120, 214, 256, 256),
          ("conv4_2", 60, 107, 512, 512), ("conv5_2", 30, 54, 512, 512), ("side1", 240, 427, 128, 16), ("side2", 120, 214, 256, 16),
          ("side3", 60, 107, 512, 16), ("side4", 30, 54, 512, 16), ("conv1_1", 480, 854, 3, 64)]
VARIANTS = [4, 5]
BLOCKS = [512, 256]


def child():
    import torch
    sys.path.insert(0, REPO)
    from osvos_pytorch_amd import ops
    from osvos_pytorch_amd._lib import F32
    out = {}
    for name, h, w, cin, cout in LAYERS:
        x = torch.randn(1, h, w, (cin + 7) // 8 * 8, device="cuda")
        dy = torch.randn(1, h, w, cout, device="cuda")
        best = 1e9
        for rnd in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops.conv3x3_wgrad(x, dy, cin, cout, dtype=F32)
            b.record()
            torch.cuda.synchronize()
            if rnd > 0:
                best = min(best, a.elapsed_time(b))
        out[name] = best
    print(json.dumps(out))


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
        sys.exit(0)
    res = {}
    for b in BLOCKS:
        for v in VARIANTS:
            env = dict(os.environ, OSVOS_WGRAD_VARIANT=str(v), OSVOS_WGRAD_BLOCKS=str(b))
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], capture_output=True, text=True, env=env, timeout=900)
            res[(b, v)] = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    print("layer     GF     | " + " ".join("v%-2d/b%-4d" % (v, b) for b in BLOCKS for v in VARIANTS))
    for name, h, w, cin, cout in LAYERS:
        gf = 2.0 * h * w * cout * 9 * cin / 1e9
        print("%-9s %6.2f | %s" % (name, gf, " ".join("%-9.3f" % res[(b, v)][name] for b in BLOCKS for v in VARIANTS)))
