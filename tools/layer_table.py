#!/usr/bin/env python3
"""Per-layer roofline table of the conv3_x / conv4_x kernels: forward, data gradient and weight gradient of conv3_2 and
conv4_2 as the network runs them, each with its algorithmic TFLOP/s, fraction of the 2.5 PFLOP/s bf16 spec peak, fraction of what the
matrix pipe of THIS box sustains on noise operands (register-only MFMA loop, measured in the same process), and -- from a rocprofv3 --pmc
pass over the same command -- the matrix pipe's busy share and the effective clock.

    python tools/layer_table.py run --mode bf16 --batch 12            # the workload (prints its launch manifest + un-profiled timings as JSON)
    rocprofv3 --kernel-trace --output-format csv -d D -o p --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python tools/layer_table.py run ...
    python tools/layer_table.py report D manifest.json                # the table

modes: bf16 = bf16 tensors / bf16 MFMA operands (configs[2]); x3 = fp32 tensors, f32x3 arithmetic (configs[1]; executed FLOPs = 6 x algorithmic)."""
import argparse
import csv
import ctypes as C
import glob
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
LAYERS = [("conv3_2", 120, 214, 256, 256), ("conv4_2", 60, 107, 512, 512)]
ALL_LAYERS = [("conv1_2", 480, 854, 64, 64), ("conv2_1", 240, 427, 64, 128), ("conv2_2", 240, 427, 128, 128), ("conv3_1", 120, 214, 128, 256),
              ("conv3_2", 120, 214, 256, 256), ("conv4_1", 60, 107, 256, 512), ("conv4_2", 60, 107, 512, 512), ("conv5_2", 30, 54, 512, 512)]
REPS = 3


def run(args):
    import torch
    from osvos_pytorch_amd import _lib, ops
    vp = C.c_void_p
    lib = _lib.lib()
    n = args.batch
    st = lambda: vp(torch.cuda.current_stream().cuda_stream)      # noqa: E731

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(REPS):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
        return best
    manifest = []
    for name, h, w, cin, cout in (ALL_LAYERS if args.all else LAYERS):
        gf = 2.0 * n * h * w * cout * 9 * cin / 1e9
        g = torch.Generator(device="cuda").manual_seed(3)
        x = torch.relu(torch.randn(n, h, w, cin, device="cuda", generator=g))
        dy = torch.randn(n, h, w, cout, device="cuda", generator=g) * (torch.rand(n, h, w, cout, device="cuda", generator=g) > 0.5)
        wt = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * 0.05
        if args.mode == "bf16":
            DT = _lib.F32_BF16MFMA
            xb, dyb = x.bfloat16(), dy.bfloat16()
            wf, wd = ops.pack_fwd(wt, DT), ops.pack_dgrad(wt, DT)
            yb = torch.empty(n, h, w, cout, device="cuda", dtype=torch.bfloat16)
            fns = {"fwd": lambda: _lib.check(lib.osvos_conv3x3_bf16io(vp(xb.data_ptr()), 1, vp(wf.data_ptr()), None, None, 0, None, vp(yb.data_ptr()), n, h, w, cin,
                                                                       cout, cout, 1, -1, st()), "conv"),
                   "dgrad": lambda: _lib.check(lib.osvos_conv3x3_bf16io(vp(dyb.data_ptr()), 1, vp(wd.data_ptr()), None, vp(xb.data_ptr()), 1, None, vp(yb.data_ptr()), n, h,
                                                                         w, cout, cin, cin, 0, -1, st()), "dgrad"),
                   "dgrad_nomask": lambda: _lib.check(lib.osvos_conv3x3_bf16io(vp(dyb.data_ptr()), 1, vp(wd.data_ptr()), None, None, 0, None, vp(yb.data_ptr()), n, h,
                                                                                w, cout, cin, cin, 0, -1, st()), "dgrad"),
                   "wgrad": lambda: ops.conv3x3_wgrad_bf16act(xb, dyb, cin, cout)}
        else:
            w3, w3d = ops.pack_x3(wt), ops.pack_x3(wt, dgrad=True)
            fns = {"fwd": lambda: ops.conv3x3_x3(x, w3, None, cout, relu=True),
                   "dgrad": lambda: ops.conv3x3_x3(dy, w3d, None, cin, mask=x),
                   "dgrad_nomask": lambda: ops.conv3x3_x3(dy, w3d, None, cin),
                   "wgrad": lambda: ops.conv3x3_wgrad(x, dy, cin, cout, dtype=_lib.F32_X3)}
        for d in (("fwd", "dgrad", "dgrad_nomask", "wgrad") if args.nomask else ("fwd", "dgrad", "wgrad")):
            manifest.append({"layer": name, "dir": d, "gflop": gf, "ms": timed(fns[d]), "launches": REPS + 1})
    # what the matrix pipe of this box sustains (register-only loop, noise operands)
    blocks, iters = 2048, 2000
    buf = torch.empty(blocks * 512, device="cuda")
    seed = (torch.rand(128 * 8) - 0.5).to(torch.bfloat16).cuda()
    pipe = timed(lambda: _lib.check(lib.osvos_debug_mfma_peak_bf16(vp(seed.data_ptr()), vp(buf.data_ptr()), blocks, iters, st())))
    pipe_tf = blocks * 8 * iters * 8 * 2.0 * 32 * 32 * 16 / (pipe * 1e-3) / 1e12
    print(json.dumps({"mode": args.mode, "batch": n, "pipe_sustained_tflops": pipe_tf, "manifest": manifest}))


def report(args):
    man = json.load(open(args.manifest))
    mode, mult = man["mode"], (6.0 if man["mode"] == "x3" else 1.0)
    disp, ctr = [], {}
    for f in glob.glob(os.path.join(args.dir, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name") or row.get("kernel_name")
            if ("conv3x3" in k or "wgrad" in k) and "reduce" not in k and "finalize" not in k and "pack" not in k:
                disp.append((float(row["Start_Timestamp"]), row.get("Dispatch_Id") or row.get("dispatch_id"), k,
                             (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3))
    for f in glob.glob(os.path.join(args.dir, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            d = row.get("Dispatch_Id") or row.get("dispatch_id")
            c = row.get("Counter_Name") or row.get("counter_name")
            ctr.setdefault(d, {}).setdefault(c, 0.0)
            ctr[d][c] += float(row.get("Counter_Value") or row.get("counter_value") or 0)
    disp.sort()
    out = ["%s, batch %d; pipe sustained on noise operands (this box, un-profiled run): %.0f TFLOP/s; spec peak 2500" % (
        {"bf16": "bf16 tensors / bf16 MFMA (configs[2])", "x3": "fp32 tensors, f32x3 (configs[1]): executed = 6 x algorithmic FLOPs"}[mode], man["batch"],
        man["pipe_sustained_tflops"]),
        "%-8s %-5s %8s %9s %9s | %8s %8s | %9s %8s %9s  %s" % ("layer", "pass", "GFLOP", "ms", "TF/s alg", "of 2500", "of pipe", "ms (pmc)", "busy %", "clock GHz", "kernel")]
    i = 0
    for m in man["manifest"]:
        rows = disp[i:i + m["launches"]]
        i += m["launches"]
        tf = m["gflop"] / m["ms"]
        line = "%-8s %-12s %8.2f %9.3f %9.1f | %8.3f %8.3f |" % (m["layer"], m["dir"], m["gflop"], m["ms"], tf, mult * tf / 2500.0, mult * tf / man["pipe_sustained_tflops"])
        use = rows[1:]                      # (the first launch of each group is the warm-up)
        if use:
            us = sum(r[3] for r in use) / len(use)
            busy = [ctr.get(r[1], {}).get("SQ_VALU_MFMA_BUSY_CYCLES") for r in use]
            act = [ctr.get(r[1], {}).get("GRBM_GUI_ACTIVE") for r in use]
            if all(b is not None for b in busy) and all(a for a in act):
                b, a = sum(busy) / len(busy), sum(act) / len(act)
                line += " %9.3f %8.1f %9.2f  %s" % (us / 1e3, 100.0 * b / 1024.0 / (a / 8.0), a / 8.0 / (us * 1e3), use[0][2].replace("(anonymous namespace)::", "")[:60])
        out.append(line)
    out.append("busy % = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); clock = elapsed shader cycles / kernel duration under the counters;")
    out.append("'of 2500' / 'of pipe' use the EXECUTED FLOPs (x 6 for f32x3) of the un-profiled timing.")
    print("\n".join(out))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    r = sub.add_parser("run")
    r.add_argument("--mode", default="bf16", choices=["bf16", "x3"])
    r.add_argument("--batch", type=int, default=12)
    r.add_argument("--nomask", action="store_true", help="also time the data gradient without its ReLU mask (what the mask read costs)")
    r.add_argument("--all", action="store_true", help="every distinct wide trunk shape instead of conv3_2 / conv4_2")
    p = sub.add_parser("report")
    p.add_argument("dir")
    p.add_argument("manifest")
    a = ap.parse_args()
    run(a) if a.cmd == "run" else report(a)
