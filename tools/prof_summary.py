#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace into the text summary kept under profiles/.
usage: prof_summary.py <results.db> <steps | 0 = count them: one cbce_count launch per step> [out.txt] [command text]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2])
rows = list(db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
if steps <= 0:      # (round 4: a run with a settle phase profiles far more steps than --steps + --warmup; a wrong divisor scaled every per-step column)
    steps = float(max([r[1] for r in rows if "cbce_count_kernel" in r[0]] or [1]))
cmd = sys.argv[4] if len(sys.argv) > 4 else "python bench.py --steps 20 --warmup 5 --no-cpu-baseline"
out = ["rocprofv3 --kernel-trace --stats  (command: %s; %d profiled steps incl. settle / warm-up steps)" % (cmd, steps),
       "%-96s %7s %12s %10s %6s" % ("kernel", "calls", "us/step", "avg_us", "%")]
# bench.py's one-off probes (the register-only MFMA loop behind roofline.pipe_sustained runs once AFTER the timed regions) are listed but kept
# out of the per-step shares
ONE_OFF = ("mfma_peak_bf16_kernel", "mfma_peak_kernel")
step_total = sum(r[2] for r in rows if not any(o in r[0] for o in ONE_OFF))
for n, c, tot, avg, pct in rows:
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    if any(o in n for o in ONE_OFF):
        out.append("%-96s %7d %12s %10.1f %6s   <- one-off probe after the timed region (roofline.pipe_sustained), not part of the steps" % (n[:96], c, "-", avg, "-"))
    else:
        out.append("%-96s %7d %12.1f %10.1f %6.2f" % (n[:96], c, tot / steps, avg, 100.0 * tot / step_total))
out.append("total kernel time per step: %.1f us" % (step_total / steps))

# forward / backward split of the convolution launches (the same kernel runs both passes): a step's forward is everything between
# its NCHW->NHWC conversion and its first loss kernel.  This is the row to hold against bench.py's roofline.families.conv_fwd
# (hipEvent-timed, un-profiled): ms_per_step / launches there = the average forward launch duration here.
try:
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    ks = [t for t in tabs if "kernel_symbol" in t.lower()][-1]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
    namecol = "kernel_name" if "kernel_name" in cols else cols[-1]
    disp = list(db.execute("select s.%s, d.start, d.end, d.stream_id from rocpd_kernel_dispatch d join %s s on d.kernel_id = s.id order by d.start" % (namecol, ks)))
    bf16_run = any("conv3x3_bf16" in d[0] for d in disp)       # a bf16 run's fp32 conv launches are bench.py's calibration forward
    x3_run = any("conv3x3_f32x3" in d[0] for d in disp)           # default precision: f32x3 kernels + the exact one for conv1_1
    conv_names = ("conv3x3_bf16_kernel", "conv3x3_bf16_dma_kernel") if bf16_run else (("conv3x3_f32x3_kernel", "conv3x3_f32_kernel") if x3_run else ("conv3x3_f32_kernel",))
    phase, acc, main = None, {"fwd": [], "bwd": []}, {"fwd": [], "bwd": []}
    main_stream = None
    for name, t0, t1, stream in disp:
        if "nchw_to_nhwc" in name:
            phase, main_stream = "fwd", stream
        elif "cbce_count" in name:
            phase = "bwd"
        elif phase and any(c in name for c in conv_names):
            acc[phase].append((t1 - t0) / 1e3)
            if stream == main_stream:
                main[phase].append((t1 - t0) / 1e3)
    for ph, label in (("fwd", "forward"), ("bwd", "backward (data gradient, overlapped with the weight-gradient stream)")):
        if acc[ph]:
            out.append("conv3x3 launches in the %s: %d (%.1f per step), average %.1f us, %.1f us per step"
                       % (label, len(acc[ph]), len(acc[ph]) / steps, sum(acc[ph]) / len(acc[ph]), sum(acc[ph]) / steps))
            out.append("    of which on the main stream (the side-branch 3x3 convs run beside them on a second stream): %d, average %.1f us, %.1f us per step"
                       % (len(main[ph]), sum(main[ph]) / max(1, len(main[ph])), sum(main[ph]) / steps))
except Exception as e:      # older trace layouts: keep the plain table
    out.append("(forward/backward split unavailable: %s)" % e)
text = "\n".join(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(text + "\n")
print(text)
