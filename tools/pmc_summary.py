#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (one directory per pass) into per-kernel, per-launch averages.
usage: pmc_summary.py <dir with p1/ p2/ ... subdirectories>.  Derived columns follow MI355X_MICROARCH.md / profiles/r01_pmc_*:
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); FETCH_SIZE x 2 on gfx950 (wide loads count half), KB units."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
vals = defaultdict(lambda: defaultdict(list))       # kernel -> counter -> [per-dispatch values]
dur = defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per_dispatch = defaultdict(float)
    names = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name") or row.get("kernel_name")
            c = row.get("Counter_Name") or row.get("counter_name")
            v = float(row.get("Counter_Value") or row.get("counter_value") or 0)
            d = row.get("Dispatch_Id") or row.get("dispatch_id")
            per_dispatch[(d, c)] += v          # counters arrive per XCD / per instance: sum them
            names[d] = k
    for (d, c), v in per_dispatch.items():
        vals[names[d]][c].append(v)
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name") or row.get("kernel_name")
            try:
                dur[k].append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3)
            except Exception:
                pass


def short(k):
    for a, b in (("void (anonymous namespace)::", ""), ("(anonymous namespace)::", "")):
        k = k.replace(a, b)
    return k[:70]


counters = sorted({c for k in vals for c in vals[k]})
print("per-launch averages (sum over XCDs / instances); us = kernel-trace duration under the counters")
for k in sorted(vals, key=lambda k: -sum(dur.get(k, [0]))):
    if not any(s in k for s in ("conv3x3", "wgrad")):
        continue
    a = {c: sum(v) / len(v) for c, v in vals[k].items()}
    us = sum(dur[k]) / len(dur[k]) if dur.get(k) else float("nan")
    line = "%-70s n=%-3d us=%7.1f" % (short(k), max(len(v) for v in vals[k].values()), us)
    if "GRBM_GUI_ACTIVE" in a:
        line += "  clock=%.2f GHz" % (a["GRBM_GUI_ACTIVE"] / 8 / (us * 1e3)) if us == us else ""
    print(line)
    print("    " + "  ".join("%s=%.4g" % (c, a[c]) for c in counters if c in a))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "GRBM_GUI_ACTIVE" in a:
        print("    MFMA busy = %.1f %% of elapsed cycles (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs))"
              % (100.0 * a["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (a["GRBM_GUI_ACTIVE"] / 8.0)))
    if "FETCH_SIZE" in a or "WRITE_SIZE" in a:
        print("    HBM-side traffic per launch: fetch %.1f MB (FETCH_SIZE KB x 2: wide loads count half on gfx950), write %.1f MB"
              % (a.get("FETCH_SIZE", float("nan")) * 2 / 1e3, a.get("WRITE_SIZE", float("nan")) / 1e3))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "SQ_BUSY_CYCLES" in a:
        # SQ_BUSY_CYCLES is summed over the shader engines: use wave-cycle based shares for the stall split
        wc = a.get("SQ_WAVE_CYCLES", float("nan"))
        print("    issue-stall share of wave cycles (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) = %.2f, waitcnt+barrier (SQ_WAIT_ANY) = %.2f, LDS conflict share of LDS cycles = %.3f"
              % (a.get("SQ_WAIT_INST_ANY", float("nan")) / wc, a.get("SQ_WAIT_ANY", float("nan")) / wc,
                 a.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(a.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0)))
