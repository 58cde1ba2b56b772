#!/usr/bin/env python3
"""HBM-side bytes per launch of the dominant kernels from rocprofv3 --pmc passes (tools/gpu_pmc_x3.sh: FETCH_SIZE and WRITE_SIZE in
their own runs), corrected as MI355X_MICROARCH.md prescribes (KB units; FETCH_SIZE counts wide coalesced loads at half on gfx950 -> x2),
written as profiles/r02_pmc_traffic.json together with the commit it was measured at -- the file bench.py's roofline.traffic loads.
usage: pmc_traffic.py <pmc dir> <out.json>"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

root, out = sys.argv[1], sys.argv[2]
per = defaultdict(lambda: defaultdict(list))        # kernel -> counter -> per-dispatch sums, in dispatch order
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    acc, names, order = defaultdict(float), {}, []
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k, c, d = row["Kernel_Name"], row["Counter_Name"], int(row["Dispatch_Id"])
            if c not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            if (d, c) not in acc:
                order.append((d, c))
            acc[(d, c)] += float(row["Counter_Value"])
            names[d] = k
    for d, c in sorted(order):
        per[names[d]][c].append(acc[(d, c)])


def find(sub):
    for k in per:
        if sub in k:
            return per[k]
    return None


# dispatch order of tools/pmc_probe_x3.py: per shape 3 x (forward, data gradient, weight gradient); shape 0 = conv3_2-like, 1 = conv1_2-like
ALG = {  # algorithmic MB per launch: input + output (+ mask for the data gradient) + weights once, fp32
    "conv3_2 fwd": 26.3 + 26.3 + 2.36, "conv3_2 dgrad": 26.3 + 26.3 + 26.3 + 2.36, "conv3_2 wgrad": 26.3 + 26.3 + 2.36,
    "conv1_2 fwd": 104.9 + 104.9 + 0.15, "conv1_2 dgrad": 104.9 * 3 + 0.15, "conv1_2 wgrad": 104.9 * 2 + 0.15,
}
res = {}
for kname, sub, shapes in (("conv3x3_f32x3_kernel", "CfgX<32, 1, 8, 4, 4, 2, 1, 1>", ["conv3_2"]), ("conv3x3_f32x3_kernel", "CfgX<32, 1, 8, 2, 4, 2, 1, 1>", ["conv1_2"])):
    v = find(sub)
    if not v:
        continue
    f, w = v.get("FETCH_SIZE", []), v.get("WRITE_SIZE", [])
    for j, role in enumerate(("fwd", "dgrad")):
        fj, wj = f[j::2], w[j::2]
        if fj and wj:
            key = "%s %s" % (shapes[0], role)
            res["%s, %s" % (kname, key)] = {"hbm_fetch_MB": round(sum(fj) / len(fj) * 2 / 1e3, 1), "hbm_write_MB": round(sum(wj) / len(wj) / 1e3, 1),
                                            "algorithmic_MB": round(ALG[key], 1)}
v = find("wgrad_f32x3_kernel")
if v:
    f, w = v.get("FETCH_SIZE", []), v.get("WRITE_SIZE", [])
    for j, shape in enumerate(("conv3_2", "conv1_2")):
        fj, wj = f[3 * j:3 * j + 3], w[3 * j:3 * j + 3]
        if fj and wj:
            res["wgrad_f32x3_kernel, %s wgrad" % shape] = {"hbm_fetch_MB": round(sum(fj) / len(fj) * 2 / 1e3, 1), "hbm_write_MB": round(sum(wj) / len(wj) / 1e3, 1),
                                                           "algorithmic_MB": round(ALG["%s wgrad" % shape], 1)}
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(os.path.abspath(__file__))).decode().strip()
except Exception:
    commit = "unknown"
doc = {"measured_at_commit": commit, "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only), KB -> MB, FETCH_SIZE x 2 (gfx950 wide loads)",
       "workload": "tools/pmc_probe_x3.py: 120x214 256->256 (conv3_2) and 480x854 64->64 (conv1_2), batch 1, per-launch averages of 3 launches",
       "per_launch": res}
with open(out, "w") as fh:
    json.dump(doc, fh, indent=1)
print(json.dumps(doc, indent=1))
