#!/usr/bin/env python3
"""HBM-side bytes PER STEP of every kernel of the real training step, from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs,
--kernel-trace only) over `bench.py` itself -- what the network launches at HEAD (pre-split packs, one-bit ReLU masks, fused pools), not a
probe that re-creates single layers.  Corrections as MI355X_MICROARCH.md prescribes: counter values are KB; FETCH_SIZE counts wide coalesced
loads at half on gfx950 (x 2).  Totals are divided by the number of steps the profiled process ran (settle + warm-up + timed, read from the
bench line in its log).  Writes the JSON bench.py attaches as roofline.traffic for that precision, with the commit.

usage: pmc_step_traffic.py <dir with fetch/ write/ sub-directories and fetch.log / write.log> <out.json> <label>"""
import csv
import glob
import json
import os
import re
import subprocess
import sys
from collections import defaultdict

root, out, label = sys.argv[1], sys.argv[2], sys.argv[3]


def steps_of(log):
    line = [l for l in open(log).read().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    return d["steps"] + d["warmup"] + d["setup_settle_steps"], d


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*\)$", "", name)[:90]


tot = {}
line = None
for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    nsteps, line = steps_of(os.path.join(root, sub + ".log"))
    acc, calls = defaultdict(float), defaultdict(set)
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                k = short(row["Kernel_Name"])
                acc[k] += float(row["Counter_Value"])
                calls[k].add(row["Dispatch_Id"])
    for k, v in acc.items():
        e = tot.setdefault(k, {"launches_per_step": 0.0, "fetch_MB_per_step": 0.0, "write_MB_per_step": 0.0})
        e["launches_per_step"] = round(len(calls[k]) / nsteps, 2)
        e["fetch_MB_per_step" if counter == "FETCH_SIZE" else "write_MB_per_step"] = round(v * (2.0 if counter == "FETCH_SIZE" else 1.0) / 1e3 / nsteps, 1)
for e in tot.values():
    e["hbm_MB_per_step"] = round(e["fetch_MB_per_step"] + e["write_MB_per_step"], 1)
order = sorted(tot.items(), key=lambda kv: -kv[1]["hbm_MB_per_step"])
conv = {k: v for k, v in tot.items() if k.startswith(("conv3x3", "wgrad_f32x3", "wgrad_bf16", "wgrad_f32_", "wgrad_c3", "wgrad_reduce", "conv_splitk", "dgrad_c3"))}
# minimum conv tensor traffic 904 MB fp32 / 452 MB bf16 per forward frame (every conv reads its input once, writes its output
# once, reads its weights once); the backward moves each tensor twice more (data gradient: dy in, dx out; weight gradient: x in, dy in)
cfg = line["config"]["workload"]
batch = int(re.search(r"batch=(\d+)", cfg).group(1))
bf16 = "bf16" in line["dtype"][:8]
wh = re.match(r"(\d+)x(\d+)", cfg)
scale = (int(wh.group(1)) * int(wh.group(2))) / (854.0 * 480.0) if wh else 1.0
passes = 1.0 if "inference" in cfg else 3.0          # forward only / forward + data gradient + weight gradient
if "WINDOW-FUSED" in cfg:                            # one step = the whole window: nAveGrad x batch frames
    m = re.search(r"the (\d+) micro-batches", cfg)
    batch = int(m.group(1)) if m else batch
alg = (452.0 if bf16 else 904.0) * passes * batch * scale
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=os.path.dirname(os.path.abspath(__file__))).decode().strip()
except Exception:
    commit = os.environ.get("OSVOS_COMMIT", "unknown")
doc = {"label": label, "measured_at_commit": commit, "workload": cfg,
       "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over bench.py; KB -> MB, FETCH_SIZE x 2 (gfx950 wide loads); per step",
       "conv_family": {"hbm_MB_per_step": round(sum(v["hbm_MB_per_step"] for v in conv.values()), 1),
                       "algorithmic_MB_per_step": round(alg, 1),
                       "ratio": round(sum(v["hbm_MB_per_step"] for v in conv.values()) / alg, 3)},
       "all_kernels_hbm_MB_per_step": round(sum(v["hbm_MB_per_step"] for v in tot.values()), 1),
       "per_kernel": {k: v for k, v in order[:24]}}
with open(out, "w") as fh:
    json.dump(doc, fh, indent=1)
print(json.dumps(doc, indent=1)[:6000])
