#!/bin/bash
# what do LDS reads / VALU work beside the MFMAs cost in sustained matrix throughput?  (register-only loop + ds_read_b128 / VALU mixes, noise operands)
set -u
mkdir -p gpurun_out/mfma3
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mfma3
tools/native/bin/mfma_probe noise 8 mix >> $O/probe.txt 2>&1
tools/native/bin/mfma_probe zero 8 mix >> $O/probe.txt 2>&1
cat $O/probe.txt
cd /tmp; export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/pmc_noise/p1 -o p1 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -- $R/tools/native/bin/mfma_probe noise 8 mix > $O/pmc_noise.log 2>&1
cd $R; python - <<PY
import csv, glob
from collections import defaultdict
v = defaultdict(lambda: defaultdict(float)); dur = defaultdict(float); cnt = defaultdict(int)
for f in glob.glob("gpurun_out/mfma3/pmc_noise/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        v[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"])
for f in glob.glob("gpurun_out/mfma3/pmc_noise/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Kernel_Name"]] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3; cnt[row["Kernel_Name"]] += 1
for k in sorted(v):
    us = dur[k]
    if "mfma" in k:
        print("noise: %-44s clock %.2f GHz  MFMA busy %.1f %%" % (k[:44], v[k]["GRBM_GUI_ACTIVE"] / 8 / (us * 1e3), 100 * v[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (v[k]["GRBM_GUI_ACTIVE"] / 8)))
PY
