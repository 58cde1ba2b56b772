#!/bin/bash
# the bf16 matrix pipe's sustained rate on this chip: register-only MFMA loop, zero / constant / noise operands, with the clock from PMC
set -u
mkdir -p gpurun_out/mfma
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mfma
for d in zero const noise zero noise; do tools/native/bin/mfma_probe $d 8 >> $O/probe.txt 2>&1; done
tools/native/bin/mfma_probe noise 4 >> $O/probe.txt 2>&1
cat $O/probe.txt
cd /tmp; export TMPDIR=/tmp
for d in zero noise; do
  timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/pmc_$d/p1 -o p1 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -- $R/tools/native/bin/mfma_probe $d 8 > $O/pmc_$d.log 2>&1
  (cd $R; python - <<PY
import csv, glob
from collections import defaultdict
v = defaultdict(float); n = 0
for f in glob.glob("gpurun_out/mfma/pmc_$d/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        v[row["Counter_Name"]] += float(row["Counter_Value"])
d0 = []
for f in glob.glob("gpurun_out/mfma/pmc_$d/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        d0.append((float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3)
us = sum(d0)
print("$d: %d launches, %.0f us total, clock %.2f GHz, MFMA busy %.1f %%" % (len(d0), us, v["GRBM_GUI_ACTIVE"] / 8 / (us * 1e3), 100 * v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (v["GRBM_GUI_ACTIVE"] / 8)))
PY
)
done
