#!/bin/bash
# does the clock follow the matrix pipe's occupancy?  register-only MFMA loop at several duty cycles (one wave per SIMD), noise and zero operands, clock + occupancy from PMC
set -u
mkdir -p gpurun_out/mfma2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mfma2
for d in noise zero; do tools/native/bin/mfma_probe $d 4 duty >> $O/probe.txt 2>&1; done
cat $O/probe.txt
cd /tmp; export TMPDIR=/tmp
for d in noise zero; do
  timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/pmc_$d/p1 -o p1 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -- $R/tools/native/bin/mfma_probe $d 4 duty > $O/pmc_$d.log 2>&1
  (cd $R; python - <<PY
import csv, glob
from collections import defaultdict
names = {}; v = defaultdict(lambda: defaultdict(float)); dur = defaultdict(float); cnt = defaultdict(int)
for f in glob.glob("gpurun_out/mfma2/pmc_$d/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        v[row["Kernel_Name"]][row["Counter_Name"]] += float(row["Counter_Value"])
for f in glob.glob("gpurun_out/mfma2/pmc_$d/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        dur[row["Kernel_Name"]] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3; cnt[row["Kernel_Name"]] += 1
for k in sorted(v):
    us = dur[k]
    print("$d: %-40s %2d launches  clock %.2f GHz  MFMA busy %.1f %%" % (k[:40], cnt[k], v[k]["GRBM_GUI_ACTIVE"] / 8 / (us * 1e3), 100 * v[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (v[k]["GRBM_GUI_ACTIVE"] / 8)))
PY
)
done
