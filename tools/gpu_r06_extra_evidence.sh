#!/bin/bash
# round-6 extra evidence (not part of the gate): every measured error of the trained-like tests per precision; entry-point throughput incl. the two-piece precisions
cd "$(dirname "$0")/.."; O=gpurun_out/r06_extra; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_trained_like.py -q -s -p no:cacheprovider 2>&1 | grep -v "amdgpu.ids\|Constructing\|Initializing" > $O/trained_like.txt; tail -3 $O/trained_like.txt
timeout 1500 bash tools/scripts_e2e.sh r06_extra/e2e > $O/scripts_e2e.txt 2>&1; tail -30 $O/scripts_e2e.txt
export OSVOS_SAVE_ROOT=$PWD/$O/save OSVOS_MODELS_DIR=$PWD/$O/save; mkdir -p $O/save
for p in fp32x3b2 fp32x3h2; do
  SEQ_NAME=blackswan timeout 600 python train_online.py --synthetic --device-augment --epochs 3000 --precision $p > $O/online_$p.log 2>&1
  python - <<PY >> $O/scripts_e2e.txt
import re
t = float(re.search(r"Online training time: ([0-9.]+)", open("$O/online_$p.log").read()).group(1))
print("train_online.py --precision $p: 3000 micro-batches in %.3f s = %.1f frames/s (the script's own timer)" % (t, 3000 / t))
PY
done
tail -3 $O/scripts_e2e.txt
