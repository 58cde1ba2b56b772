#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04l; mkdir -p $O; export TMPDIR=/tmp
for init in calibrated reference; do PROBE_INIT=$init timeout 600 python tools/feeder_probe.py bf16 2>&1 | grep -v "amdgpu.ids\|Constructing\|Initializing"; done | tee $O/feeder_probe.txt
timeout 300 python tools/bench_fields.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --mode parent --precision bf16 --batch 1 | cut -c1-150 | tee -a $O/feeder_probe.txt
