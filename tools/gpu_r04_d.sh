#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04d; mkdir -p $O; export TMPDIR=/tmp
for m in 0 1 0 1 0 1; do echo "== OSVOS_X3_STREAMK=$m"; OSVOS_X3_STREAMK=$m timeout 300 python tools/bench_fields.py --no-extra --no-cpu-baseline --steps 20 --warmup 5; done 2>&1 | cut -c1-120 | tee $O/bench_ab.txt
for m in 0 1 0 1; do echo "== 1080p infer OSVOS_X3_STREAMK=$m"; OSVOS_X3_STREAMK=$m timeout 300 python tools/bench_fields.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --mode infer --height 1080 --width 1920 --batch 4 --graph 1; done 2>&1 | cut -c1-160 | tee -a $O/bench_ab.txt
