#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04b; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q --tb=short -p no:cacheprovider -x -k "streamk or cbce or input_gradient" > $O/pytest_a.log 2>&1; echo "exit $?" >> $O/pytest_a.log; tail -8 $O/pytest_a.log
timeout 600 python tools/tune_streamk.py --grids 256,224 > $O/tune_streamk_b1.txt 2>&1; cat $O/tune_streamk_b1.txt
for m in 0 1 0 1; do echo "== OSVOS_X3_STREAMK=$m"; OSVOS_X3_STREAMK=$m timeout 300 python tools/bench_fields.py --no-extra --no-cpu-baseline --steps 20 --warmup 5; done 2>&1 | tee $O/bench_ab.txt
echo "== bf16 b12 new input gradient"; timeout 300 python tools/bench_fields.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --mode parent --precision bf16 --batch 12 2>&1 | tee -a $O/bench_ab.txt
echo "== bf16 b12 old input gradient"; OSVOS_TMP_NO_C3B=1 timeout 300 python tools/bench_fields.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --mode parent --precision bf16 --batch 12 2>&1 | tee -a $O/bench_ab.txt
echo "== bf16 b12 new input gradient"; timeout 300 python tools/bench_fields.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --mode parent --precision bf16 --batch 12 2>&1 | tee -a $O/bench_ab.txt
timeout 600 python -m pytest tests/test_gpu_net.py tests/test_gpu_baseline_configs.py -q --tb=short -p no:cacheprovider -x -k "bf16" > $O/pytest_bf16.log 2>&1; echo "exit $?" >> $O/pytest_bf16.log; tail -8 $O/pytest_bf16.log
