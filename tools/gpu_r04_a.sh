#!/bin/bash
# round 4, call A: correctness of the pruned tree + stream-K + bf16-in input gradient, then the stream-K A/Bs
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04a; mkdir -p $O; export TMPDIR=/tmp
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > $O/env.log 2>&1
timeout 900 python -m pytest tests/test_gpu_ops.py -q --tb=short -p no:cacheprovider -x -k "streamk or skinny_side_prep or input_gradient or head_lowres or x3_is_fp32_grade" > $O/pytest_a.log 2>&1; echo "exit $?" >> $O/pytest_a.log; tail -5 $O/pytest_a.log
timeout 900 python -m pytest tests/test_gpu_net.py -q --tb=short -p no:cacheprovider -x -k "golden or pooling_fused or one_bit or smoke or dropin" > $O/pytest_b.log 2>&1; echo "exit $?" >> $O/pytest_b.log; tail -5 $O/pytest_b.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python tools/tune_streamk.py > $O/tune_streamk_b1.txt 2>&1; cat $O/tune_streamk_b1.txt
for m in 0 1 2; do echo "== OSVOS_X3_STREAMK=$m"; OSVOS_X3_STREAMK=$m timeout 300 python tools/bench_fields.py --no-extra --no-cpu-baseline --steps 20 --warmup 5; done 2>&1 | tee $O/bench_ab.txt
for g in 248 240; do echo "== STREAMK=1 GRID=$g"; OSVOS_X3_STREAMK_GRID=$g timeout 300 python tools/bench_fields.py --no-extra --no-cpu-baseline --steps 20 --warmup 5; done 2>&1 | tee -a $O/bench_ab.txt
for l in 3 12; do echo "== STREAMK=1 MIN_LOSS=$l"; OSVOS_X3_STREAMK_MIN_LOSS=$l timeout 300 python tools/bench_fields.py --no-extra --no-cpu-baseline --steps 20 --warmup 5; done 2>&1 | tee -a $O/bench_ab.txt
echo "== bf16 b12"; timeout 300 python tools/bench_fields.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --mode parent --precision bf16 --batch 12 2>&1 | tee -a $O/bench_ab.txt
