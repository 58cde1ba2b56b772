cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04u; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "input_gradient" 2>&1 | tail -4 > $O/pytest.txt
timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_trained_like.py -x -q -k "bf16" 2>&1 | tail -4 >> $O/pytest.txt
for r in 1 2; do for v in 0 1; do echo "OSVOS_BF16_DX_MMA=$v" >> $O/bench.txt; OSVOS_BF16_DX_MMA=$v timeout 300 python tools/bench_fields.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --mode parent --precision bf16 --batch 12 | cut -c1-150 >> $O/bench.txt; done; done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o b -- python $GRAFT_REPO_ROOT/bench.py --no-extra --no-cpu-baseline --no-prof --min-seconds 0 --steps 10 --warmup 5 --mode parent --precision bf16 --batch 12 > /dev/null 2>&1)
python tools/step_timeline.py $(find $O/prof -name "*.db" | head -1) 10 10000 13000 2>&1 | cut -c1-150 | tail -22 >> $O/bench.txt
rm -rf $O/prof
