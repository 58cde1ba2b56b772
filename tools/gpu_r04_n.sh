cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "pack_x3 or f32x3" 2>&1 | tail -3 > $O/pytest.txt
timeout 900 python -m pytest tests/test_augment.py tests/test_gpu_scripts.py tests/test_gpu_golden.py -x -q 2>&1 | tail -3 >> $O/pytest.txt
timeout 300 python tools/bench_fields.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 | cut -c1-110 > $O/bench.txt
timeout 300 python tools/bench_fields.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 --mode parent --precision bf16 --batch 12 | cut -c1-150 >> $O/bench.txt
export OSVOS_SAVE_ROOT=/tmp/pp OSVOS_MODELS_DIR=/tmp/pp; mkdir -p /tmp/pp
for p in fp32x3 bf16; do echo "== train_parent.py --synthetic 512 --device-augment --epochs 4 --precision $p" >> $O/bench.txt; timeout 600 python train_parent.py --synthetic 512 --device-augment --epochs 4 --precision $p 2>&1 | grep "Execution" | tr '\n' ' ' >> $O/bench.txt; echo >> $O/bench.txt; done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o p -- python $GRAFT_REPO_ROOT/train_parent.py --synthetic 150 --device-augment --epochs 2 > /dev/null 2>&1)
python - <<'PY' >> gpurun_out/r04r/bench.txt
import csv, glob
f = glob.glob("gpurun_out/r04r/prof/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:40]:
    if any(k in r["Name"] for k in ("label", "augment", "pack_x3", "sgd", "fillBuffer", "copyBuffer")): print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
rm -rf $O/prof
