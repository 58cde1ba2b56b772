#!/bin/bash
# Entry-point throughput: the two scripts at 854x480 with the device input pipeline, frames/s from their OWN timers, next
# to bench.py's figure for the same loop on the same box; then a kernel trace of train_online.py with three steps' timeline.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/${1:-e2e}; mkdir -p $O; export TMPDIR=/tmp
export OSVOS_SAVE_ROOT=$O/save OSVOS_MODELS_DIR=$O/save; mkdir -p $O/save
fps() { python - "$@" <<'PY'
import re, sys
kind, path, frames = sys.argv[1], sys.argv[2], float(sys.argv[3])
txt = open(path).read()
if kind == "online":
    t = float(re.search(r"Online training time: ([0-9.]+)", txt).group(1))
    print("train_online.py: %d micro-batches in %.3f s = %.1f frames/s (the script's own timer, incl. its first un-warmed iterations)" % (frames, t, frames / t))
else:
    ts = [float(x) for x in re.findall(r"Execution time: ([0-9.]+)", txt)]
    print("train_parent.py: %d frames per epoch; epochs took %s s -> last epoch %.1f frames/s (the script's own timer)" % (frames, ["%.3f" % t for t in ts], frames / ts[-1]))
PY
}
echo "== train_online.py --synthetic --device-augment --epochs 1000 (854x480, fp32x3)"
SEQ_NAME=blackswan timeout 600 python train_online.py --synthetic --device-augment --epochs 1000 > $O/online.log 2>&1; fps online $O/online.log 1000
SEQ_NAME=blackswan timeout 600 python train_online.py --synthetic --device-augment --epochs 3000 > $O/online3k.log 2>&1; fps online $O/online3k.log 3000
echo "== train_online.py --synthetic --device-augment --window-fused --epochs 3000"
SEQ_NAME=blackswan timeout 600 python train_online.py --synthetic --device-augment --window-fused --epochs 3000 > $O/online_win.log 2>&1; fps online $O/online_win.log 3000
echo "== bench.py (same loop, frame resident, no input pipeline)"
timeout 300 python tools/bench_fields.py --no-extra --no-cpu-baseline --no-parity --steps 20 --warmup 5 | cut -c1-110
timeout 300 python tools/bench_fields.py --no-extra --no-cpu-baseline --no-parity --steps 20 --warmup 5 --window-fused 1 | cut -c1-130
echo "== train_parent.py --synthetic 512 --device-augment --precision bf16 --epochs 4 (854x480, batch 1 per micro-batch like the reference)"
timeout 900 python train_parent.py --synthetic 512 --device-augment --precision bf16 --epochs 4 > $O/parent.log 2>&1; fps parent $O/parent.log 512
echo "== train_parent.py --synthetic 512 --device-augment --epochs 3 (fp32x3)"
timeout 900 python train_parent.py --synthetic 512 --device-augment --epochs 3 > $O/parent_x3.log 2>&1; fps parent $O/parent_x3.log 512
echo "== bench.py --mode parent --precision bf16 --batch 1"
timeout 300 python tools/bench_fields.py --no-extra --no-cpu-baseline --no-parity --steps 20 --warmup 5 --mode parent --precision bf16 --batch 1 | cut -c1-150
timeout 300 python tools/bench_fields.py --no-extra --no-cpu-baseline --no-parity --steps 20 --warmup 5 --mode parent --batch 1 | cut -c1-130
echo "== kernel trace of train_online.py --device-augment (120 iterations): three steps"
(cd /tmp && SEQ_NAME=blackswan timeout 600 rocprofv3 --kernel-trace -d $O/prof -o online -- python $R/train_online.py --synthetic --device-augment --epochs 120 > $O/rocprof.log 2>&1)
DB=$(find $O/prof -name "*.db" | head -1)
python tools/step_timeline.py $DB 100 0 140 1400 1800 4000 6000 2>&1 | cut -c1-150
rm -rf $O/prof $O/save
