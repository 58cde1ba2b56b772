#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -s -k "wgrad_f32x3" > gpurun_out/pytest_o.log 2>&1; grep -E "passed|failed|FAILED|Error" gpurun_out/pytest_o.log | tail -5
OSVOS_X3_WGRAD_WAVES=4 timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q --tb=short -p no:cacheprovider -k "wgrad_f32x3" 2>&1 | tail -1
for wv in 8 4; do
OSVOS_X3_WGRAD_WAVES=$wv timeout 600 python tools/tune_x3.py --tiles 210 --wgrad-only > gpurun_out/tune_wg_$wv.log 2>&1; echo "waves $wv"; grep "wgrad" gpurun_out/tune_wg_$wv.log | cut -c1-200
OSVOS_X3_WGRAD_WAVES=$wv timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench waves $wv:', d['value'], d['sustained']['value'], d['roofline']['families'])"
done
