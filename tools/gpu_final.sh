#!/bin/bash
# end-of-session validation: full GPU test tier, smoke, the bench lines kept under profiles/
mkdir -p gpurun_out/final
timeout 200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/final/pytest.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee gpurun_out/final/smoke.txt
timeout 150 python bench.py 2>&1 | tail -1 > gpurun_out/final/bench_fp32_online.json
timeout 60 python bench.py --precision bf16 --mode parent --batch 12 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/final/bench_bf16_parent_b12.json
timeout 60 python bench.py --precision bf16 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/final/bench_bf16_online_b1.json
cut -c1-260 gpurun_out/final/bench_*.json
