#!/bin/bash
# round 4 validation: full GPU tier, bench line, settled rocprof summaries, step-level PMC traffic for configs[1] and configs[2]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04j; mkdir -p $O; export TMPDIR=/tmp
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > $O/env.log 2>&1; nproc >> $O/env.log; lscpu | grep "Model name" >> $O/env.log
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=15 > $O/pytest_gpu.log 2>&1; echo "exit $?" >> $O/pytest_gpu.log; grep -E "passed|failed|FAILED|Error|^exit" $O/pytest_gpu.log | tail -25
bash tools/gpu_pmc_step.sh r04j_pmc1 "configs[1]: 854x480 batch 1 online loop, fp32x3" > $O/pmc1.txt 2>&1; tail -c 1500 $O/pmc1.txt
bash tools/gpu_pmc_step.sh r04j_pmc2 "configs[2]: 854x480 batch 12 parent loop, bf16" --mode parent --precision bf16 --batch 12 > $O/pmc2.txt 2>&1; tail -c 1500 $O/pmc2.txt
