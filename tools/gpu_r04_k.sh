#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04k; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_scripts.py -q --tb=short -p no:cacheprovider > $O/pytest_scripts.log 2>&1; echo "exit $?" >> $O/pytest_scripts.log; tail -6 $O/pytest_scripts.log
bash tools/scripts_e2e.sh r04k_e2e 2>&1 | tee $O/scripts_e2e.txt
