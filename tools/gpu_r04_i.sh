#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04i; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_trained_like.py -q --tb=short -p no:cacheprovider -s > $O/pytest_trained.log 2>&1; echo "exit $?" >> $O/pytest_trained.log; grep "trained-like\|window-fused\|worst\|one vector\|passed\|failed\|^E  \|20-step" $O/pytest_trained.log | cut -c1-500
