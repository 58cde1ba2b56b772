#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04g; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python tools/make_trained_fixture.py --oracle-steps 30 --write 2>&1 | grep -v "amdgpu.ids\|Constructing\|Initializing\|UserWarning\|Consider using\|acc +=" | tee $O/make_fixture.txt
cp tests/golden/trained_like.json $O/
timeout 1500 python -m pytest tests/test_gpu_trained_like.py -q --tb=short -p no:cacheprovider -s > $O/pytest_trained.log 2>&1; echo "exit $?" >> $O/pytest_trained.log; grep -v "amdgpu.ids\|Constructing\|Initializing" $O/pytest_trained.log | tail -40
