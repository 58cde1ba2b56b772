#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04e; mkdir -p $O; export TMPDIR=/tmp
for lr in 1e-7 3e-7 1e-6 3e-6; do echo "== lr $lr"; timeout 300 python tools/make_trained_fixture.py --lr $lr --steps 200 2>&1 | grep -v "amdgpu.ids\|Constructing\|Initializing"; done | tee $O/fixture_search.txt
