#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04h; mkdir -p $O; export TMPDIR=/tmp
for v in "--lr 5e-8 --steps 300" "--lr 3e-8 --steps 300" "--lr 5e-8 --steps 300 --warm-factor 0.5" "--lr 7e-8 --steps 300"; do echo "== $v"; timeout 300 python tools/make_trained_fixture.py $v 2>&1 | grep -v "amdgpu.ids\|Constructing\|Initializing"; done | tee $O/fixture_search2.txt
