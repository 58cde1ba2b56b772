#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r04c; mkdir -p $O; export TMPDIR=/tmp
for o in 0 1; do echo "== ORDER=$o"; OSVOS_X3_STREAMK_ORDER=$o timeout 600 python tools/tune_streamk.py --grids 256 --dgrad 0; done > $O/tune_streamk_order.txt 2>&1; cat $O/tune_streamk_order.txt
