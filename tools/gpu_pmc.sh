#!/bin/bash
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmc/p1 -o p1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python $R/tools/pmc_probe.py > $R/gpurun_out/pmc/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmc/p2 -o p2 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -- python $R/tools/pmc_probe.py > $R/gpurun_out/pmc/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmc/p3 -o p3 --pmc WRITE_SIZE -- python $R/tools/pmc_probe.py > $R/gpurun_out/pmc/p3.log 2>&1
cd $R; find gpurun_out/pmc -name "*.csv" | head; tail -3 gpurun_out/pmc/p1.log
