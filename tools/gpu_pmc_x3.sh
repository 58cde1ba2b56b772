#!/bin/bash
# rocprofv3 PMC passes on the f32x3 kernels (separate runs per counter group, --kernel-trace only).
mkdir -p gpurun_out/pmc_x3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_x3/p1 -o p1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python $R/tools/pmc_probe_x3.py > $R/gpurun_out/pmc_x3/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_x3/p2 -o p2 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -- python $R/tools/pmc_probe_x3.py > $R/gpurun_out/pmc_x3/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_x3/p3 -o p3 --pmc WRITE_SIZE -- python $R/tools/pmc_probe_x3.py > $R/gpurun_out/pmc_x3/p3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_x3/p4 -o p4 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA -- python $R/tools/pmc_probe_x3.py > $R/gpurun_out/pmc_x3/p4.log 2>&1
cd $R; python tools/pmc_summary.py gpurun_out/pmc_x3 > gpurun_out/pmc_x3/summary.txt 2>&1; cat gpurun_out/pmc_x3/summary.txt | cut -c1-260 | head -60
