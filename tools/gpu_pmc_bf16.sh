#!/bin/bash
mkdir -p gpurun_out/pmcb
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmcb/p1 -o p1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python $R/tools/pmc_probe_bf16.py > $R/gpurun_out/pmcb/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmcb/p2 -o p2 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INSTS_MFMA -- python $R/tools/pmc_probe_bf16.py > $R/gpurun_out/pmcb/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmcb/p3 -o p3 --pmc FETCH_SIZE -- python $R/tools/pmc_probe_bf16.py > $R/gpurun_out/pmcb/p3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmcb/p4 -o p4 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- python $R/tools/pmc_probe_bf16.py > $R/gpurun_out/pmcb/p4.log 2>&1
cd $R; ls gpurun_out/pmcb/*; tail -2 gpurun_out/pmcb/p2.log
