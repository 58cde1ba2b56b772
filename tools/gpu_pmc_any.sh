#!/bin/bash
# PMC passes (separate runs, --kernel-trace only) over ANY python workload: tools/gpu_pmc_any.sh <tag> <script and args...>; summary by tools/pmc_summary.py
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_REQ_sum"; do
  i=$((i + 1))
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/p$i -o p$i --pmc $set -- python $R/"$1" "${@:2}" > $O/p$i.log 2>&1
done
cd $R
python tools/pmc_summary.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
