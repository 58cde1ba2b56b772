#!/bin/bash
# PMC passes over the native weight-gradient probe (separate runs per counter group, kernel trace only)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmcw
cd /tmp && export TMPDIR=/tmp
FORM=${1:-4}
SHAPE=${2:-"12 120 214 256 256"}
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_TOTAL_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE"; do
  for m in 0 1; do
    OSVOS_WGRAD_MAP=$m OSVOS_WGRAD_FORM=$FORM timeout 60 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/pmcw/f${FORM}_m${m}_g$i -o p --pmc $grp -- $R/tools/native/bin/wgrad_probe $SHAPE > $R/gpurun_out/pmcw/f${FORM}_m${m}_g$i.log 2>&1
  done
  i=$((i+1))
done
find $R/gpurun_out/pmcw -name "*counter_collection.csv" | head -20
