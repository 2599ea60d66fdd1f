#!/bin/bash
# Vector-memory pipeline counters of one kernel: tools/pmc_mem.sh <tag> <kernel substring> -- <python args...>
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; KER=$2; shift 3
OUT=$R/gpurun_out/pmcmem_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for CTR in "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $CTR --output-format csv -d $OUT/p$i -- python3 $R/"$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
python3 $R/tools/pmc_summary.py $OUT $KER | tee $OUT/summary.txt
