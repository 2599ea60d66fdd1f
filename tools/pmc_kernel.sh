#!/bin/bash
# PMC passes of one kernel on the GPU box: tools/pmc_kernel.sh <tag> <kernel substring> -- <python args...>   (env passes through)
# e.g. XRT_PACKET=1 XRT_SPLIT=0 tools/pmc_kernel.sh pk k_packet -- tools/blocking.py C5 6
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; KER=$2; shift 3
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"
B="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
C="SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_INSTS_SENDMSG SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_FLAT"
i=0
for CTR in "$A" "$B" "$C"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $CTR --output-format csv -d $OUT/p$i -- python3 $R/"$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 $R/tools/pmc_summary.py $OUT $KER ${PMC_GROUPS:-1} | tee $OUT/summary.txt
grep -h "blocking frame\|frame " $OUT/p1.log | tail -2
