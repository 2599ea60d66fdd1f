#!/bin/bash
# Run on the GPU box (gpurun): refreshes the profile evidence under gpurun_out/final for the current build.
#   1. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ wave/VALU set | SQ scalar set) of the bench of every config -> tools/pmc_collect.py ->
#      profiles/pmc_traversal.json (stamped with the kernel sources' hash; bench.py's roofline block reads it); XRT_SPLIT=0: every launch of a pass is a
#      whole-frame launch, as in the bench's timed region (blocking frames of two-level scenes are otherwise rendered as two bands)
#   2. rocprofv3 --kernel-trace --stats of the default `python3 bench.py` command (the bench line is kept next to it)
# Usage: tools/refresh_profiles.sh [configs...]   (default: C5 C3 C2 C4 G1)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/final
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CFGS=${@:-C5 C3 C2 C4 G1}
SC="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU"
for c in $CFGS; do
  for pass in fetch write sq sc; do
    case $pass in fetch) CTR="FETCH_SIZE";; write) CTR="WRITE_SIZE";; sq) CTR="$SQ";; sc) CTR="$SC";; esac
    echo "== pmc $pass $c"
    XRT_SPLIT=0 timeout -k 10 400 rocprofv3 --pmc $CTR --output-format csv -d $OUT/pmc_${pass}_$c -- python3 $R/bench.py --config $c --no-extra --no-cpu --no-host --steps 4 --warmup 1 > $OUT/pmc_${pass}_$c.log 2>&1
  done
done
python3 $R/tools/pmc_collect.py $OUT --out $OUT/pmc_traversal.json > /dev/null
cp $OUT/pmc_traversal.json $R/profiles/pmc_traversal.json   # so that the bench below quotes it (same build)
echo "== kernel trace of the default bench"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py > $OUT/bench_under_rocprof.log 2>&1
grep '^{"metric"' $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench_line_under_rocprof.json
echo "== kernel trace of the headline config alone: the traversal launches by their position in the frame"
XRT_SPLIT=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_C5 -- python3 $R/bench.py --config C5 --no-extra --no-cpu --no-host > $OUT/bench_C5_under_rocprof.log 2>&1
LPF=$(grep '^{"metric"' $OUT/bench_C5_under_rocprof.log | tail -1 | python3 -c "import sys, json; print(json.loads(sys.stdin.read())['roofline']['launches_per_frame'])")
python3 $R/tools/trace_classes.py $OUT/stats_C5 $LPF > $OUT/C5_launch_classes_kernel_trace.txt
cat $OUT/C5_launch_classes_kernel_trace.txt
# keep only the small summaries (gpurun merges at most 64 MiB back)
find $OUT -name "*counter_collection.csv" -size +20M -delete
find $OUT -name "*kernel_trace.csv" -size +20M -delete
du -sh $OUT
