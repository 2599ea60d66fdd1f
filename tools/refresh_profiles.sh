#!/bin/bash
# Run on the GPU box (gpurun): refreshes the profile evidence under gpurun_out/final for the current build.
#   1. rocprofv3 --kernel-trace --stats of the default `python3 bench.py` command (the bench line is kept next to it)
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of the bench of C2, C3 and C5_1spp -> HBM bytes per k_intersect launch
# tools/pmc_traffic.py turns the counter CSVs into profiles/<round>/pmc_hbm_traffic.json.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/final
rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py > $OUT/bench_under_rocprof.log 2>&1
grep '^{"metric"' $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench_line_under_rocprof.json
for c in C2 C3 C5_1spp; do
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$c -- python3 $R/bench.py --config $c --no-extra --no-cpu --steps 10 --warmup 2 > $OUT/pmc_fetch_$c.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$c -- python3 $R/bench.py --config $c --no-extra --no-cpu --steps 10 --warmup 2 > $OUT/pmc_write_$c.log 2>&1
done
python3 $R/tools/pmc_traffic.py $OUT > $OUT/pmc_hbm_traffic.json
cat $OUT/pmc_hbm_traffic.json
