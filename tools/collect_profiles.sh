#!/bin/bash
# After tools/refresh_profiles.sh ran under gpurun: copy the summaries worth keeping from gpurun_out/final into profiles/r01_final.
set -e
cd "$(dirname "$0")/.."
F=gpurun_out/final; P=profiles/r01_final
mkdir -p $P
rm -f $P/*
S=$(ls -t $F/stats/*/*kernel_stats.csv | head -1)
cp $S $P/default_bench_kernel_stats.csv
cp $F/bench_line_under_rocprof.json $P/default_bench_line_under_rocprof.json
cp $F/pmc_hbm_traffic.json $P/pmc_hbm_traffic.json
for c in C4 C5 G1; do [ -s $F/bench_$c.json ] && grep '^{"metric"' $F/bench_$c.json | tail -1 > $P/bench_line_$c.json; done
make -C xna-ray-trace_amd/csrc asm >/dev/null 2>&1 && grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy|SGPRs:|LDS Size" xna-ray-trace_amd/csrc/kernels.usage.txt | sed 's/.*remark: *//;s/ *\[-Rpass[^]]*\]//' > $P/kernel_resource_usage.txt
ls -la $P
