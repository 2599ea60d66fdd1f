#!/bin/bash
# A/B of differently built libraries on ONE GPU box (boxes differ by a few per cent, so only numbers of one call compare):
#   tools/ab_libs.sh "libxrt_a.so libxrt_b.so libxrt.so" "C3 C4 C5" [rounds]
# Every (library, config) pair is benchmarked `rounds` times, interleaved; prints ms_per_step (two frames in flight) and the blocking frame.
LIBS=${1:-libxrt.so}; CFGS=${2:-C3 C4 C5}; ROUNDS=${3:-2}
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq 1 $ROUNDS); do for c in $CFGS; do for l in $LIBS; do
  XRT_LIB_VARIANT=$l timeout -k 10 120 python3 $R/bench.py --config $c --no-extra --no-cpu --no-host --steps 30 --warmup 5 2>/dev/null | python3 -c "
import sys, json
for x in sys.stdin:
    if x.startswith('{'):
        d = json.loads(x); print('$c %-22s ms_per_step %.4f blocking %.4f parity %s' % ('$l', d['ms_per_step'], d['ms_per_step_blocking'], d.get('parity_ok')))
"
done; done; done
