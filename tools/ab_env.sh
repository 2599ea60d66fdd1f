#!/bin/bash
# A/B of environment switches of ONE library on one GPU box (as ab_libs.sh for libraries):
#   tools/ab_env.sh "XRT_NODE_CULL=0 XRT_NODE_CULL=1 XRT_NODE_CULL=2" "C5 C3 C4" [rounds] [lib]
ENVS=${1:-X=0}; CFGS=${2:-C3 C4 C5}; ROUNDS=${3:-2}; LIB=${4:-libxrt.so}
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq 1 $ROUNDS); do for c in $CFGS; do for e in $ENVS; do
  env $e XRT_LIB_VARIANT=$LIB timeout -k 10 120 python3 $R/bench.py --config $c --no-extra --no-cpu --no-host --steps 30 --warmup 5 2>/dev/null | python3 -c "
import sys, json
for x in sys.stdin:
    if x.startswith('{'):
        d = json.loads(x); print('$c %-22s ms_per_step %.4f blocking %.4f' % ('$e', d['ms_per_step'], d['ms_per_step_blocking']))
"
done; done; done
