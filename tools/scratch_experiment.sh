#!/bin/bash
# Scratch pad for one-off measurements on the GPU box (gpurun -- 'bash tools/scratch_experiment.sh'); rewritten per experiment.
set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python tools/blocking.py C3 20 > /dev/null
for c in C3 C4; do for v in "XRT_HEAVY_SPARSE=0" "XRT_HEAVY_SPARSE=1 XRT_HEAVY_SHIFT=1" "XRT_HEAVY_SPARSE=1 XRT_HEAVY_SHIFT=2" "XRT_HEAVY_SPARSE=1 XRT_HEAVY_SHIFT=3" "XRT_HEAVY_SPARSE=1 XRT_HEAVY_SHIFT=2 XRT_LONG_FRAC=6,12"; do echo $c $v; env $v timeout -k 10 200 python tools/blocking.py $c 30 | tail -1; done; done
