#!/bin/bash
# Scratch pad for one-off measurements on the GPU box (gpurun -- 'bash tools/scratch_experiment.sh'); rewritten per experiment.
set -e -o pipefail
cd $GRAFT_REPO_ROOT
for c in C3 C4 C5_1spp G2; do for v in "XRT_BATCH_MIN=64 XRT_GUIDE_DIV=2" "XRT_BATCH_MIN=4 XRT_GUIDE_DIV=1" "XRT_BATCH_MIN=16 XRT_GUIDE_DIV=1" "XRT_BATCH_MIN=32 XRT_GUIDE_DIV=1" "XRT_BATCH_MIN=64 XRT_GUIDE_DIV=1"; do echo $c $v; env $v timeout -k 10 200 python tools/blocking.py $c 30 | tail -1; done; done
