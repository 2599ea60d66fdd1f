#!/bin/bash
# Scratch pad for one-off measurements on the GPU box (gpurun -- 'bash tools/scratch_experiment.sh'); rewritten per experiment.
set -e -o pipefail
cd $GRAFT_REPO_ROOT
run() { echo "$@"; env "$@" timeout -k 10 120 python tools/blocking.py G1 60 | tail -1; }
run A=1
run XRT_TUNE=24,16,48,32
run XRT_TUNE=64,16,48,0
run XRT_TUNE=16,8,24,32
run XRT_FIRST_BATCH=16
run XRT_LEAF_CULL=0
run XRT_LAUNCH_TIMING=0
run XRT_CULL_SAFETY=0
