#!/bin/bash
# Scratch pad for one-off measurements on the GPU box (gpurun -- 'bash tools/scratch_experiment.sh'); rewritten per experiment.
set -e -o pipefail
cd $GRAFT_REPO_ROOT
bash tools/frame_trace.sh C3 | tail -12
timeout -k 10 200 python tools/stamp_lives.py C3
run() { echo "$@"; env "$@" timeout -k 10 120 python tools/blocking.py C3 40 | tail -1; }
run A=1
run XRT_FIRST_BATCH=64
run XRT_TUNE=64,16,48,24
run XRT_TUNE=64,12,48,32
run XRT_TUNE=64,16,64,32
