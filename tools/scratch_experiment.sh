#!/bin/bash
# Scratch pad for one-off measurements on the GPU box (gpurun -- 'bash tools/scratch_experiment.sh'); rewritten per experiment.
set -e -o pipefail
cd $GRAFT_REPO_ROOT
bash tools/frame_trace.sh C2 | tail -12
timeout -k 10 200 python tools/stamp_lives.py C2
