#!/bin/bash
# Scratch pad for one-off measurements on the GPU box (gpurun -- 'bash tools/scratch_experiment.sh'); rewritten per experiment.
set -e -o pipefail
cd $GRAFT_REPO_ROOT
for h in 1 0 2 3 1 0; do echo hints=$h; SOAK_CAMS=0 XRT_GRID_HINTS=$h timeout -k 10 300 python tools/soak.py C3 0.5 400; done
