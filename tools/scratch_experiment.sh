#!/bin/bash
# Scratch pad for one-off measurements on the GPU box (gpurun -- 'bash tools/scratch_experiment.sh'); rewritten per experiment.
set -e -o pipefail
cd $GRAFT_REPO_ROOT
for c in G1 G2 C3 C5_1spp; do for v in 16 8 4; do echo $c spread=$v; XRT_SPREAD_MIN=$v timeout -k 10 200 python tools/blocking.py $c 30 | tail -1; done; done
XRT_SPREAD_MIN=8 timeout -k 10 200 python tools/stamp_lives.py G1 | head -4
