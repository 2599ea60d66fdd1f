#!/bin/bash
# Scratch pad for one-off measurements on the GPU box (gpurun -- 'bash tools/scratch_experiment.sh'); rewritten per experiment.
set -e -o pipefail
cd $GRAFT_REPO_ROOT
for c in C3 C4 C5_1spp C2 G1; do timeout -k 10 120 python tools/blocking.py $c 30 | tail -1; done
for c in C3 C5_1spp; do timeout -k 10 120 python tools/hosttime.py $c 40 | tail -1; done
