#!/bin/bash
# Scratch pad for one-off measurements on the GPU box (gpurun -- 'bash tools/scratch_experiment.sh'); rewritten per experiment.
set -e -o pipefail
cd $GRAFT_REPO_ROOT
XRT_FUZZ_EXTRA=400 timeout -k 10 1100 python -m pytest tests -m gpu -q -k "random_scenes_against_the_oracle" > gpurun_out/fuzz400.log 2>&1 || { tail -30 gpurun_out/fuzz400.log; exit 1; }
tail -3 gpurun_out/fuzz400.log
