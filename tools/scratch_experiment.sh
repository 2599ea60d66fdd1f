#!/bin/bash
# Scratch pad for one-off measurements on the GPU box (gpurun -- 'bash tools/scratch_experiment.sh'); rewritten per experiment.
set -e -o pipefail
cd $GRAFT_REPO_ROOT
XRT_FUZZ_EXTRA=400 timeout -k 10 1100 python -m pytest tests -m gpu -q -k "random_scenes_against_the_oracle" > gpurun_out/fuzz400b.log 2>&1 || { tail -30 gpurun_out/fuzz400b.log; exit 1; }
tail -2 gpurun_out/fuzz400b.log
timeout -k 10 300 python tools/soak.py G1 1.0 2000
timeout -k 10 300 python tools/soak.py G2 0.5 1000
XRT_HEAP_RAY_CAP=2048 timeout -k 10 300 python tools/soak.py G1 0.25 300
timeout -k 10 300 python tools/soak.py C2 1.0 3000
timeout -k 10 300 python tools/soak.py C5 0.25 400
