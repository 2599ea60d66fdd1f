#!/bin/bash
# Scratch pad for one-off measurements on the GPU box (gpurun -- 'bash tools/scratch_experiment.sh'); rewritten per experiment.
set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/hosttime.py C2 100 > /dev/null
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/exp21_pytest.log 2>&1 || { tail -40 gpurun_out/exp21_pytest.log; exit 1; }
tail -2 gpurun_out/exp21_pytest.log
for c in G1 G2 C2 C3; do for h in 1 0; do echo $c hints=$h; XRT_GRID_HINTS=$h timeout -k 10 300 python tools/hosttime.py $c 100 | tail -1; done; done
