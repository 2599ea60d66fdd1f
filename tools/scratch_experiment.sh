#!/bin/bash
# Scratch pad for one-off measurements on the GPU box (gpurun -- 'bash tools/scratch_experiment.sh'); rewritten per experiment.
set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/exp17_pytest.log 2>&1 || { tail -40 gpurun_out/exp17_pytest.log; exit 1; }
tail -3 gpurun_out/exp17_pytest.log
for c in C2 C3 C5; do for v in 1 0; do echo $c hints=$v; XRT_GRID_HINTS=$v timeout -k 10 300 python tools/hosttime.py $c 100 | tail -1; done; done
XRT_GRID_HINTS=1 timeout -k 10 100 python tools/blocking.py C2 100 | tail -1
XRT_GRID_HINTS=0 timeout -k 10 100 python tools/blocking.py C2 100 | tail -1
