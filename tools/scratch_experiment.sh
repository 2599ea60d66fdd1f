#!/bin/bash
# Scratch pad for one-off measurements on the GPU box (gpurun -- 'bash tools/scratch_experiment.sh'); rewritten per experiment.
set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/exp11_pytest.log 2>&1 || { tail -30 gpurun_out/exp11_pytest.log; exit 1; }
tail -3 gpurun_out/exp11_pytest.log
for i in 1 2; do timeout -k 10 300 python tools/hosttime.py C5 40; XRT_PACKET=23 timeout -k 10 300 python tools/hosttime.py C5 40; done
timeout -k 10 300 python tools/blocking.py C5 30
XRT_PACKET=23 timeout -k 10 300 python tools/blocking.py C5 30
