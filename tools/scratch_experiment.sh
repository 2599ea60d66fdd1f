#!/bin/bash
# Scratch pad for one-off measurements on the GPU box (gpurun -- 'bash tools/scratch_experiment.sh'); rewritten per experiment.
set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/hosttime_host.py C5 40
timeout -k 10 300 python tools/hosttime.py C5 40
XRT_ONE_STREAM=1 timeout -k 10 300 python tools/hosttime_host.py C5 40
