#!/bin/bash
# Scratch pad for one-off measurements on the GPU box (gpurun -- 'bash tools/scratch_experiment.sh'); rewritten per experiment.
set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "host_written_in_c" > gpurun_out/exp22_pytest.log 2>&1 || { tail -40 gpurun_out/exp22_pytest.log; exit 1; }
tail -3 gpurun_out/exp22_pytest.log
