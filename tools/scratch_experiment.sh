#!/bin/bash
# Scratch pad for one-off measurements on the GPU box (gpurun -- 'bash tools/scratch_experiment.sh'); rewritten per experiment.
set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "ray_tree or refraction or default_game or G1 or content_scene" > gpurun_out/exp20_pytest.log 2>&1 || { tail -50 gpurun_out/exp20_pytest.log; exit 1; }
tail -3 gpurun_out/exp20_pytest.log
