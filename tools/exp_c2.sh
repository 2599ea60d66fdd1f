set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/exp10_pytest.log 2>&1 || { tail -30 gpurun_out/exp10_pytest.log; exit 1; }
tail -3 gpurun_out/exp10_pytest.log
for i in 1 2 3; do timeout -k 10 300 python tools/hosttime.py C5 40; done
bash tools/frame_trace.sh C5 | tail -16
