set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/exp6_pytest.log 2>&1 || { tail -30 gpurun_out/exp6_pytest.log; exit 1; }
tail -3 gpurun_out/exp6_pytest.log
for c in C3 C4 C5_1spp C2 G1; do
for v in 1 0; do
echo $c leafcull=$v; XRT_LEAF_CULL=$v timeout -k 10 300 python tools/hosttime.py $c 60
done
done
