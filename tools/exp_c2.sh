set -e -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/exp3_pytest.log 2>&1 || { tail -30 gpurun_out/exp3_pytest.log; exit 1; }
tail -3 gpurun_out/exp3_pytest.log
timeout -k 10 600 python bench.py > gpurun_out/exp3_bench.log 2>&1
grep '^{"metric"' gpurun_out/exp3_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['ms_per_step_host_output'], d['ms_per_step_blocking'], d['roofline']['ms_per_launch'], d['roofline']['serialised']['ms_per_launch'])
for k,v in d['other_configs'].items(): print(k, v['Mrays_per_s'], v['ms_per_step'], v.get('ms_per_step_serialised'), v['ms_per_launch'])
"
