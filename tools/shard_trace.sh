#!/bin/bash
# Kernel-by-kernel timeline of one blocking frame of ONE TILE SHARD: tools/shard_trace.sh <config> <rank> <count>   (on the GPU box)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace_$1_$2of$3
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
XRT_SPLIT=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/shard_frames.py $1 $2 $3 6 > $OUT/run.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = max(i for i, r in enumerate(rows) if "k_raygen" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
prev_end = t0
for r in rows[idx:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void xrt::", "").replace("xrt::", "")
    print("%-28s start %8.1f us  dur %8.1f us  gap %6.1f us  grid %s" % (name[:28], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Grid_Size", "?")))
    prev_end = e
print("frame total %.1f us" % ((prev_end - t0) / 1e3))
PY
tail -1 $OUT/run.log
find $OUT -name "*.csv" -size +5M -delete
