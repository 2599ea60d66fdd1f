#!/bin/bash
# How full is the GPU while a tile shard's frames run two in flight?  tools/shard_overlap.sh <config> <rank> <count>   (on the GPU box)
# From a kernel trace: over the steady-state window, the time during which NO kernel runs, exactly one, two or more; the sum of kernel durations.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/overlap_$1_$2of$3
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/shard_pipelined.py $1 $2 $3 24 > $OUT/run.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rg = [i for i, r in enumerate(rows) if "k_raygen" in r["Kernel_Name"]]
lo, hi = rg[len(rg) // 3], rg[-3]          # steady state: from the raygen at one third to the third last
t0, t1 = int(rows[lo]["Start_Timestamp"]), int(rows[hi]["Start_Timestamp"])
ev = []
byk = collections.defaultdict(float)
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    s, e = max(s, t0), min(e, t1)
    if e > s:
        ev.append((s, 1)); ev.append((e, -1))
        byk[r["Kernel_Name"].split("(")[0].replace("void xrt::", "").replace("xrt::", "")[:24]] += e - s
ev.sort()
depth, last, hist = 0, t0, collections.defaultdict(float)
for t, d in ev:
    hist[min(depth, 3)] += t - last
    last = t
    depth += d
hist[min(depth, 3)] += t1 - last
frames = sum(1 for i in rg if lo <= i < hi)
tot = t1 - t0
print("window %.3f ms = %d frames, period %.4f ms" % (tot / 1e6, frames, tot / 1e6 / frames))
for k in sorted(hist):
    print("  %s kernel(s) running: %5.1f %%" % (("3+" if k == 3 else str(k)), 100.0 * hist[k] / tot))
print("  sum of kernel durations / window: %.2f" % (sum(byk.values()) / tot))
for k, v in sorted(byk.items(), key=lambda kv: -kv[1]):
    print("    %-26s %.4f ms per frame" % (k, v / 1e6 / frames))
PY
tail -1 $OUT/run.log
find $OUT -name "*.csv" -size +5M -delete
