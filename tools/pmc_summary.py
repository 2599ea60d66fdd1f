"""Sum rocprofv3 --pmc counter CSVs per kernel name: python tools/pmc_summary.py <dir> [kernel substring] [N]
With N: the kernel's dispatches are taken in dispatch order and grouped by position % N (the N launches of that kernel a frame
makes -- e.g. primary / shadow / reflection packets), one table per group."""
import csv, glob, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_intersect"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = collections.defaultdict(list)   # counter -> [(dispatch id, value)]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            per[r["Counter_Name"]][int(r["Dispatch_Id"])] = per[r["Counter_Name"]].get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    for c, m in per.items():
        for pos, k in enumerate(sorted(m)):
            rows[c].append((pos % N, m[k]))
for g in range(N):
    if N > 1:
        print("-- dispatches %d mod %d" % (g, N))
    for c in sorted(rows):
        v = [x for gg, x in rows[c] if gg == g]
        if v:
            print("%-28s %16.0f  (%d dispatches, %.0f per dispatch)" % (c, sum(v), len(v), sum(v) / len(v)))
