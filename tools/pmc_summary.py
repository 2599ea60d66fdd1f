"""Sum rocprofv3 --pmc counter CSVs per kernel name: python tools/pmc_summary.py <dir> [kernel substring]"""
import csv, glob, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_intersect"
tot = collections.defaultdict(float)
n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
for k in sorted(tot):
    print("%-28s %16.0f  (%d dispatches, %.0f per dispatch)" % (k, tot[k], n[k], tot[k] / n[k]))
