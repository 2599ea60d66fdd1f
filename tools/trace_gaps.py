"""Per-frame timeline from a rocprofv3 --kernel-trace CSV: kernel durations and the idle gaps between them."""
import csv, sys, glob
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("xrt::", "")))
rows.sort()
starts = [i for i, r in enumerate(rows) if r[2].startswith("k_raygen")]
if len(starts) < 4:
    sys.exit("too few frames")
a, b = starts[-3], starts[-2]          # a steady-state frame
prev_end = rows[a - 1][1]
busy = 0
for s, e, n in rows[a:b]:
    print("%-28s dur %7.2f us   gap before %6.2f us" % (n[:28], (e - s) / 1e3, (s - prev_end) / 1e3))
    busy += e - s
    prev_end = e
print("frame period %.2f us, kernels busy %.2f us" % ((rows[b][0] - rows[a][0]) / 1e3, busy / 1e3))
