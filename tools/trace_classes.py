"""Average duration of the traversal launches of a rocprofv3 --kernel-trace by their position in the frame (the launch classes of
tools/pmc_collect.py and bench.py's roofline.dominant_launch): python tools/trace_classes.py <dir with *kernel_trace.csv> <launches per frame>"""
import csv, glob, os, re, statistics, sys
d, per = sys.argv[1], int(sys.argv[2])
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if re.search("k_intersect|k_packet", r["Kernel_Name"])]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
cls = [[] for _ in range(per)]
for j, r in enumerate(rows):
    cls[j % per].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%d traversal dispatches, %d per frame (rocprofv3 dispatch-level durations, us)" % (len(rows), per))
for i, c in enumerate(cls):
    if c:
        print("  launch %d of a frame: n %4d  average %9.1f  median %9.1f  min %9.1f  max %9.1f" % (i, len(c), sum(c) / len(c), statistics.median(c), min(c), max(c)))
