"""HBM bytes per k_intersect launch from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/refresh_profiles.sh.
gfx950: FETCH_SIZE tallies 128-byte read requests at 64 bytes, so streamed reads are twice the counter
(MI355X_MICROARCH.md, HBM section); both counters are in KB."""
import csv, glob, json, sys
out = sys.argv[1]
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel k_intersect only), KB per launch averaged over all launches of "
               "`bench.py --config <cfg> --no-extra --no-cpu --steps 10 --warmup 2`; gfx950 correction: FETCH_SIZE counts 1/2 of streamed read bytes "
               "(MI355X_MICROARCH.md HBM section) -> traffic = (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch"}
for cfg in ("C2", "C3", "C5_1spp"):
    vals = {}
    for kind, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        tot, n = 0.0, 0
        for f in glob.glob("%s/pmc_%s_%s/**/*counter_collection.csv" % (out, kind, cfg), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_intersect" in r["Kernel_Name"] and r["Counter_Name"] == name:
                    tot += float(r["Counter_Value"]); n += 1
        vals[name] = (tot / n if n else None, n)
    if vals["FETCH_SIZE"][0] is None or vals["WRITE_SIZE"][0] is None:
        continue
    res[cfg] = {"FETCH_SIZE_KB_per_launch": vals["FETCH_SIZE"][0], "launches_fetch": vals["FETCH_SIZE"][1],
                "WRITE_SIZE_KB_per_launch": vals["WRITE_SIZE"][0], "launches_write": vals["WRITE_SIZE"][1],
                "traffic_bytes_per_launch": int((2 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024)}
print(json.dumps(res, indent=1))
