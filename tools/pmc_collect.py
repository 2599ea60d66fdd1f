"""rocprofv3 --pmc counter CSVs -> profiles/pmc_traversal.json (what bench.py's roofline block reads).

    python tools/pmc_collect.py <dir> [--out profiles/pmc_traversal.json] [--merge]

<dir> holds one sub-directory per pass, named pmc_<pass>_<config> (tools/refresh_profiles.sh): separate passes for FETCH_SIZE,
WRITE_SIZE (they do not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots") and the SQ set.  Per config the counters are
averaged over ALL traversal dispatches (k_intersect, the per-lane kernel, and k_packet, the wave-packet kernel) of `bench.py --config <config>` (every frame has the same launches).  FETCH_SIZE /
WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so bench.py doubles it (MI355X_MICROARCH.md, HBM).
The file is stamped with the hash of the kernel sources: bench.py quotes it only for that build."""
import argparse, collections, csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_id():
    import hashlib
    h = hashlib.sha256()
    for f in ("kernels.hip", "packet.hip", "kernels.h", "device_util.h", "traverse.h", "xrt_core.h", "xrt_api.cpp"):
        h.update(open(os.path.join(ROOT, "xna-ray-trace_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "pmc_traversal.json"))
    ap.add_argument("--kernel", default="k_intersect|k_packet", help="regex: the traversal kernels (per-lane and wave-packet form)")
    args = ap.parse_args()
    configs = collections.defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(args.dir, "pmc_*_*"))):
        if not os.path.isdir(d):
            continue
        m = re.match(r"pmc_([a-z0-9]+)_(.+)$", os.path.basename(d))
        if not m:
            continue
        cfg = m.group(2)
        tot, n = collections.defaultdict(float), collections.Counter()
        # ... and the same per launch CLASS: the traversal launches of a frame differ (the primary rays' launch walks the octree, the launch of
        # the shadow rays and reflections of a terrain mostly ends at the mesh's normal box), so the dispatches are also grouped by their position
        # in the frame (index among the traversal dispatches modulo the launches per frame, which the pass's bench line states)
        per_frame = 0
        try:
            for line in open(d + ".log"):
                if line.startswith('{"metric"'):
                    per_frame = int(json.loads(line)["roofline"]["launches_per_frame"])
        except Exception:
            per_frame = 0
        rows = []
        files = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
        for f in files[-1:]:   # (the newest run only: a directory merged back from several gpurun calls holds the older runs' files too)
            for r in csv.DictReader(open(f)):
                if re.search(args.kernel, r["Kernel_Name"]):
                    rows.append((int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"])))
                    tot[r["Counter_Name"]] += float(r["Counter_Value"])
                    n[r["Counter_Name"]] += 1
        for k in tot:
            name = k + "_KB" if k in ("FETCH_SIZE", "WRITE_SIZE") else k
            configs[cfg][name] = tot[k] / n[k]
            configs[cfg].setdefault("dispatches", {})[k] = n[k]
        if per_frame > 1 and rows:
            ids = sorted(set(i for i, _, _ in rows))
            pos = {i: j % per_frame for j, i in enumerate(ids)}
            gt, gn = collections.defaultdict(float), collections.Counter()
            for i, k, v in rows:
                gt[(pos[i], k)] += v
                gn[(pos[i], k)] += 1
            groups = configs[cfg].setdefault("launch_classes", [dict() for _ in range(per_frame)])
            if len(groups) == per_frame:
                for (g, k), v in gt.items():
                    groups[g][k + "_KB" if k in ("FETCH_SIZE", "WRITE_SIZE") else k] = v / gn[(g, k)]
    out = {"build_id": build_id(), "kernel": args.kernel,
           "note": "per traversal launch (k_intersect and k_packet together), averaged over every launch of `bench.py --config <cfg> --no-extra --no-cpu --no-host` under rocprofv3 --pmc "
                   "(separate passes: FETCH_SIZE | WRITE_SIZE | SQ_*); FETCH_SIZE/WRITE_SIZE in KB, FETCH_SIZE counts half of streamed read bytes on gfx950",
           "configs": configs}
    need = ("FETCH_SIZE_KB", "WRITE_SIZE_KB", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU")
    for cfg, v in configs.items():
        missing = [k for k in need if k not in v]
        if missing:
            print("warning: %s lacks %s" % (cfg, missing), file=sys.stderr)
    json.dump(out, open(args.out, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
