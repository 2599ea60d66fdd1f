"""Scheduling-knob sweep of the persistent traversal loop on the GPU box (development tool)."""
import importlib, os, subprocess, sys, json
cfgs = sys.argv[1].split(";") if len(sys.argv) > 1 else ["24,16,8"]
confs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["C3", "C5_1spp"]
for t in cfgs:
    for c in confs:
        env = dict(os.environ, XRT_TUNE=t)
        out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), "--config", c, "--steps", "5", "--warmup", "1", "--no-extra", "--no-cpu"], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(t, c, "FAILED", out.stderr[-300:]); continue
        d = json.loads(line[-1])
        print("tune %-10s %-8s %8.1f Mrays/s  %7.3f ms/frame  intersect %.3f ms/launch frac %.3f" % (t, c, d["value"], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"]), flush=True)
