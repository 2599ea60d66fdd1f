"""Frame period with the frame landing in a page-locked host Color[] (xrt_render_begin/_end; bench.py time_host_output) next to the
HBM-resident period: python tools/hosttime_host.py <config> [frames]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, torch
import bench
xrt = importlib.import_module("xna-ray-trace_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
spec = xrt.configs.config(name)
scene, tracer = xrt.configs.build_product(spec)
for rep in range(3):
    print("%s: host output %.3f ms per frame" % (name, bench.time_host_output(tracer, spec, N, 4) * 1e3))
