"""Latency of ONE blocking frame (xrt_render_device; what the C# host's RenderInternal sees): python tools/blocking.py C3 [frames]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, torch
xrt = importlib.import_module("xna-ray-trace_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
spec = xrt.configs.config(name)
scene, tracer = xrt.configs.build_product(spec)
out = torch.zeros(spec.width * spec.height, dtype=torch.int32, device="cuda")
fr = tracer.PrepareDevice(out.data_ptr())
for _ in range(5):
    st = fr()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    st = fr()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print("%s: blocking frame %.3f ms (gpu ms_total %.3f, intersect %.3f, %d launches, %d piece(s)) XRT_SPLIT=%s" % (
    name, dt * 1e3, st["ms_total"], st["ms_intersect"], st["intersect_launches"], st["pieces"], os.environ.get("XRT_SPLIT", "default")))
