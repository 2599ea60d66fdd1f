"""Host cost of begin() when frames do not overlap (begin; end; begin; end) on the context's own stream."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, torch
xrt = importlib.import_module("xna-ray-trace_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "C1"
spec = xrt.configs.config(name)
scene, tracer = xrt.configs.build_product(spec)
out = torch.zeros(spec.width * spec.height, dtype=torch.int32, device="cuda")
fr = tracer.PrepareDevice(out.data_ptr())
for _ in range(5):
    fr()
tb = te = 0.0
N = 100
for i in range(N):
    a = time.perf_counter(); t = fr.begin(); b = time.perf_counter(); fr.end(t); c = time.perf_counter()
    tb += b - a; te += c - b
print("%s serial on ctx stream: begin %.1f us end %.1f us" % (name, tb / N * 1e6, te / N * 1e6))
