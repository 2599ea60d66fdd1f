"""Where does a pipelined frame spend its time: host enqueue (begin) vs waiting for the GPU (end)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, torch
xrt = importlib.import_module("xna-ray-trace_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
spec = xrt.configs.config(name)
scene, tracer = xrt.configs.build_product(spec)
outs = [torch.zeros(spec.width * spec.height, dtype=torch.int32, device="cuda") for _ in range(2)]
frs = [tracer.PrepareDevice(o.data_ptr()) for o in outs]
fr = frs[0]
for _ in range(3):
    fr()
ts = [frs[0].begin(), frs[1].begin()]   # both frame contexts sized before the clock starts
frs[0].end(ts[0]); frs[1].end(ts[1])
ts = [frs[0].begin(), frs[1].begin()]
frs[0].end(ts[0]); frs[1].end(ts[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
tb = te = 0.0
t0 = time.perf_counter()
prev = None
for i in range(N):
    a = time.perf_counter()
    t = frs[i % 2].begin()
    b = time.perf_counter()
    if prev is not None:
        st = frs[(i - 1) % 2].end(prev)
    c = time.perf_counter()
    tb += b - a
    te += c - b
    prev = t
st = frs[(N - 1) % 2].end(prev)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%s: frame %.1f us  begin %.1f us  end(wait) %.1f us  gpu ms_total %.1f us  intersect %.1f us (%d launches)" % (
    name, dt / N * 1e6, tb / N * 1e6, te / N * 1e6, st["ms_total"] * 1e3, st["ms_intersect"] * 1e3, st["intersect_launches"]))
