"""Soak: thousands of frames, two in flight, two cameras alternating at random, every frame compared with the blocking render of
its camera on the GPU (torch.equal).  python tools/soak.py <config> [scale] [frames]"""
import sys, os, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, numpy as np, torch
xrt = importlib.import_module("xna-ray-trace_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "G1"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
spec = xrt.configs.config(name, scale)
scene, tracer = xrt.configs.build_product(spec)
n = spec.width * spec.height
c = spec.camera
cams = [tracer.CurrentCamera,
        xrt.api.Camera(tuple(1.15 * x for x in c["pos"]), c["target"], c["up"], c["fov"], xrt.xna.aspect_ratio(spec.width, spec.height), c["near"], c["far"]),
        xrt.api.Camera(c["pos"], tuple(x + 400.0 for x in c["target"]), c["up"], c["fov"], xrt.xna.aspect_ratio(spec.width, spec.height), c["near"], c["far"])]
want = []
for cam in cams:
    tracer.CurrentCamera = cam
    want.append(torch.from_numpy(tracer.Render().copy().view(np.int32)).cuda())
outs = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(2)]
frs = []
for cam in cams:
    tracer.CurrentCamera = cam
    frs.append([tracer.PrepareDevice(o.data_ptr()) for o in outs])
if os.environ.get("SOAK_CAMS"):   # e.g. "0,2": only these cameras
    keep = [int(x) for x in os.environ["SOAK_CAMS"].split(",")]
    cams, want, frs = [cams[i] for i in keep], [want[i] for i in keep], [frs[i] for i in keep]
rng = random.Random(7)
open_t, bad = None, 0
t0 = time.perf_counter()
for i in range(N):
    k = rng.randrange(len(cams))
    t = frs[k][i % 2].begin()
    if open_t is not None:
        pk, pi, pt = open_t
        frs[pk][pi % 2].end(pt)
        if not torch.equal(outs[pi % 2], want[pk]):
            bad += 1
        outs[pi % 2].zero_()
        torch.cuda.current_stream().synchronize()
    open_t = (k, i, t)
pk, pi, pt = open_t
frs[pk][pi % 2].end(pt)
bad += 0 if torch.equal(outs[pi % 2], want[pk]) else 1
print("%s x%.2f: %d frames, two in flight, three cameras at random: %d wrong, %.3f ms per frame" % (name, scale, N, bad, (time.perf_counter() - t0) / N * 1e3))
sys.exit(1 if bad else 0)
