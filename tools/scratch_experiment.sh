#!/bin/bash
# Scratch pad for one-off measurements on the GPU box (gpurun -- 'bash tools/scratch_experiment.sh'); rewritten per experiment.
set -e -o pipefail
cd $GRAFT_REPO_ROOT
cat > /tmp/adapt.py <<'PY'
import sys, time, importlib, torch
sys.path.insert(0, '.')
xrt = importlib.import_module("xna-ray-trace_amd")
name, q = sys.argv[1], int(sys.argv[2])
spec = xrt.configs.config(name)
spec.multisampling, spec.multisample_quality = xrt.abi.MS_ADAPTIVE, q
scene, tracer = xrt.configs.build_product(spec)
out = torch.zeros(spec.width * spec.height, dtype=torch.int32, device="cuda")
fr = tracer.PrepareDevice(out.data_ptr())
for _ in range(3):
    st = fr()
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 10
for _ in range(N):
    st = fr()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
print("%s adaptive quality %d: %.3f ms per frame (gpu %.3f, traversal %.3f in %d launches)" % (name, q, dt * 1e3, st["ms_total"], st["ms_intersect"], st["intersect_launches"]))
PY
for m in -1 1 3 7; do echo XRT_PACKET=$m; XRT_PACKET=$m python /tmp/adapt.py C5_1spp 1; XRT_PACKET=$m python /tmp/adapt.py C5_1spp 2; done
