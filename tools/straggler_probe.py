"""How long does the traversal of C3's slowest second-bounce rays take on their own?  (development probe)
Second-generation reflection rays are built from the library's own answers; then the 4096 / 64 / 1 rays that were in flight longest
(xrt_hit.reserved) are traced alone."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, numpy as np
xrt = importlib.import_module("xna-ray-trace_amd")
spec = xrt.configs.config(sys.argv[1] if len(sys.argv) > 1 else "C3")
scene, tracer = xrt.configs.build_product(spec)
data = spec.meshes[0][0]


def reflect(rays, hits):
    m = hits["hit"] == 1
    D = rays["d"][m].astype(np.float64)
    N = data.surface_normal[hits["tri"][m]].astype(np.float64)
    R = D - 2 * (D * N).sum(axis=1, keepdims=True) * N
    R /= np.linalg.norm(R, axis=1, keepdims=True)
    return xrt.rays_array(hits["w"][m], R.astype(np.float32), hits["mesh"][m], hits["tri"][m])


def trace(rays, reps=5):
    best = 1e9
    for _ in range(reps):
        hits, st = scene.IntersectBatch(rays, stats=True)
        best = min(best, st["ms_total"])
    return hits, best


prim = tracer.GeneratePrimaryRays()
h0, t0 = trace(prim)
r1 = reflect(prim, h0)
h1, t1 = trace(r1)
r2 = reflect(r1, h1)
h2, t2 = trace(r2)
print("primary %d rays %.3f ms; generation 1: %d rays %.3f ms; generation 2: %d rays %.3f ms" % (len(prim), t0, len(r1), t1, len(r2), t2))
for name, rays, hits in (("generation 1", r1, h1), ("generation 2", r2, h2)):
    cost = hits["reserved"]
    order = np.argsort(cost)[::-1]
    print(name, "rounds in flight: median %d, 99 %% %d, max %d" % (np.median(cost), np.percentile(cost, 99), cost.max()))
    for k in (4096, 64, 8, 1):
        sub = np.ascontiguousarray(rays[order[:k]])
        _, t = trace(sub)
        print("  the %4d rays longest in flight, alone: %.1f us" % (k, t * 1e3))
    sub = np.ascontiguousarray(rays[order[len(order) // 2: len(order) // 2 + 64]])
    _, t = trace(sub)
    print("  64 median rays alone: %.1f us" % (t * 1e3))
