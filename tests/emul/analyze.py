"""TEST INFRASTRUCTURE (development aid): per-ray step counts and wave utilisation of the traversal state machine on a
configuration's primary rays, from the CPU single-stepper.  python tests/emul/analyze.py C3 [rows]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
xrt = importlib.import_module("xna-ray-trace_amd")
import emul_py
from oracle import oracle_py as orc

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 64
spec = xrt.configs.config(name)
o = orc.OracleScene(spec)
rays = o.primary_rays().reshape(spec.height, spec.width)
y0 = spec.height // 2 - rows // 2
band = rays[y0:y0 + rows]
# tile order 64x8 like the product
tiles = []
for ty in range(0, rows, 8):
    for tx in range(0, spec.width, 64):
        tiles.append(band[ty:ty + 8, tx:tx + 64].reshape(-1))
pr = np.concatenate(tiles)
e = emul_py.EmulScene(spec)
hits, st = e.intersect(pr, steps=True)
n = len(pr)
print("%s: %d primary rays, hit %.1f%%; per ray: scene steps %.2f  node steps %.2f  leaf steps %.2f" % (name, n, 100.0 * (hits["hit"] != 0).mean(), st[:, 0].mean(), st[:, 1].mean(), st[:, 2].mean()))
print("  max per ray: scene %d node %d leaf %d" % (st[:, 0].max(), st[:, 1].max(), st[:, 2].max()))
w = e.wave_sim(pr, n_waves=max(1, n // 1024))
print("  wave sim:", {k: v for k, v in w.items()})
for ph in ("scene", "node", "leaf"):
    s, l = w[ph + "_steps"], w[ph + "_lanes"]
    print("  %-5s wave-steps per 64 rays %.1f, lane utilisation %.2f" % (ph, s / (n / 64.0), l / (64.0 * s) if s else 0))
# secondary: reflection rays of the hits
