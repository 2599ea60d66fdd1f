"""Generates the committed golden vectors from the CPU oracle (the reference has no tests, fixtures or
golden images of its own — SURVEY §4 — and cannot be run; these vectors pin the oracle against
regressions and travel to the GPU box, where /root/reference does not exist).

    python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
xrt = importlib.import_module("xna-ray-trace_amd")
from oracle import oracle_py as orc  # noqa: E402


def main():
    # K10: config C1, full 256x256 frame, packed RGBA8 (256 KB)
    spec = xrt.configs.config("C1")
    rgba, rgbf, st = orc.OracleScene(spec).render()
    np.save(os.path.join(HERE, "c1_rgba.npy"), rgba)
    # sampled primary-ray hit records of the bigger scenes (every 7th pixel of a 160x90 frame)
    for name, spec in (("c3", xrt.configs.crate_grid_scene(160, 90)), ("h224", xrt.configs.heightfield_scene(160, 90, m=224))):
        o = orc.OracleScene(spec)
        rays = o.primary_rays()[::7]
        np.save(os.path.join(HERE, name + "_rays.npy"), rays)
        np.save(os.path.join(HERE, name + "_hits.npy"), o.intersect(rays))
    # small full renders with reflections + shadows (C3 scene 96x54, C5 scene 48x27 with 16 sub-rays)
    s3 = xrt.configs.crate_grid_scene(96, 54)
    np.save(os.path.join(HERE, "c3_96x54_rgba.npy"), orc.OracleScene(s3).render(nthreads=8)[0])
    s5 = xrt.configs.heightfield_scene(48, 27, m=224, multisampling=xrt.abi.MS_FIXED16)
    np.save(os.path.join(HERE, "h224_48x27_ms16_rgba.npy"), orc.OracleScene(s5).render(nthreads=8)[0])
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
