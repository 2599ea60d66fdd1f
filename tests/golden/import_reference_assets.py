"""Imports geometry of the reference's own content (RayTraceProjectContent/*.fbx) with xna-ray-trace_amd/fbx.py and
stores it as data fixtures — /root/reference does not exist on the GPU box.

    python tests/golden/import_reference_assets.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
importlib.import_module("xna-ray-trace_amd")
fbx = importlib.import_module("xna-ray-trace_amd.fbx")
CONTENT = "/root/reference/RayTraceProject/RayTraceProjectContent"


def main():
    # Sphere.fbx with the processor parameters of RayTraceProjectContent.contentproj:87-96
    meshes, up = fbx.load_fbx(os.path.join(CONTENT, "Sphere.fbx"))
    md = fbx.import_mesh(meshes[0], up, scale=2.0, diffuse_color=(255, 0, 0, 100))
    np.savez_compressed(os.path.join(HERE, "sphere_mesh.npz"), v=md.v, n=md.n, uv=md.uv, color=md.color)
    print("Sphere.fbx ->", md.ntri, "triangles")
    # more of the reference's content, each with the processor parameters the content project gives it
    # (RayTraceProjectContent.contentproj:107-195)
    fixtures = importlib.import_module("xna-ray-trace_amd.fixtures")
    out = {}
    for name, asset, kw in (("plane", "plane.fbx", dict(scale=18.0, diffuse_color=(255, 255, 255, 255))),         # contentproj:124-135 "ground" (Scale 18 instead of 3:
                            # object-space distances are compared across bodies (OSM:370-378), so the size goes into the geometry, not the body)
                            ("monkey", "monkey.fbx", dict(scale=5.0, diffuse_color=(255, 255, 255, 64))),        # contentproj:137-146
                            ("torus", "torus.fbx", dict(scale=2.0, diffuse_color=(0, 0, 255, 255))),             # contentproj:148-156
                            ("cube", "cube.fbx", dict()),                                                         # contentproj:107-110
                            # assets whose content-project entries carry ModelProcessor rotation parameters (fbx.import_mesh `rotation`)
                            ("prism", "prism2.fbx", dict(rotation=(-90.0, 0.0, 0.0), diffuse_color=(255, 255, 255, 100))),          # contentproj:112-122
                            ("chesspiece", "chesspiece.fbx", dict(scale=3.0, rotation=(-90.0, 0.0, 0.0), diffuse_color=(255, 255, 255, 255)))):   # contentproj:186-195
        meshes, up = fbx.load_fbx(os.path.join(CONTENT, asset))
        md = fbx.import_mesh(meshes[0], up, **kw)
        for k in ("v", "n", "uv", "color"):
            out[name + "_" + k] = getattr(md, k)
        print(asset, "->", md.ntri, "triangles")
    out["checkers_argb"] = fixtures.load_bmp_argb(os.path.join(CONTENT, "checkers.bmp"))   # the content project's texture asset
    np.savez_compressed(os.path.join(HERE, "content_meshes.npz"), **out)


if __name__ == "__main__":
    main()
