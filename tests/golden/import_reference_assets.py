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


if __name__ == "__main__":
    main()
