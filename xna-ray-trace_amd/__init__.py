"""xna-ray-trace_amd — MI355X-native ray / octree / triangle hot path of eitan3/xna-ray-trace.

Import with importlib (the directory name is not a Python identifier):

    import importlib; xrt = importlib.import_module("xna-ray-trace_amd")
"""
from . import _abi as abi
from . import xna, fixtures, configs, dist
from .api import (Material, Mesh, MeshOctree, SceneObject, ISpatialManager, OctreeSpatialManager, Camera, SpotLight,
                  DirectionalLight, RenderTarget, RayTracer, rays_array, RAY_DTYPE, HIT_DTYPE, NODE_DTYPE)

__all__ = ["abi", "xna", "fixtures", "configs", "dist", "Material", "Mesh", "MeshOctree", "SceneObject", "ISpatialManager",
           "OctreeSpatialManager", "Camera", "SpotLight", "DirectionalLight", "RenderTarget", "RayTracer", "rays_array",
           "RAY_DTYPE", "HIT_DTYPE", "NODE_DTYPE"]
