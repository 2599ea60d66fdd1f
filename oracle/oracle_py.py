"""TEST INFRASTRUCTURE — ctypes binding of oracle/liboracle.so (the CPU restatement of the reference's
C# hot path; *** PARITY UNPINNED ***, see oracle.cpp).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module."""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# XRT_ORACLE_LIB=liboracle_asan.so: the sanitizer build (make -C oracle liboracle_asan.so; run the CPU suite with libasan preloaded,
# tools/asan_cpu_suite.sh)
LIB_PATH = os.path.join(_HERE, os.environ.get("XRT_ORACLE_LIB", "liboracle.so"))
_pkg = importlib.import_module("xna-ray-trace_amd")
abi, xna = _pkg.abi, _pkg.xna
RAY_DTYPE, HIT_DTYPE, NODE_DTYPE = _pkg.RAY_DTYPE, _pkg.HIT_DTYPE, _pkg.NODE_DTYPE

_lib = None
_F = C.POINTER(C.c_float)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, os.path.basename(LIB_PATH)])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        l = C.CDLL(LIB_PATH)
        l.orc_scene_create.restype = C.c_void_p
        l.orc_scene_destroy.argtypes = [C.c_void_p]
        l.orc_last_error.restype = C.c_char_p
        l.orc_last_error.argtypes = [C.c_void_p]
        l.orc_scene_add_mesh.argtypes = [C.c_void_p, _F, _F, _F, _F, _F, C.c_int32, C.POINTER(abi.xrt_material), _F]
        l.orc_scene_add_object.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, _F, _F, _F, _F]
        l.orc_scene_build.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        l.orc_scene_get_tree.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64)]
        l.orc_scene_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(abi.xrt_stats)]
        l.orc_mesh_intersect.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(abi.xrt_stats)]
        l.orc_generate_primary_rays.argtypes = [C.POINTER(abi.xrt_camera), C.c_void_p]
        l.orc_render.argtypes = [C.c_void_p, C.POINTER(abi.xrt_camera), C.POINTER(abi.xrt_light), C.c_int32,
                                 C.POINTER(abi.xrt_render_opts), C.c_void_p, C.c_void_p, C.POINTER(abi.xrt_stats),
                                 C.c_int32, C.c_int32, C.c_int32]
        l.orc_kat_triangle.argtypes = [_F, _F, _F, _F, _F]
        l.orc_kat_box.argtypes = [_F, _F, _F, _F]
        l.orc_kat_pack_color.argtypes = [_F]
        l.orc_kat_pack_color.restype = C.c_uint32
        l.orc_kat_unpack_color.argtypes = [C.c_uint32, _F]
        l.orc_kat_look_at.argtypes = [_F, _F, _F, _F]
        l.orc_kat_perspective.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, _F]
        l.orc_kat_invert.argtypes = [_F, _F]
        l.orc_kat_multiply.argtypes = [_F, _F, _F]
        l.orc_kat_build_world.argtypes = [_F, _F, _F, _F, _F, _F, _F]
        l.orc_kat_spot_light.argtypes = [C.POINTER(abi.xrt_light), _F, _F, _F]
        l.orc_kat_lookup_uv.argtypes = [C.POINTER(abi.xrt_material), _F, C.c_int, C.c_int, _F]
        _lib = l
    return _lib


def _fp(a):
    return a.ctypes.data_as(_F)


def fa(*x):
    return np.array(x, dtype=np.float32).reshape(-1)


def material_abi(m):
    a = abi.xrt_material()
    a.reflectiveness, a.transparent, a.refraction_index = m["reflectiveness"], int(m["transparent"]), m["refraction_index"]
    a.interpolate_normals, a.use_texture = int(m["interpolate_normals"]), int(m["use_texture"])
    keep = None
    if m["use_texture"]:
        keep = np.ascontiguousarray(m["texture"], dtype=np.uint32)
        a.tex_height, a.tex_width = keep.shape
        a.tex_argb = keep.ctypes.data_as(C.POINTER(C.c_uint32))
        if m.get("texture_pargb") is not None:
            keep2 = np.ascontiguousarray(m["texture_pargb"], dtype=np.uint32)
            a.tex_pargb = keep2.ctypes.data_as(C.POINTER(C.c_uint32))
            keep = (keep, keep2)
    return a, keep


def light_abi(l):
    a = abi.xrt_light()
    a.kind = l["kind"]
    a.position[:], a.direction[:], a.color[:] = l["position"], l["direction"], l["color"]
    a.intensity, a.spot_angle, a.decay_exponent = l["intensity"], l["spot_angle"], l["decay_exponent"]
    return a


def camera_abi(spec):
    view, proj = _pkg.configs.camera_matrices(spec)
    cam = abi.xrt_camera()
    cam.view[:] = [float(x) for x in view]
    cam.proj[:] = [float(x) for x in proj]
    cam.vp_x, cam.vp_y, cam.vp_width, cam.vp_height = 0, 0, spec.width, spec.height
    cam.vp_min_depth, cam.vp_max_depth = 0.0, 1.0
    return cam


def opts_abi(spec):
    o = abi.xrt_render_opts()
    o.max_reflections, o.use_multisampling, o.multisample_quality = spec.max_reflections, spec.multisampling, spec.multisample_quality
    o.address_mode, o.filtering, o.shard_rank, o.shard_count = spec.address_mode, spec.filtering, 0, 1
    return o


class OracleScene:
    """The reference's OctreeSpatialManager + SceneObjects + Meshes, restated on the CPU."""

    def __init__(self, spec):
        L = lib()
        self.spec = spec
        self.h = C.c_void_p(L.orc_scene_create())
        for data, m in spec.meshes:
            a, keep = material_abi(m)
            sn = np.ascontiguousarray(data.surface_normal, dtype=np.float32)
            rc = L.orc_scene_add_mesh(self.h, _fp(data.v), _fp(data.n), _fp(data.uv), _fp(sn), _fp(data.color), data.ntri,
                                      C.byref(a), _fp(np.ascontiguousarray(data.bbox, dtype=np.float32)))
            assert rc >= 0, L.orc_last_error(self.h)
        for ids, pos, rot, scale in spec.objects:
            bb = np.zeros(6, dtype=np.float32)
            for i in ids:
                bb[:3] = np.minimum(bb[:3], spec.meshes[i][0].bbox[:3])
                bb[3:] = np.maximum(bb[3:], spec.meshes[i][0].bbox[3:])
            world, inv, wbb = xna.build_world(scale, rot, pos, bb)
            idarr = np.array(ids, dtype=np.int32)
            rc = L.orc_scene_add_object(self.h, idarr.ctypes.data_as(C.POINTER(C.c_int32)), len(ids), _fp(xna.as_array(world)),
                                        _fp(xna.as_array(inv)), _fp(bb), _fp(xna.as_array(wbb)))
            assert rc >= 0, L.orc_last_error(self.h)
        rc = L.orc_scene_build(self.h, spec.mesh_threshold, spec.scene_threshold)
        if rc != 0:
            raise RuntimeError(L.orc_last_error(self.h).decode())

    def __del__(self):
        try:
            if self.h:
                lib().orc_scene_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def tree(self, mesh_id=-1):
        nn, nr = C.c_int64(0), C.c_int64(0)
        assert lib().orc_scene_get_tree(self.h, mesh_id, None, C.byref(nn), None, C.byref(nr)) == 0
        nodes = np.zeros(nn.value, dtype=NODE_DTYPE)
        refs = np.zeros(max(nr.value, 1), dtype=np.int32)
        assert lib().orc_scene_get_tree(self.h, mesh_id, nodes.ctypes.data, C.byref(nn), refs.ctypes.data, C.byref(nr)) == 0
        return nodes, refs[: nr.value]

    def intersect(self, rays, stats=False):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        st = abi.xrt_stats()
        assert lib().orc_scene_intersect(self.h, rays.ctypes.data, rays.shape[0], hits.ctypes.data, C.byref(st)) == 0
        return (hits, st.as_dict()) if stats else hits

    def mesh_intersect(self, mesh_id, rays, stats=False):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        st = abi.xrt_stats()
        assert lib().orc_mesh_intersect(self.h, mesh_id, rays.ctypes.data, rays.shape[0], hits.ctypes.data, C.byref(st)) == 0
        return (hits, st.as_dict()) if stats else hits

    def primary_rays(self):
        cam = camera_abi(self.spec)
        rays = np.zeros(self.spec.width * self.spec.height, dtype=RAY_DTYPE)
        assert lib().orc_generate_primary_rays(C.byref(cam), rays.ctypes.data) == 0
        return rays

    def render(self, nthreads=1, rows=None, want_float=True):
        """RayTracer.RenderInternal on the CPU.  Returns (rgba uint32[H*W], rgb float32[H*W,3] or None, stats)."""
        spec = self.spec
        cam, opts = camera_abi(spec), opts_abi(spec)
        lights = (abi.xrt_light * max(len(spec.lights), 1))()
        for i, l in enumerate(spec.lights):
            lights[i] = light_abi(l)
        rgba = np.zeros(spec.width * spec.height, dtype=np.uint32)
        rgbf = np.zeros((spec.width * spec.height, 3), dtype=np.float32) if want_float else None
        st = abi.xrt_stats()
        r0, r1 = (0, spec.height) if rows is None else rows
        rc = lib().orc_render(self.h, C.byref(cam), lights, len(spec.lights), C.byref(opts), rgba.ctypes.data,
                              rgbf.ctypes.data if want_float else None, C.byref(st), nthreads, r0, r1)
        if rc != 0:
            raise RuntimeError("oracle render failed: %d" % rc)
        return rgba, rgbf, st.as_dict()
