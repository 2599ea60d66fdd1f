"""ctypes binding of include/xrt.h — the only way Python reaches the HIP library.

There is no CPU fallback: if ``csrc/libxrt.so`` is missing or cannot be loaded, importing the
library handle raises (SURVEY §8b "the product path must fail loudly").
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
import re


def _lib_name():
    """XRT_LIB_VARIANT selects a differently built libxrt next to the shipped one (tools/ experiments: `make variant NAME=..`).  Only a bare
    file name of the form libxrt[_name].so inside csrc/ is accepted: the environment must not be able to make the package load an
    arbitrary shared object (an absolute path would replace the directory in os.path.join)."""
    v = os.environ.get("XRT_LIB_VARIANT", "libxrt.so")
    if not re.fullmatch(r"libxrt(_\w+)?\.so", v):
        raise ImportError("XRT_LIB_VARIANT=%r: expected libxrt[_name].so (a file inside %s)" % (v, os.path.join(_HERE, "csrc")))
    return v


LIB_PATH = os.path.join(_HERE, "csrc", _lib_name())

XRT_OK = 0
XRT_E_INVALID_ARG = -1
XRT_E_BUSY = -2
XRT_E_NO_DEVICE = -3
XRT_E_OOM = -4
XRT_E_HIP = -5
XRT_E_RCCL = -6
XRT_E_INTERNAL = -7
XRT_E_UNSUPPORTED = -8
XRT_E_NOT_BUILT = -9

FILTER_POINT, FILTER_BILINEAR = 0, 1
ADDRESS_CLAMP, ADDRESS_WRAP, ADDRESS_MIRROR = 0, 1, 2
LIGHT_SPOT, LIGHT_DIRECTIONAL = 0, 1
MS_OFF, MS_ADAPTIVE, MS_FIXED16 = 0, 1, 2
TILE_W, TILE_H = 64, 8


class xrt_ray(C.Structure):
    _fields_ = [("o", C.c_float * 3), ("d", C.c_float * 3), ("ignore_mesh", C.c_int32), ("ignore_tri", C.c_int32)]


class xrt_hit(C.Structure):
    _fields_ = [("hit", C.c_int32), ("object", C.c_int32), ("mesh", C.c_int32), ("tri", C.c_int32),
                ("leaf", C.c_int32), ("u", C.c_float), ("v", C.c_float), ("d", C.c_float),
                ("wx", C.c_float), ("wy", C.c_float), ("wz", C.c_float), ("reserved", C.c_int32)]


class xrt_material(C.Structure):
    _fields_ = [("reflectiveness", C.c_float), ("transparent", C.c_int32), ("refraction_index", C.c_float),
                ("interpolate_normals", C.c_int32), ("use_texture", C.c_int32), ("tex_width", C.c_int32),
                ("tex_height", C.c_int32), ("reserved", C.c_int32), ("tex_argb", C.POINTER(C.c_uint32)),
                ("tex_pargb", C.POINTER(C.c_uint32))]


class xrt_camera(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("proj", C.c_float * 16), ("vp_x", C.c_int32), ("vp_y", C.c_int32),
                ("vp_width", C.c_int32), ("vp_height", C.c_int32), ("vp_min_depth", C.c_float),
                ("vp_max_depth", C.c_float)]


class xrt_light(C.Structure):
    _fields_ = [("kind", C.c_int32), ("position", C.c_float * 3), ("direction", C.c_float * 3),
                ("color", C.c_float * 3), ("intensity", C.c_float), ("spot_angle", C.c_float),
                ("decay_exponent", C.c_float)]


class xrt_render_opts(C.Structure):
    _fields_ = [("max_reflections", C.c_int32), ("use_multisampling", C.c_int32), ("multisample_quality", C.c_int32),
                ("address_mode", C.c_int32), ("filtering", C.c_int32), ("shard_rank", C.c_int32),
                ("shard_count", C.c_int32), ("collect_stats", C.c_int32), ("n_gpus", C.c_int32), ("balance_tiles", C.c_int32), ("reserved", C.c_int32 * 2)]


class xrt_stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "rays_closest", "rays_shadow", "hits_closest", "hits_shadow", "scene_node_tests", "instance_visits",
        "mesh_aabb_tests", "mesh_queries", "node_tests", "leaf_refs", "tri_tests", "shaded_hits", "pixels",
        "algorithmic_bytes")] + [("ms_total", C.c_double), ("ms_intersect", C.c_double),
                                 ("intersect_launches", C.c_uint32), ("pieces", C.c_uint32), ("rays_traversed", C.c_uint64),
                                 ("ms_intersect_longest", C.c_double), ("mesh_queries_facing_away", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class xrt_node_info(C.Structure):
    _fields_ = [("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("is_leaf", C.c_int32), ("count", C.c_int32),
                ("dfs_index", C.c_int32), ("depth", C.c_int32), ("first_ref", C.c_int32), ("reserved", C.c_int32)]


assert C.sizeof(xrt_ray) == 32 and C.sizeof(xrt_hit) == 48 and C.sizeof(xrt_render_opts) == 48
XRT_VERSION = 203

# every symbol include/xrt.h declares: name -> (restype, argtypes)
_P = C.POINTER
_F = _P(C.c_float)
SYMBOLS = {
    "xrt_version": (C.c_int, []),
    "xrt_last_error": (C.c_char_p, []),
    "xrt_device_count": (C.c_int, [_P(C.c_int)]),
    "xrt_scene_create": (C.c_int, [C.c_int, _P(C.c_void_p)]),
    "xrt_scene_destroy": (C.c_int, [C.c_void_p]),
    "xrt_scene_add_mesh": (C.c_int, [C.c_void_p, _F, _F, _F, _F, _F, C.c_int32, _P(xrt_material), _F, _P(C.c_int32)]),
    "xrt_scene_add_object": (C.c_int, [C.c_void_p, _P(C.c_int32), C.c_int32, _F, _F, _F, _F, _P(C.c_int32)]),
    "xrt_scene_build": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "xrt_scene_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "xrt_scene_load": (C.c_int, [C.c_int, C.c_char_p, _P(C.c_void_p)]),
    "xrt_scene_get_tree": (C.c_int, [C.c_void_p, C.c_int32, _P(xrt_node_info), _P(C.c_int64), _P(C.c_int32), _P(C.c_int64)]),
    "xrt_scene_intersect": (C.c_int, [C.c_void_p, _P(xrt_ray), _P(C.c_int32), C.c_int64, _P(xrt_hit), _P(xrt_stats)]),
    "xrt_scene_intersect_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "xrt_mesh_intersect": (C.c_int, [C.c_void_p, C.c_int32, _P(xrt_ray), C.c_int64, _P(xrt_hit)]),
    "xrt_render": (C.c_int, [C.c_void_p, _P(xrt_camera), _P(xrt_light), C.c_int32, _P(xrt_render_opts), _P(C.c_uint32), _F, _P(xrt_stats)]),
    "xrt_render_begin": (C.c_int, [C.c_void_p, _P(xrt_camera), _P(xrt_light), C.c_int32, _P(xrt_render_opts), _P(C.c_uint32), _P(C.c_int32)]),
    "xrt_render_end": (C.c_int, [C.c_void_p, C.c_int32, _P(xrt_stats)]),
    "xrt_host_register": (C.c_int, [C.c_void_p, C.c_uint64]),
    "xrt_host_unregister": (C.c_int, [C.c_void_p]),
    "xrt_render_device": (C.c_int, [C.c_void_p, _P(xrt_camera), _P(xrt_light), C.c_int32, _P(xrt_render_opts), C.c_void_p, C.c_void_p, _P(xrt_stats)]),
    "xrt_render_device_begin": (C.c_int, [C.c_void_p, _P(xrt_camera), _P(xrt_light), C.c_int32, _P(xrt_render_opts), C.c_void_p, C.c_void_p, _P(C.c_int32)]),
    "xrt_render_device_end": (C.c_int, [C.c_void_p, C.c_int32, _P(xrt_stats)]),
    "xrt_shard_layout": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P(C.c_int32), _P(C.c_int32), _P(C.c_int32)]),
    "xrt_detile_device": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "xrt_scene_set_tile_table": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P(C.c_int32)]),
    "xrt_scene_tile_costs": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _F, C.c_int32]),
    "xrt_balance_tiles": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _F, C.c_int32, _P(C.c_int32)]),
    "xrt_detile_table_device": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "xrt_progress": (C.c_float, [C.c_void_p]),
    "xrt_rccl_probe": (C.c_int, []),
    "xrt_split_stats": (C.c_int, [C.c_void_p, _P(C.c_uint64), C.c_int32]),
    "xrt_generate_primary_rays": (C.c_int, [C.c_void_p, _P(xrt_camera), _P(xrt_ray)]),
}

_lib = None


class XrtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libxrt error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Load csrc/libxrt.so (built by __graft_entry__.build()).  Raises if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libxrt.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` (%s)" % LIB_PATH)
        # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Whichever is loaded first
        # serves the whole process, and tensors handed to xrt_render_device must belong to the same runtime,
        # so let torch load its copy first when it is installed.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)   # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        if l.xrt_version() != XRT_VERSION:
            raise ImportError("%s is ABI version %d, this binding is %d" % (LIB_PATH, l.xrt_version(), XRT_VERSION))
        _lib = l
    return _lib


def check(rc):
    """Map the reference's exception convention onto return codes (RT:26-27,62-63; SO:123-124; MAT:85,97)."""
    if rc == XRT_OK:
        return
    msg = lib().xrt_last_error().decode("utf-8", "replace")
    if rc == XRT_E_INVALID_ARG:
        raise ValueError("libxrt: " + msg)          # ArgumentException
    if rc == XRT_E_BUSY:
        raise RuntimeError("libxrt busy: " + msg)   # InvalidOperationException
    raise XrtError(rc, msg)
