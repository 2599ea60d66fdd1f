"""Host-side logic of the product, checked on the CPU (no GPU, no compute calls):
  * libxrt.so loads and exports every symbol include/xrt.h declares;
  * the native octree build (scene_build.cpp) produces the reference's trees (vs the oracle);
  * the pruned front-to-back traversal state machine (traverse.h), single-stepped by tests/emul,
    returns the oracle's answer bit for bit — including the leaf-group quirk, ties and ignoreTriangle;
  * error conventions of the C-ABI."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from util import hits_equal, random_rays, secondary_rays, triangle_soup

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol(xrt):
    hdr = open(os.path.join(ROOT, "include", "xrt.h")).read()
    declared = set(re.findall(r"\b(xrt_[a-z_]+)\s*\(", hdr))
    declared -= {"xrt_scene"}
    assert declared == set(xrt.abi.SYMBOLS), declared ^ set(xrt.abi.SYMBOLS)
    lib = xrt.abi.lib()                      # resolves all of them or raises
    assert lib.xrt_version() == 203
    assert C.sizeof(xrt.abi.xrt_ray) == 32 and C.sizeof(xrt.abi.xrt_hit) == 48
    assert lib.xrt_last_error() is not None


def test_csharp_binding_covers_the_header():
    """csharp/XrtNative.cs (the P/Invoke shim of the reference's C# host; uncompiled here: no .NET in the image) binds EVERY export of
    include/xrt.h, each with CallingConvention.Cdecl, and its struct sizes match the header's (counted from the field lists)."""
    hdr = open(os.path.join(ROOT, "include", "xrt.h")).read()
    cs = open(os.path.join(ROOT, "csharp", "XrtNative.cs")).read()
    declared = set(re.findall(r"\b(xrt_[a-z_]+)\s*\(", hdr)) - {"xrt_scene"}
    imports = re.findall(r"\[DllImport\(Lib, CallingConvention = CallingConvention\.Cdecl\)\]\s*public static extern (?:unsafe )?\w+ (xrt_[a-z_]+)\(", cs)
    assert set(imports) == declared, set(imports) ^ declared
    assert len(imports) == len(set(imports))
    assert len(re.findall(r"\[DllImport\(", cs)) == len(imports)          # no import without the calling convention
    # xrt_render_opts: 12 ints in both
    opts_c = re.search(r"typedef struct xrt_render_opts \{(.*?)\} xrt_render_opts;", hdr, re.S).group(1)
    n_c = sum(int(m.group(1) or 1) for m in re.finditer(r"int32_t\s+\w+(?:\[(\d+)\])?;", opts_c))
    opts_cs = re.search(r"public struct XrtRenderOpts\s*\{(.*?)\n    \}", cs, re.S).group(1)
    n_cs = sum(len(m.group(1).split(",")) for m in re.finditer(r"public int ([^;]+);", opts_cs))
    assert n_c == n_cs == 12


def test_header_is_plain_c_and_the_c_host_builds(xrt, tmp_path):
    """The boundary is a C ABI: include/xrt.h compiles as strict C99 (no C++, no torch types), and a host written in C against it
    alone (tests/c_host/c_host.c) links with libxrt.so.  Without a GPU it stops at xrt_device_count, loudly."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "hdr.c"
    src.write_text('#include "xrt.h"\nint main(void) { return (int)sizeof(xrt_hit) - 48; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), "-fsyntax-only", str(src)])
    from util import build_c_host
    exe = build_c_host()
    n = C.c_int(0)
    if xrt.abi.lib().xrt_device_count(C.byref(n)) != 0 or n.value < 1:
        r = subprocess.run([exe, "none.xrts", "none.bin", str(tmp_path / "out.rgba")], capture_output=True, text=True)
        assert r.returncode == 1 and "xrt_device_count" in r.stderr


def test_no_cpu_fallback_and_error_codes(xrt):
    lib, abi = xrt.abi.lib(), xrt.abi
    h = C.c_void_p()
    n = C.c_int(-1)
    assert lib.xrt_device_count(C.byref(n)) == 0
    if n.value == 0:
        assert lib.xrt_scene_create(0, C.byref(h)) == abi.XRT_E_NO_DEVICE
        assert b"no CPU execution path" in lib.xrt_last_error()
    assert lib.xrt_scene_create(-5, C.byref(h)) == abi.XRT_E_INVALID_ARG
    # host-only scene: trees can be built and inspected, nothing can be traced
    assert lib.xrt_scene_create(-1, C.byref(h)) == 0
    spec = xrt.configs.config("C1")
    data, m = spec.meshes[0]
    from oracle.oracle_py import material_abi
    mat, keep = material_abi(m)
    mid = C.c_int32(-1)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    sn = np.ascontiguousarray(data.surface_normal)
    assert lib.xrt_scene_add_mesh(h, fp(data.v), fp(data.n), fp(data.uv), fp(sn), fp(data.color), data.ntri, C.byref(mat), fp(data.bbox), C.byref(mid)) == 0
    assert mid.value == 0
    bad = np.array([7], dtype=np.int32)
    oid = C.c_int32(-1)
    eye = np.eye(4, dtype=np.float32).reshape(-1)
    assert lib.xrt_scene_add_object(h, bad.ctypes.data_as(C.POINTER(C.c_int32)), 1, fp(eye), fp(eye), fp(data.bbox), fp(data.bbox), C.byref(oid)) == abi.XRT_E_INVALID_ARG
    rays = xrt.rays_array([(0, 0, 5)], [(0, 0, -1)])
    hits = np.zeros(1, dtype=xrt.HIT_DTYPE)
    assert lib.xrt_scene_intersect(h, rays.ctypes.data_as(C.POINTER(abi.xrt_ray)), None, 1, hits.ctypes.data_as(C.POINTER(abi.xrt_hit)), None) == abi.XRT_E_NO_DEVICE
    ok = np.array([0], dtype=np.int32)
    assert lib.xrt_scene_add_object(h, ok.ctypes.data_as(C.POINTER(C.c_int32)), 1, fp(eye), fp(eye), fp(data.bbox), fp(data.bbox), C.byref(oid)) == 0
    assert lib.xrt_scene_build(h, 0, 0) == 0
    nn, nr = C.c_int64(0), C.c_int64(0)
    assert lib.xrt_scene_get_tree(h, 0, None, C.byref(nn), None, C.byref(nr)) == 0 and (nn.value, nr.value) == (1, 12)
    cam = xrt.abi.xrt_camera()
    opts = xrt.abi.xrt_render_opts()
    out = np.zeros(4, dtype=np.uint32)
    assert lib.xrt_render(h, C.byref(cam), None, 0, C.byref(opts), out.ctypes.data_as(C.POINTER(C.c_uint32)), None, None) == abi.XRT_E_NO_DEVICE
    with pytest.raises(xrt.abi.XrtError):
        abi.check(abi.XRT_E_NO_DEVICE)
    with pytest.raises(ValueError):
        abi.check(abi.XRT_E_INVALID_ARG)
    assert lib.xrt_scene_destroy(h) == 0
    # shard layout: 64x8 tiles dealt round-robin
    tx, ty, tpr = C.c_int32(), C.c_int32(), C.c_int32()
    assert lib.xrt_shard_layout(1920, 1080, 8, C.byref(tx), C.byref(ty), C.byref(tpr)) == 0
    assert (tx.value, ty.value, tpr.value) == (30, 135, 507)
    assert lib.xrt_shard_layout(100, 20, 3, C.byref(tx), C.byref(ty), C.byref(tpr)) == 0 and (tx.value, ty.value, tpr.value) == (2, 3, 2)


SCENES = {
    "crate": lambda x: x.configs.config("C1"),
    "grid": lambda x: x.configs.crate_grid_scene(96, 54),
    "h64": lambda x: x.configs.heightfield_scene(96, 54, m=64),
    "h224": lambda x: x.configs.heightfield_scene(80, 45, m=224),
}


@pytest.mark.parametrize("name", list(SCENES))
def test_native_tree_equals_reference_tree(xrt, orc, emul, name):
    spec = SCENES[name](xrt)
    o, e = orc.OracleScene(spec), emul.EmulScene(spec)
    for mesh in (-1, 0):
        no, ro = o.tree(mesh)
        ne, re_ = e.tree(mesh)
        assert no.tobytes() == ne.tobytes() and np.array_equal(ro, re_)
    st = e.tree_stats(0)
    no, ro = o.tree(0)
    assert st["nodes"] == len(no) and st["leaves"] == int(no["is_leaf"].sum())
    assert st["empty_leaves"] == int(((no["is_leaf"] == 1) & (no["count"] == 0)).sum())


def test_tree_sizes_of_the_baseline_fixtures(xrt, emul):
    # crate(11): 137 nodes / 120 leaves / 16 empty / 2,484 refs / depth 3;
    # heightfield(224): 25,577 / 22,380 / 10,892 / 257,946 / depth 6
    e = emul.EmulScene(xrt.configs.crate_grid_scene(16, 16))
    st, (n, r) = e.tree_stats(0), e.tree(0)
    assert (st["nodes"], st["leaves"], st["empty_leaves"], len(r), st["max_depth"]) == (137, 120, 16, 2484, 3)
    ns, rs = e.tree(-1)
    assert len(ns) == 9 and ns[0]["count"] == 64 and (ns["count"][1:] == 16).all()
    e = emul.EmulScene(xrt.configs.heightfield_scene(16, 16, m=224))
    st, (n, r) = e.tree_stats(0), e.tree(0)
    assert (st["nodes"], st["leaves"], st["empty_leaves"], len(r), st["max_depth"]) == (25577, 22380, 10892, 257946, 6)


@pytest.mark.parametrize("name", list(SCENES))
def test_pruned_traversal_equals_reference_answer(xrt, orc, emul, name):
    spec = SCENES[name](xrt)
    o, e = orc.OracleScene(spec), emul.EmulScene(spec)
    rays = o.primary_rays()
    ho = o.intersect(rays)
    assert hits_equal(ho, e.intersect(rays)) == {}
    sec = secondary_rays(xrt, ho)
    if len(sec):
        assert hits_equal(o.intersect(sec), e.intersect(sec)) == {}
    # mesh-level query (MO:259) in object space
    assert hits_equal(o.mesh_intersect(0, rays[::5]), e.intersect(rays[::5], mode=1, mesh=0)) == {}
    # single-object scenes also run through the collapsed prologue the GPU uses for them (MODE_SINGLE)
    if len(spec.objects) == 1:
        assert hits_equal(ho, e.intersect(rays, mode=2)) == {}
        if len(sec):
            assert hits_equal(o.intersect(sec), e.intersect(sec, mode=2)) == {}


def soup_spec(xrt, n, seed, threshold, size):
    s = xrt.configs.SceneSpec("soup")
    s.meshes.append((triangle_soup(n, seed, size), xrt.configs.material(0.5)))
    s.objects.append(([0], (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)))
    s.camera = xrt.configs.camera((0, 3, 3), (0, 0, 0))
    s.lights = [xrt.configs.spot((0, 5, 5))]
    s.mesh_threshold = threshold
    return s.with_size(32, 32)


@pytest.mark.parametrize("n,seed,threshold,size", [(60, 3, 2, 0.9), (300, 5, 4, 0.5), (2000, 7, 20, 0.15), (500, 9, 50, 1.5)])
def test_leaf_group_quirk_and_ties_on_triangle_soups(xrt, orc, emul, n, seed, threshold, size):
    """Large overlapping triangles + tiny thresholds: hits outside their leaf box, triangles missing from
    leaves they cross (Q5), later buckets holding nearer hits (Q1) — the arg-min formulation must agree."""
    spec = soup_spec(xrt, n, seed, threshold, size)
    o, e = orc.OracleScene(spec), emul.EmulScene(spec)
    rays = random_rays(xrt, 6000, seed + 100)
    ho = o.intersect(rays)
    assert ho["hit"].sum() > 500
    assert hits_equal(ho, e.intersect(rays)) == {}
    sec = secondary_rays(xrt, ho, seed)
    assert hits_equal(o.intersect(sec), e.intersect(sec)) == {}


def test_driver_entry_point_builds():
    """__graft_entry__.build() is what the driver runs first: it must build (everything is up to date here) and agree with the header's ABI version."""
    import re
    import __graft_entry__ as g
    pkg = g.build()
    ver = int(re.search(r"#define XRT_VERSION (\d+)", open(os.path.join(ROOT, "include", "xrt.h")).read()).group(1))
    assert pkg.abi.lib().xrt_version() == ver == pkg.abi.XRT_VERSION


def test_equal_distance_ties_inside_a_leaf(xrt, orc, emul):
    """MO:293-294: of two triangles of one leaf hit at exactly the same distance the earlier in the list wins.  The device arrays store the references of a big
    leaf in runs of neighbouring triangles (scene_host.cpp spatial_runs) and the traversal settles such ties by the smaller triangle index (a leaf's list is
    ascending in it): the stepper, which walks those arrays, must give the reference's answers where exact ties are the rule."""
    from util import coplanar_tie_scene
    spec, sets = coplanar_tie_scene(xrt)
    o, e = orc.OracleScene(spec), emul.EmulScene(spec)
    tied = 0
    for rays in sets:
        ho = o.intersect(rays)
        assert hits_equal(ho, e.intersect(rays)) == {}
        tied += int((ho["hit"] != 0).sum())
    assert tied > 400


def test_instances_rotated_scaled_and_shared_ignore(xrt, orc, emul):
    """Two-level scene with rotation and non-unit scale (Q6/Q7 awake) and a mesh shared by all instances:
    ignoreTriangle applies in every instance (Q9)."""
    s = xrt.configs.SceneSpec("inst")
    s.meshes.append((xrt.fixtures.crate(3), xrt.configs.material(0.5)))
    s.meshes.append((triangle_soup(80, 11, 0.4), xrt.configs.material(0.2)))
    k = 0
    for ix in range(5):
        for iz in range(5):
            rot = (0.1 * ix, 0.37 * iz, 0.05 * (ix + iz))
            sc = (1.0 + 0.1 * ix, 1.0, 0.8 + 0.1 * iz)
            s.objects.append(([0] if (k % 3) else [0, 1], (-60.0 + 30.0 * ix, 2.0 * (k % 2), -60.0 + 30.0 * iz), rot, sc))
            k += 1
    s.camera = xrt.configs.camera((0, 80, 160), (0, 0, 0))
    s.lights = [xrt.configs.spot((0, 100, 100))]
    s = s.with_size(96, 54)
    o, e = orc.OracleScene(s), emul.EmulScene(s)
    ns, rs = o.tree(-1)
    assert len(ns) > 1, "scene octree must split (25 bodies > 20)"
    rays = o.primary_rays()
    ho = o.intersect(rays)
    assert ho["hit"].sum() > 300
    assert hits_equal(ho, e.intersect(rays)) == {}
    sec = secondary_rays(xrt, ho)
    assert hits_equal(o.intersect(sec), e.intersect(sec)) == {}


def test_degenerate_rays(xrt, orc, emul):
    spec = SCENES["h64"](xrt)
    o, e = orc.OracleScene(spec), emul.EmulScene(spec)
    nan = float("nan")
    rays = xrt.rays_array([(0, 50, 0), (0, 50, 0), (0, 50, 0), (1e30, 0, 0), (0, 50, 0), (0, 4.0, 0)],
                          [(0, 0, 0), (nan, -1, 0), (0, -1, 0), (-1, 0, 0), (1e-7, -1, 1e-7), (0, 1, 0)])
    assert hits_equal(o.intersect(rays), e.intersect(rays)) == {}


def test_conservative_skips_under_grazing_rays(xrt, orc, emul):
    """The two work-avoiding tests (world-space object pre-cull, all-back-facing leaf skip) must never change an
    answer: random rotations / non-uniform scales, rays aimed at the corners and edges of the objects' boxes
    (the pre-cull margin) and rays almost tangent to a smooth surface (N.D near zero, the leaf-skip margin)."""
    rng = np.random.default_rng(17)
    s = xrt.configs.SceneSpec("graze")
    s.meshes.append((xrt.fixtures.crate(2), xrt.configs.material(0.5)))
    s.meshes.append((xrt.fixtures.heightfield(40), xrt.configs.material(0.3)))
    pos = []
    for k in range(24):
        p = tuple(float(x) for x in rng.uniform(-150, 150, size=3))
        rot = tuple(float(x) for x in rng.uniform(-3.1, 3.1, size=3))
        sc = tuple(float(x) for x in rng.uniform(0.3, 3.0, size=3))
        s.objects.append(([k % 2], p, rot, sc))
        pos.append(p)
    s.camera = xrt.configs.camera((0, 200, 400), (0, 0, 0))
    s.lights = [xrt.configs.spot((0, 300, 300))]
    s = s.with_size(64, 36)
    o, e = orc.OracleScene(s), emul.EmulScene(s)
    # rays from far away through points jittered around each object's world-space box corners
    origins, dirs = [], []
    for k, (ids, p, rot, sc) in enumerate(s.objects):
        bb = s.meshes[ids[0]][0].bbox
        world, inv, wbb = xrt.xna.build_world(sc, rot, p, bb)
        W = xrt.xna.as_array(world).reshape(4, 4).astype(np.float64)
        for c in range(8):
            corner = np.array([bb[3 if c & 1 else 0], bb[4 if c & 2 else 1], bb[5 if c & 4 else 2], 1.0])
            wc = corner @ W
            for j in range(12):
                tgt = wc[:3] + rng.normal(scale=10.0 ** rng.uniform(-6, 0), size=3)
                org = rng.normal(size=3)
                org = org / np.linalg.norm(org) * 900.0
                d = tgt - org
                origins.append(org)
                dirs.append(d / np.linalg.norm(d))
    rays = xrt.rays_array(np.array(origins, dtype=np.float32), np.array(dirs, dtype=np.float32))
    ho = o.intersect(rays)
    assert hits_equal(ho, e.intersect(rays)) == {}
    # near-tangent rays leaving the smooth surface: directions within 1e-6 .. 1e-1 of the tangent plane
    hf = s.meshes[1][0]
    m = (ho["hit"] == 1) & (ho["mesh"] == 1)
    if m.sum() < 50:
        prim = o.primary_rays()
        hp = o.intersect(prim)
        m2 = (hp["hit"] == 1) & (hp["mesh"] == 1)
        P, T, Mh = hp["w"][m2], hp["tri"][m2], hp["mesh"][m2]
    else:
        P, T, Mh = ho["w"][m], ho["tri"][m], ho["mesh"][m]
    if len(P):
        N = hf.surface_normal[T].astype(np.float64)
        t = rng.normal(size=N.shape)
        t -= (t * N).sum(axis=1, keepdims=True) * N
        t /= np.linalg.norm(t, axis=1, keepdims=True)
        eps = (10.0 ** rng.uniform(-7, -1, size=(len(P), 1))) * rng.choice([-1.0, 1.0], size=(len(P), 1))
        d = t + eps * N
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        sec = xrt.rays_array(P, d.astype(np.float32), Mh, T)
        assert hits_equal(o.intersect(sec), e.intersect(sec)) == {}


def test_object_precull_far_origins(xrt, orc, emul):
    """with the world-space pre-cull margin independent of the ray, rays from 1e4 units away lost hits the
    reference finds (and found the wrong body).  The margin is now m(|o|) (scene_host.cpp): 96k rays from
    radius 1e3, 1e4 and 1e5 answer exactly like the oracle -- and with the margin cut to a thousandth of the proven bound
    the same rays DO show mismatches, i.e. the test reaches the regime the bound is for."""
    from util import far_origin_scene, far_origin_rays
    s = far_origin_scene(xrt)
    o, e = orc.OracleScene(s), emul.EmulScene(s)
    weak = emul.EmulScene(s, cull_safety=1e-3)
    bitten = 0
    for radius, seed in ((1e3, 1), (1e4, 2), (1e5, 3)):
        rays = far_origin_rays(xrt, s, radius, 2000, seed)
        ho = o.intersect(rays)
        assert hits_equal(ho, e.intersect(rays)) == {}, radius
        bitten += len(hits_equal(ho, weak.intersect(rays)))
    assert bitten > 0


def test_object_precull_adversarial_transforms(xrt, orc, emul):
    """Rays grazing the un-enlarged world hull of every body within 1e-7 .. 1e-3 of its size, bodies scaled by 1e-3 .. 1e3
    with condition numbers up to ~350, origins from one to 1e5 hull sizes away (tests/util.py)."""
    from util import precull_adversarial_scene, grazing_rays
    s = precull_adversarial_scene(xrt)
    o, e = orc.OracleScene(s), emul.EmulScene(s)
    rays = grazing_rays(xrt, s, 6000, 11)
    ho = o.intersect(rays)
    assert 0.02 < (ho["hit"] != 0).mean() < 0.98
    assert hits_equal(ho, e.intersect(rays)) == {}


@pytest.mark.parametrize("kind", ["heightfield", "soup", "crate"])
def test_tight_leaf_boxes_adversarial(xrt, orc, emul, kind):
    """A leaf is skipped when the ray misses the box of its triangles' vertices grown by rho(ray, leaf) (xrt_core.h
    leaf_certainly_missed): rays placed on that decision boundary -- grazing the vertex boxes, lying in
    triangle planes, through vertices and edges, from up to 1e5 leaf sizes away -- answer exactly like the oracle, which tests
    every reference of every leaf it enters.  With the margin cut to 1e-4 of the proven bound the same rays DO lose hits."""
    from util import tight_box_adversarial_rays
    if kind == "heightfield":
        spec = xrt.configs.heightfield_scene(64, 36, m=48)
    elif kind == "crate":
        spec = xrt.configs.config("C1")
        spec.meshes[0] = (xrt.fixtures.crate(7), spec.meshes[0][1])
    else:
        spec = soup_spec(xrt, 900, 5, 12, 0.3)
    o, e = orc.OracleScene(spec), emul.EmulScene(spec)
    off, weak = emul.EmulScene(spec, leaf_cull=0.0), emul.EmulScene(spec, leaf_cull=1e-4)
    nodes, refs = o.tree(0)
    rays = tight_box_adversarial_rays(xrt, spec.meshes[0][0], nodes, refs, 60, 150, 23)
    ho = o.mesh_intersect(0, rays)
    assert 0.05 < (ho["hit"] != 0).mean() < 0.95
    assert hits_equal(ho, e.intersect(rays, mode=1, mesh=0)) == {}
    assert hits_equal(ho, off.intersect(rays, mode=1, mesh=0)) == {}
    if kind == "heightfield":   # (the others' boxes are too loose for 25,000 rays to find the gap)
        assert len(hits_equal(ho, weak.intersect(rays, mode=1, mesh=0))) > 0


REF_CONTENT = "/root/reference/RayTraceProject/RayTraceProjectContent"


@pytest.mark.skipif(not os.path.isdir(REF_CONTENT), reason="the reference tree is only mounted in the build container")
def test_fbx_import_of_the_reference_assets(xrt):
    """Asset ingestion (fbx.py): the binary FBX 6100 crate imports to exactly the hand-written crate
    fixture, the ASCII FBX 6.1 sphere to the committed data fixture, and the other meshes of the content project parse."""
    import importlib
    fbx = importlib.import_module("xna-ray-trace_amd.fbx")
    cm, up = fbx.load_fbx(os.path.join(REF_CONTENT, "Crate_Fragile.FBX"))
    assert up == 2 and cm[0].normal_mapping == "ByPolygonVertex"
    got, ref = fbx.import_mesh(cm[0], up, scale=0.6, apply_node_transform=False), xrt.fixtures.crate(1)
    for f in ("v", "n", "uv", "surface_normal", "bbox"):
        assert np.array_equal(getattr(got, f), getattr(ref, f)), f
    sm, up = fbx.load_fbx(os.path.join(REF_CONTENT, "Sphere.fbx"))
    sph = fbx.import_mesh(sm[0], up, scale=2.0, diffuse_color=(255, 0, 0, 100))
    z = np.load(os.path.join(ROOT, "tests", "golden", "sphere_mesh.npz"))
    assert sph.ntri == 960 and np.array_equal(sph.v, z["v"]) and np.array_equal(sph.n, z["n"]) and np.array_equal(sph.color, z["color"])
    assert ((sph.surface_normal * sph.n[:, 0, :]).sum(axis=1) > 0.9).all()      # winding agrees with the exported normals (K4)
    for name, ntri in (("torus.fbx", 1152), ("cube.fbx", 12), ("ground.fbx", 2)):
        ms, up = fbx.load_fbx(os.path.join(REF_CONTENT, name))
        assert fbx.import_mesh(ms[0], up).ntri == ntri
    # ModelProcessor rotation parameters (contentproj:112-122,186-195,197-206): prism2 and chesspiece equal their committed
    # fixtures, a rotation of 180 degrees about Y mirrors x and z of the unrotated import, winding still agrees with the normals
    zc = np.load(os.path.join(ROOT, "tests", "golden", "content_meshes.npz"))
    for name, asset, kw in (("prism", "prism2.fbx", dict(rotation=(-90.0, 0.0, 0.0), diffuse_color=(255, 255, 255, 100))),
                            ("chesspiece", "chesspiece.fbx", dict(scale=3.0, rotation=(-90.0, 0.0, 0.0)))):
        ms, up = fbx.load_fbx(os.path.join(REF_CONTENT, asset))
        got = fbx.import_mesh(ms[0], up, **kw)
        assert np.array_equal(got.v, zc[name + "_v"]) and np.array_equal(got.n, zc[name + "_n"]) and np.array_equal(got.color, zc[name + "_color"])
        assert ((got.surface_normal * got.n[:, 0, :]).sum(axis=1) > 0).mean() > 0.99
    # binary FBX 7.1 (zlib-compressed arrays, Geometry -> Model connections): the last two of the content directory's 17 files
    ms, up = fbx.load_fbx(os.path.join(REF_CONTENT, "dna_exported_from_max2011.FBX"))
    assert len(ms) == 40 and all(len(m.polygons) == 1596 for m in ms[:3])
    dna = fbx.import_mesh(ms[0], up)
    assert dna.ntri == 1728 and ((dna.surface_normal * dna.n[:, 0, :]).sum(axis=1) > 0).all()
    ms, up = fbx.load_fbx(os.path.join(REF_CONTENT, "Sony_3D_Logo_by_Peter_Iliev_fbx.FBX"))
    assert len(ms) == 1 and fbx.import_mesh(ms[0], up).ntri == 3268
    import glob
    files = sorted(glob.glob(os.path.join(REF_CONTENT, "*.fbx")) + glob.glob(os.path.join(REF_CONTENT, "*.FBX")))
    assert len(files) == 17 and all(len(fbx.load_fbx(f)[0]) >= 1 for f in files)
    ms, up = fbx.load_fbx(os.path.join(REF_CONTENT, "wossy.fbx"))
    plain, turned = fbx.import_mesh(ms[0], up, scale=32.0), fbx.import_mesh(ms[0], up, scale=32.0, rotation=(0.0, 180.0, 0.0))
    assert turned.ntri == plain.ntri == 9420
    assert np.allclose(turned.v[..., 0], -plain.v[..., 0], atol=2e-4) and np.allclose(turned.v[..., 2], -plain.v[..., 2], atol=2e-4)
    assert np.array_equal(turned.v[..., 1], plain.v[..., 1])


def _fbx7_bytes(tree, version):
    """A minimal binary FBX 7.x writer for the tests (the layout fbx.py documents): 32-bit record headers below 7500, 64-bit from 7500 on;
    arrays uncompressed on even calls, zlib on odd ones."""
    import struct
    import zlib
    wide = version >= 7500
    state = {"n": 0}

    def prop(v):
        if isinstance(v, str):
            raw = v.encode("latin-1")
            return b"S" + struct.pack("<I", len(raw)) + raw
        if isinstance(v, bool):
            return b"C" + struct.pack("<?", v)
        if isinstance(v, int):
            return (b"L" + struct.pack("<q", v)) if abs(v) >= 1 << 31 else (b"I" + struct.pack("<i", v))
        if isinstance(v, float):
            return b"D" + struct.pack("<d", v)
        a = np.asarray(v)
        code, dt = (b"d", "<f8") if a.dtype.kind == "f" else (b"i", "<i4")
        raw = a.astype(dt).tobytes()
        state["n"] += 1
        if state["n"] % 2:
            z = zlib.compress(raw)
            return code + struct.pack("<III", a.size, 1, len(z)) + z
        return code + struct.pack("<III", a.size, 0, len(raw)) + raw

    def node(name, props, children, at):
        head = 25 if wide else 13
        pl = b"".join(prop(x) for x in props)
        body = b""
        pos = at + head + len(name) + len(pl)
        for c in children:
            cb = node(c[0], c[1], c[2], pos + len(body))
            body += cb
        if children:
            body += bytes(head)   # null record closes a list of children
        end = pos + len(body)
        return struct.pack("<QQQ" if wide else "<III", end, len(props), len(pl)) + bytes([len(name)]) + name.encode("latin-1") + pl + body

    out = b"Kaydara FBX Binary  \x00\x1a\x00" + struct.pack("<I", version)
    assert len(out) == 27
    for t in tree:
        out += node(t[0], t[1], t[2], len(out))
    return out + bytes(25 if wide else 13)


def test_fbx_75_records_and_vertex_colours(xrt, tmp_path):
    """fbx.py beyond the reference's own assets (VERDICT r2 missing #7): binary FBX 7500 (64-bit record headers) reads like 7400, and the
    processor parameter UseVertexColors (TMP:93-101, 224-227) takes a triangle's colour from the colour channel at the triangle's first
    index, through an XNA `Color` (bytes, round-half-even) -- from binary 7.x (ByPolygonVertex, IndexToDirect) and ASCII 6.1 (ByVertice, Direct)."""
    import importlib
    fbx = importlib.import_module("xna-ray-trace_amd.fbx")
    verts = [0.0, 0.0, 0.0, 2.0, 0.0, 0.0, 2.0, 2.0, 0.0, 0.0, 2.0, 0.0, 1.0, 1.0, 3.0]
    pvi = [0, 1, 2, -4, 0, 1, -5]            # a quad and a triangle
    normals = [0.0, 0.0, 1.0] * 7
    palette = [1.0, 0.0, 0.0, 1.0, 0.5, 0.1, 0.25, 1.0, 0.0, 0.0, 1.0, 0.5]     # red, an inexact one, half-transparent blue
    cidx = [1, 0, 0, 0, 2, 0, 0]             # polygon-vertex -> palette entry
    tree = [("GlobalSettings", [], [("Properties70", [], [("P", ["UpAxis", "int", "Integer", "", 1], [])])]),
            ("Objects", [], [
                ("Geometry", [1001, "quadtri\x00\x01Geometry", "Mesh"], [
                    ("Vertices", [np.array(verts)], []), ("PolygonVertexIndex", [np.array(pvi)], []),
                    ("LayerElementNormal", [0], [("MappingInformationType", ["ByPolygonVertex"], []), ("ReferenceInformationType", ["Direct"], []),
                                                 ("Normals", [np.array(normals)], [])]),
                    ("LayerElementColor", [0], [("MappingInformationType", ["ByPolygonVertex"], []), ("ReferenceInformationType", ["IndexToDirect"], []),
                                                ("Colors", [np.array(palette)], []), ("ColorIndex", [np.array(cidx)], [])])]),
                ("Model", [2002, "quadtri\x00\x01Model", "Mesh"], [("Properties70", [], [("P", ["Lcl Translation", "Lcl Translation", "", "A", 1.0, 2.0, 3.0], [])])])]),
            ("Connections", [], [("C", ["OO", 1001, 2002], [])])]
    got = {}
    for version in (7400, 7500):
        f = tmp_path / ("q%d.fbx" % version)
        f.write_bytes(_fbx7_bytes(tree, version))
        ms, up = fbx.load_fbx(str(f))
        assert up == 1 and len(ms) == 1 and ms[0].translation == (1.0, 2.0, 3.0) and ms[0].polygons == [[0, 1, 2, 3], [0, 1, 4]]
        got[version] = (fbx.import_mesh(ms[0], up, diffuse_color=(10, 20, 30, 255)), fbx.import_mesh(ms[0], up, diffuse_color=(10, 20, 30, 255), use_vertex_colors=True))
    for a, b in zip(got[7400], got[7500]):
        assert np.array_equal(a.v, b.v) and np.array_equal(a.n, b.n) and np.array_equal(a.color, b.color)
    plain, coloured = got[7500]
    assert plain.ntri == 3 and np.array_equal(plain.color, np.tile(np.array([10, 20, 30, 255], dtype=np.float32) / np.float32(255.0), (3, 1)))
    # first polygon-vertex of the quad has palette entry 1: (0.5, 0.1, 0.25, 1) -> bytes (128, 26, 64, 255): 127.5 and 63.75 round to even / nearest
    q = np.array([128, 26, 64, 255], dtype=np.float32) / np.float32(255.0)
    t = np.array([0, 0, 255, 128], dtype=np.float32) / np.float32(255.0)   # the triangle's first polygon-vertex: entry 2, alpha 0.5 -> 127.5 -> 128
    assert np.array_equal(coloured.color, np.stack([q, q, t]))
    assert fbx.xna_color_bytes((1.5, -0.2, float("nan"), 2.5 / 255.0)) == (255, 0, 0, 2)   # clamped; NaN -> 0; 2.5 -> 2 (to even)
    wide = _fbx7_bytes(tree, 7500)
    with pytest.raises(ValueError):
        fbx.load_binary7(wide[:27] + (10 ** 9).to_bytes(8, "little") + wide[35:])   # a record that ends outside the file
    # ASCII 6.1, colours by vertex, direct
    text = """FBXHeaderExtension:  { FBXVersion: 6100 }
Objects:  {
    Model: "Model::tri", "Mesh" {
        Properties60:  { Property: "Lcl Translation", "Lcl Translation", "A+",0,0,0 }
        Vertices: 0,0,0,1,0,0,0,1,0
        PolygonVertexIndex: 0,1,-3
        LayerElementNormal: 0 { MappingInformationType: "ByVertice"
            ReferenceInformationType: "Direct"
            Normals: 0,0,1,0,0,1,0,0,1 }
        LayerElementColor: 0 { MappingInformationType: "ByVertice"
            ReferenceInformationType: "Direct"
            Colors: 0.2,0.4,0.6,1,1,1,1,1,0,0,0,1 }
    }
}
"""
    f = tmp_path / "t.fbx"
    f.write_text(text)
    ms, up = fbx.load_fbx(str(f))
    m = fbx.import_mesh(ms[0], up, use_vertex_colors=True)
    assert m.ntri == 1 and np.array_equal(m.color[0], np.array([51, 102, 153, 255], dtype=np.float32) / np.float32(255.0))
    assert np.array_equal(fbx.import_mesh(ms[0], up).color[0], np.ones(4, dtype=np.float32))


def test_default_game_scene_traversal(xrt, orc, emul):
    """G1 = the scene of Game1.LoadContent (4 Transparent spheres sharing one Mesh): scene octree, interpolated
    normals and the shared-mesh ignoreTriangle (Q9) through the emulated traversal."""
    spec = xrt.configs.default_game_scene(64, 64, 2)
    o, e = orc.OracleScene(spec), emul.EmulScene(spec)
    rays = o.primary_rays()
    ho = o.intersect(rays)
    assert ho["hit"].sum() > 100
    assert hits_equal(ho, e.intersect(rays)) == {}
    sec = secondary_rays(xrt, ho)
    assert hits_equal(o.intersect(sec), e.intersect(sec)) == {}


def test_reference_content_scene2_on_the_cpu(xrt, orc, emul):
    """The assets that need the processor's rotation parameters (glass prism, chess piece) on the file-textured ground."""
    spec = xrt.configs.content_scene2(96, 54)
    o, e = orc.OracleScene(spec), emul.EmulScene(spec)
    prim = o.primary_rays()
    o_hits = o.intersect(prim)
    assert (o_hits["hit"] != 0).mean() > 0.3 and len(np.unique(o_hits["mesh"][o_hits["hit"] != 0])) == 3
    assert hits_equal(o_hits, e.intersect(prim)) == {}
    sec = secondary_rays(xrt, o_hits, seed=6)
    assert hits_equal(o.intersect(sec), e.intersect(sec)) == {}


def test_scene_file_round_trip(xrt, tmp_path):
    """xrt_scene_save / xrt_scene_load (host-only scenes, no GPU): the loaded scene rebuilds byte-identical octrees for every
    mesh and for the bodies -- textures, premultiplied copy and shared meshes included; corrupt files are refused."""
    spec = xrt.configs.content_scene2(64, 36)
    rng = np.random.default_rng(1)
    argb = rng.integers(0, 2 ** 32, size=(8, 16), dtype=np.uint64).astype(np.uint32)
    spec.meshes.append((xrt.fixtures.crate(2), xrt.configs.material(0.4, texture=argb, texture_pargb=(argb & np.uint32(0xff7f7f7f)))))
    spec.objects.append(([3, 0], (9.0, 1.0, 2.0), (0.1, 0.2, 0.3), (1.0, 2.0, 0.5)))
    scene, _ = xrt.configs.build_product(spec, device=-1)
    path = str(tmp_path / "scene.xrts")
    scene.Save(path)
    loaded = xrt.api.OctreeSpatialManager.Load(path, device=-1)
    import ctypes as C

    def trees(handle, n_meshes):
        out = []
        for mid in range(-1, n_meshes):
            nn, nr = C.c_int64(0), C.c_int64(0)
            xrt.abi.check(xrt.abi.lib().xrt_scene_get_tree(handle, mid, None, C.byref(nn), None, C.byref(nr)))
            nodes = np.zeros(nn.value, dtype=xrt.NODE_DTYPE)
            refs = np.zeros(max(nr.value, 1), dtype=np.int32)
            xrt.abi.check(xrt.abi.lib().xrt_scene_get_tree(handle, mid, nodes.ctypes.data_as(C.POINTER(xrt.abi.xrt_node_info)), C.byref(nn),
                                                           refs.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nr)))
            out.append((nodes.tobytes(), refs[: nr.value].tobytes()))
        return out
    assert trees(scene.handle, 4) == trees(loaded.handle, 4)
    scene.Save(str(tmp_path / "again.xrts"))
    loaded.Save(str(tmp_path / "loaded.xrts"))
    assert open(str(tmp_path / "again.xrts"), "rb").read() == open(str(tmp_path / "loaded.xrts"), "rb").read() == open(path, "rb").read()
    data = open(path, "rb").read()
    for bad in (data[: len(data) // 2], b"XRTSCENX" + data[8:], data[:8] + b"\x09\0\0\0" + data[12:]):
        open(str(tmp_path / "bad.xrts"), "wb").write(bad)
        with pytest.raises(ValueError):
            xrt.api.OctreeSpatialManager.Load(str(tmp_path / "bad.xrts"), device=-1)


def test_reference_content_scene_on_the_cpu():
    """The content scene (assets of the reference, imported as data): the product's host-side trees and the CPU
    single-stepper of its traversal agree with the oracle on every primary ray and on rays leaving the surfaces."""
    import importlib
    xrt = importlib.import_module("xna-ray-trace_amd")
    from oracle import oracle_py as orc
    import emul_py
    from util import hits_equal, secondary_rays
    spec = xrt.configs.content_scene(96, 54)
    o, e = orc.OracleScene(spec), emul_py.EmulScene(spec)
    prim = o.primary_rays()
    o_hits = o.intersect(prim)
    assert (o_hits["hit"] != 0).mean() > 0.3 and len(np.unique(o_hits["mesh"][o_hits["hit"] != 0])) == 5
    assert hits_equal(o_hits, e.intersect(prim)) == {}
    sec = secondary_rays(xrt, o_hits, seed=5)
    assert hits_equal(o.intersect(sec), e.intersect(sec)) == {}


def test_scene_file_hostile_and_truncated_headers(xrt, tmp_path):
    """xrt_scene_load trusts nothing in the file: counts and array sizes are bounded by the bytes the file really
    holds before anything is allocated, the records go through the checks of xrt_scene_add_mesh / _add_object, and no C++
    exception crosses the C boundary -- every bad file is XRT_E_INVALID_ARG (ValueError in the mirror) in well under a second."""
    import struct
    import time

    def load(data):
        p = str(tmp_path / "h.xrts")
        open(p, "wb").write(data)
        t0 = time.perf_counter()
        try:
            xrt.api.OctreeSpatialManager.Load(p, device=-1)
            return None, time.perf_counter() - t0
        except ValueError as e:
            return str(e), time.perf_counter() - t0
    head = b"XRTSCENE" + struct.pack("<I", 1)

    def mesh(ntri, flags, tw, th, body=b""):
        return struct.pack("<i6fffiii", ntri, 0, 0, 0, 1, 1, 1, 0.5, 0.0, flags, tw, th) + body
    tri = struct.pack("<9f", 0, 0, 0, 1, 0, 0, 0, 1, 0) + struct.pack("<9f", *([0, 0, 1] * 3)) + struct.pack("<6f", 0, 0, 1, 0, 0, 1) + \
        struct.pack("<3f", 0, 0, 1) + struct.pack("<4f", 1, 1, 1, 1)
    cases = {
        "20-byte file announcing 2^24 meshes": head + struct.pack("<II", 1 << 24, 0),
        "2^24 objects, none present": head + struct.pack("<II", 0, 1 << 24),
        "a mesh of 2^28 - 1 triangles in a 60-byte file": head + struct.pack("<II", 1, 0) + mesh((1 << 28) - 1, 0, 0, 0),
        "UseTexture with a 0 x 0 texture": head + struct.pack("<II", 1, 0) + mesh(1, 4, 0, 0, tri),
        "premultiplied texels without UseTexture": head + struct.pack("<II", 1, 0) + mesh(1, 8, 1, 1, tri + struct.pack("<I", 0)),
        "a 2^15 x 2^15 texture that is not there": head + struct.pack("<II", 1, 0) + mesh(1, 4, 1 << 15, 1 << 15, tri),
        "unknown flag bits": head + struct.pack("<II", 1, 0) + mesh(1, 64, 0, 0, tri),
        "an object naming a mesh that does not exist": head + struct.pack("<II", 1, 1) + mesh(1, 0, 0, 0, tri) + struct.pack("<ii", 1, 7) + b"\0" * (4 * 44),
        "an object with 2^20 mesh ids, none present": head + struct.pack("<II", 0, 1) + struct.pack("<i", 1 << 20),
        "empty file": b"",
    }
    for what, data in cases.items():
        err, dt = load(data)
        assert err is not None and dt < 1.0, (what, err, dt)
    # ... and the smallest honest file loads: one triangle, one body
    ok = head + struct.pack("<II", 1, 1) + mesh(1, 0, 0, 0, tri) + struct.pack("<ii", 1, 0) + \
        struct.pack("<16f", *np.eye(4).ravel()) * 2 + struct.pack("<6f", 0, 0, 0, 1, 1, 1) * 2
    assert load(ok)[0] is None


def test_rccl_load_failure_is_an_error_code_not_a_crash(xrt, tmp_path):
    """with librccl missing, RcclGather::load built its message from a second dlerror() call (NULL -> std::string(NULL),
    undefined behaviour) -- on a host without RCCL every n_gpus > 1 render would have crashed instead of returning XRT_E_RCCL.
    xrt_rccl_probe takes the same load path without touching a device: a library that does not exist and one that lacks the
    RCCL entry points both come back as XRT_E_RCCL with the loader's message.  (In a child process: the override is read from
    the environment at load time and a loaded library stays loaded.)"""
    import subprocess
    import sys
    code = ("import importlib, sys; sys.path.insert(0, %r); x = importlib.import_module('xna-ray-trace_amd'); l = x.abi.lib(); "
            "rc = l.xrt_rccl_probe(); print(rc, l.xrt_last_error().decode())") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for libname, needle in (("/nonexistent/librccl-missing.so", "cannot load librccl.so"), ("libm.so.6", "lacks ncclCommInitAll")):
        env = dict(os.environ, XRT_RCCL_LIB=libname)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        rc, _, msg = out.stdout.strip().partition(" ")
        assert int(rc) == xrt.abi.XRT_E_RCCL and needle in msg and len(msg) > len(needle), out.stdout
