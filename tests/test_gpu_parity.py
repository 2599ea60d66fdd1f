"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI, against the
CPU oracle on the same seeded inputs — bit-exact hit-triangle index, octree leaf id, u/v/d, world position,
packed RGBA8 and the fp32 colour vector (tolerance 0: the north star allows 1e-5 on fp32 RGB, we assert
equality of the bit patterns and report the max abs difference if that ever fails)."""
import ctypes as C
import math
import os
import threading

import numpy as np
import pytest

from util import hits_equal, random_rays, secondary_rays, triangle_soup

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RGB_TOL = 1e-5   # north-star tolerance for fp32 pixel RGB; we expect exactly 0


def assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf):
    if rgbf is not None:
        diff = float(np.abs(rgbf.astype(np.float64) - o_rgbf.astype(np.float64)).max()) if rgbf.size else 0.0
        assert diff <= RGB_TOL, "fp32 RGB max abs diff %g" % diff
        assert np.array_equal(rgbf.view(np.uint32), o_rgbf.view(np.uint32)), "fp32 RGB not bit-identical (max abs diff %g)" % diff
    assert np.array_equal(rgba, o_rgba), "%d RGBA8 pixels differ" % int((rgba != o_rgba).sum())


def test_library_loaded_and_device_visible(xrt):
    n = C.c_int(0)
    assert xrt.abi.lib().xrt_device_count(C.byref(n)) == 0 and n.value >= 1


SCENES = {
    "crate": lambda x: x.configs.config("C1"),
    "grid": lambda x: x.configs.crate_grid_scene(160, 90),
    "h64": lambda x: x.configs.heightfield_scene(160, 90, m=64),
    "h224": lambda x: x.configs.heightfield_scene(160, 90, m=224),
}


@pytest.mark.parametrize("name", list(SCENES))
def test_tree_raygen_and_intersection_parity(xrt, orc, name):
    spec = SCENES[name](xrt)
    scene, tracer = xrt.configs.build_product(spec)
    o = orc.OracleScene(spec)
    for mesh, mid in ((None, -1), (scene.meshes[0], 0)):
        n1, r1 = scene.tree(mesh)
        n2, r2 = o.tree(mid)
        assert n1.tobytes() == n2.tobytes() and np.array_equal(r1, r2)
    rays = tracer.GeneratePrimaryRays()          # RT:410-421 on the GPU
    o_rays = o.primary_rays()
    assert rays.tobytes() == o_rays.tobytes(), "primary rays differ"
    hits, st = scene.IntersectBatch(rays, stats=True)
    o_hits, o_st = o.intersect(rays, stats=True)
    assert hits_equal(o_hits, hits) == {}
    for k in ("rays_closest", "hits_closest", "scene_node_tests", "instance_visits", "mesh_aabb_tests", "mesh_queries", "node_tests", "leaf_refs", "tri_tests"):
        assert st[k] == o_st[k], (k, st[k], o_st[k])
    sec = secondary_rays(xrt, o_hits)
    if len(sec):
        h2, st2 = scene.IntersectBatch(sec, stats=True)
        o2, ost2 = o.intersect(sec, stats=True)
        assert hits_equal(o2, h2) == {}
        for k in ("node_tests", "leaf_refs", "tri_tests", "scene_node_tests", "instance_visits"):
            assert st2[k] == ost2[k], (k, st2[k], ost2[k])
    # MeshOctree.GetRayIntersection (MO:259) through Mesh.Init / xrt_mesh_intersect
    mesh = scene.meshes[0]
    mesh.Init()
    assert hits_equal(o.mesh_intersect(0, rays[::3]), mesh.Octree.IntersectBatch(rays[::3])) == {}


def test_golden_hit_vectors_on_gpu(xrt):
    for name, spec in (("c3", xrt.configs.crate_grid_scene(160, 90)), ("h224", xrt.configs.heightfield_scene(160, 90, m=224))):
        scene, tracer = xrt.configs.build_product(spec)
        rays = np.load(os.path.join(GOLDEN, name + "_rays.npy"))
        gold = np.load(os.path.join(GOLDEN, name + "_hits.npy"))
        assert hits_equal(gold, scene.IntersectBatch(rays)) == {}


def soup_spec(xrt, n, seed, threshold, size):
    s = xrt.configs.SceneSpec("soup")
    s.meshes.append((triangle_soup(n, seed, size), xrt.configs.material(0.5)))
    s.objects.append(([0], (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)))
    s.camera = xrt.configs.camera((0, 3, 3), (0, 0, 0))
    s.lights = [xrt.configs.spot((0, 5, 5))]
    s.mesh_threshold = threshold
    return s.with_size(64, 64)


@pytest.mark.parametrize("n,seed,threshold,size", [(60, 3, 2, 0.9), (300, 5, 4, 0.5), (2000, 7, 20, 0.15), (500, 9, 50, 1.5)])
def test_leaf_group_quirk_ties_and_ignore_on_soups(xrt, orc, n, seed, threshold, size):
    spec = soup_spec(xrt, n, seed, threshold, size)
    scene, tracer = xrt.configs.build_product(spec)
    o = orc.OracleScene(spec)
    rays = random_rays(xrt, 20000, seed + 100)
    o_hits = o.intersect(rays)
    assert hits_equal(o_hits, scene.IntersectBatch(rays)) == {}
    sec = secondary_rays(xrt, o_hits, seed)
    assert hits_equal(o.intersect(sec), scene.IntersectBatch(sec)) == {}
    rgba, rgbf = tracer_render(tracer, 2)
    o_rgba, o_rgbf, _ = orc.OracleScene(spec_with(spec, 2)).render(nthreads=8)
    assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)


def test_equal_distance_ties_inside_a_leaf(xrt, orc, monkeypatch):
    """MO:293-294: of two triangles of ONE leaf hit at exactly the same distance the earlier in the leaf's list wins (strict '<').  The device stores the references
    of a big leaf in runs of neighbouring triangles, not in list order (scene_host.cpp spatial_runs), and settles such a tie by the smaller triangle index -- a leaf's
    list is ascending in the index.  Here exact ties are the rule: hundreds of overlapping coplanar triangles with small integer coordinates (two stacked sheets, so that
    leaves differ), rays straight down and along the diagonals from integer heights -- every quantity of RE:42-75 is exact, and up to dozens of triangles of a leaf report
    the very same distance.  Per-lane kernel, packet kernel (XRT_PACKET=31), split walks; also with the list order kept (XRT_LEAF_ORDER=0: same answers)."""
    from util import coplanar_tie_scene
    spec, sets = coplanar_tie_scene(xrt)
    orc_scene = orc.OracleScene(spec)
    want = [orc_scene.intersect(r) for r in sets]
    tied = 0
    for env in ({}, {"XRT_PACKET": "31"}, {"XRT_PACKET": "31", "XRT_PK_SPLIT": "1", "XRT_PK_BUDGET": "0", "XRT_PK_BUDGET_ITEM": "0"}, {"XRT_LEAF_ORDER": "0"}, {"XRT_LEAF_ORDER": "0", "XRT_PACKET": "31"}):
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        scene, tracer = xrt.configs.build_product(spec)
        for r, w in zip(sets, want):
            assert hits_equal(w, scene.IntersectBatch(r)) == {}, env
        rgba, rgbf = tracer_render(tracer, 2)
        o_rgba, o_rgbf, _ = orc.OracleScene(spec_with(spec, 2)).render(nthreads=8)
        assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)
        for k in env:
            monkeypatch.delenv(k)
    # (the rays do meet ties: many of them hit, and a hit point lies in several triangles of its sheet)
    hit = want[0]["hit"] != 0
    assert hit.sum() > 200


def spec_with(spec, R):
    spec.max_reflections = R
    return spec


def tracer_render(tracer, R):
    tracer.MaxReflections = R
    return tracer.Render(want_float=True)


def test_single_ray_interface_and_edge_batches(xrt, orc):
    spec = xrt.configs.config("C1")
    scene, tracer = xrt.configs.build_product(spec)
    o = orc.OracleScene(spec)
    # ISpatialManager.GetRayIntersection single-ray signature (ISM:15)
    found, res = scene.GetRayIntersection(((0, 32, 64), tuple(xrt.xna.as_array(xrt.xna.normalize(xrt.xna.vec3(0, -24, -64))))))
    assert found and res["mesh"] is scene.meshes[0] and 0 <= res["triangle"] < 12
    found, res = scene.GetRayIntersection(((0, 32, 64), (0, 1, 0)))
    assert not found and res is None
    # empty batch, one ray, a batch that is not a multiple of the wave size
    assert len(scene.IntersectBatch(xrt.rays_array(np.zeros((0, 3)), np.zeros((0, 3))))) == 0
    rays = o.primary_rays()
    for n in (1, 63, 65, 257, 1000):
        assert hits_equal(o.intersect(rays[30000:30000 + n]), scene.IntersectBatch(rays[30000:30000 + n])) == {}
    # degenerate rays: zero / NaN direction, origin far away, near-parallel direction
    nan = float("nan")
    bad = xrt.rays_array([(0, 50, 0), (0, 50, 0), (1e30, 0, 0), (0, 50, 0)], [(0, 0, 0), (nan, -1, 0), (-1, 0, 0), (1e-7, -1, 1e-7)])
    assert hits_equal(o.intersect(bad), scene.IntersectBatch(bad)) == {}


def test_instances_rotated_scaled_two_meshes(xrt, orc):
    s = xrt.configs.SceneSpec("inst")
    s.meshes.append((xrt.fixtures.crate(3), xrt.configs.material(0.5, texture=xrt.fixtures.crate_texture())))
    s.meshes.append((triangle_soup(80, 11, 0.4), xrt.configs.material(0.2, interpolate_normals=True)))
    k = 0
    for ix in range(5):
        for iz in range(5):
            s.objects.append(([0] if (k % 3) else [0, 1], (-60.0 + 30.0 * ix, 2.0 * (k % 2), -60.0 + 30.0 * iz),
                              (0.1 * ix, 0.37 * iz, 0.05 * (ix + iz)), (1.0 + 0.1 * ix, 1.0, 0.8 + 0.1 * iz)))
            k += 1
    s.camera = xrt.configs.camera((0, 80, 160), (0, 0, 0))
    s.lights = [xrt.configs.spot((0, 100, 100)), xrt.configs.directional((0.3, 0.8, 0.5), (0.4, 0.5, 0.6), 0.7)]
    s.max_reflections = 3
    s = s.with_size(160, 90)
    scene, tracer = xrt.configs.build_product(s)
    o = orc.OracleScene(s)
    rays = o.primary_rays()
    o_hits = o.intersect(rays)
    assert hits_equal(o_hits, scene.IntersectBatch(rays)) == {}
    sec = secondary_rays(xrt, o_hits)
    assert hits_equal(o.intersect(sec), scene.IntersectBatch(sec)) == {}
    tracer.collect_stats = True
    rgba, rgbf = tracer.Render(want_float=True)
    o_rgba, o_rgbf, o_st = o.render(nthreads=8)
    assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)
    for k in ("rays_closest", "rays_shadow", "hits_closest", "hits_shadow", "node_tests", "leaf_refs", "tri_tests", "shaded_hits",
              "scene_node_tests", "instance_visits", "mesh_aabb_tests", "mesh_queries", "algorithmic_bytes"):
        assert tracer.last_stats[k] == o_st[k], (k, tracer.last_stats[k], o_st[k])


RENDERS = {
    "C1": lambda x: x.configs.config("C1"),                                          # the reference-runnable case, full size
    "C2_small": lambda x: x.configs.crate_scene(240, 136, max_reflections=2),
    "C2_flat": lambda x: x.configs.crate_scene(240, 136, max_reflections=2, textured=False),
    "C3_small": lambda x: x.configs.crate_grid_scene(192, 108),
    "H224_R8": lambda x: x.configs.heightfield_scene(128, 72, m=224, max_reflections=8),
    "C5_small_ms16": lambda x: x.configs.heightfield_scene(64, 36, m=224, multisampling=x.abi.MS_FIXED16),
    "odd_size": lambda x: x.configs.crate_grid_scene(101, 37),                        # edge tiles with dead lanes
}


@pytest.mark.parametrize("name", list(RENDERS))
def test_render_parity(xrt, orc, name):
    spec = RENDERS[name](xrt)
    scene, tracer = xrt.configs.build_product(spec)
    tracer.collect_stats = True
    rgba, rgbf = tracer.Render(want_float=True)
    o_rgba, o_rgbf, o_st = orc.OracleScene(spec).render(nthreads=8)
    assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)
    st = tracer.last_stats
    for k in ("rays_closest", "rays_shadow", "hits_closest", "hits_shadow", "node_tests", "leaf_refs", "tri_tests", "shaded_hits", "pixels", "algorithmic_bytes"):
        assert st[k] == o_st[k], (k, st[k], o_st[k])
    assert st["ms_intersect"] > 0 and st["intersect_launches"] >= 2


def test_live_ray_compaction_corners(xrt, orc):
    """The generation-0 ray arrays hold live rays only and are sized by the screen rectangle of the scene's root box (k_raygen compacts,
    DESIGN.md §4): a camera INSIDE the root box (every pixel is live, the rectangle is the image), one that looks away (no live ray at all),
    one that sees the scene in a corner of the image only (a rectangle of a few pixels), 1 and 16 sub-rays, blocking and two in flight."""
    import copy
    import torch
    base = xrt.configs.crate_grid_scene(96, 54)
    inside = copy.deepcopy(base); inside.camera = dict(base.camera); inside.camera["pos"] = (3.0, 30.0, 5.0); inside.camera["target"] = (60.0, 10.0, 40.0)
    away = copy.deepcopy(base); away.camera = dict(base.camera); away.camera["target"] = (0.0, 240.0, 520.0)
    corner = copy.deepcopy(xrt.configs.crate_scene(96, 64, max_reflections=2)); corner.camera = dict(corner.camera)
    corner.camera["pos"] = (0.0, 128.0, 256.0); corner.camera["target"] = (120.0, -60.0, 0.0)   # far away and looking past the crate: ~75 pixels of it, off centre
    inside16 = copy.deepcopy(inside); inside16.multisampling = xrt.abi.MS_FIXED16
    corner16 = copy.deepcopy(corner); corner16.multisampling = xrt.abi.MS_FIXED16
    seen = set()
    for spec in (inside, away, corner, inside16, corner16):
        o_rgba, _, o_st = orc.OracleScene(spec).render(want_float=False, nthreads=8)
        scene, tracer = xrt.configs.build_product(spec)
        rgba = tracer.Render()
        assert np.array_equal(rgba, o_rgba), spec.name
        st = tracer.last_stats
        assert st["rays_closest"] == o_st["rays_closest"] and st["rays_shadow"] == o_st["rays_shadow"]
        live = st["rays_traversed"] - st["rays_shadow"] - (st["rays_closest"] - spec.width * spec.height * (16 if spec.multisampling == xrt.abi.MS_FIXED16 else 1))
        seen.add("none" if live == 0 else ("all" if live == spec.width * spec.height * (16 if spec.multisampling == xrt.abi.MS_FIXED16 else 1) else "some"))
        outs = [torch.zeros(spec.width * spec.height, dtype=torch.int32, device="cuda") for _ in range(2)]
        frs = [tracer.PrepareDevice(o.data_ptr()) for o in outs]
        t0 = frs[0].begin(); t1 = frs[1].begin(); frs[0].end(t0); frs[1].end(t1)
        torch.cuda.synchronize()
        for o in outs:
            assert np.array_equal(o.cpu().numpy().view(np.uint32).reshape(o_rgba.shape), o_rgba), spec.name
    assert {"none", "some"} <= seen, seen   # (the corners this test is about really occurred)


def test_no_light_many_lights_and_deep_chains(xrt, orc):
    """Corners of the frame schedule: a scene without lights (no shadow segment at all), one with three lights (three shadow rays
    and hit / miss words per hit), and reflection chains of depth 12 on a mirror-like crate grid (fourteen traversal steps, the
    later ones nearly empty) -- plain frames (no counting pass: the launches time themselves) and counted ones."""
    import copy
    base = xrt.configs.crate_grid_scene(96, 54)
    none = copy.deepcopy(base); none.lights = []
    three = copy.deepcopy(base)
    three.lights = [xrt.configs.spot((0, 300, 300)), xrt.configs.spot((200, 150, -100)), xrt.configs.directional((0.3, -1.0, 0.2))]
    deep = copy.deepcopy(base); deep.max_reflections = 12
    for m in deep.meshes:
        m[1]["reflectiveness"] = 0.9
    deep16 = copy.deepcopy(xrt.configs.heightfield_scene(48, 27, m=64, multisampling=xrt.abi.MS_FIXED16)); deep16.max_reflections = 6
    for spec in (none, three, deep, deep16):
        o_rgba, o_rgbf, o_st = orc.OracleScene(spec).render(nthreads=8)
        for collect in (False, True):
            scene, tracer = xrt.configs.build_product(spec)
            tracer.collect_stats = collect
            rgba, rgbf = tracer.Render(want_float=True)
            assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)
            st = tracer.last_stats
            for k in ("rays_closest", "rays_shadow", "shaded_hits", "pixels"):
                assert st[k] == o_st[k], (k, st[k], o_st[k])
            assert st["intersect_launches"] >= 1 and st["ms_intersect"] > 0


def test_host_written_in_c(xrt, tmp_path):
    """A host in plain C99 (tests/c_host/c_host.c: include/xrt.h and libxrt.so, nothing else) loads a scene file, builds it and
    renders the frame the Python binding renders -- the FFI path of the reference's C# host (csharp/XrtNative.cs), without Python."""
    import subprocess
    import ctypes as C
    from util import build_c_host
    exe = build_c_host()
    for spec in (xrt.configs.config("C1"), xrt.configs.content_scene(96, 54)):
        scene, tracer = xrt.configs.build_product(spec)
        want = tracer.Render().copy()
        path = str(tmp_path / (spec.name + ".xrts"))
        scene.Save(path)
        cam, opts, lights = tracer._camera_abi(), tracer._opts_abi(), tracer._lights_abi()
        n = len(tracer.Lights)
        with open(str(tmp_path / "frame.bin"), "wb") as f:
            f.write(bytes(cam)); f.write(C.c_int32(n)); f.write(bytes(lights)[: n * C.sizeof(xrt.abi.xrt_light)]); f.write(bytes(opts))
        out = str(tmp_path / "out.rgba")
        r = subprocess.run([exe, path, str(tmp_path / "frame.bin"), out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert np.array_equal(np.fromfile(out, dtype=np.uint32), want), spec.name


def test_golden_frames_on_gpu(xrt):
    for fname, spec in (("c1_rgba.npy", xrt.configs.config("C1")), ("c3_96x54_rgba.npy", xrt.configs.crate_grid_scene(96, 54)),
                        ("h224_48x27_ms16_rgba.npy", xrt.configs.heightfield_scene(48, 27, m=224, multisampling=xrt.abi.MS_FIXED16))):
        scene, tracer = xrt.configs.build_product(spec)
        assert np.array_equal(tracer.Render(), np.load(os.path.join(GOLDEN, fname))), fname


def test_directional_light_and_transparent_blocker(xrt, orc):
    """DIR:23-30 (+Direction shading, -Direction shadow rays, Q18) and the alpha of a Transparent blocker
    as shadow attenuation (RT:489-492).  MaxReflections 0: refraction itself is a 'next' row."""
    s = xrt.configs.SceneSpec("lights")
    s.meshes.append((xrt.fixtures.heightfield(48), xrt.configs.material(0.3)))
    soup = triangle_soup(40, 21, 6.0)
    soup.v[:, :, 1] += np.float32(12.0)
    soup = xrt.fixtures.MeshData(soup.v, soup.n, soup.uv, soup.color)
    s.meshes.append((soup, xrt.configs.material(0.1, transparent=True, refraction_index=1.32)))
    s.objects.append(([0], (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)))
    s.objects.append(([1], (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (8.0, 1.0, 8.0)))
    s.camera = xrt.configs.camera((0, 60, 110), (0, 0, 0))
    s.lights = [xrt.configs.directional((0.2, 0.9, 0.1), (1.0, 0.9, 0.8), 0.9), xrt.configs.spot((0, 120, 160))]
    s.max_reflections = 0
    s = s.with_size(160, 90)
    scene, tracer = xrt.configs.build_product(s)
    rgba, rgbf = tracer.Render(want_float=True)
    o_rgba, o_rgbf, o_st = orc.OracleScene(s).render(nthreads=8)
    assert o_st["hits_shadow"] > 0
    assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)


def glass_scene(xrt, w, h, R):
    """The shape of the reference's default scene (Game1.cs:98-138): a 2x2 block of Transparent objects
    (alpha 100/255, refraction index 1.32, reflectiveness 0.7; contentproj:87-96) over an opaque floor."""
    s = xrt.configs.SceneSpec("glass")
    glass = xrt.fixtures.crate(2)
    glass.color[:, 3] = np.float32(100.0) / np.float32(255.0)
    glass.color[:, :3] = np.array([0.9, 0.95, 1.0], dtype=np.float32)
    s.meshes.append((glass, xrt.configs.material(0.7, transparent=True, refraction_index=1.32)))
    s.meshes.append((xrt.fixtures.heightfield(24), xrt.configs.material(0.3)))
    for x in range(2):
        for y in range(2):
            s.objects.append(([0], (-22.0 + 44.0 * x, 5.0, -22.0 + 44.0 * y), (0.0, 0.3 * x, 0.0), (1.0, 1.0, 1.0)))
    s.objects.append(([1], (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)))
    s.camera = xrt.configs.camera((0, 60, 120), (0, 5, 0))
    s.lights = [xrt.configs.spot((0, 90, 100))]
    s.max_reflections = R
    return s.with_size(w, h)


@pytest.mark.parametrize("R", [1, 3, 5])
def test_refraction_ray_tree(xrt, orc, R):
    """RT:586-702: Transparent materials turn CastRay into a binary tree (reflection + refraction per hit),
    with the double-precision Snell terms and the currentRefIndex bookkeeping."""
    spec = glass_scene(xrt, 96, 54, R)
    scene, tracer = xrt.configs.build_product(spec)
    tracer.collect_stats = True
    rgba, rgbf = tracer.Render(want_float=True)
    o_rgba, o_rgbf, o_st = orc.OracleScene(spec).render(nthreads=8)
    assert o_st["rays_closest"] > 96 * 54 + o_st["shaded_hits"] * 0.9, "fixture does not refract"
    assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)
    for k in ("rays_closest", "rays_shadow", "hits_closest", "hits_shadow", "shaded_hits", "tri_tests", "algorithmic_bytes"):
        assert tracer.last_stats[k] == o_st[k], (k, tracer.last_stats[k], o_st[k])


@pytest.mark.parametrize("quality", [0, 1, 2])
def test_adaptive_supersampling(xrt, orc, quality):
    """RenderInternalWithMultisampling (RT:128-168, 170-311): four rays per quadrant, corners deviating by more
    than 0.5 in colour length are subdivided down to MultisampleQuality, including the RT:305 slip that stores
    the lower-right recursion in the upper-right colour."""
    for spec in (xrt.configs.crate_grid_scene(96, 54), xrt.configs.heightfield_scene(80, 45, m=64), xrt.configs.crate_grid_scene(70, 33, n=2, grid=3)):
        spec.multisampling, spec.multisample_quality = xrt.abi.MS_ADAPTIVE, quality
        scene, tracer = xrt.configs.build_product(spec)
        tracer.collect_stats = True
        rgba, rgbf = tracer.Render(want_float=True)
        o_rgba, o_rgbf, o_st = orc.OracleScene(spec).render(nthreads=8)
        if quality > 0:
            assert o_st["rays_closest"] > 4 * spec.width * spec.height + o_st["shaded_hits"] * 0.5, "no quadrant was subdivided"
        assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)
        for k in ("rays_closest", "rays_shadow", "shaded_hits", "tri_tests", "pixels", "algorithmic_bytes"):
            assert tracer.last_stats[k] == o_st[k], (k, tracer.last_stats[k], o_st[k])


@pytest.mark.parametrize("quality", [0, 1, 3])
def test_adaptive_frames_in_flight(xrt, orc, monkeypatch, quality):
    """Adaptive supersampling without host round trips (RT:128-168 RenderAsync over RT:170-311): the size of every
    deeper quadrant level stays on the device, so an adaptive frame takes a ticket like a plain one -- two in flight, to device
    and to host memory, every frame the oracle's, the ray accounting the oracle's (no counting pass: the level sizes come back
    with the frame's counters).  A level that does not fit its optimistically sized buffers (XRT_ADAPTIVE_CAP forces that) makes
    the frame render again the careful way -- same pixels -- and XRT_ADAPTIVE_FAST=0 is the careful way from the start."""
    import torch
    for spec in (xrt.configs.crate_grid_scene(96, 54), xrt.configs.heightfield_scene(80, 45, m=64)):
        spec.multisampling, spec.multisample_quality = xrt.abi.MS_ADAPTIVE, quality
        o_rgba, _, o_st = orc.OracleScene(spec).render(nthreads=8, want_float=False)
        scene, tracer = xrt.configs.build_product(spec)
        got = tracer.Render().copy()            # no counting pass: the whole frame is enqueued at once
        assert np.array_equal(got, o_rgba)
        for k in ("rays_closest", "rays_shadow", "shaded_hits", "pixels"):
            assert tracer.last_stats[k] == o_st[k], (k, tracer.last_stats[k], o_st[k])
        n = spec.width * spec.height
        outs = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(2)]
        frs = [tracer.PrepareDevice(o.data_ptr()) for o in outs]
        for rep in range(3):
            t0 = frs[0].begin()
            t1 = frs[1].begin()                 # the second adaptive frame is enqueued while the first is in flight
            st0 = frs[0].end(t0)
            st1 = frs[1].end(t1)
            for o in outs:
                assert np.array_equal(o.cpu().numpy().view(np.uint32), o_rgba), rep
                o.zero_()
            assert st0["rays_closest"] == st1["rays_closest"] == o_st["rays_closest"]
        hosts = [np.zeros(n, dtype=np.uint32) for _ in range(2)]
        hfr = [tracer.PrepareHost(h) for h in hosts]
        t0, t1 = hfr[0].begin(), hfr[1].begin()
        hfr[0].end(t0); hfr[1].end(t1)
        for h in hosts:
            assert np.array_equal(h, o_rgba)
        if quality > 0:
            monkeypatch.setenv("XRT_ADAPTIVE_CAP", "8")   # eight quadrants per deeper level: overflows, the frame is rendered again
            _, tr_small = xrt.configs.build_product(spec)
            monkeypatch.delenv("XRT_ADAPTIVE_CAP")
            for _ in range(2):   # (the second frame takes the careful way from the start)
                assert np.array_equal(tr_small.Render(), o_rgba)
                assert tr_small.last_stats["rays_closest"] == o_st["rays_closest"]
            h2 = np.zeros(n, dtype=np.uint32)
            monkeypatch.setenv("XRT_ADAPTIVE_CAP", "8")
            _, tr_small2 = xrt.configs.build_product(spec)
            monkeypatch.delenv("XRT_ADAPTIVE_CAP")
            f2 = tr_small2.PrepareHost(h2)      # the redo of a host-output frame copies the right pixels again
            f2.end(f2.begin())
            assert np.array_equal(h2, o_rgba)
        monkeypatch.setenv("XRT_ADAPTIVE_FAST", "0")
        _, tr_careful = xrt.configs.build_product(spec)
        monkeypatch.delenv("XRT_ADAPTIVE_FAST")
        assert np.array_equal(tr_careful.Render(), o_rgba)


@pytest.mark.parametrize("address", [0, 1, 2])
@pytest.mark.parametrize("filtering", [0, 1])
def test_texture_address_modes_and_filters(xrt, orc, address, filtering):
    """MAT:71-232: Clamp / Wrap / Mirror addressing with point and bilinear filtering (IEEERemainder in double).
    UVs are scaled beyond [0,1] so the address modes matter."""
    spec = xrt.configs.crate_grid_scene(120, 68, n=2, grid=3)
    data, m = spec.meshes[0]
    data.uv[:] = data.uv * np.float32(2.5) - np.float32(0.75)
    spec.address_mode, spec.filtering = address, filtering
    scene, tracer = xrt.configs.build_product(spec)
    rgba, rgbf = tracer.Render(want_float=True)
    o_rgba, o_rgbf, _ = orc.OracleScene(spec).render(nthreads=8)
    assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)


def test_bilinear_filter_reads_the_premultiplied_copy(xrt, orc):
    """GetColorBilinear indexes Material.Texture.ColorData -- the Format32bppPArgb copy RayTracerTexture makes of the file
    (TEX:24-33, MAT:186-189) -- while the point sampler reads the Format32bppArgb lock (MAT:150).  For a texture with alpha the
    two arrays differ; the host passes both (xrt_material.tex_pargb).  How GDI+ rounds the premultiplication is the host's
    business (closed source): this test makes its own premultiplied copy and checks that each filter reads its own array."""
    rng = np.random.default_rng(3)
    argb = rng.integers(0, 2 ** 32, size=(64, 128), dtype=np.uint64).astype(np.uint32)
    a = (argb >> 24) & 0xff
    pre = lambda c: ((c * a + 127) // 255).astype(np.uint32)
    pargb = (a << 24) | (pre((argb >> 16) & 0xff) << 16) | (pre((argb >> 8) & 0xff) << 8) | pre(argb & 0xff)
    frames = {}
    for filtering in (xrt.abi.FILTER_POINT, xrt.abi.FILTER_BILINEAR):
        for with_p in (False, True):
            spec = xrt.configs.crate_scene(160, 90, max_reflections=1)
            spec.meshes[0] = (spec.meshes[0][0], xrt.configs.material(0.5, texture=argb, texture_pargb=pargb.astype(np.uint32) if with_p else None))
            spec.filtering = filtering
            _, tracer = xrt.configs.build_product(spec)
            rgba, rgbf = tracer.Render(want_float=True)
            o_rgba, o_rgbf, _ = orc.OracleScene(spec).render(nthreads=8)
            assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)
            frames[(filtering, with_p)] = rgba.copy()
    assert np.array_equal(frames[(xrt.abi.FILTER_POINT, False)], frames[(xrt.abi.FILTER_POINT, True)])          # the point sampler never reads it
    assert not np.array_equal(frames[(xrt.abi.FILTER_BILINEAR, False)], frames[(xrt.abi.FILTER_BILINEAR, True)])  # the bilinear filter does


def test_default_game_scene(xrt, orc):
    """G1: the reference's own default workload (Game1.cs:98-138: 4 Transparent spheres of Sphere.fbx, interpolated
    normals, MaxReflections 8 -> a 511-node ray tree per pixel), at reduced resolution against the oracle."""
    spec = xrt.configs.default_game_scene(96, 96, 8)
    scene, tracer = xrt.configs.build_product(spec)
    tracer.collect_stats = True
    rgba, rgbf = tracer.Render(want_float=True)
    o_rgba, o_rgbf, o_st = orc.OracleScene(spec).render(nthreads=8)
    assert o_st["rays_closest"] > 96 * 96 + 2 * o_st["shaded_hits"] * 0.4   # the ray tree branches
    assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)
    for k in ("rays_closest", "rays_shadow", "shaded_hits", "algorithmic_bytes"):
        assert tracer.last_stats[k] == o_st[k], (k, tracer.last_stats[k], o_st[k])


def test_default_game_scene_as_the_reference_renders_it(xrt, orc):
    """G1 at the size the reference's game really renders (Game1.cs:44-45: a 512x512 back buffer; Game1.cs:126: MaxReflections 8): the
    whole frame, RGBA8 and the fp32 colours, blocking and two frames in flight, plus the oracle's ray accounting."""
    import torch
    spec = xrt.configs.default_game_scene(512, 512, 8)
    scene, tracer = xrt.configs.build_product(spec)
    o_rgba, o_rgbf, o_st = orc.OracleScene(spec).render(nthreads=16)
    rgba, rgbf = tracer.Render(want_float=True)
    assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)
    st = tracer.last_stats
    for k in ("rays_closest", "rays_shadow", "shaded_hits", "pixels"):
        assert st[k] == o_st[k], (k, st[k], o_st[k])
    outs = [torch.zeros(512 * 512, dtype=torch.int32, device="cuda") for _ in range(2)]
    frs = [tracer.PrepareDevice(o.data_ptr()) for o in outs]
    for _ in range(3):
        t0 = frs[0].begin(); t1 = frs[1].begin(); frs[0].end(t0); frs[1].end(t1)
    torch.cuda.synchronize()
    for o in outs:
        assert np.array_equal(o.cpu().numpy().view(np.uint32).reshape(o_rgba.shape), o_rgba)


def test_work_buffers_under_guards(xrt, orc, monkeypatch):
    """XRT_GUARD=1: every device buffer of scenes created from now on ends in 4 KB of a known pattern that the end of every frame and
    batched query checks (xrt_api.cpp guards_check) -- a kernel that writes past an array it was given is XRT_E_INTERNAL here instead
    of a corrupted neighbour or a process abort.  The frame modes whose arrays are sized tightly: generation-0 arrays sized by the root
    box's screen rectangle (a soup seen from close by: the case that aborted once during round 3, a camera inside the root box, a scene
    in the image's corner), 16 sub-rays, adaptive levels, ray trees, many lights, tile shards, two frames in flight."""
    import copy
    import torch
    monkeypatch.setenv("XRT_GUARD", "1")
    try:
        grid = xrt.configs.crate_grid_scene(96, 54)
        inside = copy.deepcopy(grid); inside.camera = dict(grid.camera); inside.camera["pos"] = (3.0, 30.0, 5.0); inside.camera["target"] = (60.0, 10.0, 40.0)
        corner = copy.deepcopy(xrt.configs.crate_scene(96, 64, max_reflections=2)); corner.camera = dict(corner.camera)
        corner.camera["pos"] = (0.0, 128.0, 256.0); corner.camera["target"] = (120.0, -60.0, 0.0)
        grid16 = copy.deepcopy(grid); grid16.multisampling = xrt.abi.MS_FIXED16
        hf16 = xrt.configs.heightfield_scene(80, 45, m=64, multisampling=xrt.abi.MS_FIXED16)
        adaptive = xrt.configs.heightfield_scene(64, 36, m=48, multisampling=xrt.abi.MS_ADAPTIVE); adaptive.multisample_quality = 2
        glass = xrt.configs.default_game_scene(64, 64, 4)
        lights = copy.deepcopy(grid); lights.lights = [xrt.configs.spot((40.0 * np.cos(i), 200.0 + i, 40.0 * np.sin(i))) for i in range(5)]
        specs = [soup_spec(xrt, 60, 3, 2, 0.9), soup_spec(xrt, 2000, 7, 20, 0.15), inside, corner, grid16, hf16, adaptive, glass, lights]
        for spec in specs:
            spec = spec_with(spec, 2) if spec.name == "soup" else spec
            o_rgba, _, _ = orc.OracleScene(spec).render(nthreads=8, want_float=False)
            scene, tracer = xrt.configs.build_product(spec)
            assert np.array_equal(tracer.Render(), o_rgba), spec.name          # (a violated guard raises from xrt.abi.check)
            outs = [torch.zeros(spec.width * spec.height, dtype=torch.int32, device="cuda") for _ in range(2)]
            frs = [tracer.PrepareDevice(o.data_ptr()) for o in outs]
            t0 = frs[0].begin(); t1 = frs[1].begin(); frs[0].end(t0); frs[1].end(t1)
            torch.cuda.synchronize()
            for o in outs:
                assert np.array_equal(o.cpu().numpy().view(np.uint32).reshape(o_rgba.shape), o_rgba), spec.name
            rays = random_rays(xrt, 3000, 11)
            assert hits_equal(orc.OracleScene(spec).intersect(rays), scene.IntersectBatch(rays)) == {}
            if spec.multisampling != xrt.abi.MS_ADAPTIVE:   # tile shards of the same frame
                tx, ty, tpr = C.c_int32(), C.c_int32(), C.c_int32()
                xrt.abi.lib().xrt_shard_layout(spec.width, spec.height, 3, C.byref(tx), C.byref(ty), C.byref(tpr))
                part = torch.zeros(tpr.value * 512, dtype=torch.int32, device="cuda")
                for r in range(3):
                    tracer.RenderDevice(part.data_ptr(), shard_rank=r, shard_count=3)
    finally:
        monkeypatch.delenv("XRT_GUARD")
        s0, _ = xrt.configs.build_product(xrt.configs.crate_scene(32, 32, 0))   # (xrt_scene_create reads the switch: guards off again for the tests that follow)


def test_many_lights_shrink_the_chunk(xrt, orc, monkeypatch):
    """The reference iterates a List<ILight> of any length (RT:534-542).  Shadow rays, hits and words are 84 bytes per path and light in
    every frame context: a light count whose arrays exceed the byte budget (XRT_SHADOW_BYTES, here forced small) shrinks the chunk and the
    frame takes the multi-chunk path instead of failing with XRT_E_OOM -- 128 lights on a small image."""
    import copy
    spec = copy.deepcopy(xrt.configs.crate_grid_scene(160, 96))
    spec.lights = [xrt.configs.spot((300.0 * np.cos(0.37 * i), 150.0 + 2.0 * i, 300.0 * np.sin(0.37 * i))) for i in range(128)]
    for l in spec.lights:
        l["intensity"] = 0.02
    o_rgba, o_rgbf, o_st = orc.OracleScene(spec).render(nthreads=16)
    monkeypatch.setenv("XRT_SHADOW_BYTES", str(84 * 128 * 8192))   # room for 8192 paths' shadow rays: the 160x96 frame (15,360 paths + tile padding) needs three chunks
    scene, tracer = xrt.configs.build_product(spec)
    rgba, rgbf = tracer.Render(want_float=True)
    assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)
    st = tracer.last_stats
    for k in ("rays_closest", "rays_shadow", "shaded_hits"):
        assert st[k] == o_st[k], (k, st[k], o_st[k])
    assert st["rays_shadow"] == 128 * st["shaded_hits"]


def test_empty_and_tiny_scenes(xrt, orc):
    """Edge cases of the containers: no bodies at all, a body without meshes, a mesh without triangles, one triangle."""
    def spec_of(meshes, objects):
        s = xrt.configs.SceneSpec("edge")
        s.meshes, s.objects = meshes, objects
        s.camera = xrt.configs.camera((0, 3, 6), (0, 0, 0))
        s.lights = [xrt.configs.spot((0, 8, 8))]
        s.max_reflections = 1
        return s.with_size(70, 33)
    one = xrt.fixtures.MeshData(np.array([[(-2, 0, -2), (2, 0, -2), (0, 0, 2)]], dtype=np.float32), np.zeros((1, 3, 3), np.float32),
                                np.zeros((1, 3, 2), np.float32), np.ones((1, 4), np.float32))
    none = xrt.fixtures.MeshData(np.zeros((0, 3, 3), np.float32), np.zeros((0, 3, 3), np.float32), np.zeros((0, 3, 2), np.float32),
                                 np.zeros((0, 4), np.float32))
    ident = ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    cases = {
        "no bodies": spec_of([(one, xrt.configs.material(0.5))], []),
        "body without meshes": spec_of([(one, xrt.configs.material(0.5))], [([],) + ident, ([0],) + ident]),
        "empty mesh": spec_of([(none, xrt.configs.material(0.5)), (one, xrt.configs.material(0.2))], [([0, 1],) + ident]),
        "one triangle": spec_of([(one, xrt.configs.material(0.5))], [([0],) + ident]),
    }
    for name, spec in cases.items():
        scene, tracer = xrt.configs.build_product(spec)
        tracer.collect_stats = True
        rgba, rgbf = tracer.Render(want_float=True)
        o = orc.OracleScene(spec)
        o_rgba, o_rgbf, o_st = o.render()
        assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)
        for k in ("rays_closest", "rays_shadow", "scene_node_tests", "instance_visits", "mesh_aabb_tests", "node_tests", "tri_tests", "algorithmic_bytes"):
            assert tracer.last_stats[k] == o_st[k], (name, k, tracer.last_stats[k], o_st[k])
        rays = o.primary_rays()
        assert hits_equal(o.intersect(rays), scene.IntersectBatch(rays)) == {}, name


def test_error_conventions_on_gpu(xrt):
    spec = xrt.configs.config("C1")
    scene, tracer = xrt.configs.build_product(spec)
    tracer.AddressMode = 9
    with pytest.raises(ValueError):       # ArgumentException of MAT:85
        tracer.Render()
    tracer.AddressMode = xrt.abi.ADDRESS_WRAP
    # RenderAsync while busy -> InvalidOperationException (RT:62-63); completion callback fires (RT:435-436)
    done = threading.Event()
    tracer.RenderCompleted = lambda t: done.set()
    th = tracer.RenderAsync()
    try:
        with pytest.raises(RuntimeError):
            if tracer.IsBusy:
                tracer.RenderAsync()
            else:
                raise RuntimeError("finished already")
    finally:
        th.join()
    assert done.is_set() and not tracer.IsBusy and 0.99 <= tracer.Progress <= 1.0


def test_full_size_properties_c2(xrt):
    """C2 at its full 1920x1080 size: idempotence, ray accounting, EVERY row against the oracle (2.7 M rays, 0.2 s on the host's cores),
    and sharded == unsharded."""
    import torch
    spec = xrt.configs.config("C2")
    scene, tracer = xrt.configs.build_product(spec)
    a = tracer.Render().copy()
    st = dict(tracer.last_stats)
    b = tracer.Render().copy()
    assert np.array_equal(a, b)
    assert st["pixels"] == 1920 * 1080 and st["rays_closest"] >= st["pixels"] and st["rays_shadow"] == st["shaded_hits"]
    from oracle import oracle_py as orc
    o = orc.OracleScene(spec)
    o_rgba, _, o_st = o.render(nthreads=16, want_float=False)
    assert np.array_equal(a.reshape(1080, 1920), o_rgba.reshape(1080, 1920))
    assert st["rays_closest"] == o_st["rays_closest"] and st["rays_shadow"] == o_st["rays_shadow"]
    # image-tile shards rendered one after the other on this GPU, gathered and de-tiled = the whole frame
    world = 4
    tx, ty, tpr = C.c_int32(), C.c_int32(), C.c_int32()
    xrt.abi.lib().xrt_shard_layout(1920, 1080, world, C.byref(tx), C.byref(ty), C.byref(tpr))
    gathered = torch.zeros(world * tpr.value * 512, dtype=torch.int32, device="cuda")
    for r in range(world):
        part = gathered[r * tpr.value * 512:(r + 1) * tpr.value * 512]
        tracer.RenderDevice(part.data_ptr(), shard_rank=r, shard_count=world)
    out = torch.zeros(1920 * 1080, dtype=torch.int32, device="cuda")
    xrt.abi.check(xrt.abi.lib().xrt_detile_device(1920, 1080, world, C.c_void_p(gathered.data_ptr()), 0, C.c_void_p(out.data_ptr()), None))
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), a)
    # strided form (one gather carrying several frames): the same tiles as frame 1 of 3 per rank
    n = tpr.value * 512
    wide = torch.full((world * 3 * n,), -1, dtype=torch.int32, device="cuda")
    wide.view(world, 3, n)[:, 1, :] = gathered.view(world, n)
    out.zero_()
    xrt.dist.detile_device(wide, 1920, 1080, world, out, rank_stride=3 * n, offset=n)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), a)


def test_seam1_is_reentrant_across_host_threads(xrt, orc):
    """ISpatialManager.GetRayIntersection is called from N render threads at once (RT:105-113): four host threads query one
    scene with different ray sets, many times each, and every thread must get the hits of ITS rays (the staging buffers
    and the queue word are shared inside the scene).  Also while a begin/end ticket is open -- where only a call
    that wants stats answers BUSY."""
    import torch
    spec = xrt.configs.crate_grid_scene(160, 90)
    scene, tracer = xrt.configs.build_product(spec)
    o = orc.OracleScene(spec)
    prim = o.primary_rays()
    sets = [np.ascontiguousarray(prim[i::4][: 3000 + 257 * i]) for i in range(4)]
    want = [o.intersect(r) for r in sets]
    errors = []

    def worker(i):
        try:
            for _ in range(12):
                got = scene.IntersectBatch(sets[i])
                bad = hits_equal(want[i], got)
                if bad:
                    errors.append((i, bad))
                    return
        except Exception as e:   # noqa: BLE001
            errors.append((i, repr(e)))
    out = torch.zeros(spec.width * spec.height, dtype=torch.int32, device="cuda")
    fr = tracer.PrepareDevice(out.data_ptr())
    ticket = fr.begin()
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    with pytest.raises(RuntimeError):
        scene.IntersectBatch(sets[0], stats=True)        # shared counters: BUSY while the ticket is open
    for t in threads:
        t.join()
    fr.end(ticket)
    assert errors == []
    hits, st = scene.IntersectBatch(sets[0], stats=True)   # and works again afterwards
    assert hits_equal(want[0], hits) == {} and st["rays_closest"] == len(sets[0])


def test_render_argument_limits(xrt, orc):
    """rgb_f32_out in a supersampled mode is Color.ToVector3() of the final colour (k_resolve writes it; ADVICE r1 suspected
    stale memory).  The number of lights is not limited (the reference iterates a List<ILight>, RT:534-542; round 2 refused
    more than 32): 40 lights render like the oracle's 40 lights, and a light count whose shadow rays cannot fit one generation
    is XRT_E_UNSUPPORTED, not an overflow."""
    import copy
    spec = xrt.configs.config("C1", 0.25)
    spec.multisampling = xrt.abi.MS_FIXED16
    scene, tracer = xrt.configs.build_product(spec)
    rgba, rgbf = tracer.Render(want_float=True)
    unpacked = np.stack([(rgba >> s & 0xff).astype(np.float32) / np.float32(255.0) for s in (0, 8, 16)], axis=1)
    assert np.array_equal(rgbf, unpacked) and (rgba & 0xffffff).any()
    rgba = rgba.copy()
    lib, abi = xrt.abi.lib(), xrt.abi
    out = np.zeros(spec.width * spec.height, dtype=np.uint32)
    cam, lights, n, opts = tracer._camera_abi(), tracer._lights_abi(), len(tracer.Lights), tracer._opts_abi()
    huge = 200000
    many = (abi.xrt_light * huge)(*([lights[0]] * huge))
    rc = lib.xrt_render(scene.handle, C.byref(cam), many, huge, C.byref(opts), out.ctypes.data_as(C.POINTER(C.c_uint32)), None, None)
    assert rc == abi.XRT_E_UNSUPPORTED and b"lights" in lib.xrt_last_error()
    rc = lib.xrt_render(scene.handle, C.byref(cam), lights, n, C.byref(opts), out.ctypes.data_as(C.POINTER(C.c_uint32)), None, None)
    assert rc == 0 and np.array_equal(out, rgba)
    forty = copy.deepcopy(xrt.configs.crate_grid_scene(64, 36))
    forty.lights = [xrt.configs.spot((30.0 * math.cos(0.7 * i), 150 + 5 * i, 300.0 * math.sin(0.7 * i) + 40)) for i in range(39)] + [xrt.configs.directional((0.3, -1.0, 0.2))]
    o_rgba, o_rgbf, o_st = orc.OracleScene(forty).render(nthreads=8)
    _, tr40 = xrt.configs.build_product(forty)
    g_rgba, g_rgbf = tr40.Render(want_float=True)
    assert_frames_equal(g_rgba, g_rgbf, o_rgba, o_rgbf)
    assert tr40.last_stats["rays_shadow"] == o_st["rays_shadow"] == 40 * o_st["shaded_hits"]


def test_device_pointer_intersect(xrt, orc):
    import torch
    spec = xrt.configs.heightfield_scene(160, 90, m=64)
    scene, tracer = xrt.configs.build_product(spec)
    o = orc.OracleScene(spec)
    rays = o.primary_rays()
    d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1, 32).copy()).cuda()
    d_hits = torch.zeros((len(rays), 48), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    xrt.abi.check(xrt.abi.lib().xrt_scene_intersect_device(scene.handle, C.c_void_p(d_rays.data_ptr()), len(rays), C.c_void_p(d_hits.data_ptr()), None))
    torch.cuda.synchronize()
    hits = d_hits.cpu().numpy().reshape(-1).view(xrt.HIT_DTYPE)
    assert hits_equal(o.intersect(rays), hits) == {}


def test_pipelined_frames_begin_end(xrt):
    """xrt_render_device_begin / _end: two frames in flight give the frames and the accounting of the blocking call;
    a third begin, or any other render, answers BUSY until a ticket is closed."""
    import torch
    spec = xrt.configs.config("C4", 0.25)
    scene, tracer = xrt.configs.build_product(spec)
    want = tracer.Render().copy()
    st_want = dict(tracer.last_stats)
    n = spec.width * spec.height
    outs = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(2)]
    fr = [tracer.PrepareDevice(o.data_ptr()) for o in outs]
    t0 = fr[0].begin()
    t1 = fr[1].begin()
    assert {t0, t1} == {0, 1}
    with pytest.raises(RuntimeError):
        fr[0].begin()                        # both contexts busy
    with pytest.raises(RuntimeError):
        tracer.Render()                      # RT:62-63 while frames are open
    st0 = fr[0].end(t0)
    with pytest.raises(ValueError):
        fr[0].end(t0)                        # closed already
    t2 = fr[0].begin()                       # slot is free again, frame 1 still open
    st1 = fr[1].end(t1)
    st2 = fr[0].end(t2)
    with pytest.raises(ValueError):
        fr[0].end(7)
    torch.cuda.synchronize()
    for o in outs:
        assert np.array_equal(o.cpu().numpy().view(np.uint32), want)
    for st in (st0, st1, st2):
        for k in ("rays_closest", "rays_shadow", "shaded_hits", "pixels", "intersect_launches"):
            assert st[k] == st_want[k], (k, st[k], st_want[k])
        assert st["ms_total"] > 0 and st["ms_intersect"] > 0
    assert np.array_equal(tracer.Render(), want)   # the blocking call works again


def test_multi_chunk_frames(xrt, monkeypatch):
    """Frames larger than one chunk of paths (33.5 M by default) are rendered chunk by chunk; XRT_CHUNK_PATHS forces the
    same path at a testable size: same frame and same accounting as the single-chunk render, for 1 and 16 sub-rays."""
    for ms in (False, True):
        spec = xrt.configs.config("C3", 0.2)
        spec.multisampling = xrt.abi.MS_FIXED16 if ms else xrt.abi.MS_OFF   # (configs.build_product reads it)
        scene, tracer = xrt.configs.build_product(spec)
        want = tracer.Render().copy()
        st_want = dict(tracer.last_stats)
        monkeypatch.setenv("XRT_CHUNK_PATHS", "16384")
        scene2, tracer2 = xrt.configs.build_product(spec)
        monkeypatch.delenv("XRT_CHUNK_PATHS")
        got = tracer2.Render()
        assert np.array_equal(got, want)
        for k in ("rays_closest", "rays_shadow", "shaded_hits", "pixels"):
            assert tracer2.last_stats[k] == st_want[k], (k, tracer2.last_stats[k], st_want[k])
        assert tracer2.last_stats["intersect_launches"] > st_want["intersect_launches"]


def test_launch_timing_device_clock_vs_events(xrt, monkeypatch):
    """Plain single-chunk frames time their traversal launches on the device clock (device_util.h stamp_begin / stamp_end, folded by
    k_compose's epilogue) instead of carrying two events per launch; XRT_LAUNCH_EVENTS=1 is the old way.  Same frame, same launch
    count -- one launch per step where the closest-hit and the shadow rays of a step are both packets (XRT_PK_MERGE=1, the default), two
    with XRT_PK_MERGE=0 --, and the two clocks agree loosely (bench.py's roofline fractions divide by these durations: a stamp fold that
    dropped a launch's row would show here; the bound is wide enough for a shared or throttled box)."""
    spec = xrt.configs.config("C3", 0.5)
    scene, tracer = xrt.configs.build_product(spec)
    monkeypatch.setenv("XRT_LAUNCH_EVENTS", "1")
    scene_e, tracer_e = xrt.configs.build_product(spec)
    monkeypatch.delenv("XRT_LAUNCH_EVENTS")
    monkeypatch.setenv("XRT_PK_MERGE", "0")
    scene_u, tracer_u = xrt.configs.build_product(spec)
    monkeypatch.delenv("XRT_PK_MERGE")
    R = spec.max_reflections
    ms_c, ms_e = [], []
    for i in range(10):
        a = tracer.Render().copy()
        st_c = dict(tracer.last_stats)
        b = tracer_e.Render().copy()
        st_e = dict(tracer_e.last_stats)
        assert np.array_equal(a, b)
        # a two-level scene: every generation is packets -- R + 2 steps, each ONE launch (the first has no shadow rays, the last no closest-hit rays)
        assert st_c["intersect_launches"] == st_e["intersect_launches"] == R + 2
        if i >= 2:
            ms_c.append(st_c["ms_intersect"]); ms_e.append(st_e["ms_intersect"])
            assert 0 < st_c["ms_intersect_longest"] <= st_c["ms_intersect"] < st_c["ms_total"]
            assert 0 < st_e["ms_intersect_longest"] <= st_e["ms_intersect"]
    c, e = min(ms_c), min(ms_e)
    assert 0.5 * e <= c <= 1.5 * e + 0.05, (ms_c, ms_e)
    a = tracer_u.Render().copy()
    assert np.array_equal(a, b)
    assert tracer_u.last_stats["intersect_launches"] == 2 * (R + 2) - 2   # the steps 1 .. R carry two ray populations: two launches each when not merged


def test_launch_timing_mixed_stamps_and_events(xrt, monkeypatch):
    """A frame with more traversal launches than stamp rows (XRT_STAMP_ROWS lowers the 256 of a build to 3) times the rest with
    events: same frames, same launch count, a sum of the same order -- for a plain frame, a ray-tree frame (whose chunks fold
    their rows in k_compose_tree) and an adaptive frame (several passes)."""
    import copy
    specs = [xrt.configs.config("C3", 0.25), xrt.configs.config("G1", 0.25)]
    ad = copy.deepcopy(xrt.configs.config("C3", 0.25))
    ad.multisampling, ad.multisample_quality = xrt.abi.MS_ADAPTIVE, 2
    specs.append(ad)
    for spec in specs:
        scene, tracer = xrt.configs.build_product(spec)
        monkeypatch.setenv("XRT_STAMP_ROWS", "3")
        scene_m, tracer_m = xrt.configs.build_product(spec)
        monkeypatch.delenv("XRT_STAMP_ROWS")
        for i in range(3):
            a = tracer.Render().copy(); sa = dict(tracer.last_stats)
            b = tracer_m.Render().copy(); sb = dict(tracer_m.last_stats)
        assert np.array_equal(a, b)
        assert sa["intersect_launches"] == sb["intersect_launches"] > 3
        assert 0 < sa["ms_intersect"] <= sa["ms_total"] and 0 < sb["ms_intersect"]   # (sanity bounds only, no timing ratios)


def test_grid_hints_follow_a_camera_that_looks_away(xrt):
    """The launches of a frame's later generations are sized for four times the generation sizes of the last finished frame
    (xrt_api.cpp genRays / genShade).  A camera that looks away from the scene (every generation empty) and back again makes those
    hints as wrong as they can be; sizing never touches a result: the frames equal the ones rendered without hints."""
    import torch
    spec = xrt.configs.config("C3", 0.25)
    scene, tracer = xrt.configs.build_product(spec)
    full = tracer.CurrentCamera
    away = xrt.api.Camera((0, 200, 400), (0, 800, 1200), (0.0, 1.0, 0.0), 0.7853981852531433, xrt.xna.aspect_ratio(spec.width, spec.height), 1.0, 1000.0)
    want = {}
    os.environ["XRT_GRID_HINTS"] = "0"
    try:
        scene0, tracer0 = xrt.configs.build_product(spec)
    finally:
        del os.environ["XRT_GRID_HINTS"]
    for name, cam in (("full", full), ("away", away)):
        tracer0.CurrentCamera = cam
        want[name] = tracer0.Render().copy()
    assert want["away"].max() == want["away"].min() and not np.array_equal(want["full"], want["away"])
    seq = ["full", "full", "away", "away", "full", "away", "full", "full"]
    for name in seq:   # blocking frames
        tracer.CurrentCamera = full if name == "full" else away
        assert np.array_equal(tracer.Render(), want[name]), name
    n = spec.width * spec.height
    outs = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(2)]
    frs = {}
    for name, cam in (("full", full), ("away", away)):
        tracer.CurrentCamera = cam
        frs[name] = [tracer.PrepareDevice(o.data_ptr()) for o in outs]
    open_t = None
    for i, name in enumerate(seq * 3):   # two frames in flight
        t = frs[name][i % 2].begin()
        if open_t is not None:
            pn, pi, pt = open_t
            frs[pn][pi % 2].end(pt)
            assert np.array_equal(outs[pi % 2].cpu().numpy().view(np.uint32), want[pn]), (pi, pn)
        open_t = (name, i, t)
    pn, pi, pt = open_t
    frs[pn][pi % 2].end(pt)
    assert np.array_equal(outs[pi % 2].cpu().numpy().view(np.uint32), want[pn])


def test_overlapping_frames_on_two_streams(xrt, monkeypatch):
    """Frames that run long enough get one stream per frame context and overlap on the GPU (XRT_OVERLAP_MS=0 forces it
    for a test-sized frame): two different cameras in flight at once give the frames of the blocking renders, over
    many alternations, with the long-ray list and its cost feedback active (deep meshes)."""
    import torch
    monkeypatch.setenv("XRT_OVERLAP_MS", "0")
    monkeypatch.setenv("XRT_HEAVY", "0.2")
    spec = xrt.configs.config("C3", 0.25)
    scene, tracer = xrt.configs.build_product(spec)
    monkeypatch.delenv("XRT_OVERLAP_MS")
    monkeypatch.delenv("XRT_HEAVY")
    n = spec.width * spec.height
    cams = [tracer.CurrentCamera, xrt.api.Camera((40, 60, 90), (0, 8, 0), (0.0, 1.0, 0.0), 0.7853981852531433,
                                                 xrt.xna.aspect_ratio(spec.width, spec.height), 1.0, 1000.0)]
    want, outs, frs = [], [], []
    for c in cams:
        tracer.CurrentCamera = c
        want.append(tracer.Render().copy())
        outs.append(torch.zeros(n, dtype=torch.int32, device="cuda"))
        frs.append(tracer.PrepareDevice(outs[-1].data_ptr()))     # stream None: the library picks the context's stream
    assert not np.array_equal(want[0], want[1])
    open_t = None
    for i in range(24):
        t = frs[i % 2].begin()
        if open_t is not None:
            frs[(i - 1) % 2].end(open_t)
            got = outs[(i - 1) % 2].cpu().numpy().view(np.uint32)
            assert np.array_equal(got, want[(i - 1) % 2]), i
            outs[(i - 1) % 2].zero_()
            torch.cuda.current_stream().synchronize()   # (only torch's stream: the other frame stays in flight)
        open_t = t
    frs[23 % 2].end(open_t)
    assert np.array_equal(outs[23 % 2].cpu().numpy().view(np.uint32), want[23 % 2])


@pytest.mark.parametrize("env", [{"XRT_TUNE": "8,4,4,0"}, {"XRT_TUNE": "40,64,64,64"}, {"XRT_HEAVY": "0.02"}, {"XRT_HEAVY": "0.02", "XRT_NO_FEEDBACK": "1"},
                                 {"XRT_NO_RECT_CULL": "1"}, {"XRT_NO_SINGLE": "1"}, {"XRT_LONG_FRAC": "40,60"},
                                 {"XRT_NODE_CULL": "0"}, {"XRT_NODE_CULL": "2"}, {"XRT_AE": "0"}, {"XRT_AE": "0", "XRT_NODE_CULL": "2", "XRT_PACKET": "31"}, {"XRT_SPLIT": "2"},
                                 {"XRT_PK_PREFETCH": "1", "XRT_PACKET": "31"}, {"XRT_PK_PREFETCH": "0", "XRT_PACKET": "31"}, {"XRT_LEAF_ORDER": "0", "XRT_PACKET": "31"}, {"XRT_LEAF_ORDER": "0"}, {"XRT_LEVEL_MAP": "0"}])
def test_scheduling_switches_never_change_results(xrt, monkeypatch, env):
    """Knobs of the persistent loop, the cooperative leaf step, the long-ray list (geometric estimate and cost
    feedback), the screen-rectangle cull, the single-object mode, the normal boxes of nodes and meshes (off / rays leaving a surface / every
    ray), answering at emission, frame bands, the prefetches of small packet launches, the storage order of big leaves and the compact level records are scheduling / work-avoidance / layout only: hits of a
    secondary-ray population and three consecutive frames (the feedback needs history) are those of the default build."""
    spec = xrt.configs.heightfield_scene(320, 180, m=96)
    scene, tracer = xrt.configs.build_product(spec)
    rays = tracer.GeneratePrimaryRays()
    hits = scene.IntersectBatch(rays)
    sec = secondary_rays(xrt, hits, seed=3)
    want_hits, want_sec = hits, scene.IntersectBatch(sec)
    want = tracer.Render().copy()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    scene2, tracer2 = xrt.configs.build_product(spec)
    assert hits_equal(want_hits, scene2.IntersectBatch(rays)) == {}
    assert hits_equal(want_sec, scene2.IntersectBatch(sec)) == {}
    for _ in range(3):
        assert np.array_equal(tracer2.Render(), want)


# 32 committed seeds in the default run; XRT_FUZZ_EXTRA=n adds n more scenes
def fuzz_spec(xrt, seed, rng=None):
    """Random bodies (1-7) of random triangle soups (one to three meshes each, 20-600 triangles, leaf thresholds 2-40) under
    random rotations, non-uniform scales and translations."""
    rng = rng if rng is not None else np.random.default_rng(seed)
    s = xrt.configs.SceneSpec("fuzz%d" % seed)
    n_mesh = int(rng.integers(1, 4))
    for m in range(n_mesh):
        soup = triangle_soup(int(rng.integers(20, 600)), seed + m, float(rng.uniform(0.1, 1.2)))
        s.meshes.append((soup, xrt.configs.material(float(rng.uniform(0, 1)), interpolate_normals=bool(rng.integers(0, 2)))))
    for b in range(int(rng.integers(1, 8))):
        ids = sorted(set(int(i) for i in rng.integers(0, n_mesh, size=int(rng.integers(1, 4)))))
        s.objects.append((ids, tuple(float(x) for x in rng.uniform(-4, 4, 3)), tuple(float(x) for x in rng.uniform(-3.2, 3.2, 3)),
                          tuple(float(x) for x in rng.uniform(1.0, 4.0, 3))))
    s.mesh_threshold = int(rng.integers(2, 41))
    s.scene_threshold = int(rng.integers(1, 5))
    s.camera = xrt.configs.camera((0, 9, 17), (0, 0, 0))
    s.lights = [xrt.configs.spot((3, 20, 12)), xrt.configs.directional((0.2, 0.9, 0.3), (0.5, 0.5, 0.4), 0.6)]
    s.max_reflections = 2
    return s.with_size(96, 54)


FUZZ_SEEDS = [101, 202, 303, 404, 505, 606, 707, 808, 909, 1111, 1212, 1313, 1414, 1515, 1616, 1717, 1818, 1919, 2020, 2121, 2222, 2323, 2424, 2525,
              2626, 2727, 2828, 2929, 3030, 3131, 3232, 3333] + list(range(5000, 5000 + int(os.environ.get("XRT_FUZZ_EXTRA", "0"))))


@pytest.mark.parametrize("seed", FUZZ_SEEDS)
def test_random_scenes_against_the_oracle(xrt, orc, seed):
    """Fuzz: random bodies (1-7) of random triangle soups (one to three meshes each, 20-600 triangles, leaf thresholds
    2-40) under random rotations, non-uniform scales and translations; rays from outside, from inside the boxes, exactly
    axis-parallel, with zero and non-finite components, and the secondary rays of the hits.  Hit triangle, leaf, u/v/d
    and world position bit for bit; one small frame with every shading term."""
    rng = np.random.default_rng(seed)
    s = fuzz_spec(xrt, seed, rng)
    try:
        scene, tracer = xrt.configs.build_product(s)
    except xrt.abi.XrtError as e:
        # bodies that overlap more than the scene threshold allows can never be separated: OSM:101-113 has no depth limit,
        # the reference would recurse forever and the library refuses the scene
        assert "would not terminate" in str(e)
        pytest.skip("the reference's BuildTree would not terminate on this scene")
    o = orc.OracleScene(s)
    # mesh ids are handles: the host mirror numbers meshes by first use in the bodies (OctreeSpatialManager.Build), the
    # oracle wrapper by their position in the spec
    first_use = []
    for ids, _, _, _ in s.objects:
        first_use += [i for i in ids if i not in first_use]
    to_spec = np.array(first_use + [-1], dtype=np.int32)

    def product_hits(r):
        h = scene.IntersectBatch(r).copy()
        h["mesh"] = np.where(h["hit"] != 0, to_spec[np.clip(h["mesh"], 0, len(first_use))], h["mesh"])
        return h
    n = 6000
    O = rng.uniform(-9, 9, size=(n, 3)).astype(np.float32)
    D = rng.normal(size=(n, 3)).astype(np.float32)
    D /= np.linalg.norm(D, axis=1, keepdims=True).astype(np.float32)
    O[: n // 3] *= 0.25                                     # origins inside the bodies' boxes
    ax = rng.integers(0, 3, size=n // 6)
    D[n // 3: n // 3 + n // 6] = 0.0                        # exactly axis-parallel
    D[np.arange(n // 3, n // 3 + n // 6), ax] = rng.choice([-1.0, 1.0], size=n // 6)
    D[-8:-6] = 0.0                                          # zero direction
    D[-6, 0] = np.nan; D[-5, 1] = np.inf; O[-4, 2] = np.nan; O[-3, 0] = -np.inf; D[-2] = (1e-7, -1.0, 1e-7); O[-1] = (1e30, 0, 0)
    rays = xrt.rays_array(O, D)
    o_hits = o.intersect(rays)
    assert hits_equal(o_hits, product_hits(rays)) == {}
    inv = {v: k for k, v in enumerate(first_use)}
    sec = secondary_rays(xrt, o_hits, seed)                 # ignore (mesh, tri) in the oracle's numbering
    sec_p = sec.copy()
    sec_p["ignore_mesh"] = np.array([inv.get(int(m), -1) for m in sec["ignore_mesh"]], dtype=np.int32)
    assert hits_equal(o.intersect(sec), product_hits(sec_p)) == {}
    prim = o.primary_rays()
    assert hits_equal(o.intersect(prim), product_hits(prim)) == {}
    rgba, rgbf = tracer.Render(want_float=True)
    o_rgba, o_rgbf, _ = o.render(nthreads=8)
    assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)
    # the same bodies in the other frame modes: fixed 16 sub-rays, adaptive quadrants, a ray tree through glass
    for mode in ("ms16", "adaptive", "glass"):
        s2 = xrt.configs.SceneSpec("fuzz%d_%s" % (seed, mode))
        s2.meshes = [(d, dict(m)) for d, m in s.meshes]
        s2.objects, s2.camera, s2.lights = s.objects, s.camera, s.lights
        s2.mesh_threshold, s2.scene_threshold = s.mesh_threshold, s.scene_threshold
        s2.max_reflections = int(rng.integers(1, 4))
        if mode == "ms16":
            s2.multisampling = xrt.abi.MS_FIXED16
        elif mode == "adaptive":
            s2.multisampling, s2.multisample_quality = xrt.abi.MS_ADAPTIVE, int(rng.integers(0, 3))
        else:
            for _, m in s2.meshes[::2]:
                m["transparent"], m["refraction_index"] = True, float(rng.uniform(1.1, 1.8))
        s2 = s2.with_size(64, 36)
        _, tracer2 = xrt.configs.build_product(s2)
        rgba, rgbf = tracer2.Render(want_float=True)
        o_rgba, o_rgbf, _ = orc.OracleScene(s2).render(nthreads=8)
        if mode == "glass":
            assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)
        else:
            assert np.array_equal(rgba, o_rgba), (seed, mode)   # (no fp32 colour vector in the supersampled modes)


def test_object_precull_far_origins_and_grazing_rays(xrt, orc):
    """The world-space object pre-cull (traverse.h advance_scene, margin m(|o|) of DESIGN.md §3) on the GPU: rays from 1e3,
    1e4 and 1e5 units away aimed around the corners of two small bodies near the origin (the regime where the reference's
    own ray transform, OSM:358-364, loses precision like eps * |o|^2), and rays grazing the un-enlarged world hull of
    bodies scaled 1e-3 .. 1e3 with condition numbers up to ~350 within 1e-7 .. 1e-3 of its size."""
    from util import far_origin_scene, far_origin_rays, precull_adversarial_scene, grazing_rays
    s = far_origin_scene(xrt)
    scene, _ = xrt.configs.build_product(s)
    o = orc.OracleScene(s)
    for radius, seed in ((1e3, 1), (1e4, 2), (1e5, 3)):
        rays = far_origin_rays(xrt, s, radius, 2000, seed)
        ho = o.intersect(rays)
        assert hits_equal(ho, scene.IntersectBatch(rays)) == {}, radius
    s = precull_adversarial_scene(xrt)
    scene, _ = xrt.configs.build_product(s)
    o = orc.OracleScene(s)
    for seed in (11, 12):
        rays = grazing_rays(xrt, s, 6000, seed)
        ho = o.intersect(rays)
        assert 0.02 < (ho["hit"] != 0).mean() < 0.98
        assert hits_equal(ho, scene.IntersectBatch(rays)) == {}, seed


_ORACLE_FRAMES = {}


def oracle_whole_frame(xrt, name):
    """The oracle's render of EVERY row of a BASELINE configuration at its full size, once per test session (16 host threads:
    C5 -- 54.7 M rays -- takes about 4 s, C4 about as long, C3 and the 1-sample C5 frame about a second)."""
    if name not in _ORACLE_FRAMES:
        from oracle import oracle_py as orc
        spec = xrt.configs.config(name)
        rgba, _, st = orc.OracleScene(spec).render(nthreads=16, rows=(0, spec.height), want_float=False)
        _ORACLE_FRAMES[name] = (rgba.reshape(spec.height, spec.width), st)
    return _ORACLE_FRAMES[name]


@pytest.mark.parametrize("name", ["C3", "C5_1spp", "C5"])
def test_full_size_pipelined_frames_equal_the_oracle_whole_frame(xrt, name):
    """C3, the C5 scene at one sample and C5 as specified (16 sub-rays) at their full 1080p size, rendered the way bench.py times
    them: eight frames, two in flight -- they run long enough to overlap on two streams, with the long-ray list and its cost
    feedback building up history (per-lane launches) and the wave-packet launches of two frames side by side (C5).  EVERY frame
    equals the oracle's render of ALL 1080 rows (four to sixteen centre rows were compared; the horizon rows --
    grazing rays, the long-ray list, tight-leaf-box margins at large distances -- never were), and so does the blocking render."""
    import torch
    spec = xrt.configs.config(name)
    o_rgba, o_st = oracle_whole_frame(xrt, name)
    scene, tracer = xrt.configs.build_product(spec)
    want = tracer.Render().copy()
    W, H = spec.width, spec.height
    bad = want.reshape(H, W) != o_rgba
    assert not bad.any(), "%d pixels differ from the oracle, first rows %s" % (int(bad.sum()), np.unique(np.nonzero(bad)[0])[:8])
    for k in ("rays_closest", "rays_shadow", "shaded_hits", "pixels"):
        assert tracer.last_stats[k] == o_st[k], (k, tracer.last_stats[k], o_st[k])
    assert 0 < tracer.last_stats["rays_traversed"] <= o_st["rays_closest"] + o_st["rays_shadow"]
    n = W * H
    outs = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(2)]
    frs = [tracer.PrepareDevice(o.data_ptr()) for o in outs]
    open_t = None
    for i in range(8):
        t = frs[i % 2].begin()
        if open_t is not None:
            frs[(i - 1) % 2].end(open_t)
            assert np.array_equal(outs[(i - 1) % 2].cpu().numpy().view(np.uint32).reshape(H, W), o_rgba), i
        open_t = t
    frs[7 % 2].end(open_t)
    assert np.array_equal(outs[7 % 2].cpu().numpy().view(np.uint32).reshape(H, W), o_rgba)


@pytest.mark.parametrize("name", ["C4", "C5"])
def test_full_size_c4_c5_as_specified(xrt, name):
    """BASELINE configs[3] and [4] at their full sizes: C4 = the 64-instance grid at 3840x2160, C5 = the 999,698-triangle
    heightfield at 1920x1080 with 16 sub-rays per pixel (XRT_MS_FIXED16), depth 3: the WHOLE frame bit-equal to the oracle's
    (all 2160 / 1080 rows), the oracle's ray accounting, idempotence, and the 4- and 8-way image-tile shards rendered in turn,
    gathered and de-tiled == the unsharded frame, their ray counts adding up to the whole frame's."""
    import torch
    spec = xrt.configs.config(name)
    assert (spec.width, spec.height) == ((3840, 2160) if name == "C4" else (1920, 1080))
    assert name != "C5" or (spec.multisampling == xrt.abi.MS_FIXED16 and sum(m[0].ntri for m in spec.meshes) == 999698)
    scene, tracer = xrt.configs.build_product(spec)
    W, H = spec.width, spec.height
    whole = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    st = dict(tracer.RenderDevice(whole.data_ptr()))
    want = whole.cpu().numpy().view(np.uint32).copy()
    samples = 16 if name == "C5" else 1
    assert st["pixels"] == W * H and st["rays_closest"] >= W * H * samples and st["rays_shadow"] == st["shaded_hits"]
    o_rgba, o_st = oracle_whole_frame(xrt, name)
    bad = want.reshape(H, W) != o_rgba
    assert not bad.any(), "%d pixels differ from the oracle, first rows %s" % (int(bad.sum()), np.unique(np.nonzero(bad)[0])[:8])
    assert (want & 0xffffff).any()
    for k in ("rays_closest", "rays_shadow", "shaded_hits", "pixels"):
        assert st[k] == o_st[k], (k, st[k], o_st[k])
    if name == "C5":   # the counting pass also says how many mesh queries the mesh's normal box answers: most shadow rays and reflections of this terrain
        tracer.collect_stats = True
        st_c = dict(tracer.RenderDevice(whole.data_ptr()))
        tracer.collect_stats = False
        assert st_c["mesh_queries"] == o_st["mesh_queries"] and 0.4 * st_c["mesh_queries"] < st_c["mesh_queries_facing_away"] < st_c["mesh_queries"]
        assert np.array_equal(whole.cpu().numpy().view(np.uint32), want)
    again = torch.zeros_like(whole)
    tracer.RenderDevice(again.data_ptr())
    assert torch.equal(again, whole)
    for world in (4, 8):
        tx, ty, tpr = C.c_int32(), C.c_int32(), C.c_int32()
        xrt.abi.lib().xrt_shard_layout(W, H, world, C.byref(tx), C.byref(ty), C.byref(tpr))
        n = tpr.value * 512
        gathered = torch.zeros(world * n, dtype=torch.int32, device="cuda")
        acc = dict(rays_closest=0, rays_shadow=0, shaded_hits=0, pixels=0)
        for r in range(world):
            s_r = tracer.RenderDevice(gathered[r * n:(r + 1) * n].data_ptr(), shard_rank=r, shard_count=world)
            for k in acc:
                acc[k] += s_r[k]
        for k in acc:
            assert acc[k] == st[k], (world, k, acc[k], st[k])
        out = torch.zeros(W * H, dtype=torch.int32, device="cuda")
        xrt.abi.check(xrt.abi.lib().xrt_detile_device(W, H, world, C.c_void_p(gathered.data_ptr()), 0, C.c_void_p(out.data_ptr()), None))
        torch.cuda.synchronize()
        assert torch.equal(out, whole), world
    # Cost-aware tile assignment (xrt.h xrt_scene_set_tile_table): the whole frame's tile costs -> a longest-first table for 4 and 8 ranks; the
    # shards rendered under it, gathered and de-tiled through the table == the unsharded frame, ray counts adding up; the table is better
    # balanced than round-robin BY THE MEASURED COSTS (what tools/shard_predict.py then times); then back to round-robin.
    tracer.TileCosts(reset=True)
    tracer.RenderDevice(again.data_ptr())
    cost = tracer.TileCosts(reset=True)
    assert cost.size == ((W + 63) // 64) * ((H + 7) // 8) and (cost > 0).sum() > 0.3 * cost.size and np.isfinite(cost).all()
    for world in (4, 8):
        tprb, table = xrt.dist.balanced_table(W, H, world, cost)
        rows = table.reshape(world, tprb)
        assert np.array_equal(np.sort(table[table >= 0]), np.arange(cost.size))
        loads = np.array([cost[r[r >= 0]].sum() for r in rows])
        _, rr = xrt.dist.round_robin_table(W, H, world)
        rr_loads = np.array([cost[r[r >= 0]].sum() for r in rr.reshape(world, -1)])
        assert loads.mean() / loads.max() >= 0.99 and loads.mean() / loads.max() >= rr_loads.mean() / rr_loads.max()
        tracer.SetTileTable(world, tprb, table)
        n = tprb * 512
        gathered = torch.zeros(world * n, dtype=torch.int32, device="cuda")
        acc = dict(rays_closest=0, rays_shadow=0, shaded_hits=0, pixels=0)
        for r in range(world):
            s_r = tracer.RenderDevice(gathered[r * n:(r + 1) * n].data_ptr(), shard_rank=r, shard_count=world)
            for k in acc:
                acc[k] += s_r[k]
        for k in acc:
            assert acc[k] == st[k], (world, k, acc[k], st[k])
        out = torch.zeros(W * H, dtype=torch.int32, device="cuda")
        xrt.dist.detile_device(gathered, W, H, world, out, table_dev=torch.from_numpy(table).cuda(), tiles_per_rank=tprb)
        torch.cuda.synchronize()
        assert torch.equal(out, whole), ("balanced", world)
        tracer.SetTileTable(world, tprb, None)
    tracer.RenderDevice(again.data_ptr())
    assert torch.equal(again, whole)


def test_tile_table_validation_and_small_frames(xrt, orc):
    """xrt_scene_set_tile_table refuses a table that misses or repeats a tile; a hand-made (reversed, uneven) table on a small frame renders
    the same pixels; frames of another size keep the round-robin layout while the table is installed; a scene traced by the per-lane kernel
    reports no costs and xrt_balance_tiles then answers round-robin."""
    import torch
    spec = xrt.configs.crate_grid_scene(200, 120)   # 4 x 15 = 60 tiles
    spec.multisampling = xrt.abi.MS_FIXED16
    scene, tracer = xrt.configs.build_product(spec)
    want = tracer.Render().copy()
    W, H, world, tiles = 200, 120, 3, 60
    tpr = 26
    table = np.full(world * tpr, -1, dtype=np.int32)
    order = np.arange(tiles)[::-1]
    table[0:26] = order[:26]; table[26:26 + 20] = order[26:46]; table[52:52 + 14] = order[46:]
    for bad in (table[:world * tpr - 1].tolist() + [table[0]], np.where(table == 7, -1, table)):
        with pytest.raises(ValueError):
            tracer.SetTileTable(world, tpr, np.asarray(bad, dtype=np.int32))
    tracer.SetTileTable(world, tpr, table)
    n = tpr * 512
    gathered = torch.zeros(world * n, dtype=torch.int32, device="cuda")
    for r in range(world):
        tracer.RenderDevice(gathered[r * n:(r + 1) * n].data_ptr(), shard_rank=r, shard_count=world)
    out = torch.zeros(W * H, dtype=torch.int32, device="cuda")
    xrt.dist.detile_device(gathered, W, H, world, out, table_dev=torch.from_numpy(table).cuda(), tiles_per_rank=tpr)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), want)
    assert np.array_equal(xrt.dist.detile_host(gathered.cpu().numpy().view(np.uint32), W, H, world, table=table, tiles_per_rank=tpr), want)
    # two shards of the same frame: not the table's shard count -> round-robin
    tx, ty, tpr2 = xrt.dist.shard_layout(W, H, 2)
    g2 = torch.zeros(2 * tpr2 * 512, dtype=torch.int32, device="cuda")
    for r in range(2):
        tracer.RenderDevice(g2[r * tpr2 * 512:(r + 1) * tpr2 * 512].data_ptr(), shard_rank=r, shard_count=2)
    xrt.dist.detile_device(g2, W, H, 2, out)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint32), want)
    # a crate traced at one sample per pixel goes to the per-lane kernel: no costs
    s1 = xrt.configs.crate_scene(128, 64, 2)
    sc1, tr1 = xrt.configs.build_product(s1)
    tr1.Render()
    c1 = tr1.TileCosts()
    assert not c1.any()
    tprb, t1 = xrt.dist.balanced_table(128, 64, 4, c1, slack=0.0)
    assert np.array_equal(t1, xrt.dist.round_robin_table(128, 64, 4)[1])


def test_in_library_multi_gpu(xrt, monkeypatch):
    """xrt_render_opts.n_gpus: ONE call from ONE host thread (RT:103-126) spreads the frame's tiles over N devices, one grouped
    RCCL send/recv gathers them on the scene's device, k_detile writes the frame.  On this one-GPU box: n_gpus = 1 is today's
    path; asking for more devices than are visible is XRT_E_NO_DEVICE; with XRT_FAKE_GPUS=1 the N scene replicas live on
    the one device and the exchange is RCCL send-to-self, which runs replica upload, the per-device host threads, the
    shard parameters, the RCCL calls, the de-tile and the statistics for N = 2, 3, 8 -- for plain, 16-sub-ray, adaptive and
    ray-tree frames, to host and to device memory, blocking and pipelined."""
    import torch
    spec = xrt.configs.crate_grid_scene(200, 120)
    scene, tracer = xrt.configs.build_product(spec)
    want = tracer.Render().copy()
    st_want = dict(tracer.last_stats)
    tracer.NumGpus = 1
    assert np.array_equal(tracer.Render(), want)
    ndev = torch.cuda.device_count()
    tracer.NumGpus = ndev + 1
    with pytest.raises(xrt.abi.XrtError) as e:
        tracer.Render()
    assert e.value.code == xrt.abi.XRT_E_NO_DEVICE
    tracer.NumGpus = 2
    with pytest.raises(ValueError):
        xrt.abi.check(_render_with(xrt, scene, tracer, n_gpus=2, shard_count=2))   # the library shards the frame itself
    tracer.NumGpus = 1
    monkeypatch.setenv("XRT_FAKE_GPUS", "1")
    keys = ("rays_closest", "rays_shadow", "hits_closest", "shaded_hits", "pixels")
    for mode in ("plain", "ms16", "adaptive", "glass"):
        s2 = xrt.configs.crate_grid_scene(200, 120) if mode != "glass" else glass_scene(xrt, 96, 64, 3)
        if mode == "ms16":
            s2.multisampling = xrt.abi.MS_FIXED16
        if mode == "adaptive":
            s2.multisampling, s2.multisample_quality = xrt.abi.MS_ADAPTIVE, 2
        scene2, tracer2 = xrt.configs.build_product(s2)
        ref = tracer2.Render().copy()
        st_ref = dict(tracer2.last_stats)
        for n in (2, 3, 8):
            tracer2.NumGpus = n
            got = tracer2.Render().copy()                                   # xrt_render: host Color[]
            assert np.array_equal(got, ref), (mode, n)
            for k in keys:
                assert tracer2.last_stats[k] == st_ref[k], (mode, n, k)
            assert tracer2.last_stats["pieces"] == n
            d = torch.zeros(s2.width * s2.height, dtype=torch.int32, device="cuda")
            tracer2.RenderDevice(d.data_ptr())                              # xrt_render_device
            assert np.array_equal(d.cpu().numpy().view(np.uint32), ref), (mode, n)
        # balance_tiles: the second frame onwards deals the tiles by the frame before's costs (adaptive and ray-tree frames report none: round-robin)
        tracer2.BalanceTiles = True
        for n in (3, 8):
            tracer2.NumGpus = n
            for rep in range(3):
                assert np.array_equal(tracer2.Render(), ref), (mode, n, rep, "balanced")
                for k in keys:
                    assert tracer2.last_stats[k] == st_ref[k], (mode, n, k, "balanced")
            d = torch.zeros(s2.width * s2.height, dtype=torch.int32, device="cuda")
            tracer2.RenderDevice(d.data_ptr())
            assert np.array_equal(d.cpu().numpy().view(np.uint32), ref), (mode, n, "balanced")
        tracer2.NumGpus = 2
        outs = [torch.zeros(s2.width * s2.height, dtype=torch.int32, device="cuda") for _ in range(2)]
        frs = [tracer2.PrepareDevice(o.data_ptr()) for o in outs]
        for rep in range(2):   # two tickets open under the installed table
            t0, t1 = frs[0].begin(), frs[1].begin()
            frs[0].end(t0); frs[1].end(t1)
            for o in outs:
                assert np.array_equal(o.cpu().numpy().view(np.uint32), ref), (mode, "balanced, pipelined")
        tracer2.BalanceTiles = False
        assert np.array_equal(tracer2.Render(), ref), (mode, "round-robin again")
        # pipelined, two tickets open, to device and to host memory
        tracer2.NumGpus = 2
        outs = [torch.zeros(s2.width * s2.height, dtype=torch.int32, device="cuda") for _ in range(2)]
        frs = [tracer2.PrepareDevice(o.data_ptr()) for o in outs]
        t0, t1 = frs[0].begin(), frs[1].begin()
        frs[0].end(t0); frs[1].end(t1)
        for o in outs:
            assert np.array_equal(o.cpu().numpy().view(np.uint32), ref), mode
        hosts = [np.zeros(s2.width * s2.height, dtype=np.uint32) for _ in range(2)]
        hfr = [tracer2.PrepareHost(h) for h in hosts]
        t0, t1 = hfr[0].begin(), hfr[1].begin()
        hfr[0].end(t0); hfr[1].end(t1)
        for h in hosts:
            assert np.array_equal(h, ref), mode
        tracer2.NumGpus = 1
    assert st_want["pixels"] == 200 * 120
    # RCCL cannot be loaded (the message used to be built from a second dlerror() call, std::string(NULL)): the render
    # fails with XRT_E_RCCL and its loader message, leaves no frame context pending, and the scene renders on afterwards
    monkeypatch.setenv("XRT_RCCL_LIB", "/nonexistent/librccl-missing.so")
    scene3, tracer3 = xrt.configs.build_product(spec)
    tracer3.NumGpus = 2
    with pytest.raises(xrt.abi.XrtError) as e:
        tracer3.Render()
    assert e.value.code == xrt.abi.XRT_E_RCCL and "cannot load librccl.so" in str(e.value)
    tracer3.NumGpus = 1
    assert np.array_equal(tracer3.Render(), want)
    monkeypatch.delenv("XRT_RCCL_LIB")
    tracer3.NumGpus = 2
    assert np.array_equal(tracer3.Render(), want)


def _render_with(xrt, scene, tracer, n_gpus, shard_count):
    cam, lights, n, opts = tracer._camera_abi(), tracer._lights_abi(), len(tracer.Lights), tracer._opts_abi()
    opts.n_gpus, opts.shard_count = n_gpus, shard_count
    out = np.zeros(tracer._target.Width * tracer._target.Height, dtype=np.uint32)
    return xrt.abi.lib().xrt_render(scene.handle, C.byref(cam), lights, n, C.byref(opts), out.ctypes.data_as(C.POINTER(C.c_uint32)), None, None)


def test_pipelined_host_output_frames(xrt):
    """xrt_render_begin / xrt_render_end: RenderAsync with the frame ending in the host's Color[] (RT:122-123).  Two tickets
    open, page-locked (xrt_host_register) and pageable buffers, equal to the blocking xrt_render; a third begin is BUSY."""
    spec = xrt.configs.config("C3", 0.25)
    scene, tracer = xrt.configs.build_product(spec)
    want = tracer.Render().copy()
    n = spec.width * spec.height
    lib = xrt.abi.lib()
    pinned = [np.zeros(n, dtype=np.uint32) for _ in range(2)]
    for b in pinned:
        xrt.abi.check(lib.xrt_host_register(C.c_void_p(b.ctypes.data), b.nbytes))
    try:
        for bufs in (pinned, [np.zeros(n, dtype=np.uint32) for _ in range(2)]):
            fr = [tracer.PrepareHost(b) for b in bufs]
            for rep in range(3):
                t0 = fr[0].begin()
                t1 = fr[1].begin()
                with pytest.raises(RuntimeError):
                    fr[0].begin()
                st0 = fr[0].end(t0)
                st1 = fr[1].end(t1)
                assert np.array_equal(bufs[0], want) and np.array_equal(bufs[1], want)
                assert st0["rays_closest"] == st1["rays_closest"] == tracer.last_stats["rays_closest"] > 0
                bufs[0][:] = 0
                bufs[1][:] = 0
    finally:
        for b in pinned:
            xrt.abi.check(lib.xrt_host_unregister(C.c_void_p(b.ctypes.data)))
    assert np.array_equal(tracer.Render(), want)


@pytest.mark.parametrize("split", ["1", "2"])
def test_frames_split_in_two_halves_on_two_streams(xrt, monkeypatch, split):
    """A blocking frame has no other frame to fill the drain of its launches, so libxrt renders it as two halves of its tiles
    in two frame contexts on two streams (XRT_SPLIT, default 1; thresholds forced to zero here so that test-sized frames take
    the path).  Same frame, same fp32 colour vector, same accounting as the unsplit render -- for 1 and 16 sub-rays, odd sizes,
    host and device output, and (XRT_SPLIT=2) also for pipelined frames with four contexts busy."""
    import torch
    for spec in (xrt.configs.crate_grid_scene(413, 230), xrt.configs.heightfield_scene(400, 225, m=96),
                 xrt.configs.heightfield_scene(200, 120, m=64, multisampling=xrt.abi.MS_FIXED16)):
        monkeypatch.setenv("XRT_SPLIT", "0")
        _, ref_tracer = xrt.configs.build_product(spec)
        ms = spec.multisampling != xrt.abi.MS_OFF
        want, want_f = ref_tracer.Render(want_float=True)
        want, want_f, st_want = want.copy(), want_f.copy(), dict(ref_tracer.last_stats)
        monkeypatch.setenv("XRT_SPLIT", split)
        monkeypatch.setenv("XRT_SPLIT_MS", "0")
        monkeypatch.setenv("XRT_OVERLAP_MS", "0")
        scene, tracer = xrt.configs.build_product(spec)
        for _ in range(3):   # the first frame of a scene is never split (no frame time known yet)
            got, got_f = tracer.Render(want_float=True)
            assert np.array_equal(got, want) and np.array_equal(got_f.view(np.uint32), want_f.view(np.uint32))
            for k in ("rays_closest", "rays_shadow", "hits_closest", "shaded_hits", "pixels"):
                assert tracer.last_stats[k] == st_want[k], (k, tracer.last_stats[k], st_want[k])
        assert tracer.last_stats["intersect_launches"] == st_want["intersect_launches"]
        assert tracer.last_stats["pieces"] == 2 and st_want["pieces"] == 1
        n = spec.width * spec.height
        outs = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(2)]
        frs = [tracer.PrepareDevice(o.data_ptr()) for o in outs]
        for rep in range(3):
            t0 = frs[0].begin()
            t1 = frs[1].begin()
            frs[0].end(t0)
            frs[1].end(t1)
            for o in outs:
                assert np.array_equal(o.cpu().numpy().view(np.uint32), want), rep
                o.zero_()
        frs[0]()   # blocking device render
        assert np.array_equal(outs[0].cpu().numpy().view(np.uint32), want)
        for k in ("XRT_SPLIT", "XRT_SPLIT_MS", "XRT_OVERLAP_MS"):
            monkeypatch.delenv(k)


def test_wave_packet_kernel_against_the_oracle(xrt, orc, monkeypatch):
    """k_packet (packet.hip): one wavefront walks the mesh octree once for 64 rays.  XRT_PACKET=31 routes every ray population
    of a one-body scene through it -- seam-1 batches included, i.e. incoherent random rays, rays leaving surfaces with an
    ignored triangle, axis-parallel, zero and non-finite rays: the worst case for a packet, and it must still give the
    reference's answers bit for bit (hit triangle, leaf id, u/v/d, world position), for the scene query and the per-mesh query;
    frames with 1 and 16 sub-rays equal the oracle's too."""
    monkeypatch.setenv("XRT_PACKET", "31")
    specs = {"h64": xrt.configs.heightfield_scene(160, 90, m=64), "h224": xrt.configs.heightfield_scene(160, 90, m=224),
             "soup": soup_spec(xrt, 2000, 7, 20, 0.15), "soup_deep": soup_spec(xrt, 300, 5, 4, 0.5)}
    t = xrt.configs.SceneSpec("crate5_rot")
    t.meshes.append((xrt.fixtures.crate(5), xrt.configs.material(0.5)))
    t.objects.append(([0], (3.0, -2.0, 5.0), (0.3, 1.1, -0.4), (1.0, 1.3, 0.8)))
    t.camera = xrt.configs.camera((0, 32, 64), (0, 8, 0))
    t.lights = [xrt.configs.spot((0, 40, 60))]
    specs["crate5_rot"] = t.with_size(128, 72)
    for name, spec in specs.items():
        scene, tracer = xrt.configs.build_product(spec)
        o = orc.OracleScene(spec)
        prim = tracer.GeneratePrimaryRays()
        sets = [prim, random_rays(xrt, 20000, 5, radius=60.0 if name.startswith("h") else 3.0)]
        nan = float("nan")
        sets.append(xrt.rays_array([(0, 50, 0), (0, 50, 0), (0, 50, 0), (1e30, 0, 0), (0, 50, 0), (0, 4.0, 0), (0, 2, 0), (1, 2, 1)] * 9,
                                   [(0, 0, 0), (nan, -1, 0), (0, -1, 0), (-1, 0, 0), (1e-7, -1, 1e-7), (0, 1, 0), (1, 0, 0), (0, 0, -1)] * 9))
        for rays in sets:
            ho = o.intersect(rays)
            assert hits_equal(ho, scene.IntersectBatch(rays)) == {}, name
            sec = secondary_rays(xrt, ho, seed=4)
            if len(sec):
                assert hits_equal(o.intersect(sec), scene.IntersectBatch(sec)) == {}, name
        if spec.mesh_threshold == 50:   # Mesh.Init builds with the reference's threshold (MO:42)
            mesh = scene.meshes[0]
            mesh.Init()
            assert hits_equal(o.mesh_intersect(0, prim[::2]), mesh.Octree.IntersectBatch(prim[::2])) == {}, name
        rgba, rgbf = tracer_render(tracer, 2)
        o_rgba, o_rgbf, o_st = orc.OracleScene(spec_with(spec, 2)).render(nthreads=8)
        assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)
        for k in ("rays_closest", "rays_shadow", "hits_closest", "shaded_hits"):
            assert tracer.last_stats[k] == o_st[k], (name, k)
    spec = xrt.configs.heightfield_scene(96, 54, m=224, multisampling=xrt.abi.MS_FIXED16)
    _, tracer = xrt.configs.build_product(spec)
    o_rgba, _, _ = orc.OracleScene(spec).render(nthreads=8, want_float=False)
    assert np.array_equal(tracer.Render(), o_rgba)


@pytest.mark.parametrize("env", [{"XRT_PK_SPLIT": "1", "XRT_PK_BUDGET": "0", "XRT_PK_BUDGET_ITEM": "0"}, {"XRT_PK_SPLIT": "1", "XRT_PK_BUDGET": "0", "XRT_PK_BUDGET_ITEM": "0", "XRT_PK_SPLIT_ITEMS": "48"},
                                 {"XRT_PK_SPLIT": "1", "XRT_PK_BUDGET": "0", "XRT_PK_BUDGET_ITEM": "1000000"}, {"XRT_PK_SPLIT": "1", "XRT_PK_BUDGET": "2", "XRT_PK_BUDGET_ITEM": "1"},
                                 {"XRT_PK_SPLIT": "1", "XRT_PK_LONG": "1", "XRT_PK_BUDGET_LONG": "0"}, {"XRT_PK_SPLIT": "0"}])
def test_split_walks_never_change_results(xrt, orc, monkeypatch, env):
    """Split walks (packet.hip): a packet whose walk has outlasted its budget hands the pending subtrees of its stacked levels to other
    waves, takers may split again, the last participant merges the partial answers.  With a budget of ZERO every block a walk enters
    with something pending above gives all of it away (until the packet's record or the launch's arena is full -- a tiny arena is
    one of the cases), so nearly every (leaf, triangle) test of a packet is made by another wave than the one that writes the result:
    answers must still be the oracle's bit for bit -- seam-1 batches of primary, incoherent, axis-parallel / zero / non-finite rays and rays
    leaving surfaces with an ignored triangle, and whole frames with 1 and 16 sub-rays (shadow rays in a compact list with scattered
    answers, two populations in one launch)."""
    monkeypatch.setenv("XRT_PACKET", "31")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    specs = {"h224": xrt.configs.heightfield_scene(160, 90, m=224), "soup_deep": soup_spec(xrt, 300, 5, 4, 0.5), "soup": soup_spec(xrt, 2000, 7, 20, 0.15)}
    nan = float("nan")
    for name, spec in specs.items():
        scene, tracer = xrt.configs.build_product(spec)
        scene.SplitStats()
        o = orc.OracleScene(spec)
        prim = tracer.GeneratePrimaryRays()
        sets = [prim, random_rays(xrt, 20000, 5, radius=60.0 if name.startswith("h") else 3.0),
                xrt.rays_array([(0, 50, 0), (0, 50, 0), (0, 50, 0), (1e30, 0, 0), (0, 50, 0), (0, 4.0, 0), (0, 2, 0), (1, 2, 1)] * 9,
                               [(0, 0, 0), (nan, -1, 0), (0, -1, 0), (-1, 0, 0), (1e-7, -1, 1e-7), (0, 1, 0), (1, 0, 0), (0, 0, -1)] * 9)]
        for rays in sets:
            ho = o.intersect(rays)
            assert hits_equal(ho, scene.IntersectBatch(rays)) == {}, name
            sec = secondary_rays(xrt, ho, seed=4)
            if len(sec):
                assert hits_equal(o.intersect(sec), scene.IntersectBatch(sec)) == {}, name
        rgba, rgbf = tracer_render(tracer, 2)
        o_rgba, o_rgbf, o_st = orc.OracleScene(spec_with(spec, 2)).render(nthreads=8)
        assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)
        for k in ("rays_closest", "rays_shadow", "hits_closest", "shaded_hits"):
            assert tracer.last_stats[k] == o_st[k], (name, k)
        given, taken, packets, by_taker = scene.SplitStats()
        if env.get("XRT_PK_SPLIT") == "0": assert (given, taken, packets, by_taker) == (0, 0, 0, 0)
        else:   # every subtree handed over was walked by someone; with a budget of zero the walks really were split and takers wrote results
            assert given == taken and given >= packets, (name, given, taken, packets, by_taker)
            if env.get("XRT_PK_BUDGET") == "0": assert packets > 0 and by_taker > 0, (name, given, taken, packets, by_taker)
    spec = xrt.configs.heightfield_scene(96, 54, m=224, multisampling=xrt.abi.MS_FIXED16)
    scene, tracer = xrt.configs.build_product(spec)
    o_rgba, _, _ = orc.OracleScene(spec).render(nthreads=8, want_float=False)
    for _ in range(4):   # (frames in a row: the arena is reused, only the launch's serial number tells old items from new; from the second frame of a
        assert np.array_equal(tracer.Render(), o_rgba)   # context on, the packets that cost more than XRT_PK_LONG microseconds last time are split from the start)
    given, taken, packets, _ = scene.SplitStats()
    assert given == taken and (env.get("XRT_PK_BUDGET") != "0" or packets > 0) and (env.get("XRT_PK_SPLIT") != "0" or packets == 0)


def test_scene_packets_against_the_oracle(xrt, orc, monkeypatch):
    """k_packet<MODE_SCENE> (packet.hip): one wavefront walks the SCENE octree, the bodies of its leaves and each body's mesh
    octree once for 64 rays (OSM:312-455 -> MO:259-353).  XRT_PACKET=31 routes every ray population of two-level scenes through
    it -- seam-1 batches included: incoherent random rays, rays from inside the bodies, axis-parallel / zero / non-finite rays,
    rays leaving surfaces with an ignored triangle, and the pre-cull's far-origin and grazing rays -- the worst cases for a packet,
    and it must still give the reference's answers bit for bit (object, mesh, triangle, leaf id, u/v/d, world position); bodies
    that share meshes, meshes whose octree root is a leaf (MO:265 with a single bucket), rotated / non-uniformly scaled bodies,
    two meshes per body; frames with 1 and 16 sub-rays equal the oracle's."""
    from util import far_origin_scene, far_origin_rays, precull_adversarial_scene, grazing_rays
    monkeypatch.setenv("XRT_PACKET", "31")
    specs = {}
    s = xrt.configs.SceneSpec("inst")
    s.meshes.append((xrt.fixtures.crate(3), xrt.configs.material(0.5, texture=xrt.fixtures.crate_texture())))
    s.meshes.append((triangle_soup(80, 11, 0.4), xrt.configs.material(0.2, interpolate_normals=True)))
    s.meshes.append((xrt.fixtures.crate(1), xrt.configs.material(0.6)))   # 12 triangles: the octree root is a leaf
    k = 0
    for ix in range(5):
        for iz in range(5):
            ids = [[0], [0, 1], [2], [2, 0]][k % 4]
            s.objects.append((ids, (-60.0 + 30.0 * ix, 2.0 * (k % 2), -60.0 + 30.0 * iz), (0.1 * ix, 0.37 * iz, 0.05 * (ix + iz)), (1.0 + 0.1 * ix, 1.0, 0.8 + 0.1 * iz)))
            k += 1
    s.camera = xrt.configs.camera((0, 80, 160), (0, 0, 0))
    s.lights = [xrt.configs.spot((0, 100, 100)), xrt.configs.directional((0.3, 0.8, 0.5), (0.4, 0.5, 0.6), 0.7)]
    s.max_reflections = 3
    specs["inst"] = s.with_size(160, 90)
    specs["grid"] = xrt.configs.crate_grid_scene(160, 90)
    three = xrt.configs.SceneSpec("leaf_roots")
    three.meshes.append((xrt.fixtures.crate(1), xrt.configs.material(0.5)))
    for i in range(3):
        three.objects.append(([0], (-20.0 + 20.0 * i, 0.0, 3.0 * i), (0.0, 0.5 * i, 0.0), (1.0, 1.0 + 0.2 * i, 1.0)))
    three.camera = xrt.configs.camera((0, 32, 64), (0, 8, 0))
    three.lights = [xrt.configs.spot((0, 40, 60))]
    three.max_reflections = 2
    specs["leaf_roots"] = three.with_size(128, 72)
    for seed in (101, 404, 909, 1717, 2626):
        fs = fuzz_spec(xrt, seed)
        if fs is not None:
            specs["fuzz%d" % seed] = fs
    nan = float("nan")
    for name, spec in specs.items():
        try:
            scene, tracer = xrt.configs.build_product(spec)
        except xrt.abi.XrtError as e:
            assert "would not terminate" in str(e)
            continue
        o = orc.OracleScene(spec)
        first_use = []
        for ids, _, _, _ in spec.objects:
            first_use += [i for i in ids if i not in first_use]
        to_spec = np.array(first_use + [-1], dtype=np.int32)
        inv = {v: i for i, v in enumerate(first_use)}

        def product_hits(r):
            h = scene.IntersectBatch(r).copy()
            h["mesh"] = np.where(h["hit"] != 0, to_spec[np.clip(h["mesh"], 0, len(first_use))], h["mesh"])
            return h
        prim = o.primary_rays()
        radius = 12.0 if name.startswith("fuzz") else 150.0
        def wide_rays(n, seed, r):   # origins on a sphere of radius r, aimed anywhere into the scene (random_rays aims at the origin)
            g = np.random.default_rng(seed)
            oo = g.normal(size=(n, 3)).astype(np.float32)
            oo *= (r / np.linalg.norm(oo, axis=1, keepdims=True)).astype(np.float32)
            dd = g.uniform(-0.6 * r, 0.6 * r, size=(n, 3)).astype(np.float32) * np.array([1.0, 0.1, 1.0], dtype=np.float32) - oo
            dd /= np.linalg.norm(dd, axis=1, keepdims=True).astype(np.float32)
            return xrt.rays_array(oo, dd.astype(np.float32))
        sets = [prim, wide_rays(20000, 5, radius), wide_rays(6000, 6, radius * 0.3), random_rays(xrt, 4000, 7, radius=radius)]
        sets.append(xrt.rays_array([(0, 50, 0), (0, 50, 0), (0, 50, 0), (1e30, 0, 0), (0, 50, 0), (0, 4.0, 0), (0, 2, 0), (1, 2, 1)] * 9,
                                   [(0, 0, 0), (nan, -1, 0), (0, -1, 0), (-1, 0, 0), (1e-7, -1, 1e-7), (0, 1, 0), (1, 0, 0), (0, 0, -1)] * 9))
        for rays in sets:
            ho = o.intersect(rays)
            assert hits_equal(ho, product_hits(rays)) == {}, name
            sec = secondary_rays(xrt, ho, seed=4)
            if len(sec):
                sec_p = sec.copy()
                sec_p["ignore_mesh"] = np.array([inv.get(int(m), -1) for m in sec["ignore_mesh"]], dtype=np.int32)
                assert hits_equal(o.intersect(sec), product_hits(sec_p)) == {}, name
        rgba, rgbf = tracer.Render(want_float=True)
        o_rgba, o_rgbf, o_st = o.render(nthreads=8)
        assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)
        for k in ("rays_closest", "rays_shadow", "hits_closest", "shaded_hits"):
            assert tracer.last_stats[k] == o_st[k], (name, k)
        import copy
        s16 = copy.deepcopy(spec).with_size(64, 36)
        s16.multisampling = xrt.abi.MS_FIXED16
        _, tr16 = xrt.configs.build_product(s16)
        o16, _, _ = orc.OracleScene(s16).render(nthreads=8, want_float=False)
        assert np.array_equal(tr16.Render(), o16), name
    # the object pre-cull's adversarial rays through the packet kernel (it applies the same margin, traverse.h precull_hit)
    s = far_origin_scene(xrt)
    scene, _ = xrt.configs.build_product(s)
    o = orc.OracleScene(s)
    for radius, seed in ((1e3, 1), (1e4, 2), (1e5, 3)):
        rays = far_origin_rays(xrt, s, radius, 2000, seed)
        assert hits_equal(o.intersect(rays), scene.IntersectBatch(rays)) == {}, radius
    s = precull_adversarial_scene(xrt)
    scene, _ = xrt.configs.build_product(s)
    o = orc.OracleScene(s)
    rays = grazing_rays(xrt, s, 6000, 11)
    assert hits_equal(o.intersect(rays), scene.IntersectBatch(rays)) == {}


def test_changing_frame_parameters_between_pipelined_frames(xrt):
    """One scene, frames in flight while the caller changes what a frame is: reflection depth, lights, target size, a
    counting pass in between.  Per-context counters, the light cache, the cost map and the work buffers must follow."""
    import torch
    spec = xrt.configs.heightfield_scene(160, 90, m=96)
    scene, tracer = xrt.configs.build_product(spec)

    def setup(step):
        tracer.MaxReflections = (2, 4, 1, 3)[step % 4]
        tracer.CurrentTarget = xrt.api.RenderTarget(*((160, 90), (160, 90), (224, 126), (96, 54))[step % 4])
        c = spec.camera
        tracer.CurrentCamera = xrt.api.Camera(c["pos"], c["target"], c["up"], c["fov"],
                                              xrt.xna.aspect_ratio(tracer.CurrentTarget.Width, tracer.CurrentTarget.Height), c["near"], c["far"])
        lights = list(tracer.Lights)
        if step % 4 == 1 and len(lights) < 2:
            L = xrt.api.DirectionalLight()
            L.Direction, L.Color, L.Intensity = (0.3, 0.8, 0.5), (0.4, 0.5, 0.6), 0.7
            tracer.Lights = lights + [L]
        if step % 4 == 3:
            tracer.Lights = lights[:1]
    want = {}
    for step in range(4):
        setup(step)
        want[step] = tracer.Render().copy()
    assert len({w.tobytes() for w in want.values()}) == 4
    open_frames = []
    for i in range(16):
        step = (i // 2) % 4                      # two frames per parameter set, so consecutive frames in flight differ
        setup(step)
        if i == 9:                               # a counting pass (a blocking call: close the open ticket first) in the middle
            f, t, o, s = open_frames.pop(0)
            f.end(t)
            assert np.array_equal(o.cpu().numpy().view(np.uint32), want[s]), (i, s)
            tracer.collect_stats = True
            assert np.array_equal(tracer.Render(), want[step])
            assert tracer.last_stats["tri_tests"] > 0
            tracer.collect_stats = False
        out = torch.zeros(tracer.CurrentTarget.Width * tracer.CurrentTarget.Height, dtype=torch.int32, device="cuda")
        fr = tracer.PrepareDevice(out.data_ptr())
        open_frames.append((fr, fr.begin(), out, step))
        while len(open_frames) >= 2:
            f, t, o, s = open_frames.pop(0)
            f.end(t)
            assert np.array_equal(o.cpu().numpy().view(np.uint32), want[s]), (i, s)
    while open_frames:
        f, t, o, s = open_frames.pop(0)
        f.end(t)
        assert np.array_equal(o.cpu().numpy().view(np.uint32), want[s]), s


def test_reference_content_scene(xrt, orc):
    """The reference's own assets (monkey, torus, plane, cube, Sphere from RayTraceProjectContent, imported by fbx.py and
    committed as data) with the parameters of its content project: glass monkey and sphere (ray tree), textured ground,
    two lights.  Hits and frames (plain, bilinear + mirror addressing, adaptive supersampling) against the oracle."""
    spec = xrt.configs.content_scene(192, 108, max_reflections=4)
    scene, tracer = xrt.configs.build_product(spec)
    o = orc.OracleScene(spec)
    prim = o.primary_rays()
    o_hits = o.intersect(prim)
    assert (o_hits["hit"] != 0).mean() > 0.3
    assert hits_equal(o_hits, scene.IntersectBatch(prim)) == {}     # bodies list their meshes in spec order here
    tracer.collect_stats = True
    rgba, rgbf = tracer.Render(want_float=True)
    o_rgba, o_rgbf, o_st = o.render(nthreads=8)
    assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)
    assert o_st["rays_closest"] > 192 * 108 * 1.5, "no ray tree"
    for k in ("rays_closest", "rays_shadow", "shaded_hits", "tri_tests", "algorithmic_bytes"):
        assert tracer.last_stats[k] == o_st[k], (k, tracer.last_stats[k], o_st[k])
    spec2 = xrt.configs.content_scene(128, 72, max_reflections=2)
    spec2.address_mode, spec2.filtering = xrt.abi.ADDRESS_MIRROR, xrt.abi.FILTER_BILINEAR
    spec2.multisampling, spec2.multisample_quality = xrt.abi.MS_ADAPTIVE, 1
    _, tracer2 = xrt.configs.build_product(spec2)
    assert np.array_equal(tracer2.Render(), orc.OracleScene(spec2).render(nthreads=8, want_float=False)[0])


def test_reference_content_scene2(xrt, orc):
    """Assets imported with the ModelProcessor rotation parameters of the content project (glass prism2.fbx, chesspiece.fbx,
    fbx.import_mesh `rotation`) on a ground whose texture the host mirror reads from a FILE (Material(textureFilePath) ->
    Material.Init, MAT:59-69 / TMP:121-131): hits and frames bit-identical to the oracle, plain / ray tree / 16 sub-rays /
    adaptive, point and bilinear filtering."""
    spec = xrt.configs.content_scene2(192, 108)
    scene, tracer = xrt.configs.build_product(spec)
    assert scene.meshes[0].MeshMaterial.TextureFilePath and scene.meshes[0].MeshMaterial.Texture.shape == (512, 512)
    o = orc.OracleScene(spec)
    prim = tracer.GeneratePrimaryRays()
    assert prim.tobytes() == o.primary_rays().tobytes()
    ho = o.intersect(prim)
    assert hits_equal(ho, scene.IntersectBatch(prim)) == {}
    sec = secondary_rays(xrt, ho, seed=9)
    assert hits_equal(o.intersect(sec), scene.IntersectBatch(sec)) == {}
    for ms, filt in ((xrt.abi.MS_OFF, xrt.abi.FILTER_POINT), (xrt.abi.MS_OFF, xrt.abi.FILTER_BILINEAR), (xrt.abi.MS_FIXED16, xrt.abi.FILTER_POINT),
                     (xrt.abi.MS_ADAPTIVE, xrt.abi.FILTER_BILINEAR)):
        s2 = xrt.configs.content_scene2(96 if ms != xrt.abi.MS_OFF else 192, 54 if ms != xrt.abi.MS_OFF else 108)
        s2.multisampling, s2.multisample_quality, s2.filtering = ms, 1, filt
        _, t2 = xrt.configs.build_product(s2)
        rgba, rgbf = t2.Render(want_float=True)
        o_rgba, o_rgbf, o_st = orc.OracleScene(s2).render(nthreads=8)
        assert_frames_equal(rgba, rgbf, o_rgba, o_rgbf)
        assert t2.last_stats["rays_closest"] == o_st["rays_closest"] and t2.last_stats["rays_shadow"] == o_st["rays_shadow"]


def test_scene_file_renders_like_the_scene_it_was_saved_from(xrt, tmp_path):
    """xrt_scene_save -> xrt_scene_load -> xrt_scene_build on the GPU: same hits, same frame."""
    spec = xrt.configs.content_scene2(160, 90)
    scene, tracer = xrt.configs.build_product(spec)
    want = tracer.Render().copy()
    rays = tracer.GeneratePrimaryRays()
    hits = scene.IntersectBatch(rays)
    path = str(tmp_path / "content2.xrts")
    scene.Save(path)
    tracer.CurrentScene = xrt.api.OctreeSpatialManager.Load(path)
    assert hits_equal(hits, tracer.CurrentScene.IntersectBatch(rays)) == {}
    assert np.array_equal(tracer.Render(), want)


def test_ray_tree_overflow_retries_with_fewer_paths(xrt, monkeypatch):
    """A generation of the ray tree that does not fit the ray buffers (forced here by XRT_HEAP_RAY_CAP: one ray per path,
    while a glass sphere filling the view doubles the rays of most paths) is discarded and its chunk retried with a
    quarter of the paths: same frame, same accounting as with room to spare."""
    spec = xrt.configs.default_game_scene(96, 96, max_reflections=4)
    spec.objects = [([0], (0.0, 0.0, 0.0), (0.0, 0.0, 0.0), (4.0, 4.0, 4.0))]
    spec.camera = xrt.configs.camera((0, 0, 14), (0, 0, 0))
    scene, tracer = xrt.configs.build_product(spec)
    want = tracer.Render().copy()
    st_want = dict(tracer.last_stats)
    assert st_want["rays_closest"] > 3 * 96 * 96, "the sphere does not fill the view"
    monkeypatch.setenv("XRT_HEAP_RAY_CAP", "1024")
    scene2, tracer2 = xrt.configs.build_product(spec)
    monkeypatch.delenv("XRT_HEAP_RAY_CAP")
    got = tracer2.Render()
    assert np.array_equal(got, want)
    for k in ("rays_closest", "rays_shadow", "shaded_hits", "pixels"):
        assert tracer2.last_stats[k] == st_want[k], (k, tracer2.last_stats[k], st_want[k])
    assert tracer2.last_stats["intersect_launches"] > st_want["intersect_launches"], "no chunk was split"
    assert np.array_equal(tracer2.Render(), want)   # (the scene's later ray-tree frames go the careful way from the start)
    # A ray-tree frame of one chunk is enqueued optimistically, without a host round trip, so that two can be in flight; one that
    # overflows is rendered again when it is waited for.  Two such frames in flight, HBM and host output:
    import torch
    monkeypatch.setenv("XRT_HEAP_RAY_CAP", "1024")
    scene3, tracer3 = xrt.configs.build_product(spec)
    scene4, tracer4 = xrt.configs.build_product(spec)
    monkeypatch.delenv("XRT_HEAP_RAY_CAP")
    n = spec.width * spec.height
    outs = [torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(2)]
    frs = [tracer3.PrepareDevice(o.data_ptr()) for o in outs]
    t0, t1 = frs[0].begin(), frs[1].begin()
    st0 = frs[0].end(t0)
    st1 = frs[1].end(t1)
    for o, st in zip(outs, (st0, st1)):
        assert np.array_equal(o.cpu().numpy().view(np.uint32), want)
        assert st["rays_closest"] == st_want["rays_closest"] and st["shaded_hits"] == st_want["shaded_hits"]
    bufs = [np.zeros(n, dtype=np.uint32) for _ in range(2)]
    hfr = [tracer4.PrepareHost(b) for b in bufs]
    t0, t1 = hfr[0].begin(), hfr[1].begin()
    hfr[0].end(t0); hfr[1].end(t1)
    assert np.array_equal(bufs[0], want) and np.array_equal(bufs[1], want)
    # and with room to spare two optimistic frames in flight are simply the frame, twice
    frs = [tracer.PrepareDevice(o.data_ptr()) for o in outs]
    for o in outs:
        o.zero_()
    torch.cuda.synchronize()
    t0, t1 = frs[0].begin(), frs[1].begin()
    frs[0].end(t0); frs[1].end(t1)
    assert np.array_equal(outs[0].cpu().numpy().view(np.uint32), want) and np.array_equal(outs[1].cpu().numpy().view(np.uint32), want)
