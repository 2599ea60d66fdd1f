"""Known-answer tests of the CPU oracle.  The reference has no tests; these are
authored by the build with hand-computed answers, plus the committed golden vectors."""
import ctypes as C
import math
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def fa(*x):
    return np.array(x, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def tri(orc, o, d, v, sn):
    out = np.zeros(3, dtype=np.float32)
    hit = orc.lib().orc_kat_triangle(_p(fa(*o)), _p(fa(*d)), _p(fa(*v)), _p(fa(*sn)), _p(out))
    return hit, out


def test_k1_triangle_known_answer(orc):
    # unit right triangle in z=0, ray straight down at (0.25, 0.25): t=1, u=v=0.25 (hand computed)
    hit, uvd = tri(orc, (0.25, 0.25, 1), (0, 0, -1), (0, 0, 0, 1, 0, 0, 0, 1, 0), (0, 0, 1))
    assert hit == 1 and tuple(uvd) == (0.25, 0.25, 1.0)
    # edge inclusive: u + v == 1 accepted (RE:74), u < 0 rejected
    hit, uvd = tri(orc, (0.5, 0.5, 2), (0, 0, -1), (0, 0, 0, 1, 0, 0, 0, 1, 0), (0, 0, 1))
    assert hit == 1 and tuple(uvd) == (0.5, 0.5, 2.0)
    hit, _ = tri(orc, (-0.25, 0.25, 1), (0, 0, -1), (0, 0, 0, 1, 0, 0, 0, 1, 0), (0, 0, 1))
    assert hit == 0
    # behind the origin: distance < 0 rejected; distance == 0 accepted (Q10)
    hit, _ = tri(orc, (0.25, 0.25, -1), (0, 0, -1), (0, 0, 0, 1, 0, 0, 0, 1, 0), (0, 0, 1))
    assert hit == 0
    hit, uvd = tri(orc, (0.25, 0.25, 0), (0, 0, -1), (0, 0, 0, 1, 0, 0, 0, 1, 0), (0, 0, 1))
    assert hit == 1 and uvd[2] == 0.0


def test_k2_backface_and_grazing(orc):
    # N.D > 0 -> culled (RE:48-51)
    hit, _ = tri(orc, (0.25, 0.25, 1), (0, 0, -1), (0, 0, 0, 1, 0, 0, 0, 1, 0), (0, 0, -1))
    assert hit == 0
    # N.D == 0 passes the culling test, then 1/0 -> inf/NaN fails every comparison
    hit, uvd = tri(orc, (-1, 0.25, 0), (1, 0, 0), (0, 0, 0, 1, 0, 0, 0, 1, 0), (0, 0, 1))
    assert hit == 0 and not np.isfinite(uvd).all()


def box(orc, o, d, b):
    key = C.c_float(-1)
    hit = orc.lib().orc_kat_box(_p(fa(*o)), _p(fa(*d)), _p(fa(*b)), C.byref(key))
    return hit, key.value


def test_k3_box_entry_distance(orc):
    assert box(orc, (0, 0, 5), (0, 0, -1), (-1, -1, -1, 1, 1, 1)) == (1, 4.0)
    assert box(orc, (0, 0, 0), (0, 0, -1), (-1, -1, -1, 1, 1, 1)) == (1, 0.0)      # origin inside -> key 0 (Q2)
    assert box(orc, (0, 0, -5), (0, 0, -1), (-1, -1, -1, 1, 1, 1))[0] == 0         # box behind the ray
    assert box(orc, (2, 0, 5), (0, 0, -1), (-1, -1, -1, 1, 1, 1))[0] == 0          # parallel axis, origin outside the slab
    assert box(orc, (1, 0, 5), (5e-7, 0, -1), (-1, -1, -1, 1, 1, 1)) == (1, 4.0)   # |d| < 1e-6 takes the parallel branch, face inclusive
    hit, key = box(orc, (-3, 0, 0), (1, 0, 0), (-1, -1, -1, 1, 1, 1))
    assert (hit, key) == (1, 2.0)
    # -0 from (min - o) * inv is replaced by +0 through Math.Max(t, 0)
    hit, key = box(orc, (-1, 0, 0), (1, 0, 0), (-1, -1, -1, 1, 1, 1))
    assert hit == 1 and key == 0.0 and not math.copysign(1, key) < 0


def test_k4_crate_winding(xrt):
    # the import definition must give outward surface normals: dot(surfaceNormal, FBX vertex normal) > 0
    m = xrt.fixtures.crate(1)
    assert m.ntri == 12
    assert ((m.surface_normal * m.n[:, 0, :]).sum(axis=1) > 0.999).all()
    m11 = xrt.fixtures.crate(11)
    assert m11.ntri == 1452
    assert ((m11.surface_normal * m11.n[:, 0, :]).sum(axis=1) > 0.999).all()
    assert np.array_equal(xrt.fixtures.crate(1).bbox, m11.bbox)


def strip_mesh(xrt, n):
    """n small triangles in a row along x (no shared vertices between neighbours beyond two)."""
    v = np.zeros((n, 3, 3), dtype=np.float32)
    for i in range(n):
        x = np.float32(i)
        v[i] = [(x, 0, 0), (x, 0, 1), (x + np.float32(0.5), 0, 0)]
    return xrt.fixtures.MeshData(v, np.zeros((n, 3, 3), np.float32), np.zeros((n, 3, 2), np.float32), np.ones((n, 4), np.float32))


def spec_of(xrt, mesh, threshold=50):
    s = xrt.configs.SceneSpec("kat")
    s.meshes.append((mesh, xrt.configs.material(0.5)))
    s.objects.append(([0], (0, 0, 0), (0, 0, 0), (1, 1, 1)))
    s.camera = xrt.configs.camera((0, 5, 5), (0, 0, 0))
    s.lights = [xrt.configs.spot((0, 10, 10))]
    s.mesh_threshold = threshold
    return s.with_size(16, 16)


def test_k5_octree_one_split(xrt, orc):
    for n, expect_nodes in ((50, 1), (51, 9)):
        nodes, refs = orc.OracleScene(spec_of(xrt, strip_mesh(xrt, n))).tree(0)
        assert len(nodes) == expect_nodes
    nodes, refs = orc.OracleScene(spec_of(xrt, strip_mesh(xrt, 51))).tree(0)
    root = nodes[0]
    assert root["is_leaf"] == 0 and root["count"] == 51 and list(nodes["dfs_index"]) == list(range(9))
    half = (root["bmax"] - root["bmin"]) * np.float32(0.5)
    for c in range(8):   # child c = 4i+2j+k with i<->X, j<->Y, k<->Z (MO:210-222)
        i, j, k = (c >> 2) & 1, (c >> 1) & 1, c & 1
        mn = root["bmin"] + half * np.array([i, j, k], dtype=np.float32)
        assert np.array_equal(nodes[1 + c]["bmin"], mn) and np.array_equal(nodes[1 + c]["bmax"], mn + half)
        assert nodes[1 + c]["is_leaf"] == 1
    # lists keep the parent's order; a triangle with a vertex on the split plane is in both halves (Q5)
    for c in range(8):
        lst = refs[nodes[1 + c]["first_ref"]: nodes[1 + c]["first_ref"] + nodes[1 + c]["count"]]
        assert list(lst) == sorted(lst)
    lo = set(refs[nodes[1]["first_ref"]: nodes[1]["first_ref"] + nodes[1]["count"]])
    hi = set(refs[nodes[5]["first_ref"]: nodes[5]["first_ref"] + nodes[5]["count"]])
    assert lo & hi, "boundary triangle must be duplicated into both children"


def test_k5b_unbounded_recursion_is_reported(xrt, orc):
    # > threshold triangles sharing one vertex: MO:84-96 would recurse forever (Q5)
    n = 60
    v = np.zeros((n, 3, 3), dtype=np.float32)
    for i in range(n):
        a = np.float32(i) * np.float32(0.1)
        v[i] = [(0, 0, 0), (np.cos(a), 0, np.sin(a)), (np.cos(a + 0.05), 0, np.sin(a + 0.05))]
    md = xrt.fixtures.MeshData(v, np.zeros((n, 3, 3), np.float32), np.zeros((n, 3, 2), np.float32), np.ones((n, 4), np.float32))
    with pytest.raises(RuntimeError):
        orc.OracleScene(spec_of(xrt, md))


def test_k6_tie_break_first_wins(xrt, orc):
    # two coplanar duplicate triangles: strict '<' keeps the lower list index (MO:294)
    base = [(0, 0, 0), (1, 0, 0), (0, 0, 1)]
    v = np.array([base, base], dtype=np.float32)
    md = xrt.fixtures.MeshData(v, np.zeros((2, 3, 3), np.float32), np.zeros((2, 3, 2), np.float32), np.ones((2, 4), np.float32))
    assert md.surface_normal[0][1] > 0
    o = orc.OracleScene(spec_of(xrt, md))
    h = o.intersect(xrt.rays_array([(0.25, 2, 0.25)], [(0, -1, 0)]))[0]
    assert h["hit"] == 1 and h["tri"] == 0 and h["d"] == 2.0
    # K8: ignoreTriangle is (mesh, index) identity: ignoring triangle 0 exposes its duplicate
    h = o.intersect(xrt.rays_array([(0.25, 2, 0.25)], [(0, -1, 0)], 0, 0))[0]
    assert h["hit"] == 1 and h["tri"] == 1


def test_k7_leaf_group_is_not_nearest_hit(xrt, orc):
    # Q1: the first bucket with a hit wins even when a later bucket holds a nearer hit.  Random soups
    # with a tiny threshold make this happen; compare with a brute-force nearest-hit search.
    from util import triangle_soup, random_rays
    md = triangle_soup(60, seed=3, size=0.9)
    o = orc.OracleScene(spec_of(xrt, md, threshold=2))
    rays = random_rays(xrt, 4000, seed=4)
    hits = o.intersect(rays)
    differs = 0
    out = np.zeros(3, dtype=np.float32)
    for r, h in zip(rays[:600], hits[:600]):
        best = None
        for t in range(md.ntri):
            if orc.lib().orc_kat_triangle(_p(fa(*r["o"])), _p(fa(*r["d"])), _p(md.v[t].reshape(-1).copy()), _p(md.surface_normal[t].copy()), _p(out)):
                if best is None or out[2] < best[0]:
                    best = (float(out[2]), t)
        if h["hit"] and best is not None and best[1] != h["tri"]:
            assert best[0] <= h["d"]
            differs += 1
        if best is None:
            assert h["hit"] == 0
    assert differs > 0, "fixture does not exercise the leaf-group quirk"


def test_k9_color_pack(orc):
    pk = lambda *c: orc.lib().orc_kat_pack_color(_p(fa(*c)))
    assert pk(0, 0, 0) == 0xFF000000
    assert pk(1, 1, 1) == 0xFFFFFFFF
    assert pk(2, -1, float("nan")) == 0xFF0000FF          # clamp high, clamp low, NaN -> 0 ; R in the low byte
    assert pk(float("inf"), float("-inf"), 0.5) & 0xFFFF == 0x00FF
    # round half to even: 0.5/255 -> 0, 1.5/255 -> 2, 2.5/255 -> 2
    assert pk(0.5 / 255, 1.5 / 255, 2.5 / 255) & 0xFFFFFF == (2 << 16) | (2 << 8) | 0
    out = np.zeros(3, dtype=np.float32)
    orc.lib().orc_kat_unpack_color(0xFF804020, _p(out))
    assert tuple(out) == (np.float32(0x20) / np.float32(255), np.float32(0x40) / np.float32(255), np.float32(0x80) / np.float32(255))


def test_xna_host_math_mirror_matches_oracle(xrt, orc):
    """The Python host mirror (Camera / SceneObject matrices) and the oracle's C++ restatement are two
    independent implementations of the same XNA formulas: they must agree bit for bit."""
    L = orc.lib()
    out = np.zeros(16, dtype=np.float32)
    for pos, tgt in (((0, 32, 64), (0, 8, 0)), ((0, 120, 260), (0, 0, 0)), ((3.5, -2.25, 7.125), (1, 2, 3))):
        L.orc_kat_look_at(_p(fa(*pos)), _p(fa(*tgt)), _p(fa(0, 1, 0)), _p(out))
        assert np.array_equal(out.view(np.uint32), xrt.xna.as_array(xrt.xna.create_look_at(pos, tgt, (0, 1, 0))).view(np.uint32))
    for fov, asp in ((math.pi / 4, 1.0), (math.pi / 4, 1920 / 1080), (1.1, 0.75)):
        L.orc_kat_perspective(fov, asp, 1.0, 1000.0, _p(out))
        assert np.array_equal(out.view(np.uint32), xrt.xna.as_array(xrt.xna.create_perspective_fov(fov, asp, 1.0, 1000.0)).view(np.uint32))
    rng = np.random.default_rng(0)
    for _ in range(20):
        a = rng.normal(size=16).astype(np.float32)
        b = rng.normal(size=16).astype(np.float32)
        L.orc_kat_multiply(_p(a), _p(b), _p(out))
        assert np.array_equal(out.view(np.uint32), xrt.xna.as_array(xrt.xna.multiply(list(a), list(b))).view(np.uint32))
        L.orc_kat_invert(_p(a), _p(out))
        assert np.array_equal(out.view(np.uint32), xrt.xna.as_array(xrt.xna.invert(list(a))).view(np.uint32))
    w, iw, wbb = np.zeros(16, np.float32), np.zeros(16, np.float32), np.zeros(6, np.float32)
    bbox = fa(-1, 0, -2, 3, 4, 5)
    for scale, rot, pos in (((1, 1, 1), (0, 0, 0), (5, 0, -7)), ((2, 0.5, 1.5), (0.3, -1.2, 2.0), (1, 2, 3))):
        L.orc_kat_build_world(_p(fa(*scale)), _p(fa(*rot)), _p(fa(*pos)), _p(bbox), _p(w), _p(iw), _p(wbb))
        W, IW, WBB = xrt.xna.build_world(scale, rot, pos, bbox)
        assert np.array_equal(w.view(np.uint32), xrt.xna.as_array(W).view(np.uint32))
        assert np.array_equal(iw.view(np.uint32), xrt.xna.as_array(IW).view(np.uint32))
        assert np.array_equal(wbb.view(np.uint32), xrt.xna.as_array(WBB).view(np.uint32))


def test_spot_light_known_values(xrt, orc):
    from oracle.oracle_py import light_abi
    l = light_abi(xrt.configs.spot((0, 10, 0)))
    out = np.zeros(3, dtype=np.float32)
    # fragment straight below the light, facing it: surfaceDot = 1, lightDot = 1
    orc.lib().orc_kat_spot_light(C.byref(l), _p(fa(0, 0, 0)), _p(fa(0, 1, 0)), _p(out))
    cosA = np.float32(math.cos(float(np.float32(math.pi / 2) * np.float32(0.5))))
    spot = np.float32((float(np.float32(1) - cosA)) / math.pow(float(np.float32(1) - cosA), float(np.float32(1.3))))
    assert out[0] == spot * np.float32(1) + np.float32(1)
    # facing away -> zero (SPOT:45-48); outside the cone -> zero (SPOT:52,57-60)
    orc.lib().orc_kat_spot_light(C.byref(l), _p(fa(0, 0, 0)), _p(fa(0, -1, 0)), _p(out))
    assert tuple(out) == (0, 0, 0)
    orc.lib().orc_kat_spot_light(C.byref(l), _p(fa(100, 0, 0)), _p(fa(0, 1, 0)), _p(out))
    assert tuple(out) == (0, 0, 0)


def test_texture_lookup_wrap_point(xrt, orc):
    tex = (np.arange(16, dtype=np.uint32).reshape(4, 4) * 0x010203 + 0xFF000000).astype(np.uint32)
    m = xrt.abi.xrt_material()
    m.use_texture, m.tex_width, m.tex_height = 1, 4, 4
    m.tex_argb = tex.ctypes.data_as(C.POINTER(C.c_uint32))
    out = np.zeros(3, dtype=np.float32)

    def look(u, v, mode=xrt.abi.ADDRESS_WRAP):
        assert orc.lib().orc_kat_lookup_uv(C.byref(m), _p(fa(u, v)), mode, xrt.abi.FILTER_POINT, _p(out)) == 0
        return tuple(out)

    def texel(x, y):
        w = int(tex[y, x])
        k = np.float32(1.0) / np.float32(255.0)
        return (np.float32((w >> 16) & 255) * k, np.float32((w >> 8) & 255) * k, np.float32(w & 255) * k)

    assert look(0, 0) == texel(0, 0)
    assert look(1, 1) == texel(3, 3)                  # x = (int)(u * (W - 1)) (MAT:147)
    assert look(0.5, 0.99) == texel(1, 2)
    assert look(1.25, -0.25) == look(0.25, 0.75)      # wrap (MAT:127-135)
    assert look(7.0, 0) == texel(0, 0)                # 7 % 1 == 0
    assert look(1.5, 2.0, xrt.abi.ADDRESS_CLAMP) == texel(3, 3)
    # bad enum -> the C# throws ArgumentException (MAT:85,97)
    assert orc.lib().orc_kat_lookup_uv(C.byref(m), _p(fa(0, 0)), 7, xrt.abi.FILTER_POINT, _p(out)) == -1
    assert orc.lib().orc_kat_lookup_uv(C.byref(m), _p(fa(0, 0)), xrt.abi.ADDRESS_WRAP, 5, _p(out)) == -1
    # bilinear (MAT:162-232) on a 2x2 texture {black, red / green, blue}: at uv (0.5, 0.5) IEEERemainder(0.5, 0.5) = 0,
    # dx = dy = 0.5, the four texels blend with weight 1/4 each -> 63.75 / 255 = 0.25 per channel (hand computed)
    t2 = np.array([[0xFF000000, 0xFFFF0000], [0xFF00FF00, 0xFF0000FF]], dtype=np.uint32)
    m2 = xrt.abi.xrt_material()
    m2.use_texture, m2.tex_width, m2.tex_height = 1, 2, 2
    m2.tex_argb = t2.ctypes.data_as(C.POINTER(C.c_uint32))
    assert orc.lib().orc_kat_lookup_uv(C.byref(m2), _p(fa(0.5, 0.5)), xrt.abi.ADDRESS_WRAP, xrt.abi.FILTER_BILINEAR, _p(out)) == 0
    assert tuple(out) == (0.25, 0.25, 0.25)
    # at uv (1, 1): remainder 0, x = x2 = 1 -> 0.25 * 4 * blue
    assert orc.lib().orc_kat_lookup_uv(C.byref(m2), _p(fa(1.0, 1.0)), xrt.abi.ADDRESS_WRAP, xrt.abi.FILTER_BILINEAR, _p(out)) == 0
    assert tuple(out) == (0.0, 0.0, 1.0)


def test_k10_golden_c1_frame(xrt, orc):
    """The committed C1 frame (256x256, R=0) is reproduced exactly by the oracle."""
    spec = xrt.configs.config("C1")
    rgba, rgbf, st = orc.OracleScene(spec).render()
    gold = np.load(os.path.join(GOLDEN, "c1_rgba.npy"))
    assert np.array_equal(rgba, gold)
    assert st["rays_closest"] == 256 * 256 and st["rays_shadow"] == st["hits_closest"] > 10000
    # miss colour is opaque black (RT:732)
    assert rgba[0] == 0xFF000000


def test_golden_hit_vectors(xrt, orc):
    for name, spec in (("c3", xrt.configs.crate_grid_scene(160, 90)), ("h224", xrt.configs.heightfield_scene(160, 90, m=224))):
        rays = np.load(os.path.join(GOLDEN, name + "_rays.npy"))
        gold = np.load(os.path.join(GOLDEN, name + "_hits.npy"))
        assert orc.OracleScene(spec).intersect(rays).tobytes() == gold.tobytes()


def test_oracle_threads_and_rows_are_consistent(xrt, orc):
    spec = xrt.configs.crate_grid_scene(64, 36)
    o = orc.OracleScene(spec)
    a, af, sa = o.render(nthreads=1)
    b, bf, sb = o.render(nthreads=4)
    assert np.array_equal(a, b) and np.array_equal(af.view(np.uint32), bf.view(np.uint32))
    for k in ("rays_closest", "rays_shadow", "node_tests", "tri_tests", "algorithmic_bytes"):
        assert sa[k] == sb[k]
    c, _, sc = o.render(rows=(10, 20))
    assert np.array_equal(c.reshape(36, 64)[10:20], a.reshape(36, 64)[10:20]) and sc["pixels"] == 640


def test_multisample_modes(xrt, orc):
    # the faithful adaptive mode (RT:215-311, incl. the RT:305 bug) and the fixed 16-position mode
    spec = xrt.configs.heightfield_scene(24, 16, m=64, multisampling=xrt.abi.MS_FIXED16)
    o = orc.OracleScene(spec)
    rgba16, _, st16 = o.render()
    assert st16["rays_closest"] >= 16 * 24 * 16
    spec.multisampling, spec.multisample_quality = xrt.abi.MS_ADAPTIVE, 1
    o2 = orc.OracleScene(spec)
    rgbaA, _, stA = o2.render()
    primary = 4 * 24 * 16
    assert stA["rays_closest"] >= primary
    spec.multisample_quality = 0   # no subdivision: exactly 4 primary rays per pixel
    rgba4, _, st4 = orc.OracleScene(spec).render()
    spec.multisampling = xrt.abi.MS_OFF
    rgba1, _, st1 = orc.OracleScene(spec).render()
    assert st4["rays_closest"] - 4 * 24 * 16 <= 4 * (st1["rays_closest"] - 24 * 16) + 4 * 24 * 16
