"""Deterministic fixture geometry and the five BASELINE.json configs.

No RNG, no libm in geometry: everything is float32 add/mul (numpy, one rounding per op), so the same
arrays feed the CPU oracle and the HIP library.

The crate is the data of the reference asset RayTraceProject/RayTraceProjectContent/Crate_Fragile.FBX
(8 vertices, 6 quads, per-polygon-vertex normals and UVs), imported with the build's definition of the
closed-source XNA importer (Z-up -> Y-up, x0.6 scale, fan triangulation, clockwise winding so that
TracerModelProcessor's surfaceNormal = normalize(cross(v3-v1, v2-v1)) (TMP:199-203) points outward).
"""
import os
import struct
import numpy as np

f32 = np.float32

# ---- Crate_Fragile.FBX, parsed as data -------------------------------------------------------------
_A = 19.685039520263672
_H = 39.370079040527344
CRATE_VERTICES = [(-_A, -_A, 0.0), (-_A, -_A, _H), (_A, -_A, _H), (_A, -_A, 0.0),
                  (_A, _A, 0.0), (_A, _A, _H), (-_A, _A, _H), (-_A, _A, 0.0)]
CRATE_QUADS = [(3, 2, 1, 0), (7, 6, 5, 4), (0, 1, 6, 7), (4, 5, 2, 3), (2, 5, 6, 1), (4, 3, 0, 7)]
CRATE_NORMALS = [(0, -1, 0), (0, 1, 0), (-1, 0, 0), (1, 0, 0), (0, 0, 1), (0, 0, -1)]
CRATE_UVS = [(1.0, 0.0), (1.0, 1.0), (0.0, 1.0), (0.0, 0.0)]
CRATE_SCALE = 0.6   # RayTraceProjectContent.contentproj:65-70


def _yup(p):
    """Z-up -> Y-up: (x, y, z) -> (x, z, -y)."""
    return (p[0], p[2], -p[1])


def surface_normals(v):
    """TMP:199-203: normalize(cross(v3 - v1, v2 - v1)) in float32. v: (n,3,3)."""
    e1 = v[:, 1] - v[:, 0]
    e2 = v[:, 2] - v[:, 0]
    a, b = e2, e1
    cx = a[:, 1] * b[:, 2] - a[:, 2] * b[:, 1]
    cy = a[:, 2] * b[:, 0] - a[:, 0] * b[:, 2]
    cz = a[:, 0] * b[:, 1] - a[:, 1] * b[:, 0]
    num = (cx * cx + cy * cy) + cz * cz
    inv = f32(1.0) / np.sqrt(num.astype(np.float64)).astype(np.float32)
    return np.stack([cx * inv, cy * inv, cz * inv], axis=1).astype(np.float32)


def mesh_bbox(v):
    """TracerModelProcessor.CreateBoundingBox (TMP:244-307): starts from new BoundingBox() = origin."""
    pts = v.reshape(-1, 3)
    if pts.shape[0] == 0:
        return np.zeros(6, dtype=np.float32)
    mn = np.minimum(pts.min(axis=0), f32(0.0))
    mx = np.maximum(pts.max(axis=0), f32(0.0))
    return np.concatenate([mn, mx]).astype(np.float32)


class MeshData:
    """Plain arrays of one Mesh (TRI:14-25 fields the path reads)."""

    def __init__(self, v, n, uv, color):
        self.v = np.ascontiguousarray(v, dtype=np.float32)          # (ntri,3,3)
        self.n = np.ascontiguousarray(n, dtype=np.float32)          # (ntri,3,3)
        self.uv = np.ascontiguousarray(uv, dtype=np.float32)        # (ntri,3,2)
        self.color = np.ascontiguousarray(color, dtype=np.float32)  # (ntri,4)
        self.surface_normal = surface_normals(self.v)               # (ntri,3)
        self.bbox = mesh_bbox(self.v)

    @property
    def ntri(self):
        return self.v.shape[0]


def crate(n=1):
    """Crate tessellated n x n cells per face, 2 triangles per cell -> 12 n^2 triangles.
    n = 1 is the FBX mesh itself."""
    s = f32(CRATE_SCALE)
    verts = [tuple(f32(c) * s for c in _yup(tuple(f32(x) for x in p))) for p in CRATE_VERTICES]
    fn = f32(n)
    tris, nrm, uvs = [], [], []
    for q, quad in enumerate(CRATE_QUADS):
        p0, p1, p2, p3 = (verts[i] for i in quad)
        normal = tuple(f32(c) for c in _yup(CRATE_NORMALS[q]))
        uvq = [(f32(u), f32(1.0) - f32(w)) for (u, w) in CRATE_UVS]   # v' = 1 - v

        def canon(lo, hi, m):
            if m == 0:
                return lo
            if m == n:
                return hi
            return lo + (hi - lo) * (f32(m) / fn)

        def grid(i, j):
            out = []
            for k in range(3):
                a0, a1, a3 = p0[k], p1[k], p3[k]
                if a1 != a0:
                    lo, hi = min(a0, a1), max(a0, a1)
                    out.append(canon(lo, hi, i if a0 == lo else n - i))
                elif a3 != a0:
                    lo, hi = min(a0, a3), max(a0, a3)
                    out.append(canon(lo, hi, j if a0 == lo else n - j))
                else:
                    out.append(a0)
            return tuple(out)

        def guv(i, j):
            u0, u1, u3 = uvq[0], uvq[1], uvq[3]
            fi, fj = f32(i) / fn, f32(j) / fn
            return (u0[0] + (u1[0] - u0[0]) * fi + (u3[0] - u0[0]) * fj,
                    u0[1] + (u1[1] - u0[1]) * fi + (u3[1] - u0[1]) * fj)

        for i in range(n):
            for j in range(n):
                q0, q1, q2, q3 = grid(i, j), grid(i + 1, j), grid(i + 1, j + 1), grid(i, j + 1)
                t0, t1, t2, t3 = guv(i, j), guv(i + 1, j), guv(i + 1, j + 1), guv(i, j + 1)
                # fan (q0,q1,q2),(q0,q2,q3) reversed to clockwise
                tris += [(q0, q2, q1), (q0, q3, q2)]
                uvs += [(t0, t2, t1), (t0, t3, t2)]
                nrm += [(normal,) * 3, (normal,) * 3]
    ntri = len(tris)
    color = np.ones((ntri, 4), dtype=np.float32)
    return MeshData(np.array(tris, dtype=np.float32), np.array(nrm, dtype=np.float32),
                    np.array(uvs, dtype=np.float32), color)


def heightfield(m):
    """(m+1)^2 grid over x,z in [-50,50], y = 4 p(x/50) p(z/50), p(t) = t (1 - t^2) 2.598; two
    clockwise-from-above triangles per cell -> 2 m^2 triangles (m=224: 100,352; m=707: 999,698)."""
    idx = np.arange(m + 1, dtype=np.float32)
    c = f32(-50.0) + f32(100.0) * (idx / f32(m))
    c[0], c[m] = f32(-50.0), f32(50.0)
    t = c / f32(50.0)
    p = (t * (f32(1.0) - t * t)) * f32(2.598)
    y = (f32(4.0) * p[:, None]) * p[None, :]          # y[i (x), j (z)]
    X = np.broadcast_to(c[:, None], (m + 1, m + 1))
    Z = np.broadcast_to(c[None, :], (m + 1, m + 1))
    P = np.stack([X, y, Z], axis=-1).astype(np.float32)   # (m+1, m+1, 3)
    a = P[:-1, :-1]
    b = P[1:, :-1]
    cc = P[1:, 1:]
    d = P[:-1, 1:]
    t1 = np.stack([a, b, cc], axis=2)   # (m, m, 3, 3)
    t2 = np.stack([a, cc, d], axis=2)
    v = np.stack([t1, t2], axis=2).reshape(-1, 3, 3)   # cell-major, 2 tris per cell
    ntri = v.shape[0]
    sn = surface_normals(np.ascontiguousarray(v, dtype=np.float32))
    n = np.repeat(sn[:, None, :], 3, axis=1)
    uv = np.zeros((ntri, 3, 2), dtype=np.float32)
    i = np.arange(ntri, dtype=np.uint64)
    word = (i * np.uint64(2654435761)) % np.uint64(1 << 24)
    r = ((word >> np.uint64(16)) & np.uint64(255)).astype(np.float32) / f32(255.0)
    g = ((word >> np.uint64(8)) & np.uint64(255)).astype(np.float32) / f32(255.0)
    bl = (word & np.uint64(255)).astype(np.float32) / f32(255.0)
    color = np.stack([r, g, bl, np.ones(ntri, dtype=np.float32)], axis=1)
    return MeshData(v, n, uv, color)


# ---- texture ---------------------------------------------------------------------------------------------
def load_bmp_argb(path):
    """24-bpp uncompressed BMP -> top-down 0xFFRRGGBB words (what Bitmap.LockBits(Format32bppArgb) exposes,
    MAT:63-65; GDI+ behaviour is the build's definition)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:2] != b"BM":
        raise ValueError("not a BMP")
    off = struct.unpack_from("<I", data, 10)[0]
    w, h = struct.unpack_from("<ii", data, 18)
    bpp = struct.unpack_from("<H", data, 28)[0]
    comp = struct.unpack_from("<I", data, 30)[0]
    if bpp != 24 or comp != 0:
        raise ValueError("only 24-bpp uncompressed BMP")
    stride = (w * 3 + 3) & ~3
    rows = np.frombuffer(data, dtype=np.uint8, count=stride * abs(h), offset=off).reshape(abs(h), stride)[:, : w * 3]
    bgr = rows.reshape(abs(h), w, 3).astype(np.uint32)
    if h > 0:
        bgr = bgr[::-1]   # stored bottom-up
    argb = np.uint32(0xFF000000) | (bgr[..., 2] << np.uint32(16)) | (bgr[..., 1] << np.uint32(8)) | bgr[..., 0]
    return np.ascontiguousarray(argb, dtype=np.uint32)


_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
CRATE_TEXTURE = os.path.join(_GOLDEN, "Free_crate_Diffuse.bmp")


def crate_texture():
    """Free_crate/Diffuse.bmp (512x512x24) committed as a data fixture; a procedural 512x512 stand-in of
    the same shape is used if the file is absent."""
    if os.path.exists(CRATE_TEXTURE):
        return load_bmp_argb(CRATE_TEXTURE)
    yy, xx = np.mgrid[0:512, 0:512].astype(np.uint32)
    r = (xx * 7 + yy * 3) & 255
    g = (xx ^ yy) & 255
    b = (xx * yy >> 3) & 255
    return (np.uint32(0xFF000000) | (r << 16) | (g << 8) | b).astype(np.uint32)
