"""Asset ingestion: FBX 6.x (ASCII 6.1 as written by Blender, binary 6100 as written by 3ds Max) and binary 7.x (7100-7400
with 32-bit record offsets, 7500 and later with 64-bit ones) -> the triangle arrays a `Mesh` carries, with the semantics of RayTracePipeline/TracerModelProcessor.cs:
vertices are baked with the node's absolute transform (TMP:179-181), normals with its inverse transpose and
re-normalised (TMP:191-197), surfaceNormal = normalize(cross(v3-v1, v2-v1)) (TMP:199-203), the mesh AABB starts at
the origin (TMP:244-307), `Scale` is the ModelProcessor parameter of the .contentproj.

The XNA FbxImporter itself is closed source; the behaviour fixed here is the build's definition:
node transform = Scaling * RotX * RotY * RotZ * Translation in XNA's row-vector convention (FBX eEulerXYZ), Z-up
files are turned Y-up by (x, y, z) -> (x, z, -y), polygons are fan-triangulated and their winding reversed to
clockwise so that surfaceNormal agrees with the exported vertex normals.  All arithmetic in binary32.
"""
import re
import struct

import numpy as np

from . import xna
from .fixtures import MeshData

f32 = np.float32


class FbxMesh:
    def __init__(self, name):
        self.name = name
        self.vertices = None          # (n, 3) float64 as written in the file
        self.polygons = []            # lists of vertex indices
        self.normals = None           # (m, 3)
        self.normal_mapping = None    # "ByVertice" | "ByPolygonVertex"
        self.uvs = None               # (k, 2)
        self.uv_index = None
        self.uv_mapping = None
        self.colors = None            # (c, 4) vertex colour channel (LayerElementColor), or None
        self.color_index = None
        self.color_mapping = None     # "ByVertice" | "ByPolygonVertex"
        self.translation = (0.0, 0.0, 0.0)
        self.rotation = (0.0, 0.0, 0.0)   # degrees
        self.scaling = (1.0, 1.0, 1.0)


def _split_polygons(idx):
    polys, cur = [], []
    for i in idx:
        if i < 0:
            cur.append(~i)
            polys.append(cur)
            cur = []
        else:
            cur.append(i)
    return polys


def _numbers(text):
    return [float(x) for x in re.findall(r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?", text)]


def _ascii_block(text, start):
    """text[start] is just after an opening '{'; returns the index of the matching '}'."""
    depth, i = 1, start
    while depth and i < len(text):
        c = text[i]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
        elif c == '"':
            i = text.index('"', i + 1)
        i += 1
    return i - 1


def _ascii_array(body, key):
    m = re.search(r"\b" + key + r":\s*([^A-Za-z{}\"]*)", body)
    return _numbers(m.group(1)) if m else None


def load_ascii(text):
    up = re.search(r'Property:\s*"UpAxis",\s*"int",\s*"",\s*(-?\d+)', text)
    up_axis = int(up.group(1)) if up else 1
    meshes = []
    for m in re.finditer(r'Model:\s*"Model::([^"]*)",\s*"Mesh"\s*\{', text):
        end = _ascii_block(text, m.end())
        body = text[m.end():end]
        fm = FbxMesh(m.group(1))
        for prop, attr in (("Lcl Translation", "translation"), ("Lcl Rotation", "rotation"), ("Lcl Scaling", "scaling")):
            pm = re.search(r'Property:\s*"' + prop + r'",\s*"[^"]*",\s*"[^"]*",\s*([^\n]*)', body)
            if pm:
                setattr(fm, attr, tuple(_numbers(pm.group(1))[:3]))
        verts, pvi = _ascii_array(body, "Vertices"), _ascii_array(body, "PolygonVertexIndex")
        if not verts or not pvi:
            continue   # a Model of type Mesh without geometry
        fm.vertices = np.array(verts, dtype=np.float64).reshape(-1, 3)
        fm.polygons = _split_polygons([int(x) for x in pvi])
        nm = re.search(r"LayerElementNormal:\s*\d+\s*\{", body)
        if nm:
            nb = body[nm.end():_ascii_block(body, nm.end())]
            fm.normal_mapping = re.search(r'MappingInformationType:\s*"([^"]*)"', nb).group(1)
            fm.normals = np.array(_ascii_array(nb, "Normals"), dtype=np.float64).reshape(-1, 3)
        um = re.search(r"LayerElementUV:\s*\d+\s*\{", body)
        if um:
            ub = body[um.end():_ascii_block(body, um.end())]
            fm.uv_mapping = re.search(r'MappingInformationType:\s*"([^"]*)"', ub).group(1)
            fm.uvs = np.array(_ascii_array(ub, "UV"), dtype=np.float64).reshape(-1, 2)
            ui = _ascii_array(ub, "UVIndex")
            fm.uv_index = [int(x) for x in ui] if ui else None
        cm = re.search(r"LayerElementColor:\s*\d+\s*\{", body)
        if cm:
            cb = body[cm.end():_ascii_block(body, cm.end())]
            cols = _ascii_array(cb, "Colors")
            if cols:
                fm.color_mapping = re.search(r'MappingInformationType:\s*"([^"]*)"', cb).group(1)
                fm.colors = np.array(cols, dtype=np.float64).reshape(-1, 4)
                ci = _ascii_array(cb, "ColorIndex")
                fm.color_index = [int(x) for x in ci] if ci else None
        meshes.append(fm)
    return meshes, up_axis


def _binary_values(b, pos, limit=1 << 20):
    vals, j = [], pos
    while j < len(b) and len(vals) < limit:
        t = b[j:j + 1]
        if t == b"D":
            vals.append(struct.unpack_from("<d", b, j + 1)[0]); j += 9
        elif t == b"I":
            vals.append(struct.unpack_from("<i", b, j + 1)[0]); j += 5
        elif t == b"F":
            vals.append(struct.unpack_from("<f", b, j + 1)[0]); j += 5
        elif t == b"S" and j + 5 <= len(b) and struct.unpack_from("<I", b, j + 1)[0] < 256:   # string property: skipped
            j += 5 + struct.unpack_from("<I", b, j + 1)[0]
        else:
            break
    return vals


def load_binary(b):
    """Binary FBX 6100: node properties are typed scalars ('D' double, 'I' int32, 'F' float) following the node
    name; enough to read the mesh arrays of a single-mesh file such as Crate_Fragile.FBX."""
    def after(name, nth=0, min_len=1):
        found = 0
        for m in re.finditer(re.escape(name), b):
            v = _binary_values(b, m.end())
            if len(v) >= min_len:
                if found == nth:
                    return v
                found += 1
        return None
    fm = FbxMesh("mesh")
    fm.vertices = np.array(after(b"Vertices", min_len=9), dtype=np.float64).reshape(-1, 3)
    fm.polygons = _split_polygons([int(x) for x in after(b"PolygonVertexIndex", min_len=3)])
    nv = after(b"Normals", min_len=9)
    if nv:
        fm.normals = np.array(nv, dtype=np.float64).reshape(-1, 3)
        fm.normal_mapping = "ByPolygonVertex" if len(fm.normals) == sum(len(p) for p in fm.polygons) else "ByVertice"
    uv = after(b"UV", min_len=8)
    if uv:
        fm.uvs = np.array(uv, dtype=np.float64).reshape(-1, 2)
        ui = after(b"UVIndex", min_len=3)
        fm.uv_index = [int(x) for x in ui] if ui else None
        fm.uv_mapping = "ByPolygonVertex"
    t = after(b"Lcl Translation", min_len=3)
    if t:
        fm.translation = tuple(t[:3])
    up = after(b"UpAxis")
    return [fm], (int(up[0]) if up else 1)


# ---- binary FBX 7.x (7100-7400: 32-bit record offsets; 7500+: 64-bit) ---------------------------------------------------------------------
# Header: 21-byte magic, 2 bytes, uint32 version.  Node record: uint32 endOffset, numProperties, propertyListLen; uint8 nameLen; name;
# properties; nested records up to endOffset (a 13-byte zero record closes a list).  Property: a type byte, then Y int16, C bool,
# I int32, F float, D double, L int64; S / R: uint32 length + bytes; arrays f d l i b: uint32 count, encoding (1 = zlib), byte
# length, data.
class _Node:
    __slots__ = ("name", "props", "children")

    def __init__(self, name, props, children):
        self.name, self.props, self.children = name, props, children

    def find(self, name):
        return [c for c in self.children if c.name == name]

    def first(self, name):
        r = self.find(name)
        return r[0] if r else None


def _read_prop7(b, pos):
    import zlib
    t = b[pos:pos + 1]
    pos += 1
    scalar = {b"Y": "<h", b"C": "<?", b"I": "<i", b"F": "<f", b"D": "<d", b"L": "<q"}
    if t in scalar:
        fmt = scalar[t]
        return struct.unpack_from(fmt, b, pos)[0], pos + struct.calcsize(fmt)
    if t in (b"S", b"R"):
        n = struct.unpack_from("<I", b, pos)[0]
        raw = bytes(b[pos + 4:pos + 4 + n])
        return (raw.decode("latin-1") if t == b"S" else raw), pos + 4 + n
    arr = {b"f": "<f4", b"d": "<f8", b"l": "<i8", b"i": "<i4", b"b": "u1"}
    if t in arr:
        count, enc, nbytes = struct.unpack_from("<III", b, pos)
        raw = bytes(b[pos + 12:pos + 12 + nbytes])
        if enc == 1:
            raw = zlib.decompress(raw)
        return np.frombuffer(raw, dtype=arr[t], count=count), pos + 12 + nbytes
    raise ValueError("FBX 7: unknown property type %r at %d" % (t, pos - 1))


def _read_node7(b, pos, wide=False):
    """One node record; `wide`: FBX >= 7500, whose three header fields are uint64 (a 25-byte header and null record instead of 13)."""
    head = 25 if wide else 13
    end, nprops, _plen = struct.unpack_from("<QQQ" if wide else "<III", b, pos)
    nlen = b[pos + head - 1]
    if end == 0:
        return None, pos + head
    if end > len(b) or end <= pos:
        raise ValueError("FBX node record at %d ends at %d in a file of %d bytes" % (pos, end, len(b)))
    name = bytes(b[pos + head:pos + head + nlen]).decode("latin-1")
    p = pos + head + nlen
    props = []
    for _ in range(nprops):
        v, p = _read_prop7(b, p)
        props.append(v)
    children = []
    while p < end:
        c, p = _read_node7(b, p, wide)
        if c is None:
            break
        children.append(c)
    return _Node(name, props, children), end


def load_binary7(b):
    """Binary FBX 7.x: Objects/Geometry (Vertices, PolygonVertexIndex, LayerElementNormal, LayerElementUV, LayerElementColor), the Model each
    geometry is connected to (Connections "OO") for its Lcl transform, GlobalSettings/UpAxis."""
    version = struct.unpack_from("<I", b, 23)[0]
    wide = version >= 7500
    pos, top = 27, []
    while pos < len(b) - (25 if wide else 13):
        n, pos = _read_node7(b, pos, wide)
        if n is None:
            break
        top.append(n)
    root = _Node("", [], top)

    def p70(node, key, default):
        pr = node.first("Properties70") if node is not None else None
        if pr is not None:
            for p in pr.find("P"):
                if p.props and p.props[0] == key:
                    return [x for x in p.props[4:]]
        return default
    gs = root.first("GlobalSettings")
    up = int(p70(gs, "UpAxis", [1])[0])
    objects = root.first("Objects")
    models = {m.props[0]: m for m in objects.find("Model")} if objects is not None else {}
    parent = {}
    conns = root.first("Connections")
    if conns is not None:
        for c in conns.find("C"):
            if c.props and c.props[0] == "OO":
                parent[c.props[1]] = c.props[2]
    meshes = []
    for g in (objects.find("Geometry") if objects is not None else []):
        verts, pvi = g.first("Vertices"), g.first("PolygonVertexIndex")
        if verts is None or pvi is None:
            continue
        fm = FbxMesh(str(g.props[1]).split("\x00")[0] if len(g.props) > 1 else "mesh")   # ("name\0\1Geometry")
        fm.vertices = np.asarray(verts.props[0], dtype=np.float64).reshape(-1, 3)
        fm.polygons = _split_polygons([int(x) for x in pvi.props[0]])
        ln = g.first("LayerElementNormal")
        if ln is not None and ln.first("Normals") is not None:
            fm.normals = np.asarray(ln.first("Normals").props[0], dtype=np.float64).reshape(-1, 3)
            fm.normal_mapping = ln.first("MappingInformationType").props[0]
            ni = ln.first("NormalsIndex")
            if ni is not None and ln.first("ReferenceInformationType").props[0] == "IndexToDirect":
                fm.normals = fm.normals[np.asarray(ni.props[0], dtype=np.int64)]
        lu = g.first("LayerElementUV")
        if lu is not None and lu.first("UV") is not None:
            fm.uvs = np.asarray(lu.first("UV").props[0], dtype=np.float64).reshape(-1, 2)
            fm.uv_mapping = lu.first("MappingInformationType").props[0]
            ui = lu.first("UVIndex")
            fm.uv_index = [int(x) for x in ui.props[0]] if ui is not None else None
        lc = g.first("LayerElementColor")
        if lc is not None and lc.first("Colors") is not None:
            fm.colors = np.asarray(lc.first("Colors").props[0], dtype=np.float64).reshape(-1, 4)
            fm.color_mapping = lc.first("MappingInformationType").props[0]
            ci = lc.first("ColorIndex")
            fm.color_index = [int(x) for x in ci.props[0]] if (ci is not None and lc.first("ReferenceInformationType").props[0] == "IndexToDirect") else None
        model = models.get(parent.get(g.props[0]))
        fm.translation = tuple(float(x) for x in p70(model, "Lcl Translation", [0.0, 0.0, 0.0])[:3])
        fm.rotation = tuple(float(x) for x in p70(model, "Lcl Rotation", [0.0, 0.0, 0.0])[:3])
        fm.scaling = tuple(float(x) for x in p70(model, "Lcl Scaling", [1.0, 1.0, 1.0])[:3])
        meshes.append(fm)
    return meshes, up


def load_fbx(path):
    data = open(path, "rb").read()
    if data.startswith(b"Kaydara FBX Binary"):
        if struct.unpack_from("<I", data, 23)[0] >= 7000:
            return load_binary7(data)
        return load_binary(data)
    return load_ascii(data.decode("latin-1"))


def xna_color_bytes(rgba):
    """`new Color(Vector4)` of XNA 4: every component times 255, clamped to [0, 255], Math.Round to nearest-even (PackUtils.PackUNorm)."""
    out = []
    for c in rgba:
        v = f32(c) * f32(255.0)
        v = f32(0.0) if not (v == v) else min(max(v, f32(0.0)), f32(255.0))
        out.append(int(np.rint(np.float64(v))))
    return tuple(out)


def import_mesh(fm, up_axis=1, scale=1.0, diffuse_color=(255, 255, 255, 255), apply_node_transform=True, flip_v=True,
                rotation=(0.0, 0.0, 0.0), use_vertex_colors=False):
    """FbxMesh -> fixtures.MeshData with TracerModelProcessor semantics (see module docstring).

    `scale` and `rotation` (degrees) are the ModelProcessor parameters Scale / RotationX / RotationY / RotationZ of the content
    project (contentproj:116,190,203,224): the base processor transforms the whole scene before TracerModelProcessor bakes the
    absolute transforms into the vertices (TMP:105-107,179-181).  XNA's ModelProcessor is closed source; the build's definition:
    scene transform = CreateScale(Scale) * CreateRotationX * CreateRotationY * CreateRotationZ, applied after the importer's
    axis conversion; normals take the rotation part (its inverse transpose is itself).

    `use_vertex_colors` = the processor parameter UseVertexColors (TMP:93-101): when the geometry has a colour channel, a triangle's colour is
    the channel's value at the triangle's FIRST index, as an XNA `Color` (quantised to bytes) turned back into a Vector4 (TMP:224-225);
    otherwise DiffuseColor (TMP:227)."""
    if apply_node_transform:
        rad = [f32(np.deg2rad(float(a))) for a in fm.rotation]
        world, _, _ = xna.build_world(fm.scaling, rad, fm.translation, np.zeros(6, dtype=np.float32))
    else:
        world = xna.identity()
    # normals: transpose(invert(absoluteTransform)) (TMP:141)
    inv = xna.invert(world)
    invT = [inv[4 * j + i] for i in range(4) for j in range(4)]
    s = f32(scale)
    rotated = any(float(a) != 0.0 for a in rotation)
    rot = None
    if rotated:   # CreateRotationX * CreateRotationY * CreateRotationZ (MathHelper.ToRadians of the parameters), no translation, unit scale
        rot, _, _ = xna.build_world((1.0, 1.0, 1.0), [f32(np.deg2rad(float(a))) for a in rotation], (0.0, 0.0, 0.0), np.zeros(6, dtype=np.float32))

    def pos(p):
        v = xna.transform(xna.vec3(*[float(x) for x in p]), world)
        if up_axis == 2:
            v = [v[0], v[2], -v[1]]
        v = [c * s for c in v]
        if rotated:
            v = xna.transform(xna.vec3(*[float(c) for c in v]), rot)
        return tuple(float(c) for c in v)

    def nrm(nv):
        v = xna.transform(xna.vec3(*[float(x) for x in nv]), invT)
        if up_axis == 2:
            v = [v[0], v[2], -v[1]]
        if rotated:
            v = xna.transform(xna.vec3(*[float(c) for c in v]), rot)
        return tuple(float(c) for c in xna.normalize(v))
    tris, nrms, uvs, cols = [], [], [], []
    vertex_colors = use_vertex_colors and fm.colors is not None
    pv = 0   # polygon-vertex counter
    for poly in fm.polygons:
        corner = []
        for k, vi in enumerate(poly):
            n = (0.0, 0.0, 0.0)
            if fm.normals is not None:
                n = nrm(fm.normals[pv + k] if fm.normal_mapping == "ByPolygonVertex" else fm.normals[vi])
            uv = (0.0, 0.0)
            if fm.uvs is not None:
                ui = fm.uv_index[pv + k] if fm.uv_index is not None else pv + k
                u, w = fm.uvs[ui]
                uv = (float(f32(u)), float(f32(1.0) - f32(w)) if flip_v else float(f32(w)))
            rgba = None
            if vertex_colors:
                at = pv + k if fm.color_mapping == "ByPolygonVertex" else vi
                rgba = fm.colors[fm.color_index[at] if fm.color_index is not None else at]
            corner.append((pos(fm.vertices[vi]), n, uv, rgba))
        pv += len(poly)
        for k in range(1, len(poly) - 1):   # fan (p0, pk, pk+1), reversed to clockwise: (p0, pk+1, pk)
            a, b, c = corner[0], corner[k + 1], corner[k]
            tris.append((a[0], b[0], c[0])); nrms.append((a[1], b[1], c[1])); uvs.append((a[2], b[2], c[2]))
            if vertex_colors:
                cols.append([f32(x) / f32(255.0) for x in xna_color_bytes(a[3])])   # ((Color)colors[Indices[i]]).ToVector4() (TMP:225)
    n = len(tris)
    if vertex_colors:
        col = np.array(cols, dtype=np.float32).reshape(n, 4)
    else:
        col = np.tile(np.array([[f32(c) / f32(255.0) for c in diffuse_color]], dtype=np.float32), (n, 1))   # Color.ToVector4 (TMP:228)
    return MeshData(np.array(tris, dtype=np.float32), np.array(nrms, dtype=np.float32), np.array(uvs, dtype=np.float32), col)
