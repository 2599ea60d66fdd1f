"""Host-side mirror of the reference's API surface for the hot path, forwarding to libxrt through the
C-ABI (include/xrt.h).  Same names, argument meaning and error behaviour as the C# classes:

    Material (RayTracerTypeLibrary/Material.cs:25), Mesh (Mesh.cs:9), MeshOctree (MeshOctree.cs:9),
    SceneObject (RayTraceProject/SceneObject.cs:12), ISpatialManager / OctreeSpatialManager
    (Spatial/ISpatialManager.cs:10, OctreeSpatialManager.cs:35), Camera (Camera.cs), SpotLight,
    DirectionalLight (SpotLight.cs:10, DirectionalLight.cs:10), RayTracer (RayTracer.cs:13).

The real host is C#; its P/Invoke shim is in csharp/ and INTEGRATION.md.  This mirror exists because the
image has no .NET toolchain; it contains no arithmetic of the path itself — all of that runs in HIP.
"""
import ctypes as C
import threading
import numpy as np

from . import _abi as abi
from . import xna

f32 = np.float32


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Material:
    """Material.cs:25-269 (shading inputs; texels as the Format32bppArgb lock of MAT:65)."""

    def __init__(self, reflectiveness=0.0, useTexture=False, transparent=False, refractionIndex=0.0, texture=None, texture_pargb=None, textureFilePath=None):
        self.Reflectiveness = float(reflectiveness)
        self.UseTexture = bool(useTexture)
        self.Transparent = bool(transparent)
        self.RefractionIndex = float(refractionIndex)
        self.InterpolateNormals = False
        self.Texture = None if texture is None else np.ascontiguousarray(texture, dtype=np.uint32)
        # Material.Texture.ColorData: the premultiplied Format32bppPArgb copy the bilinear filter reads (TEX:24-33, MAT:186-189)
        self.TexturePArgb = None if texture_pargb is None else np.ascontiguousarray(texture_pargb, dtype=np.uint32)
        self.TextureFilePath = textureFilePath   # MAT:45-53; TracerModelProcessor.CreateMaterial passes its TextureFilePath parameter (TMP:121-131)

    def Init(self):
        """Material.Init (MAT:59-69): Bitmap.FromFile(textureFilePath) locked as Format32bppArgb.  Files without an alpha channel
        (24-bpp BMP, what the reference's content uses) need no separate premultiplied copy."""
        if self.UseTexture and self.Texture is None:
            if not self.TextureFilePath:
                raise ValueError("UseTexture without a texture file")   # Bitmap.FromFile(null) throws, MAT:63
            from . import fixtures
            self.Texture = fixtures.load_bmp_argb(self.TextureFilePath)

    def _to_abi(self):
        m = abi.xrt_material()
        m.reflectiveness = self.Reflectiveness
        m.transparent = int(self.Transparent)
        m.refraction_index = self.RefractionIndex
        m.interpolate_normals = int(self.InterpolateNormals)
        m.use_texture = int(self.UseTexture)
        if self.UseTexture:
            if self.Texture is None:
                self.Init()
            m.tex_height, m.tex_width = self.Texture.shape
            m.tex_argb = self.Texture.ctypes.data_as(C.POINTER(C.c_uint32))
            if self.TexturePArgb is not None:
                if self.TexturePArgb.shape != self.Texture.shape:
                    raise ValueError("Texture.ColorData must have the size of the bitmap")
                m.tex_pargb = self.TexturePArgb.ctypes.data_as(C.POINTER(C.c_uint32))
        return m


class _Scene:
    """Owner of one xrt_scene handle."""

    def __init__(self, device=0):
        self.handle = C.c_void_p()
        abi.check(abi.lib().xrt_scene_create(int(device), C.byref(self.handle)))

    def close(self):
        if self.handle:
            abi.lib().xrt_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_mesh(self, mesh):
        d = mesh.data
        mat = mesh.MeshMaterial._to_abi()
        mid = C.c_int32(-1)
        sn = np.ascontiguousarray(d.surface_normal, dtype=np.float32)
        bbox = np.ascontiguousarray(mesh.MeshBoundingBox, dtype=np.float32)
        abi.check(abi.lib().xrt_scene_add_mesh(self.handle, _fp(d.v), _fp(d.n), _fp(d.uv), _fp(sn), _fp(d.color),
                                               d.ntri, C.byref(mat), _fp(bbox), C.byref(mid)))
        return mid.value


def rays_array(origins, directions, ignore_mesh=None, ignore_tri=None):
    """Pack (n,3) origins / directions into an xrt_ray array."""
    o = np.asarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.asarray(directions, dtype=np.float32).reshape(-1, 3)
    n = o.shape[0]
    dt = np.dtype([("o", np.float32, 3), ("d", np.float32, 3), ("ignore_mesh", np.int32), ("ignore_tri", np.int32)])
    r = np.zeros(n, dtype=dt)
    r["o"], r["d"] = o, d
    r["ignore_mesh"] = -1 if ignore_mesh is None else ignore_mesh
    r["ignore_tri"] = -1 if ignore_tri is None else ignore_tri
    return r


RAY_DTYPE = np.dtype([("o", np.float32, 3), ("d", np.float32, 3), ("ignore_mesh", np.int32), ("ignore_tri", np.int32)])
HIT_DTYPE = np.dtype([("hit", np.int32), ("object", np.int32), ("mesh", np.int32), ("tri", np.int32), ("leaf", np.int32),
                      ("u", np.float32), ("v", np.float32), ("d", np.float32), ("w", np.float32, 3), ("reserved", np.int32)])
NODE_DTYPE = np.dtype([("bmin", np.float32, 3), ("bmax", np.float32, 3), ("is_leaf", np.int32), ("count", np.int32),
                       ("dfs_index", np.int32), ("depth", np.int32), ("first_ref", np.int32), ("reserved", np.int32)])
assert RAY_DTYPE.itemsize == 32 and HIT_DTYPE.itemsize == 48 and NODE_DTYPE.itemsize == C.sizeof(abi.xrt_node_info)


class MeshOctree:
    """MeshOctree.cs:9-355.  Bodies = the mesh's triangles; Build() builds the tree natively with the
    MO:56-96/204-236 semantics; GetRayIntersection is MO:259-326."""

    def __init__(self, mesh):
        self._mesh = mesh
        self._scene = None
        self._mesh_id = -1
        self.itemTreshold = 50   # MO:42

    @property
    def Bodies(self):
        return self._mesh.data

    def Build(self):
        self._scene = _Scene(self._mesh.device)
        self._mesh_id = self._scene.add_mesh(self._mesh)
        abi.check(abi.lib().xrt_scene_build(self._scene.handle, self.itemTreshold, 0))

    def IntersectBatch(self, rays):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        abi.check(abi.lib().xrt_mesh_intersect(self._scene.handle, self._mesh_id,
                                               rays.ctypes.data_as(C.POINTER(abi.xrt_ray)), rays.shape[0],
                                               hits.ctypes.data_as(C.POINTER(abi.xrt_hit))))
        return hits

    def GetRayIntersection(self, ray, ignoreTriangle=None):
        """ray = (position, direction); ignoreTriangle = triangle index or None.
        Returns (found, result) with result = dict(triangle, u, v, d, objectSpacePosition) or None."""
        r = rays_array([ray[0]], [ray[1]], self._mesh_id if ignoreTriangle is not None else -1,
                       -1 if ignoreTriangle is None else int(ignoreTriangle))
        h = self.IntersectBatch(r)[0]
        if not h["hit"]:
            return False, None
        return True, dict(triangle=int(h["tri"]), u=h["u"], v=h["v"], d=h["d"], objectSpacePosition=h["w"].copy(), leaf=int(h["leaf"]))

    def tree(self):
        return _get_tree(self._scene, self._mesh_id)


def _get_tree(scene, mesh_id):
    nn, nr = C.c_int64(0), C.c_int64(0)
    abi.check(abi.lib().xrt_scene_get_tree(scene.handle, mesh_id, None, C.byref(nn), None, C.byref(nr)))
    nodes = np.zeros(nn.value, dtype=NODE_DTYPE)
    refs = np.zeros(max(nr.value, 1), dtype=np.int32)
    abi.check(abi.lib().xrt_scene_get_tree(scene.handle, mesh_id, nodes.ctypes.data_as(C.POINTER(abi.xrt_node_info)),
                                           C.byref(nn), refs.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nr)))
    return nodes, refs[: nr.value]


class Mesh:
    """Mesh.cs:9-40: Triangles[] (as arrays), MeshMaterial, MeshBoundingBox, Octree, Init(), RayIntersects()."""

    def __init__(self, triangles, material, boundingBox=None, device=0):
        self.data = triangles            # fixtures.MeshData
        self.Triangles = triangles
        self.MeshMaterial = material
        self.MeshBoundingBox = np.asarray(triangles.bbox if boundingBox is None else boundingBox, dtype=np.float32)
        self.Octree = None
        self.device = device

    def Init(self):   # MESH:27-32
        self.Octree = MeshOctree(self)
        self.Octree.Build()


class SceneObject:
    """SceneObject.cs:12 — transform + shared mesh list (SO:126-127); BuildWorld is SO:183-199."""

    def __init__(self, meshes, pos=(0, 0, 0), rot=(0, 0, 0), name=None):
        self.Meshes = list(meshes)
        self.Position = tuple(pos)
        self.Rotation = tuple(rot)
        self.Scale = (1.0, 1.0, 1.0)
        self.Name = name
        bb = np.zeros(6, dtype=np.float32)   # default(BoundingBox) merged with every mesh box (SO:131)
        for m in self.Meshes:
            bb[:3] = np.minimum(bb[:3], m.MeshBoundingBox[:3])
            bb[3:] = np.maximum(bb[3:], m.MeshBoundingBox[3:])
        self.BoundingBox = bb

    def _build_world(self):
        world, inv, wbb = xna.build_world(self.Scale, self.Rotation, self.Position, self.BoundingBox)
        return xna.as_array(world), xna.as_array(inv), xna.as_array(wbb)

    @property
    def World(self):
        return self._build_world()[0]

    @property
    def InverseWorld(self):
        return self._build_world()[1]

    @property
    def WorldBoundingBox(self):
        return self._build_world()[2]


class ISpatialManager:
    """Spatial/ISpatialManager.cs:10-16."""

    def Build(self):
        raise NotImplementedError

    def GetRayIntersection(self, ray, ignoreTriangle=None, ignoreObject=None):
        raise NotImplementedError


class OctreeSpatialManager(ISpatialManager):
    """OctreeSpatialManager.cs:35 — the GPU implementation of the reference's plug-in seam."""

    def __init__(self, device=0):
        self.Bodies = []
        self.itemTreshold = 20          # OSM:50
        self.meshItemTreshold = 50      # MO:42
        self.device = device
        self._scene = None
        self._mesh_ids = {}
        self.meshes = []

    def Build(self):   # OSM:64-99 (+ Mesh.Init of every distinct mesh, SO:132)
        self._scene = _Scene(self.device)
        self._mesh_ids = {}
        self.meshes = []
        for body in self.Bodies:
            for m in body.Meshes:
                if id(m) not in self._mesh_ids:
                    self._mesh_ids[id(m)] = self._scene.add_mesh(m)
                    self.meshes.append(m)
        for body in self.Bodies:
            ids = np.array([self._mesh_ids[id(m)] for m in body.Meshes], dtype=np.int32)
            world, inv, wbb = body._build_world()
            bb = np.ascontiguousarray(body.BoundingBox, dtype=np.float32)
            oid = C.c_int32(-1)
            abi.check(abi.lib().xrt_scene_add_object(self._scene.handle, ids.ctypes.data_as(C.POINTER(C.c_int32)), len(ids),
                                                     _fp(world), _fp(inv), _fp(bb), _fp(wbb), C.byref(oid)))
        abi.check(abi.lib().xrt_scene_build(self._scene.handle, self.meshItemTreshold, self.itemTreshold))

    def Save(self, path):
        """xrt_scene_save: the built scene's meshes, materials, texels and bodies as one file (the reference's .xnb content)."""
        abi.check(abi.lib().xrt_scene_save(self.handle, str(path).encode()))

    @classmethod
    def Load(cls, path, device=0):
        """xrt_scene_load + xrt_scene_build: a spatial manager whose geometry lives only in the library (Bodies stays empty;
        meshes are known by id).  What a game does at start-up with the file its content build wrote."""
        sm = cls(device)
        sm._scene = _Scene.__new__(_Scene)
        sm._scene.handle = C.c_void_p()
        abi.check(abi.lib().xrt_scene_load(int(device), str(path).encode(), C.byref(sm._scene.handle)))
        abi.check(abi.lib().xrt_scene_build(sm._scene.handle, sm.meshItemTreshold, sm.itemTreshold))
        return sm

    @property
    def handle(self):
        if self._scene is None:
            raise RuntimeError("scene not built")
        return self._scene.handle

    def mesh_id(self, mesh):
        return self._mesh_ids[id(mesh)]

    def SplitStats(self, reset=True):
        """xrt_split_stats: (subtrees handed over, taken, packets split, packets written by a taker) -- diagnostics of the split walks of long packets."""
        out = (C.c_uint64 * 4)()
        abi.check(abi.lib().xrt_split_stats(self.handle, out, 1 if reset else 0))
        return tuple(int(x) for x in out)

    def IntersectBatch(self, rays, stats=False):
        """Batched ISpatialManager.GetRayIntersection (ISM:15): xrt_ray array -> xrt_hit array."""
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        st = abi.xrt_stats()
        abi.check(abi.lib().xrt_scene_intersect(self.handle, rays.ctypes.data_as(C.POINTER(abi.xrt_ray)), None, rays.shape[0],
                                                hits.ctypes.data_as(C.POINTER(abi.xrt_hit)), C.byref(st) if stats else None))
        return (hits, st.as_dict()) if stats else hits

    def GetRayIntersection(self, ray, ignoreTriangle=None, ignoreObject=None):
        """Single-ray signature of ISM:15, forwarded as a batch of one (correct, slow).
        ignoreTriangle = (mesh, triangle index) or None; ignoreObject is dead in the reference (OSM:343)."""
        im, it = (-1, -1) if ignoreTriangle is None else (self.mesh_id(ignoreTriangle[0]), int(ignoreTriangle[1]))
        h = self.IntersectBatch(rays_array([ray[0]], [ray[1]], im, it))[0]
        if not h["hit"]:
            return False, None
        return True, dict(mesh=self.meshes[h["mesh"]], triangle=int(h["tri"]), u=h["u"], v=h["v"], d=h["d"],
                          worldPosition=h["w"].copy(), object=int(h["object"]), leaf=int(h["leaf"]))

    def tree(self, mesh=None):
        return _get_tree(self._scene, -1 if mesh is None else self.mesh_id(mesh))


class Camera:
    """Camera.cs: LookAt view + perspective projection (CAM:40-54), rebuilt on every access."""

    def __init__(self, position, target, up, fieldOfView, aspectRatio, nearClippingPlane, farClippingPlane):
        self.Position, self.Target, self.Up = position, target, up
        self.FieldOfView, self.AspectRatio = fieldOfView, aspectRatio
        self.NearClippingPlane, self.FarClippingPlane = nearClippingPlane, farClippingPlane

    @property
    def View(self):
        return xna.as_array(xna.create_look_at(self.Position, self.Target, self.Up))

    @property
    def Projection(self):
        return xna.as_array(xna.create_perspective_fov(self.FieldOfView, self.AspectRatio, self.NearClippingPlane, self.FarClippingPlane))


class SpotLight:
    """SpotLight.cs:10-63."""
    IsPositionable = True

    def __init__(self):
        self.Position = (0.0, 0.0, 0.0)
        self.Direction = (0.0, -1.0, 0.0)
        self.Color = (1.0, 1.0, 1.0)
        self.DecayExponent = 1.3   # SPOT:33
        self.Intensity = 1.0       # SPOT:34
        self.SpotAngle = 0.0

    def _to_abi(self):
        l = abi.xrt_light()
        l.kind = abi.LIGHT_SPOT
        l.position[:] = self.Position
        l.direction[:] = self.Direction
        l.color[:] = self.Color
        l.intensity, l.spot_angle, l.decay_exponent = self.Intensity, self.SpotAngle, self.DecayExponent
        return l


class DirectionalLight:
    """DirectionalLight.cs:10-31."""
    IsPositionable = False

    def __init__(self):
        self.Direction = (0.0, -1.0, 0.0)
        self.Color = (1.0, 1.0, 1.0)
        self.Intensity = 1.0   # DIR:20

    def _to_abi(self):
        l = abi.xrt_light()
        l.kind = abi.LIGHT_DIRECTIONAL
        l.direction[:] = self.Direction
        l.color[:] = self.Color
        l.intensity = self.Intensity
        l.decay_exponent = 1.3
        return l


class RenderTarget:
    """Stand-in for RenderTarget2D: Width, Height and the Color[] the tracer fills (RT:29,123)."""

    def __init__(self, width, height):
        self.Width, self.Height = int(width), int(height)
        self.data = None


class RayTracer:
    """RayTracer.cs:13 — the properties of RT:19-46 and RenderAsync / Render; the body of
    RenderInternal (RT:105-120) is one call into libxrt."""

    def __init__(self):
        self.CurrentScene = None
        self.CurrentCamera = None
        self._target = None
        self.MaxReflections = 0
        self.IsBusy = False
        self.RenderCompleted = None
        self.TextureFiltering = abi.FILTER_POINT
        self.AddressMode = abi.ADDRESS_WRAP
        self.Lights = []
        self.UseMultisampling = False
        self.MultisampleQuality = 0
        self.MultisampleMode = None     # None: ADAPTIVE when UseMultisampling (the reference); or abi.MS_FIXED16
        self.renderTargetData = None
        self.last_stats = None
        self.collect_stats = False
        self.NumGpus = 1                # xrt_render_opts.n_gpus: one process, the frame's tiles dealt to this many devices
        self.BalanceTiles = False       # xrt_render_opts.balance_tiles (with NumGpus > 1): tiles dealt by the last frame's costs instead of round-robin

    @property
    def CurrentTarget(self):
        return self._target

    @CurrentTarget.setter
    def CurrentTarget(self, value):
        if self.IsBusy:
            raise RuntimeError("Can not change RenderTarget while RayTracer is busy.")   # RT:26-27
        self._target = value
        self.renderTargetData = np.zeros(value.Width * value.Height, dtype=np.uint32)     # RT:29

    @property
    def Progress(self):   # RT:43-46
        return float(abi.lib().xrt_progress(self.CurrentScene.handle))

    def _camera_abi(self):
        cam = abi.xrt_camera()
        cam.view[:] = [float(x) for x in self.CurrentCamera.View]
        cam.proj[:] = [float(x) for x in self.CurrentCamera.Projection]
        cam.vp_x, cam.vp_y, cam.vp_width, cam.vp_height = 0, 0, self._target.Width, self._target.Height
        cam.vp_min_depth, cam.vp_max_depth = 0.0, 1.0
        return cam

    def _opts_abi(self, shard_rank=0, shard_count=1):
        o = abi.xrt_render_opts()
        o.max_reflections = int(self.MaxReflections)
        if self.UseMultisampling:
            o.use_multisampling = abi.MS_ADAPTIVE if self.MultisampleMode is None else self.MultisampleMode
        else:
            o.use_multisampling = abi.MS_OFF
        o.multisample_quality = int(self.MultisampleQuality)
        o.address_mode, o.filtering = int(self.AddressMode), int(self.TextureFiltering)
        o.shard_rank, o.shard_count = shard_rank, shard_count
        o.collect_stats = int(self.collect_stats)
        o.n_gpus = int(self.NumGpus) if shard_count <= 1 else 0
        o.balance_tiles = 1 if (self.BalanceTiles and o.n_gpus > 1) else 0
        return o

    def _lights_abi(self):
        arr = (abi.xrt_light * max(len(self.Lights), 1))()
        for i, l in enumerate(self.Lights):
            arr[i] = l._to_abi()
        return arr

    def Render(self, want_float=False):
        """Blocking RenderInternal (RT:103-126).  Returns the packed Color[] (and the float colorVector)."""
        cam, opts, lights = self._camera_abi(), self._opts_abi(), self._lights_abi()
        st = abi.xrt_stats()
        rgbf = np.zeros(self._target.Width * self._target.Height * 3, dtype=np.float32) if want_float else None
        abi.check(abi.lib().xrt_render(self.CurrentScene.handle, C.byref(cam), lights, len(self.Lights), C.byref(opts),
                                       self.renderTargetData.ctypes.data_as(C.POINTER(C.c_uint32)),
                                       _fp(rgbf) if want_float else None, C.byref(st)))
        self.last_stats = st.as_dict()
        self._target.data = self.renderTargetData
        return (self.renderTargetData, rgbf.reshape(-1, 3)) if want_float else self.renderTargetData

    def RenderAsync(self):   # RT:59-79
        if self.IsBusy:
            raise RuntimeError("Current render operation not finished.")   # RT:62-63
        self.IsBusy = True

        def run():
            try:
                self.Render()
            finally:
                self.IsBusy = False                     # RT:433
                if self.RenderCompleted is not None:    # RT:435-436
                    self.RenderCompleted(self)

        t = threading.Thread(target=run, name="RenderDispatcherThread", daemon=True)   # RT:76-77
        t.start()
        return t

    def RenderDevice(self, d_rgba_ptr, stream=None, shard_rank=0, shard_count=1):
        """xrt_render_device: output stays in HBM (a torch tensor's data_ptr())."""
        return self.PrepareDevice(d_rgba_ptr, stream, shard_rank, shard_count)()

    def PrepareDevice(self, d_rgba_ptr, stream=None, shard_rank=0, shard_count=1):
        """Marshal camera / lights / options once (what the C# host does with its own XNA matrices) and return a
        callable that renders one frame into HBM per call and returns the xrt_stats dict."""
        cam, opts, lights = self._camera_abi(), self._opts_abi(shard_rank, shard_count), self._lights_abi()
        st = abi.xrt_stats()
        fn, handle, n = abi.lib().xrt_render_device, self.CurrentScene.handle, len(self.Lights)
        args = (handle, C.byref(cam), lights, n, C.byref(opts), C.c_void_p(d_rgba_ptr), C.c_void_p(stream or 0), C.byref(st))

        def frame():
            abi.check(fn(*args))
            self.last_stats = st.as_dict()
            return self.last_stats
        frame.keepalive = (cam, opts, lights, st)

        # pipelined form: begin() enqueues a frame and returns its ticket, end(ticket) waits for it (two may be open)
        lib = abi.lib()
        ticket = C.c_int32(0)
        bargs = args[:-1] + (C.byref(ticket),)

        def begin():
            abi.check(lib.xrt_render_device_begin(*bargs))
            return ticket.value

        def end(t):
            abi.check(lib.xrt_render_device_end(handle, t, C.byref(st)))
            self.last_stats = st.as_dict()
            return self.last_stats
        frame.begin, frame.end = begin, end
        return frame

    # ---- cost-aware tile assignment (xrt.h: the static counterpart of the reference's dynamic row stealing, RT:48-52) ----
    def TileCosts(self, reset=True):
        """xrt_scene_tile_costs: ticks per tile (row-major tile order) of the frames of the current target's size rendered since the last reset."""
        w, h = self._target.Width, self._target.Height
        tx, ty = (w + abi.TILE_W - 1) // abi.TILE_W, (h + abi.TILE_H - 1) // abi.TILE_H
        cost = np.zeros(tx * ty, dtype=np.float32)
        abi.check(abi.lib().xrt_scene_tile_costs(self.CurrentScene.handle, w, h, _fp(cost), 1 if reset else 0))
        return cost

    def SetTileTable(self, shard_count, tiles_per_rank, table):
        """xrt_scene_set_tile_table for frames of the current target's size (table None: back to round-robin)."""
        w, h = self._target.Width, self._target.Height
        if table is None:
            abi.check(abi.lib().xrt_scene_set_tile_table(self.CurrentScene.handle, w, h, 0, 0, None))
            return
        t = np.ascontiguousarray(table, dtype=np.int32)
        abi.check(abi.lib().xrt_scene_set_tile_table(self.CurrentScene.handle, w, h, int(shard_count), int(tiles_per_rank), t.ctypes.data_as(C.POINTER(C.c_int32))))

    def PrepareHost(self, host_array):
        """Pipelined host-output frames (xrt_render_begin / xrt_render_end): RenderAsync with the frame ending in the host's
        Color[] (RT:122-123).  `host_array`: a uint32 numpy array of W*H elements, ideally registered with
        xrt_host_register so that the device-to-host copy of frame i runs under the rendering of frame i+1."""
        cam, opts, lights = self._camera_abi(), self._opts_abi(), self._lights_abi()
        st, ticket, lib = abi.xrt_stats(), C.c_int32(0), abi.lib()
        handle, n = self.CurrentScene.handle, len(self.Lights)
        ptr = host_array.ctypes.data_as(C.POINTER(C.c_uint32))

        def begin():
            abi.check(lib.xrt_render_begin(handle, C.byref(cam), lights, n, C.byref(opts), ptr, C.byref(ticket)))
            return ticket.value

        def end(t):
            abi.check(lib.xrt_render_end(handle, t, C.byref(st)))
            self.last_stats = st.as_dict()
            return self.last_stats
        begin.keepalive = (cam, opts, lights, st, host_array)
        begin.begin, begin.end = begin, end
        return begin

    def GeneratePrimaryRays(self):
        """The rays of RT:410-421 for the whole target."""
        cam = self._camera_abi()
        rays = np.zeros(self._target.Width * self._target.Height, dtype=RAY_DTYPE)
        abi.check(abi.lib().xrt_generate_primary_rays(self.CurrentScene.handle, C.byref(cam), rays.ctypes.data_as(C.POINTER(abi.xrt_ray))))
        return rays
