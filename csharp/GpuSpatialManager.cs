// GpuSpatialManager.cs — ISpatialManager (Spatial/ISpatialManager.cs:10-16) implemented on libxrt.
// NOT COMPILED HERE (no .NET toolchain in the build image).  Drop next to OctreeSpatialManager.cs and assign
// `tracer.CurrentScene = new GpuSpatialManager()` in Game1.LoadContent (Game1.cs:96,123).
using System;
using System.Collections.Generic;
using Microsoft.Xna.Framework;
using RayTracerTypeLibrary;
using RayTraceProject.Native;

namespace RayTraceProject.Spatial
{
    class GpuSpatialManager : ISpatialManager, IDisposable
    {
        readonly List<ISpatialBody> objects = new List<ISpatialBody>();
        readonly Dictionary<Mesh, int> meshIds = new Dictionary<Mesh, int>();
        readonly List<Mesh> meshesById = new List<Mesh>();
        IntPtr scene;

        public List<ISpatialBody> Bodies { get { return this.objects; } }
        public IntPtr Handle { get { return this.scene; } }

        static float[] M(Matrix m)
        {
            return new float[] { m.M11, m.M12, m.M13, m.M14, m.M21, m.M22, m.M23, m.M24, m.M31, m.M32, m.M33, m.M34, m.M41, m.M42, m.M43, m.M44 };
        }
        static float[] B(BoundingBox b) { return new float[] { b.Min.X, b.Min.Y, b.Min.Z, b.Max.X, b.Max.Y, b.Max.Z }; }

        // OctreeSpatialManager.Build (OctreeSpatialManager.cs:64-99) + Mesh.Init of every distinct mesh (SceneObject.cs:132)
        public void Build()
        {
            if (this.scene != IntPtr.Zero) Xrt.Check(Xrt.xrt_scene_destroy(this.scene));
            Xrt.Check(Xrt.xrt_scene_create(0, out this.scene));
            this.meshIds.Clear(); this.meshesById.Clear();
            foreach (SceneObject so in this.objects)
                foreach (Mesh mesh in so.Meshes)
                {
                    if (this.meshIds.ContainsKey(mesh)) continue;          // meshes are shared by reference (SceneObject.cs:126-127)
                    Triangle[] t = mesh.Triangles;
                    float[] v = new float[t.Length * 9], n = new float[t.Length * 9], uv = new float[t.Length * 6],
                            sn = new float[t.Length * 3], col = new float[t.Length * 4];
                    for (int i = 0; i < t.Length; i++)
                    {
                        Put3(v, i * 9, t[i].v1); Put3(v, i * 9 + 3, t[i].v2); Put3(v, i * 9 + 6, t[i].v3);
                        Put3(n, i * 9, t[i].n1); Put3(n, i * 9 + 3, t[i].n2); Put3(n, i * 9 + 6, t[i].n3);
                        uv[i * 6] = t[i].uv1.X; uv[i * 6 + 1] = t[i].uv1.Y; uv[i * 6 + 2] = t[i].uv2.X; uv[i * 6 + 3] = t[i].uv2.Y;
                        uv[i * 6 + 4] = t[i].uv3.X; uv[i * 6 + 5] = t[i].uv3.Y;
                        Put3(sn, i * 3, t[i].surfaceNormal);
                        col[i * 4] = t[i].color.X; col[i * 4 + 1] = t[i].color.Y; col[i * 4 + 2] = t[i].color.Z; col[i * 4 + 3] = t[i].color.W;
                    }
                    Material mat = mesh.MeshMaterial;
                    XrtMaterial xm = new XrtMaterial
                    {
                        reflectiveness = mat.Reflectiveness, transparent = mat.Transparent ? 1 : 0, refractionIndex = mat.RefractionIndex,
                        interpolateNormals = mat.InterpolateNormals ? 1 : 0, useTexture = mat.UseTexture ? 1 : 0,
                        // Material.Init (Material.cs:59-69) keeps the locked bitmap in private fields; expose Scan0 / Width / Height
                        // through three internal getters on Material (one-line additions) and pass them here:
                        texWidth = mat.UseTexture ? mat.TextureWidth : 0, texHeight = mat.UseTexture ? mat.TextureHeight : 0,
                        texArgb = mat.UseTexture ? mat.TextureScan0 : IntPtr.Zero
                    };
                    // the bilinear filter reads Material.Texture.ColorData, the premultiplied Format32bppPArgb copy (RayTracerTexture.cs:24-33,
                    // Material.cs:186-189): pinned for the call (the library copies it)
                    System.Runtime.InteropServices.GCHandle pin = default(System.Runtime.InteropServices.GCHandle);
                    if (mat.UseTexture && mat.Texture != null && mat.Texture.ColorData != null)
                    {
                        pin = System.Runtime.InteropServices.GCHandle.Alloc(mat.Texture.ColorData, System.Runtime.InteropServices.GCHandleType.Pinned);
                        xm.texPArgb = pin.AddrOfPinnedObject();
                    }
                    int id;
                    try { Xrt.Check(Xrt.xrt_scene_add_mesh(this.scene, v, n, uv, sn, col, t.Length, ref xm, B(mesh.MeshBoundingBox), out id)); }
                    finally { if (pin.IsAllocated) pin.Free(); }
                    this.meshIds[mesh] = id; this.meshesById.Add(mesh);
                }
            foreach (SceneObject so in this.objects)
            {
                int[] ids = new int[so.Meshes.Count];
                for (int i = 0; i < ids.Length; i++) ids[i] = this.meshIds[so.Meshes[i]];
                int oid;
                Xrt.Check(Xrt.xrt_scene_add_object(this.scene, ids, ids.Length, M(so.World), M(so.InverseWorld), B(so.BoundingBox),
                                                   B(so.WorldBoundingBox), out oid));
            }
            Xrt.Check(Xrt.xrt_scene_build(this.scene, 0, 0));   // thresholds 50 / 20 (MeshOctree.cs:42, OctreeSpatialManager.cs:50)
        }

        // Scene file (what the content pipeline would write instead of .xnb reflection data, TracerModelProcessor.cs:113-117):
        // Save after Build(); Load replaces Build() at start-up (Bodies stays as the game filled it: object ids follow its order).
        public void Save(string path) { Xrt.Check(Xrt.xrt_scene_save(this.scene, path)); }
        public void Load(string path)
        {
            if (this.scene != IntPtr.Zero) Xrt.Check(Xrt.xrt_scene_destroy(this.scene));
            Xrt.Check(Xrt.xrt_scene_load(0, path, out this.scene));
            Xrt.Check(Xrt.xrt_scene_build(this.scene, 0, 0));
        }

        static void Put3(float[] a, int o, Vector3 p) { a[o] = p.X; a[o + 1] = p.Y; a[o + 2] = p.Z; }

        // Batched form used by a wavefront renderer.
        public void IntersectBatch(XrtRay[] rays, XrtHit[] hits)
        {
            Xrt.Check(Xrt.xrt_scene_intersect(this.scene, rays, null, rays.Length, hits, IntPtr.Zero));
        }

        // The single-ray signature of the interface, forwarded as a batch of one (correct, slow: a P/Invoke plus a kernel
        // launch per ray; RayTracer.RenderInternal should call xrt_render instead, see INTEGRATION.md).
        public bool GetRayIntersection(ref Ray ray, out IntersectionResult? result, Triangle ignoreTriangle, Mesh ignoreObject)
        {
            result = null;
            XrtRay[] r = new XrtRay[1];
            r[0].ox = ray.Position.X; r[0].oy = ray.Position.Y; r[0].oz = ray.Position.Z;
            r[0].dx = ray.Direction.X; r[0].dy = ray.Direction.Y; r[0].dz = ray.Direction.Z;
            r[0].ignoreMesh = -1; r[0].ignoreTri = -1;
            if (ignoreTriangle != null)
                foreach (KeyValuePair<Mesh, int> kv in this.meshIds)   // reference identity -> (mesh id, index in Mesh.Triangles[])
                {
                    int idx = Array.IndexOf(kv.Key.Triangles, ignoreTriangle);
                    if (idx >= 0) { r[0].ignoreMesh = kv.Value; r[0].ignoreTri = idx; break; }
                }
            XrtHit[] h = new XrtHit[1];
            IntersectBatch(r, h);
            if (h[0].hit == 0) return false;
            Mesh mesh = this.meshesById[h[0].mesh];
            result = new IntersectionResult(mesh, mesh.Triangles[h[0].tri], h[0].u, h[0].v, h[0].d, new Vector3(h[0].wx, h[0].wy, h[0].wz));
            return true;
        }

        public void Dispose()
        {
            if (this.scene != IntPtr.Zero) { Xrt.xrt_scene_destroy(this.scene); this.scene = IntPtr.Zero; }
        }
    }
}
