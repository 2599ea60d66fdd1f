// XrtNative.cs — P/Invoke binding of include/xrt.h for the xna-ray-trace C# host.
// NOT COMPILED HERE: the build image has no .NET toolchain (dotnet / mono / csc absent).  Style follows the
// reference's own native binding precedent (aviFileWrapper_src/Avi.cs:41-185: [StructLayout(Sequential)],
// [DllImport] returning int, wrapper throws on non-zero).
using System;
using System.Runtime.InteropServices;

namespace RayTraceProject.Native
{
    [StructLayout(LayoutKind.Sequential)]
    public struct XrtRay { public float ox, oy, oz, dx, dy, dz; public int ignoreMesh, ignoreTri; }          // 32 B

    [StructLayout(LayoutKind.Sequential)]
    public struct XrtHit { public int hit, obj, mesh, tri, leaf; public float u, v, d, wx, wy, wz; public int reserved; }   // 48 B

    [StructLayout(LayoutKind.Sequential)]
    public struct XrtMaterial
    {
        public float reflectiveness; public int transparent; public float refractionIndex;
        public int interpolateNormals, useTexture, texWidth, texHeight, reserved;
        public IntPtr texArgb;   // BitmapData.Scan0 of the Format32bppArgb lock (Material.cs:65)
        public IntPtr texPArgb;  // Material.Texture.ColorData pinned (RayTracerTexture.cs:24-33: the premultiplied copy GetColorBilinear reads); IntPtr.Zero = same as texArgb
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct XrtCamera
    {
        public fixed float view[16]; public fixed float proj[16];
        public int vpX, vpY, vpWidth, vpHeight; public float vpMinDepth, vpMaxDepth;
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct XrtLight
    {
        public int kind; public fixed float position[3]; public fixed float direction[3]; public fixed float color[3];
        public float intensity, spotAngle, decayExponent;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct XrtRenderOpts
    {
        public int maxReflections, useMultisampling, multisampleQuality, addressMode, filtering, shardRank, shardCount, collectStats;
        public int nGpus;                          // > 1: the library spreads the frame's tiles over that many devices and gathers them with RCCL
        public int balanceTiles;                   // with nGpus > 1: 1 = the tiles are dealt by the previous frame's costs (the static counterpart of GetNextScanline, RayTracer.cs:48-52), 0 = round-robin
        public int reserved0, reserved1;           // zero
    }

    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct XrtNodeInfo                // xrt_scene_get_tree: the private CubeNode (MeshOctree.cs:32-40) flattened, DFS pre-order
    {
        public fixed float bmin[3]; public fixed float bmax[3];
        public int isLeaf, count, dfsIndex, depth, firstRef, reserved;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct XrtStats
    {
        public ulong raysClosest, raysShadow, hitsClosest, hitsShadow, sceneNodeTests, instanceVisits, meshAabbTests, meshQueries,
                     nodeTests, leafRefs, triTests, shadedHits, pixels, algorithmicBytes;
        public double msTotal, msIntersect; public uint intersectLaunches, pieces; public ulong raysTraversed; public double msIntersectLongest; public ulong meshQueriesFacingAway;
    }

    public static class Xrt
    {
        const string Lib = "xrt";   // libxrt.so / xrt.dll
        // include/xrt.h promises cdecl.  The reference builds x86 (RayTraceProject.csproj:47,60), where P/Invoke defaults to StdCall
        // and a cdecl callee would unbalance the stack: every import names its convention.  (The Avi.cs precedent binds Win32
        // stdcall APIs and does not carry over.)
        public const int OK = 0, E_INVALID_ARG = -1, E_BUSY = -2, E_NO_DEVICE = -3;

        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_version();
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern IntPtr xrt_last_error();
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_scene_create(int device, out IntPtr scene);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_scene_destroy(IntPtr scene);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_scene_add_mesh(IntPtr scene, float[] v, float[] n, float[] uv, float[] surfN, float[] color,
                                                                    int ntri, ref XrtMaterial material, float[] bbox, out int meshId);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_scene_add_object(IntPtr scene, int[] meshIds, int nMeshes, float[] world, float[] invWorld,
                                                                      float[] bbox, float[] worldBbox, out int objectId);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_scene_build(IntPtr scene, int meshThreshold, int sceneThreshold);
        // scene file: the content-pipeline step writes it once (instead of .xnb reflection serialisation of Model.Tag), the game loads it
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_scene_save(IntPtr scene, [MarshalAs(UnmanagedType.LPStr)] string path);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_scene_load(int device, [MarshalAs(UnmanagedType.LPStr)] string path, out IntPtr scene);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_scene_intersect(IntPtr scene, [In] XrtRay[] rays, int[] ignoreObject, long n,
                                                                     [Out] XrtHit[] hits, IntPtr stats);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern unsafe int xrt_render(IntPtr scene, ref XrtCamera camera, XrtLight[] lights, int nLights,
                                                                   ref XrtRenderOpts opts, uint* rgbaOut, float* rgbF32Out, IntPtr stats);
        // pipelined form (two frames in flight; device output): RenderAsync / RenderCompleted without a host round trip
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_render_device_begin(IntPtr scene, ref XrtCamera camera, XrtLight[] lights, int nLights,
                                                                            ref XrtRenderOpts opts, IntPtr dRgbaOut, IntPtr stream, out int ticket);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_render_device_end(IntPtr scene, int ticket, IntPtr statsOut);
        // pipelined form with the frame ending in the host's Color[] (RenderAsync + CurrentTarget.SetData, RayTracer.cs:59-79,122-123):
        // pin renderTargetData (GCHandle.Alloc(.., GCHandleType.Pinned)) and xrt_host_register it once; up to two tickets open
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_render_begin(IntPtr scene, ref XrtCamera camera, XrtLight[] lights, int nLights,
                                                                     ref XrtRenderOpts opts, IntPtr rgbaOut, out int ticket);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_render_end(IntPtr scene, int ticket, IntPtr statsOut);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_host_register(IntPtr hostPtr, ulong bytes);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_host_unregister(IntPtr hostPtr);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern float xrt_progress(IntPtr scene);
        // ---- the rest of include/xrt.h (every export is bound: tests/test_host_logic.py checks this file against the header) ----
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_device_count(out int count);
        // inspection of the built trees (meshId -1: the scene octree); call with null arrays for the counts
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_scene_get_tree(IntPtr scene, int meshId, [Out] XrtNodeInfo[] nodes, ref long nNodes,
                                                                    [Out] int[] refs, ref long nRefs);
        // seam 1 with HBM-resident rays / hits (device pointers, asynchronous on `stream`), and MeshOctree.GetRayIntersection of one mesh (MeshOctree.cs:259-326)
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_scene_intersect_device(IntPtr scene, IntPtr dRays, long n, IntPtr dHitsOut, IntPtr stream);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_mesh_intersect(IntPtr scene, int meshId, [In] XrtRay[] rays, long n, [Out] XrtHit[] hits);
        // blocking frame into device memory (or this process's tile shard)
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_render_device(IntPtr scene, ref XrtCamera camera, XrtLight[] lights, int nLights,
                                                                      ref XrtRenderOpts opts, IntPtr dRgbaOut, IntPtr stream, IntPtr statsOut);
        // image-tile shards (one process per GPU): layout, de-tile of the gathered buffers
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_shard_layout(int width, int height, int shardCount, out int tilesX, out int tilesY, out int tilesPerRank);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_detile_device(int width, int height, int shardCount, IntPtr dGathered, long rankStride,
                                                                      IntPtr dRgbaOut, IntPtr stream);
        // cost-aware tile assignment: the previous frame's tile costs -> a longest-first table -> installed for the next frames
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_scene_set_tile_table(IntPtr scene, int width, int height, int shardCount, int tilesPerRank, int[] tileOfSlot);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_scene_tile_costs(IntPtr scene, int width, int height, [Out] float[] costOut, int reset);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_balance_tiles(int width, int height, int shardCount, float[] tileCost, int tilesPerRank, [Out] int[] tileOfSlotOut);
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_detile_table_device(int width, int height, int shardCount, int tilesPerRank, IntPtr dTileOfSlot,
                                                                            IntPtr dGathered, long rankStride, IntPtr dRgbaOut, IntPtr stream);
        // the primary rays of RayTracer.Render (RayTracer.cs:410-421), row-major
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_generate_primary_rays(IntPtr scene, ref XrtCamera camera, [Out] XrtRay[] raysOut);
        // can n_gpus > 1 load RCCL?  OK or E_RCCL (-6) with the loader's message; no device is touched
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_rccl_probe();
        // diagnostics of the split walks of long packets (results never depend on them): subtrees handed over, taken, packets split, packets written by a taker
        [DllImport(Lib, CallingConvention = CallingConvention.Cdecl)] public static extern int xrt_split_stats(IntPtr scene, [Out] ulong[] out4, int reset);

        // error convention of the reference: InvalidOperationException when busy (RayTracer.cs:26-27,62-63),
        // ArgumentException for bad arguments (SceneObject.cs:123-124, Material.cs:85,97)
        public static void Check(int rc)
        {
            if (rc == OK) return;
            string msg = Marshal.PtrToStringAnsi(xrt_last_error());
            if (rc == E_BUSY) throw new InvalidOperationException(msg);
            if (rc == E_INVALID_ARG) throw new ArgumentException(msg);
            throw new Exception("Exception in libxrt: " + rc + " " + msg);
        }
    }
}
