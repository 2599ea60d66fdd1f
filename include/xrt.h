/*
 * xrt.h — C-ABI of libxrt, the MI355X (gfx950) ray / octree / triangle hot path that drops in
 * behind the xna-ray-trace C# host (P/Invoke), see INTEGRATION.md.
 *
 * Every entry point names the reference interface it replaces (file:line relative to the reference
 * repository; aliases: RT = RayTraceProject/RayTraceProject/RayTracer.cs,
 * OSM = RayTraceProject/RayTraceProject/Spatial/OctreeSpatialManager.cs,
 * ISM = .../Spatial/ISpatialManager.cs, SO = .../SceneObject.cs, MO = RayTracerTypeLibrary/MeshOctree.cs,
 * MESH = RayTracerTypeLibrary/Mesh.cs, MAT = RayTracerTypeLibrary/Material.cs,
 * TRI = RayTracerTypeLibrary/Triangle.cs, SPOT/DIR = .../SpotLight.cs / DirectionalLight.cs).
 *
 * Conventions
 *   - plain C, cdecl, POD structs with natural alignment, no C++ or torch types;
 *   - every function returns int: XRT_OK (0) or a negative XRT_E_* code; xrt_last_error() gives the
 *     thread-local message (precedent for the style: aviFileWrapper_src/Avi.cs:175-185 int HRESULTs);
 *   - all host arrays are borrowed for the duration of the call and copied; the library owns all
 *     device memory; handles are opaque;
 *   - matrices are XNA row-vector convention (v' = v * M), 16 floats M11,M12,...,M44.
 *   - there is NO CPU fallback: without a HIP device every compute entry point returns
 *     XRT_E_NO_DEVICE.
 */
#ifndef XRT_H
#define XRT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XRT_VERSION 203

/* error codes (C# shim: BUSY -> InvalidOperationException (RT:26-27,62-63),
 * INVALID_ARG -> ArgumentException (SO:123-124, MAT:85,97)) */
#define XRT_OK              0
#define XRT_E_INVALID_ARG  -1
#define XRT_E_BUSY         -2
#define XRT_E_NO_DEVICE    -3
#define XRT_E_OOM          -4
#define XRT_E_HIP          -5
#define XRT_E_RCCL         -6
#define XRT_E_INTERNAL     -7
#define XRT_E_UNSUPPORTED  -8
#define XRT_E_NOT_BUILT    -9

/* enums mirror MAT:12-23 */
#define XRT_FILTER_POINT     0
#define XRT_FILTER_BILINEAR  1
#define XRT_ADDRESS_CLAMP    0
#define XRT_ADDRESS_WRAP     1
#define XRT_ADDRESS_MIRROR   2

#define XRT_LIGHT_SPOT         0   /* SPOT: IsPositionable == true  */
#define XRT_LIGHT_DIRECTIONAL  1   /* DIR:  IsPositionable == false */

/* multisampling modes of xrt_render_opts.use_multisampling */
#define XRT_MS_OFF       0   /* RT:103-126 RenderInternal                                       */
#define XRT_MS_ADAPTIVE  1   /* RT:128-168,170-311 RenderInternalWithMultisampling (faithful)   */
#define XRT_MS_FIXED16   2   /* the 16 level-1 positions of RT:218-305, mean of 4 means of 4    */

typedef struct xrt_scene xrt_scene;   /* replaces an OctreeSpatialManager + its SceneObjects (OSM:35, SO:12) */

/* Ray of one ISpatialManager.GetRayIntersection query (ISM:15): Ray{Position,Direction} + the
 * `ignoreTriangle` reference identity expressed as (mesh id, index in Mesh.Triangles[]) — instances
 * of one Model share Triangle objects, so the pair is instance independent (SO:126-127, MO:290).
 * ignore_tri < 0 means null. 32 bytes. */
typedef struct xrt_ray {
    float   o[3];
    float   d[3];
    int32_t ignore_mesh;
    int32_t ignore_tri;
} xrt_ray;

/* IntersectionResult (OSM:11-33) with object / mesh / triangle references turned into indices.
 * tri  = position in Mesh.Triangles[];
 * leaf = DFS pre-order index (root = 0, children in 4i+2j+k order, MO:210-222) of the MeshOctree
 *        leaf in which the winning strict-'<' test happened (MO:294);
 * d    = OBJECT-space distance (OSM:373,450); w = worldPosition (OSM:443). 48 bytes. */
typedef struct xrt_hit {
    int32_t hit;
    int32_t object;
    int32_t mesh;
    int32_t tri;
    int32_t leaf;
    float   u, v, d;
    float   wx, wy, wz;
    int32_t reserved;   /* not part of the answer: scheduling feedback (rounds of the traversal loop the ray was in flight) */
} xrt_hit;

/* Material (MAT:25-69, 234-268). tex_argb = the locked Format32bppArgb bitmap (MAT:65): row-major,
 * top-down, 0xAARRGGBB, tex_width*tex_height words; may be NULL when use_texture == 0.
 * tex_pargb = Material.Texture.ColorData, the Format32bppPArgb (premultiplied) copy RayTracerTexture makes of the same
 * file (TEX:24-33): the bilinear filter reads ITS words (MAT:186-189) while the point sampler reads tex_argb (MAT:150).
 * The host has both arrays; NULL means "same as tex_argb" (true for every image without an alpha channel). */
typedef struct xrt_material {
    float           reflectiveness;
    int32_t         transparent;
    float           refraction_index;
    int32_t         interpolate_normals;
    int32_t         use_texture;
    int32_t         tex_width;
    int32_t         tex_height;
    int32_t         reserved;
    const uint32_t *tex_argb;
    const uint32_t *tex_pargb;
} xrt_material;

/* Inputs of RayTracer.Render ray generation (RT:395-397): Camera.View, Camera.Projection and the
 * GraphicsDevice.Viewport. */
typedef struct xrt_camera {
    float   view[16];
    float   proj[16];
    int32_t vp_x, vp_y, vp_width, vp_height;
    float   vp_min_depth, vp_max_depth;
} xrt_camera;

/* ILight (ILight.cs:9-15) — SpotLight (SPOT:12-35) or DirectionalLight (DIR:10-20). */
typedef struct xrt_light {
    int32_t kind;
    float   position[3];
    float   direction[3];
    float   color[3];
    float   intensity;
    float   spot_angle;       /* SPOT:19-27, radians */
    float   decay_exponent;   /* SPOT:33, default 1.3f */
} xrt_light;

/* The RayTracer properties a frame reads (RT:19-41) + how the frame is spread over GPUs. */
typedef struct xrt_render_opts {
    int32_t max_reflections;      /* RT:33  */
    int32_t use_multisampling;    /* RT:40, XRT_MS_* */
    int32_t multisample_quality;  /* RT:41  */
    int32_t address_mode;         /* RT:37, XRT_ADDRESS_* */
    int32_t filtering;            /* RT:36, XRT_FILTER_*  */
    int32_t shard_rank;           /* one process per GPU: this process renders tiles t with t % shard_count == shard_rank */
    int32_t shard_count;          /* 0 or 1 = whole frame */
    int32_t collect_stats;        /* 1: also run the (untimed) reference-work counting pass */
    /* One process, several GPUs -- the drop-in for the C# host, whose RenderInternal is ONE call on ONE thread
     * (RT:103-126).  0 or 1: the scene's device only.  N > 1: the frame's 64x8 tiles are dealt round-robin to devices
     * d, d+1, .. d+N-1 (d = the scene's device; the scene is replicated on first use), every device renders its
     * tiles, one grouped RCCL send/recv over xGMI gathers the tile buffers on device d, a de-tile kernel writes the
     * W*H frame there.  Needs N visible devices (XRT_E_NO_DEVICE) and shard_count <= 1 (XRT_E_INVALID_ARG); an RCCL
     * failure is XRT_E_RCCL. */
    int32_t n_gpus;
    /* n_gpus > 1 only.  1: cost-aware tile assignment -- every frame leaves the cost of its tiles (xrt_scene_tile_costs), and the next
     * frame of the same size deals the tiles to the devices longest-first (xrt_balance_tiles) instead of round-robin: the reference's only
     * parallel idea is DYNAMIC row stealing (RT:48-52, 105-120), which a static t % N assignment does not reproduce where rows differ in
     * cost (the horizon).  0: round-robin.  Scheduling only: the frame is the same. */
    int32_t balance_tiles;
    int32_t reserved[2];          /* zero */
} xrt_render_opts;

/* Exact work counters of the REFERENCE algorithm for the rays of one call and the
 * measured kernel times. Counts are those of the C# code path, not of the pruned GPU traversal. */
typedef struct xrt_stats {
    uint64_t rays_closest;        /* CastRay queries, RT:512 */
    uint64_t rays_shadow;         /* IsLightPathObstructed queries, RT:485 */
    uint64_t hits_closest;
    uint64_t hits_shadow;
    uint64_t scene_node_tests;    /* OSM:460 slab tests */
    uint64_t instance_visits;     /* OSM:349-364 ray transforms */
    uint64_t mesh_aabb_tests;     /* MESH:37 */
    uint64_t mesh_queries;        /* MO:259 calls */
    uint64_t node_tests;          /* MO:331 slab tests */
    uint64_t leaf_refs;           /* MO:288 loop iterations in visited buckets */
    uint64_t tri_tests;           /* RE:42 calls */
    uint64_t shaded_hits;
    uint64_t pixels;
    uint64_t algorithmic_bytes;   /* SURVEY §8d formula over the counters above */
    double   ms_total;            /* device time of the whole call */
    double   ms_intersect;        /* summed durations of the traversal kernel launches: first wave's start to last wave's end on the
                                     device clock for plain single-chunk frames (an event on a dispatch packet costs ~5 us, so those
                                     launches carry none; ~4 us per launch less than a profiler's dispatch-level duration), HIP events
                                     on the launches otherwise, or always with XRT_LAUNCH_EVENTS=1.  The start stamp is the clock of
                                     workgroup 0's first wave, which need not be the first wave of the launch to start (a 1024-block
                                     grid starts first to last within ~0.3-0.7 us): the figure may be short by that much per launch */
    uint32_t intersect_launches;
    uint32_t pieces;              /* the frame was rendered in this many concurrent pieces (halves on two streams, GPUs); 1 otherwise */
    uint64_t rays_traversed;      /* queries (of rays_closest + rays_shadow) that were handed to the traversal kernels: all of them except
                                     the primary rays the ray-generation kernel answers itself because they cannot reach the scene
                                     octree's root box (OSM:318-320 returns false for them before any node is visited).  Not a
                                     counter of the reference: how much of the ray count is real traversal work on this library */
    double   ms_intersect_longest;   /* the longest single traversal launch of the frame (of those ms_intersect sums): on every configuration here
                                     the launch that traces the primary rays -- the launch class a roofline of "the dominant kernel" is about */
    uint64_t mesh_queries_facing_away;   /* with collect_stats: of mesh_queries, those whose ray is inside the mesh octree's root box and meets ONLY back
                                     faces in the whole mesh (the box of the mesh's surface normals says so, with the margin of the leaf
                                     test): RE:48-51 rejects every triangle, the reference walks the octree to find that out, this
                                     library answers "no intersection" from the normal box where the test is made (rays leaving a
                                     surface; the shadow rays and reflections of a terrain seen from above are nearly all of this kind).
                                     Not a counter of the reference either */
} xrt_stats;

/* Flattened octree node for inspection by tests (mirrors the private CubeNode, MO:32-40 / OSM:37-48). */
typedef struct xrt_node_info {
    float   bmin[3];
    float   bmax[3];
    int32_t is_leaf;
    int32_t count;        /* containingObjects.Count */
    int32_t dfs_index;    /* pre-order index */
    int32_t depth;
    int32_t first_ref;    /* offset of this node's list in the ref array returned next to it (leaves only) */
    int32_t reserved;
} xrt_node_info;

/* ---- library ------------------------------------------------------------------------------- */
int         xrt_version(void);
const char *xrt_last_error(void);                      /* thread-local, never NULL */
int         xrt_device_count(int *count_out);          /* number of HIP devices visible */

/* ---- scene upload: replaces SceneObject / Mesh / Triangle / Material graphs (SO:117-134,
 *      MESH:17-32, TRI:14-25, MAT:45-69) -------------------------------------------------------- */
int xrt_scene_create(int device, xrt_scene **scene_out);
int xrt_scene_destroy(xrt_scene *scene);

/* One Mesh (MESH:9-40). v,n: ntri*9 floats (v1,v2,v3 / n1,n2,n3), uv: ntri*6, surf_n: ntri*3
 * (Triangle.surfaceNormal, TMP:199-203), color: ntri*4 (Triangle.color), bbox: Mesh.MeshBoundingBox
 * min xyz max xyz (TMP:244-307). */
int xrt_scene_add_mesh(xrt_scene *scene, const float *v, const float *n, const float *uv,
                       const float *surf_n, const float *color, int32_t ntri,
                       const xrt_material *material, const float bbox[6], int32_t *mesh_id_out);

/* One SceneObject (SO:12): its shared mesh list (SO:126-127), World / InverseWorld (SO:183-199),
 * BoundingBox (SO:131) and the un-normalised WorldBoundingBox (SO:195-196). */
int xrt_scene_add_object(xrt_scene *scene, const int32_t *mesh_ids, int32_t n_meshes,
                         const float world[16], const float inv_world[16],
                         const float bbox[6], const float world_bbox[6], int32_t *object_id_out);

/* Mesh.Init -> MeshOctree.Build for every mesh (MESH:27-32, MO:56-96, 204-236; threshold MO:42) and
 * OctreeSpatialManager.Build (OSM:64-113, 218-248; threshold OSM:50); then uploads to HBM.
 * Pass 0 for the reference defaults (50 / 20). */
int xrt_scene_build(xrt_scene *scene, int32_t mesh_threshold, int32_t scene_threshold);

/* Scene file -- what the reference keeps as .xnb content (the processor's Model.Tag, TMP:113-117: meshes with their materials):
 * xrt_scene_save writes the meshes, materials, texels and bodies of `scene` exactly as they were added (little-endian, version
 * tagged; format in scene_host.cpp); xrt_scene_load is xrt_scene_create followed by the same add_mesh / add_object calls.  The
 * octrees are not stored: call xrt_scene_build after loading (it reproduces them, MO:56-96 / OSM:64-113). */
int xrt_scene_save(const xrt_scene *scene, const char *path);
int xrt_scene_load(int device, const char *path, xrt_scene **scene_out);

/* Inspection of the built trees (test support). mesh_id >= 0: that mesh's MeshOctree; mesh_id == -1:
 * the scene octree (refs are object ids). Call with NULL arrays to get the counts. Nodes are returned
 * in DFS pre-order. */
int xrt_scene_get_tree(const xrt_scene *scene, int32_t mesh_id, xrt_node_info *nodes,
                       int64_t *n_nodes_inout, int32_t *refs, int64_t *n_refs_inout);

/* ---- seam 1: ISpatialManager.GetRayIntersection (ISM:15 = OSM:312-455), batched --------------
 * ignore_object (nullable, n entries) is the dead `Mesh ignoreObject` parameter (OSM:343 compares a
 * Mesh with an ISpatialBody and can never be equal); it is accepted and ignored. Host buffers.
 * Re-entrant like the reference's (its N render threads call it concurrently, RT:105-113): concurrent
 * seam-1 calls on one scene are serialised inside the library and every caller gets its own answers.
 * They may also run while a begin/end ticket is open, except with stats_out (the counters are shared
 * with the frames' counting pass): XRT_E_BUSY.  The scene must not be rebuilt or destroyed meanwhile. */
int xrt_scene_intersect(xrt_scene *scene, const xrt_ray *rays, const int32_t *ignore_object,
                        int64_t n, xrt_hit *hits_out, xrt_stats *stats_out /* nullable */);

/* Same with device pointers (HBM resident rays / hits), asynchronous on `stream` (a hipStream_t,
 * NULL = default stream). Nothing is copied; the caller synchronises. */
int xrt_scene_intersect_device(xrt_scene *scene, const void *d_rays, int64_t n, void *d_hits_out,
                               void *stream);

/* MeshOctree.GetRayIntersection of one mesh in object space (MO:259-326); hit.object = -1,
 * w = objectSpacePosition (MO:310-312). */
int xrt_mesh_intersect(xrt_scene *scene, int32_t mesh_id, const xrt_ray *rays, int64_t n,
                       xrt_hit *hits_out);

/* ---- seam 2: RayTracer.RenderInternal (RT:103-126) / RenderInternalWithMultisampling
 *      (RT:128-168) -------------------------------------------------------------------------------
 * Fills rgba_out[y*W + x] with the packed XNA Color (R in the low byte, RT:425). Blocking.
 * rgb_f32_out (nullable, W*H*3) receives the iteration-0 colorVector before packing (RT:705/726); in the supersampled
 * modes, which average PACKED colours (RT:309), it receives Color.ToVector3() of the pixel's final colour (written by
 * the resolve kernel).  Not available with n_gpus > 1 (XRT_E_UNSUPPORTED).
 * A second concurrent call on the same scene returns XRT_E_BUSY (RT:62-63). */
int xrt_render(xrt_scene *scene, const xrt_camera *camera, const xrt_light *lights, int32_t n_lights,
               const xrt_render_opts *opts, uint32_t *rgba_out, float *rgb_f32_out,
               xrt_stats *stats_out /* nullable */);

/* Pipelined form of xrt_render -- RenderAsync / RenderCompleted (RT:59-79, 431-437) with the frame still ending in the
 * host's Color[] (CurrentTarget.SetData, RT:122-123).  _begin enqueues the frame and its device-to-host copy and returns
 * a ticket; _end waits for both and fills stats_out (may be NULL).  Up to two frames may be open (tickets are shared with
 * xrt_render_device_begin), so the copy of frame i runs under the rendering of frame i+1.  rgba_out must stay valid
 * until _end; the copy overlaps only if the buffer is page-locked: xrt_host_register it once (the C# host pins its
 * renderTargetData with GCHandle first), otherwise the runtime stages it. */
int xrt_render_begin(xrt_scene *scene, const xrt_camera *camera, const xrt_light *lights, int32_t n_lights,
                     const xrt_render_opts *opts, uint32_t *rgba_out, int32_t *ticket_out);
int xrt_render_end(xrt_scene *scene, int32_t ticket, xrt_stats *stats_out);

/* Page-lock / release a caller-owned host buffer for DMA (hipHostRegister): the host Color[] of xrt_render[_begin]. */
int xrt_host_register(void *host_ptr, uint64_t bytes);
int xrt_host_unregister(void *host_ptr);

/* Same, writing to device memory. With opts->shard_count > 1 only this rank's tiles are rendered and
 * d_rgba_out receives them contiguously, tile after tile (see xrt_shard_layout); otherwise the full
 * W*H frame. The call enqueues on `stream` and synchronises it before returning (stats need it). */
int xrt_render_device(xrt_scene *scene, const xrt_camera *camera, const xrt_light *lights,
                      int32_t n_lights, const xrt_render_opts *opts, void *d_rgba_out,
                      void *stream, xrt_stats *stats_out /* nullable */);

/* Pipelined form of xrt_render_device.  _begin enqueues the frame and returns a ticket while the GPU is still working;
 * _end waits for that frame and fills stats_out (may be NULL).  Up to two frames may be in flight, so the host side of
 * frame i+1 (argument marshalling, ~25 launches) overlaps the GPU side of frame i -- the reference's game loop does the
 * same with RenderAsync (RT:92-104) and one frame of latency.  With stream == NULL each of the two frame contexts runs
 * on a stream of its own and the two frames also overlap ON the GPU: a launch of persistent waves leaves the machine
 * half empty while its last rays finish, and the other frame's kernels fill it (a given stream serialises them
 * instead).  d_rgba_out of a frame is complete when its _end returns; give the two frames in flight different output
 * buffers.  Frames that need host decisions between passes (adaptive supersampling, more than one chunk) finish inside
 * _begin and never overlap another frame; a frame with Transparent materials (a ray tree per pixel) of one chunk is enqueued
 * like a plain frame with optimistically sized ray buffers and, should a generation not fit, rendered again inside _end.  While a ticket is open every other call that renders
 * or mutates the scene returns XRT_E_BUSY. */
int xrt_render_device_begin(xrt_scene *scene, const xrt_camera *camera, const xrt_light *lights,
                            int32_t n_lights, const xrt_render_opts *opts, void *d_rgba_out, void *stream,
                            int32_t *ticket_out);
int xrt_render_device_end(xrt_scene *scene, int32_t ticket, xrt_stats *stats_out);

/* Image-tile shard geometry: tiles are XRT_TILE_W x XRT_TILE_H pixels, numbered row-major, tile t is
 * owned by rank t % shard_count and stored at slot t / shard_count of that rank's buffer.  Inside a tile's 512 words
 * the pixels are stored as eight 8x8-pixel blocks, left to right, Z-order inside a block (the order the wavefronts
 * trace them in); xrt_detile_device undoes it. */
#define XRT_TILE_W 64
#define XRT_TILE_H 8
int xrt_shard_layout(int32_t width, int32_t height, int32_t shard_count, int32_t *tiles_x_out,
                     int32_t *tiles_y_out, int32_t *tiles_per_rank_out);

/* ---- cost-aware tile assignment (one process per GPU; xrt_render_opts.balance_tiles does the same inside the library) ------------
 * The reference balances its render threads dynamically: each takes the next scanline from a shared counter (RT:48-52, 105-120).
 * Across GPUs the counterpart is a TILE TABLE made from the previous frame's costs, replacing the round-robin default:
 *   tile_of_slot[r * tiles_per_rank + s] = the tile (row-major number, as xrt_shard_layout) rank r renders at slot s of its buffer,
 *                                          or -1 for an unused slot; every tile of the frame appears exactly once.
 * Results never depend on the table (every tile is rendered exactly once, by the same code); only which rank renders it does. */

/* Install (tile_of_slot != NULL) or remove (NULL) the table of frames of width x height pixels rendered with shard_count shards.  The
 * table is copied.  While it is installed, xrt_render_device[_begin] calls of that size and shard count render the tiles of row
 * shard_rank of the table, tiles_per_rank slots (512 pixels each, unused slots zero) -- other sizes and shard counts keep the
 * round-robin layout.  XRT_E_INVALID_ARG unless every tile appears exactly once.  XRT_E_BUSY while a frame is in flight. */
int xrt_scene_set_tile_table(xrt_scene *scene, int32_t width, int32_t height, int32_t shard_count, int32_t tiles_per_rank,
                             const int32_t *tile_of_slot);

/* Cost of every tile (tiles_x * tiles_y floats, row-major tile order) of the frames of width x height pixels this scene object rendered
 * since the last call with reset != 0: device-clock ticks its wave packets spent on the tile's rays, all generations (tiles another
 * rank rendered: 0 -- sum the ranks' arrays).  Frames the per-lane kernel traces (one sample per pixel on one-body scenes, adaptive
 * supersampling, ray trees) report no costs: all zeros, and xrt_balance_tiles then returns the round-robin table.  XRT_E_BUSY while a
 * frame is in flight. */
int xrt_scene_tile_costs(xrt_scene *scene, int32_t width, int32_t height, float *cost_out, int32_t reset);

/* Greedy longest-processing-time-first assignment of the frame's tiles to shard_count ranks: tiles in descending order of cost (ties:
 * ascending tile number), each to the rank with the least cost so far that still has a free slot (ties: the lowest rank); the slots
 * of a rank are then filled in ascending tile order (neighbouring waves work on neighbouring tiles).  tiles_per_rank must be at least
 * ceil(tiles / shard_count) -- the slack above that is what lets a rank take more cheap tiles than another takes expensive ones
 * (xrt_shard_layout's value + 25 % is what bench.py uses).  A cost array without a positive finite entry gives the round-robin
 * table.  Pure host arithmetic, deterministic: every rank computes the same table from the same costs.  tile_of_slot_out:
 * shard_count * tiles_per_rank entries. */
int xrt_balance_tiles(int32_t width, int32_t height, int32_t shard_count, const float *tile_cost, int32_t tiles_per_rank,
                      int32_t *tile_of_slot_out);

/* xrt_detile_device for buffers rendered under a tile table: d_tile_of_slot is the table in DEVICE memory (shard_count *
 * tiles_per_rank int32). */
int xrt_detile_table_device(int32_t width, int32_t height, int32_t shard_count, int32_t tiles_per_rank, const void *d_tile_of_slot,
                            const void *d_gathered, int64_t rank_stride, void *d_rgba_out, void *stream);

/* De-tile the gathered per-rank buffers (rank-major: rank r's tiles_per_rank * 512 pixels start at pixel
 * r * rank_stride; rank_stride 0 = tiles_per_rank * 512, i.e. contiguous) into a W*H frame on the device (used on
 * rank 0 after the RCCL gather; a stride lets one gather carry the tiles of several frames). */
int xrt_detile_device(int32_t width, int32_t height, int32_t shard_count, const void *d_gathered, int64_t rank_stride,
                      void *d_rgba_out, void *stream);

/* Can the in-library multi-GPU path (xrt_render_opts.n_gpus > 1) load RCCL?  XRT_OK, or XRT_E_RCCL with the loader's message in
 * xrt_last_error().  Touches no device: a host can ask before it offers the option (the reference has no counterpart; the call
 * exists so that the failure a C# host would otherwise meet inside RenderInternal, RT:103-126, can be met up front). */
int xrt_rccl_probe(void);

/* Diagnostics of the split walks of long packets (no counterpart in the reference: its scanline threads never share a ray).  The traversal kernel
 * of one-body scenes lets a packet whose octree walk has outlasted its budget hand pending subtrees to other wavefronts; results never depend on it.
 * out[0] subtrees handed over, out[1] taken by another (or, at the end of a launch, the same) wavefront, out[2] packets that were split, out[3] packets
 * whose results were written by a taker -- counted on the scene's device since the library was loaded or since the last call with reset != 0. */
int xrt_split_stats(xrt_scene *scene, uint64_t out[4], int32_t reset);

/* RayTracer.Progress (RT:43-46): fraction of the frame's ray generations completed; callable from
 * another thread during xrt_render. */
float xrt_progress(const xrt_scene *scene);

/* Primary rays of RayTracer.Render (RT:410-421): two Viewport.Unproject per pixel, row-major. */
int xrt_generate_primary_rays(xrt_scene *scene, const xrt_camera *camera, xrt_ray *rays_out /* W*H */);

#ifdef __cplusplus
}
#endif
#endif /* XRT_H */
