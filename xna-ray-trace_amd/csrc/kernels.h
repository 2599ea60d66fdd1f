// kernels.h — launch interface between xrt_api.cpp and kernels.hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/xrt.h"
#include "traverse.h"

namespace xrt {

// device view of the shading-side arrays
struct ShadeView {
    const f4 *shade;              // SHADE_F4 per global triangle
    const MaterialRec *materials;
    const uint32_t *texels;
    const MeshRec *meshes;
    const LightRec *lights;
    int nLights;
    int addressMode, filtering;
};

struct IntersectArgs {
    const xrt_ray *rays;
    xrt_hit *hits;
    const int *index;   // optional compact list of ray indices to trace (others were answered already)
    const int *nDev;    // when non-null the ray count is (*nDev) * nMul, else n
    int nMul;
    int n;
    int nCap;           // upper bound of the ray count (buffer capacity); 0 = none
    unsigned *queue;    // zeroed work-queue head of this launch
    int mode, meshId;
    int firstBatch;     // rays of the static first batch of every wave
    int refillMin, nodeBurst, leafBurst;   // scheduling knobs of the persistent loop (defaults in xrt_api.cpp; XRT_TUNE overrides)
};

// reference-work counters accumulated on the device (same order as the head of xrt_stats)
enum { C_RAYS = 0, C_HITS, C_SCENE_NODES, C_INSTANCES, C_MESH_AABB, C_MESH_QUERIES, C_NODES, C_REFS, C_TRIS, C_COUNT };

constexpr int FLAG_MISS = 0, FLAG_HIT = 1, FLAG_TRANSPARENT = 2;

// Ray-tree bookkeeping of scenes with Transparent materials (RT:586-702): heap node id and the refraction
// index of the medium each ray travels in.  heap == 0: plain reflection chain, node == generation.
struct TreeArgs {
    int heap;
    int cap;         // capacity of the next generation's ray buffers
    int *overflow;   // set when a generation does not fit (the host retries the chunk with fewer paths)
    const int *rayNode;
    const float *rayRef;
    int *nextNode;
    float *nextRef;
    float *lvlAlpha;
};

struct FrameBuffers {
    xrt_ray *rays[2];
    int *rayPath[2];
    xrt_hit *hits;
    xrt_ray *shadowRays;
    xrt_hit *shadowHits;
    int *shadowSrc;
    f4 *lvlA, *lvlB;          // [(R+1) * P]
    uint32_t *sampleColor;    // [P]
    float *sampleF32;         // [3 P] or null
    int *cnt;                 // ray count per level  [R+2]
    int *scnt;                // shaded hits per level [R+1]
    unsigned *queues;         // one per intersect launch [2 (R+1)]
};

int  intersect_stack_capacity(int needed);   // smallest compiled capacity >= needed, or -1
void launch_intersect(const SceneView &S, const IntersectArgs &A, int stackNeeded, int gridBlocks, hipStream_t st, hipEvent_t e0 = nullptr,
                      hipEvent_t e1 = nullptr);
int  intersect_blocks_per_cu(int stackNeeded, int mode);
void launch_count(const SceneView &S, const IntersectArgs &A, unsigned long long *counters, hipStream_t st);
void launch_raygen(const RayGenParams &g, const SceneView &S, xrt_ray *rays, f4 *lvlB0, int *index, int *count, int P, long long pathBase, hipStream_t st, hipEvent_t startEvent = nullptr);
void launch_shade_a(const SceneView &S, const ShadeView &V, const xrt_ray *rays, const xrt_hit *hits, const int *nDev, int nHost,
                    const int *index, const int *rayPath, const int *rayNode, f4 *lvlB, xrt_ray *shadowRays, int *shadowSrc, int *scnt, int P, int level,
                    int cap, int *overflow, hipStream_t st);
void launch_shade_b(const SceneView &S, const ShadeView &V, const xrt_ray *rays, const xrt_hit *hits, const int *rayPath,
                    const int *scnt, const int *shadowSrc, const xrt_hit *shadowHits, f4 *lvlA, f4 *lvlB, xrt_ray *nextRays,
                    int *nextPath, int *nextCnt, int P, int level, int maxReflections, const TreeArgs &T, hipStream_t st);
void launch_compose_tree(const f4 *lvlA, const f4 *lvlB, const float *lvlAlpha, int count, int P, int maxReflections, uint32_t *sampleColor,
                         float *sampleF32, hipStream_t st);
// compose can write the framebuffer itself when there is one sample per pixel
struct ResolveArgs {
    int fused;
    RayGenParams g;
    long long pixelBase;
    uint32_t *out;
    float *outF32;
    // frame epilogue (single-chunk frames): the first cntWords ray counters go to host-visible memory and
    // zeroWords counter / queue words are cleared for the next frame -- no copy or fill commands on the stream
    int *cntSrc = nullptr;
    int *hostCnt = nullptr;
    int cntWords = 0, zeroWords = 0;
};
void launch_compose(const f4 *lvlA, const f4 *lvlB, int count, int P /* level stride */, int maxReflections, uint32_t *sampleColor, float *sampleF32,
                    const ResolveArgs &RA, hipStream_t st, hipEvent_t stopEvent = nullptr);
void launch_resolve(const RayGenParams &g, const uint32_t *sampleColor, const float *sampleF32, int pixels, long long pixelBase,
                    uint32_t *out, float *outF32, hipStream_t st);
void launch_ms_decide(const RayGenParams &g, const uint32_t *quadColor, const int *nQuadsDev, int nQuadsHost, long long pixelBase, int *childBase,
                      int *childMask, float *nextCx, float *nextCy, int *nextCount, hipStream_t st);
void launch_ms_fold(uint32_t *quadColor, const uint32_t *childColor, const int *childBase, const int *childMask, int n, hipStream_t st);
void launch_detile(int width, int height, int shardCount, int tilesPerRank, const uint32_t *gathered, uint32_t *out, hipStream_t st);

}  // namespace xrt
