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

// "long ray first" list of a producer kernel (kernels.hip, predict_heavy); list == nullptr: off
struct HeavyArgs {
    int *list = nullptr;    // indices (into the ray array) of the rays estimated to be long
    int *count = nullptr;
    float path = 0.0f;      // geometric estimate: long when the stretch inside the scene's root box exceeds this (0: off)
    // feedback estimate: what the ray of the same path and generation cost in a recent frame (kernels.hip, cost map)
    const unsigned *costMap = nullptr;   // [paths] of this generation: (frame number << 16) | cost
    unsigned epoch = 0;                  // this frame's number
    int costThreshold = 0;               // long when the remembered cost exceeds this
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
    const int *heavyIdx = nullptr, *nHeavy = nullptr;   // rays of segment 1 to take first (marked in their records)
    unsigned long long *debugTimes = nullptr;   // [3 * waves]: start / out-of-rays / exit clocks (100 MHz) per wave, development aid
    int coopMax = 32;   // at most this many lanes in a leaf: their triangle lists are dealt to the whole wave
    int batchMax = 64;  // largest guided batch a wave takes per queue atomic (multiple of 16)
    int heavyShift = 3; // one in 2^heavyShift of the first work items is a listed (long) ray
    int batchMin = 64;  // smallest guided batch (multiple of 16, 16 .. 64): below 64 the tail of a launch is dealt in part-filled waves
    int spreadMin = 4;  // a launch of fewer than 64 rays per wave is split evenly over all waves in multiples of this (4 .. 64)
    unsigned long long *stamps = nullptr;   // this launch's row of device-clock stamps (device_util.h), or null
    // optional second segment traced by the same launch: rays2[0 .. (*nDev2) * nMul2) -> hits2 (no index list)
    const xrt_ray *rays2 = nullptr;
    xrt_hit *hits2 = nullptr;
    const int *nDev2 = nullptr;
    int nMul2 = 0, nCap2 = 0;
    // Inside a frame: one word per ray says hit / miss (dense, what k_shade reads first) and a miss writes no 48-byte record
    // unless its scheduling-feedback word is wanted (missRecords); null: every ray gets its full record (seam 1)
    int *flags = nullptr, *flags2 = nullptr;
    int missRecords = 0;
    // where the answer of ray i goes: hits / flags [scatter[i]] (null: [i]).  The shadow rays of a frame whose producer answered some of them itself
    // (ShadeArgs::ae) are a compact list, their answers go back to (slot, light).
    const int *scatter = nullptr, *scatter2 = nullptr;
};

// Row of device-clock stamps of one traversal launch (device_util.h stamp_begin / stamp_end): [start, waves, end[waves]]
constexpr int STAMP_HEADER = 2, STAMP_SLOTS = 8192, STAMP_STRIDE = STAMP_HEADER + STAMP_SLOTS;

// k_packet (packet.hip): one wavefront traces 64 consecutive rays of a coherent population together -- a shared walk of the
// mesh octree with per-lane box / triangle tests.  Same answers as k_intersect (same tests, same arg-min rule).
// What part B of k_shade needs of a shaded hit (see below, ShadeArgs): world position, path, fragment normal, Reflectiveness.
struct alignas(16) SlotRec { float wx, wy, wz; int path; float nx, ny, nz, refl; };
struct PacketArgs {
    const xrt_ray *rays = nullptr;
    xrt_hit *hits = nullptr;
    const int *index = nullptr;   // optional compact list of ray indices
    const int *nDev = nullptr;    // when non-null the ray count is (*nDev) * nMul, else n
    int nMul = 1, n = 0, nCap = 0;
    unsigned *queue = nullptr;    // PACKET_QUEUE_WORDS zeroed work-queue heads of this launch (packet.hip: interleaved)
    int mode = MODE_SINGLE, meshId = 0;
    int unmark = 0;               // rays may carry the long-ray mark of their producer (device_util.h)
    int staticDiv = 4;            // 1/staticDiv of the packets are dealt statically (0: none)
    int grabMax = 2;              // most packets a wave takes per queue atomic
    int *flags = nullptr;         // inside a frame: hit / miss word per ray, no record for a miss (IntersectArgs::flags)
    int prefetch = 0;             // small launches (a tile shard, a late generation): a block's children and a leaf's triangle records are asked for with one vector load each,
                                  // a level ahead of the scalar loads that use them (packet.hip pk_prefetch) -- their walks go through parts of the tree no other wave keeps warm
    int bundle = 1;               // one-body scenes: big leaves are scanned through the bundle prefilter (packet.hip; XRT_PK_BUNDLE=0: run by run as in round 3)
    int cullMin = 4;              // leaves of at least this many references are tested against their tight box first (the test costs about two triangles)
    unsigned long long *stamps = nullptr;   // this launch's row of device-clock stamps (device_util.h), or null
    // optional second segment traced by the same launch (the shadow rays of generation k-1 beside the closest-hit rays of generation k, as
    // IntersectArgs::rays2): packets nPk1 .. are rays2[0 .. (*nDev2) * nMul2) -> hits2 / flags2 (no index list); one launch's tail instead of two
    const xrt_ray *rays2 = nullptr;
    xrt_hit *hits2 = nullptr;
    int *flags2 = nullptr;
    const int *nDev2 = nullptr;
    int nMul2 = 0, nCap2 = 0;
    // Cost of the frame's tiles (xrt.h xrt_scene_tile_costs): every packet adds the device-clock ticks it took to the tile of its first ray --
    // tileCost[(tileBase + path) >> tileShift], path = pathOf1[ray] (or slotOf1[ray / nL1].path, or the ray's index) for the first segment, slotOf2[ray / nL2].path for the second.
    const int *scatter = nullptr, *scatter2 = nullptr;   // as IntersectArgs::scatter
    unsigned *tileCost = nullptr;
    int tileShift = 9, nL1 = 1, nL2 = 1, tileBase = 0;   // (tileBase: first path of this launch's part of the frame)
    const int *pathOf1 = nullptr;
    const SlotRec *slotOf1 = nullptr;   // (a launch whose FIRST segment is shadow rays)
    const SlotRec *slotOf2 = nullptr;
    // Split walks (packet.hip "split walks"): a walk that has outlasted splitBudget ticks of the 100 MHz device clock hands the pending subtrees of its
    // upper levels to other waves as ITEMS of this launch; the last participant of a packet merges the partial answers (the arg-min rule of DESIGN.md §3
    // does not depend on who visited what).  splitCtl == nullptr: off.
    unsigned *splitCtl = nullptr;      // 128-byte aligned, zeroed with the queue heads: a 128-byte line per XCD x -- [32 x] items reserved, [32 x + 1] items taken in the XCD's
                                       // share of the arena -- and on a ninth line [256] records allocated.  An item is taken by a wave of the XCD it was given on (a packet's
                                       // participants share an L2), and every wave asking for work asks one of eight addresses instead of one
    unsigned *splitItems = nullptr;    // SPLIT_ITEM_WORDS per item
    unsigned *splitRecs = nullptr;     // SPLIT_REC_WORDS per packet that was split
    int splitNI = 0, splitNR = 0;      // capacities
    unsigned splitSerial = 0;          // this launch's number: the value of an item's `ready` word (the arena is never cleared)
    int splitBudget = 0, splitBudgetItem = 0;   // ticks a packet / an item may walk before it looks for pending subtrees to hand over
    // ... and which packets are split EAGERLY: a packet remembers the block entries it made (a split packet keeps the larger of that and what it remembered) in
    // splitCost[packet number], and the same packet of the context's next frame, if that is above splitLong, hands its pending subtrees over every splitBudgetLong
    // block entries from the start, as do the takers of its items.  (Waiting for a walk to PROVE long costs half of it; splitting every walk early is speculation --
    // the far siblings of a ray that is about to find a near hit would have been pruned; and a cost in TICKS is contagious: the speculative work slows every packet,
    // more of them count as long.  profiles/r04/split_walks.txt.)
    unsigned *splitCost = nullptr;
    int splitLong = 0, splitBudgetLong = 0;
};
constexpr int PACKET_QUEUE_HEADS = 8, PACKET_HEAD_STRIDE = 64;   // every head on a 256-byte line of its own: atomics on one line serialise whatever the word
constexpr int PACKET_SPLIT_WORDS = 320;                          // ... and behind the heads the lines of the split-walk counters (PacketArgs::splitCtl: 9 x 128 bytes, aligned)
constexpr int PACKET_QUEUE_WORDS = PACKET_QUEUE_HEADS * PACKET_HEAD_STRIDE + PACKET_SPLIT_WORDS;
// an item: [0] ready (== PacketArgs::splitSerial), [1] packet, [2] record, [3] block, [4] pending children (front-to-back bits), [5..6] lanes, [7] the taker's budget (ticks), [8] its eager interval (block entries), [32..95] the lanes' accepted
// children of that block (cb), [96..] the lanes' best answers so far, 7 words each ([word][lane]: found, key, distance, u, v, reference, leaf) -- on the way in what the
// giver had when it gave, on the way out what the taker has; a record: [0] units outstanding (the packet itself + its items), [1] items -- a line that only atomics
// touch --, [32..63] their indices, [64..] the packet's own partial answers
constexpr int SPLIT_HEAD_WORDS = 32, SPLIT_PART_WORDS = 7 * 64, SPLIT_ITEM_WORDS = SPLIT_HEAD_WORDS + 64 + SPLIT_PART_WORDS, SPLIT_REC_HEAD = 64, SPLIT_REC_WORDS = SPLIT_REC_HEAD + SPLIT_PART_WORDS;
constexpr int SPLIT_REC_ITEMS = 32;
int  packet_split_stats(unsigned long long out[4], bool reset);   // (packet.hip g_splitStats on the current device)
bool packet_supported(int mode, int meshDepth, int sceneDepth);
int  packet_blocks_per_cu(int mode);
void launch_packet(const SceneView &S, const PacketArgs &A, int gridBlocks, hipStream_t st, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);

// reference-work counters accumulated on the device (same order as the head of xrt_stats)
enum { C_RAYS = 0, C_HITS, C_SCENE_NODES, C_INSTANCES, C_MESH_AABB, C_MESH_QUERIES, C_NODES, C_REFS, C_TRIS, C_MESH_AWAY, C_COUNT };   // (C_MESH_AWAY: not a counter of the reference, see xrt_stats.mesh_queries_facing_away)

constexpr int FLAG_MISS = 0, FLAG_HIT = 1, FLAG_TRANSPARENT = 2;

// One shaded hit of a generation: which ray found it, the path (pixel sample) it belongs to and its node in the
// ray tree (== generation for a plain reflection chain).
// What part B of k_shade needs of a shaded hit, left by part A at the hit's slot (the slots of a generation are dense, so both sides stream
// 32 bytes per hit; the level record lvlA[node][path] is written once, by part B): world position, path, fragment normal, Reflectiveness.

// k_shade: part A works on generation `level`, part B on generation level-1 (kernels.hip).
struct ShadeArgs {
    int level, doA, doB;
    int maxReflections, P, heap;
    int *overflow;            // set when a generation does not fit (the host retries the chunk with fewer paths)
    // part A
    const xrt_ray *rays; const xrt_hit *hits;
    const int *nDev; int nHost, cap;
    const int *index, *rayPath, *rayNode; const float *rayRef;
    SlotRec *slotOut; int *slotNodeOut; int *scnt; int shadowCap; xrt_ray *shadowRays;   // slotNodeOut: ray-tree frames only (a chain's node is its level)
    xrt_ray *nextRays; int *nextPath, *nextNode; float *nextRef; int *nextCnt; int nextCap;
    // part B
    const SlotRec *slotPrev; const int *slotNodePrev; const int *scntPrev; const xrt_hit *shadowHits;
    f4 *lvlA, *lvlB; float *lvlAlpha;
    LvlMap lvl;        // where a path's level records live (xrt_core.h lvl_at; P is the stride of a level)
    HeavyArgs heavy;   // for the rays of generation level+1
    unsigned *costOut = nullptr;   // cost map of generation `level` (part A writes what its rays cost), tagged with `epoch`
    unsigned epoch = 0;
    // hit / miss words of `hits` and `shadowHits` (IntersectArgs::flags): the 48-byte record of a miss does not exist
    const int *hitFlags = nullptr, *shadowFlags = nullptr;
    // Answered at emission (one-body scenes, plain frames): part A asks, for every ray it is about to emit, what the traversal kernel would ask
    // first once it has the object-space ray -- do ALL the mesh's triangles face away from it (traverse.h all_back_facing on MeshRec::nbMin/nbMax)?
    // RE:48-51 then rejects every triangle, the query's answer is "no intersection", and the ray is not emitted: a shadow ray's hit / miss word is
    // written here (shadowFlagsOut), a reflection's level record too (the path ends, RT:729-733).  The rays that are left form COMPACT lists: shadow
    // rays at shadowRays[0 .. *shadowCnt) with shadowOut[i] = slot * nLights + light (where their answers go), reflections at nextRays[0 .. *nextCnt).
    int ae = 0;
    int *shadowCnt = nullptr, *shadowOut = nullptr, *shadowFlagsOut = nullptr;
};

int  intersect_stack_capacity(int needed);   // smallest compiled capacity >= needed, or -1
void launch_intersect(const SceneView &S, const IntersectArgs &A, int stackNeeded, int gridBlocks, hipStream_t st, hipEvent_t e0 = nullptr,
                      hipEvent_t e1 = nullptr);
int  intersect_blocks_per_cu(int stackNeeded, int mode);
void launch_count(const SceneView &S, const IntersectArgs &A, unsigned long long *counters, hipStream_t st);
void launch_raygen(const RayGenParams &g, const SceneView &S, xrt_ray *rays, f4 *lvlB0, int *index, int *count, int P, long long pathBase,
                   const HeavyArgs &H, hipStream_t st, hipEvent_t startEvent = nullptr, int liveCap = 0x7fffffff);
void launch_shade(const SceneView &S, const ShadeView &V, const ShadeArgs &X, hipStream_t st, int blocks = 1024, int threads = 1024);   // (any grid is correct: grid-stride loops)
// Frame epilogue of the compose kernels: the clock stamps of traversal launches row0 .. row1-1 are folded into (start, latest
// end) pairs in host-visible memory (device_util.h)
struct StampFold {
    const unsigned long long *src = nullptr;
    unsigned long long *host = nullptr;
    int row0 = 0, row1 = 0;
};
// Frame epilogue of a single-chunk frame (block 0 of the compose kernel): the first cntWords ray counters -- and, for a ray-tree
// frame, the word that says a generation overflowed its buffers, as hostCnt[cntWords] -- go to host-visible memory, and zeroWords
// counter / queue words and that flag are cleared for the next frame: no copy or fill commands on the stream
struct FrameEpilogue {
    int *cntSrc = nullptr;
    int *hostCnt = nullptr;
    int cntWords = 0, zeroWords = 0;
    int *flagSrc = nullptr;
    int zeroFrom = 0;   // words [zeroFrom, zeroWords) are cleared (an adaptive frame keeps its level counts for the fold kernels)
};
void launch_compose_tree(const f4 *lvlA, const f4 *lvlB, const float *lvlAlpha, int count, int P, int maxReflections, uint32_t *sampleColor,
                         float *sampleF32, const StampFold &stamps, const FrameEpilogue &epilogue, hipStream_t st);
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
    int cntWords = 0, zeroWords = 0, zeroFrom = 0;
    StampFold stamps;
};
void launch_compose(const f4 *lvlA, const f4 *lvlB, int count, int P /* level stride */, int maxReflections, uint32_t *sampleColor, float *sampleF32,
                    const ResolveArgs &RA, hipStream_t st, hipEvent_t stopEvent = nullptr);
void launch_resolve(const RayGenParams &g, const uint32_t *sampleColor, const float *sampleF32, int pixels, long long pixelBase,
                    uint32_t *out, float *outF32, hipStream_t st, int *zeroPtr = nullptr, int zeroN = 0);
void launch_ms_decide(const RayGenParams &g, const uint32_t *quadColor, const int *nQuadsDev, int nQuadsHost, long long pixelBase, int *childBase,
                      int *childMask, float *nextCx, float *nextCy, int *nextCount, hipStream_t st, int nextCap = 0, int *overflow = nullptr);
void launch_ms_fold(uint32_t *quadColor, const uint32_t *childColor, const int *childBase, const int *childMask, int n, hipStream_t st,
                    const int *nDev = nullptr);
void launch_detile(int width, int height, int shardCount, int tilesPerRank, const uint32_t *gathered, long long rankStride, uint32_t *out,
                   hipStream_t st, const int *table = nullptr);

}  // namespace xrt
