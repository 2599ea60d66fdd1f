// kernels.hip — hand-written HIP kernels for gfx950 (MI355X, wave64).
//
//   k_intersect   the hot kernel: persistent wavefronts pull rays from a global queue; one lane = one
//                 ISpatialManager.GetRayIntersection query (OSM:312 -> MO:259 -> RE:42); per-lane octree
//                 stack in LDS ([level][lane], bank == lane, conflict free); lanes that finish are refilled
//                 in groups chosen with __ballot so the wave stays populated (active-lane compaction).
//                 One launch traces two ray arrays (closest-hit rays of a generation + shadow rays of the
//                 previous one) and takes the rays listed as long first; the scene-level half of a query
//                 lives in LDS; children a ray misses are filtered eight at a time; with few lanes in a leaf
//                 their triangle lists are dealt to the whole wave.  Branchy scalar fp32 — no MFMA by design.
//   k_count       the REFERENCE algorithm's work counters, untimed.
//   k_raygen      RayTracer.Render ray generation (RT:410-421) in 64x8 tile order.
//   k_shade       CastRay shading (RT:516-584, 708-727), IsLightPathObstructed (RT:465-502), lights: hits of
//                 generation k -> shadow rays + rays of generation k+1, shadow answers of k-1 -> level records.
//   k_compose     the recursion's return path: per-level RGBA8 quantisation (RT:584,705,726,732).
//   k_resolve     supersample averaging (RT:309) and the framebuffer write (RT:425).
//
// Built with -ffp-contract=off: every result must be bit-identical to the oracle.
#include "kernels.h"
#include "device_util.h"

#include <hip/hip_ext.h>

namespace xrt {

struct LdsStack {
    unsigned *base;   // &stk[wave][0][lane]
    __device__ __forceinline__ unsigned get(int i) const { return base[i * 64]; }
    __device__ __forceinline__ void set(int i, unsigned v) { base[i * 64] = v; }
};

__device__ __forceinline__ bool predict_heavy(const SceneView &S, v3 o, v3 d, float heavyPath) {
    const f4 lo = S.snodes[0], hi = S.snodes[1];
    float tmin = 0.0f, tmax = FLT_MAX;
    const float ox[3] = {o.x, o.y, o.z}, dx[3] = {d.x, d.y, d.z}, bl[3] = {lo.x, lo.y, lo.z}, bh[3] = {hi.x, hi.y, hi.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (fabsf(dx[k]) > 1e-6f) {
            const float inv = __frcp_rn(dx[k]);
            const float t1 = (bl[k] - ox[k]) * inv, t2 = (bh[k] - ox[k]) * inv;
            tmin = fmaxf(tmin, fminf(t1, t2));
            tmax = fminf(tmax, fmaxf(t1, t2));
        }
    }
    return (tmax - tmin) > heavyPath;   // NaN compares false: not listed
}
// Feedback: a frame remembers, per path and generation, how many rounds of the traversal loop its ray was in flight
// (the hit record's reserved word -> k_shade -> the cost map).  The next frames list a ray as long when the ray that
// stood in its place cost more than a threshold the host steers to a few percent of the rays.  A camera that moves
// makes the memory slightly stale, never wrong: the list only decides which rays start first.
__device__ __forceinline__ bool long_ray(const SceneView &S, const HeavyArgs &H, int path, v3 o, v3 d) {
    if (H.costMap) {
        const unsigned e = H.costMap[path];
        if (((H.epoch - (e >> 16)) & 0xffffu) - 1u < 4u) return (int)(e & 0xffffu) > H.costThreshold;   // written one to four frames ago
    }
    return H.path > 0.0f && predict_heavy(S, o, d, H.path);
}

// wave64 inclusive scans on the DPP network (row shifts, then the two row broadcasts): no LDS round trip
__device__ __forceinline__ int wave_scan_add(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int wave_scan_max(int v) {   // values >= 0
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));
    return v;
}

// Stream compaction with ONE global atomic per 1024-thread block and round (atomics on one word serialise at
// ~11 ns each, MI355X_MICROARCH.md "dequeue"): waves post their ballot counts to LDS, the first wave scans them on the
// DPP network and reserves the block's range, every flagged lane gets base + (lanes of earlier waves) + (earlier
// lanes of its wave).  Must be called by all threads of the block.
constexpr int APPEND_BLOCK = 1024;
__device__ __forceinline__ int block_append(int *counter, bool flag, int *ldsCounts /* [17] */) {
    const unsigned long long m = __ballot(flag);
    const int wave = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6), lane = lane_id();
    if (lane == 0) ldsCounts[wave] = (int)__popcll(m);
    __syncthreads();
    if (wave == 0) {
        const int c = lane < nw ? ldsCounts[lane] : 0;
        const int incl = wave_scan_add(c);
        const int total = __builtin_amdgcn_readlane(incl, 63);
        int base = 0;
        if (lane == 0 && total) base = atomicAdd(counter, total);
        base = __builtin_amdgcn_readfirstlane(base);
        if (lane < nw) ldsCounts[lane] = base + incl - c;
    }
    __syncthreads();
    const int slot = ldsCounts[wave] + lanes_below(m);
    __syncthreads();   // ldsCounts is reused by the next round
    return slot;
}

// ---- the hot kernel ------------------------------------------------------------------------------------------

// Scene mode keeps the scene-level half of every lane's query (SceneLane, 27 words: world ray, scene cursor, best answer
// so far) in LDS: only advance_scene and the final answer touch it, through the proxies of device_util.h (LdsField, LdsRay), one
// word at a time where it is needed.  The node and leaf loops then fit the 128-register budget of four waves per SIMD without
// scratch traffic.  Layout [word][lane] -- one bank per lane, like the traversal stack.
constexpr int PARK_WORDS = 27;
struct ParkedScene {
    LdsRay w;
    LdsField<int> sblk, smask, ssp, sRef, sRefEnd;
    LdsField<float> sKey;
    LdsField<int> obj, mPtr, mEnd, sfound;
    LdsField<float> sbKey, sbD, sbU, sbV;
    LdsField<int> sbRef, sbLeaf, sbObj, sbMesh;
    __device__ __forceinline__ explicit ParkedScene(unsigned *b)   // b = &park[wave][0][lane]
        : w{b}, sblk{b + 9 * 64}, smask{b + 10 * 64}, ssp{b + 11 * 64}, sRef{b + 12 * 64}, sRefEnd{b + 13 * 64}, sKey{b + 14 * 64},
          obj{b + 15 * 64}, mPtr{b + 16 * 64}, mEnd{b + 17 * 64}, sfound{b + 18 * 64}, sbKey{b + 19 * 64}, sbD{b + 20 * 64},
          sbU{b + 21 * 64}, sbV{b + 22 * 64}, sbRef{b + 23 * 64}, sbLeaf{b + 24 * 64}, sbObj{b + 25 * 64}, sbMesh{b + 26 * 64} {}
};
// what the kernel instantiates the scene-level functions with: the LDS proxies in scene mode, plain registers otherwise
template <int M> struct SceneHome { using type = SceneLane; };
template <> struct SceneHome<MODE_SCENE> { using type = ParkedScene; };

// ---- leaf phase, wave-cooperative ------------------------------------------------------------------------------------
// The triangle lists of the lanes that are in a leaf (MO:288-304) are laid end to end and dealt to the 64 lanes of the
// wave, one reference per lane and step: slot s tests reference ref_i + (s - first_i) of owner lane i with the owner's
// ray.  A lane alone in a 40-triangle leaf is done in one step instead of twenty, and a step's arithmetic runs with
// as many lanes as there are references left, not as there are lanes in a leaf.  Candidates go back to their owner in
// slot order, which is list order per owner, so MO:293's strict '<' keeps the first of equal distances.
constexpr int COOP_ROUNDS = 2;   // references per lane and cooperative step
__device__ __forceinline__ void coop_leaf_step(Lane &L, const SceneView &S, unsigned char *own /* [64 * COOP_ROUNDS] of this wave */) {
    constexpr int CAP = 64 * COOP_ROUNDS;
    const int lane = lane_id();
    const bool inLeaf = L.state == ST_LEAF;
    const int cnt = inLeaf ? L.refEnd - L.ref : 0;
    const int incl = wave_scan_add(cnt);
    const int excl = incl - cnt;
    const int total = __builtin_amdgcn_readlane(incl, 63);
    // owner of slot s: the last lane whose list starts at or before s -- every owner drops its lane number at its
    // first slot, a running maximum spreads it over the slots that follow
    // (lanes talk to each other through `own`: wave-level fences, or the compiler forwards a lane's own zero to its read)
#pragma unroll
    for (int r = 0; r < COOP_ROUNDS; r++) own[lane + 64 * r] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (cnt > 0 && excl < CAP) own[excl] = (unsigned char)lane;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int used = inLeaf ? ((CAP - excl) <= 0 ? 0 : (cnt < CAP - excl ? cnt : CAP - excl)) : 0;
    int carry = 0, spec = 0;
#pragma unroll
    for (int r = 0; r < COOP_ROUNDS; r++) {
        if (r > 0 && total <= 64 * r) break;   // wave-uniform
        const int s = lane + 64 * r;
        int owner = wave_scan_max((int)own[s]);
        owner = owner > carry ? owner : carry;
        carry = __builtin_amdgcn_readlane(owner, 63);
        const bool valid = s < total;
        const int ref = __shfl(L.ref, owner) + (s - __shfl(excl, owner));
        const v3 o = mk(__shfl(L.r.o.x, owner), __shfl(L.r.o.y, owner), __shfl(L.r.o.z, owner));
        const v3 d = mk(__shfl(L.r.d.x, owner), __shfl(L.r.d.y, owner), __shfl(L.r.d.z, owner));
        const int ignoreId = __shfl(L.ignoreId, owner);
        bool f = false, pass = false;
        float u = 0, v = 0, t = 0;
        if (valid) {
            const f4 n = S.refN[ref];
            const g3 ga = S.refG[3 * (size_t)ref], gb = S.refG[3 * (size_t)ref + 1], gc = S.refG[3 * (size_t)ref + 2];
            f = !(facing(mk(n.x, n.y, n.z), d) > 0.0f) && f2i(n.w) != ignoreId;   // RE:48-51, MO:290
            if (f) pass = tri_test_front(o, d, mk(ga.x, ga.y, ga.z), mk(gb.x, gb.y, gb.z), mk(gc.x, gc.y, gc.z), u, v, t) && t < FLT_MAX;   // MO:293
        }
        unsigned long long pm = __ballot(pass);
        while (pm) {   // rare: hand each passing candidate to its owner, in list order
            const int ps = __builtin_ctzll(pm);
            pm &= pm - 1;
            const int so = __builtin_amdgcn_readlane(owner, ps);
            const int sr = __builtin_amdgcn_readlane(ref, ps);
            const float su = i2f(__builtin_amdgcn_readlane(f2i(u), ps)), sv = i2f(__builtin_amdgcn_readlane(f2i(v), ps)),
                        sd = i2f(__builtin_amdgcn_readlane(f2i(t), ps));
            if (lane == so) leaf_candidate(L, S, sr, -2, true, su, sv, sd);   // (the ignored triangle was filtered above: -2 matches no id)
        }
        // did one of this lane's own references face its ray?  (hint for the whole-leaf back-face skip)
        const unsigned long long fm = __ballot(f);
        const int a0 = excl - 64 * r, b0 = excl + used - 64 * r;
        const int sa = a0 < 0 ? 0 : a0, sb = b0 > 64 ? 64 : b0;
        if (sa < sb) {
            const unsigned long long hi = sb >= 64 ? ~0ull : ((1ull << sb) - 1ull), lo = (1ull << sa) - 1ull;
            if (fm & hi & ~lo) spec = 1;
        }
    }
    if (used > 0) {
        L.spec = spec;
        L.ref += used;
        if (L.ref >= L.refEnd) L.state = ST_NODE;
    }
}

// LDS that only some variants need is declared where only they instantiate it: the deepest stack (40 levels, 40 KB a
// block) leaves no room for anything else at four blocks per CU.
template <int M> __device__ __forceinline__ unsigned *park_memory() {
    if constexpr (M == MODE_SCENE) { __shared__ unsigned mem[4 * PARK_WORDS * 64]; return mem; }
    else return nullptr;
}
template <int T> __device__ __forceinline__ unsigned char *coop_memory() {
    if constexpr (T < 40) { __shared__ unsigned char mem[4 * 64 * COOP_ROUNDS]; return mem; }
    else return nullptr;
}

template <int T, int M>
__global__ __launch_bounds__(256, (M == MODE_SCENE) ? 4 : 5) void k_intersect(SceneView S, IntersectArgs A) {
    __shared__ unsigned stk[4 * T * 64];
    stamp_begin(A.stamps);
    unsigned *const parkMem = park_memory<M>();
    unsigned char *const coopOwn = coop_memory<T>();
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    LdsStack st{&stk[wave * T * 64 + lane]};
    SceneLane plainScene;
    auto &&C = [&]() -> decltype(auto) {
        if constexpr (M == MODE_SCENE) return ParkedScene(&parkMem[wave * PARK_WORDS * 64 + lane]);
        else return (plainScene);
    }();
    int n1 = A.nDev ? (*A.nDev) * A.nMul : A.n;
    if (A.nCap > 0 && n1 > A.nCap) n1 = A.nCap;   // an overflowed generation is discarded by the host; stay inside the buffers
    // optional second segment (the shadow rays of the previous generation ride in the same launch as this generation's
    // closest-hit rays): ray g >= n1 is rays2[g - n1]
    int n2 = A.nDev2 ? (*A.nDev2) * A.nMul2 : 0;
    if (n2 > A.nCap2) n2 = A.nCap2;
    // Rays of segment 1 that their producer listed as long are taken first -- but not 64 to a wave: long rays come in clusters
    // (neighbouring pixels), a wave full of them takes five times as long as one of them alone (C3: 64 rays 320-450 us, one ray
    // 80 us) and ends the launch long after the other waves have left.  So every 2^heavyShift-th of the first work items is a
    // listed ray, the others are rays in array order: item w < nH << hs is listed ray w >> hs when w % 2^hs == 0.
    int nH = A.nHeavy ? *A.nHeavy : 0;
    if (nH > n1) nH = n1;
    const int n = nH + n1 + n2;
    int hs = A.heavyShift;
    while (hs > 0 && ((long long)nH << hs) > (long long)n) hs--;
    const int hRegion = nH << hs, hMask = (1 << hs) - 1;
    Lane L;
    L.state = ST_IDLE;
    // Work distribution.  The first 64 rays of every wave are static (no atomic: a grid-wide burst on one word
    // costs ~11 ns each); consecutive static batches go to different workgroups, hence different CUs, because
    // image regions with expensive rays are contiguous in the ray stream and must not pile up on one CU.
    // Further batches come from the queue, guided (large while much work remains, 64 rays near the end), and
    // the next one is requested while the current one is being traced so the atomic's latency is hidden.
    const int nWaves = (int)gridDim.x * 4;
    // static share: at most A.firstBatch rays (64 for deep octrees where dynamic balance matters, 256 for trivial
    // scenes where queue traffic matters), but no more than an even split of the launch over the resident waves
    // (a small launch is spread over all waves, spreadMin rays at least, rather than packed 64 to a wave: a batch takes as
    // long as its slowest ray, slow rays come in clusters, and idle waves cost nothing)
    const int bmin = A.batchMin < 16 ? 16 : (A.batchMin > 64 ? 64 : (A.batchMin & ~15));   // smallest guided batch
    const int smin = A.spreadMin < 4 ? 4 : (A.spreadMin > 64 ? 64 : (A.spreadMin & ~3));    // granule of the static share
    const int even = ((n + nWaves - 1) / nWaves + smin - 1) & ~(smin - 1);
    const int first = even < smin ? smin : (even < A.firstBatch ? even : A.firstBatch);
    const unsigned qOffset = (unsigned)(nWaves * first);
    int batchNext = (wave * (int)gridDim.x + (int)blockIdx.x) * first;
    int batchEnd = min(batchNext + first, n);
    bool exhausted = batchNext >= n;
#ifdef XRT_WAVE_TIMES   // development aid (make WAVE_TIMES=1): when this wave ran out of new rays / ended
    unsigned long long tStart = 0, tDry = 0;
    if (A.debugTimes) tStart = wall_clock64();
#endif
    auto guided = [&](int done) { int c = (n - done) / (nWaves * 2); c &= ~15; const int lo = A.batchMax < bmin ? A.batchMax : bmin; return c < lo ? lo : (c > A.batchMax ? A.batchMax : c); };
    unsigned pfBase = 0;
    int pfChunk = 0;
    if (!exhausted && (int)qOffset < n) {   // there is dynamic work beyond the static batches
        pfChunk = guided((int)qOffset);
        if (lane == 0) pfBase = atomicAdd(A.queue, (unsigned)pfChunk);
    }
    for (;;) {
        if (L.state != ST_IDLE) L.cost++;
        const unsigned long long idle = __ballot(L.state == ST_IDLE);
        if (idle != 0ull) {
            const int nIdle = __popcll(idle);
            if (!exhausted && (nIdle >= A.refillMin || idle == ~0ull)) {
                if (batchNext >= batchEnd) {
                    const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)pfBase) + qOffset;
                    batchNext = (int)base;
                    batchEnd = min((int)base + pfChunk, n);
                    if (pfChunk == 0 || (int)base >= n || (int)base < 0) {
                        exhausted = true;
#ifdef XRT_WAVE_TIMES
                        if (A.debugTimes) tDry = wall_clock64();
#endif
                    }
                    else {
                        pfChunk = guided(batchEnd);
                        if (lane == 0) pfBase = atomicAdd(A.queue, (unsigned)pfChunk);
                    }
                }
                if (!exhausted) {
                    const int take = min(nIdle, batchEnd - batchNext);
                    const int rank = lanes_below(idle);
                    if (L.state == ST_IDLE && rank < take) {
                        const int w = batchNext + rank;
                        const bool listed = w < hRegion && (w & hMask) == 0;
                        const int g = w < hRegion ? w - (w >> hs) - 1 : w - nH;   // (unlisted items only)
                        int idx;
                        const xrt_ray *src;
                        if (listed) { idx = A.heavyIdx[w >> hs]; src = A.rays + idx; }
                        else if (g < n1) { idx = A.index ? A.index[g] : g; src = A.rays + idx; }
                        else { src = A.rays2 + (g - n1); idx = ~(g - n1); }   // answers of segment 2 go to hits2
                        v3 o, d; int im, it;
                        load_ray(src, o, d, im, it);
                        bool skip = false;
                        if (A.nHeavy && idx >= 0 && heavy_marked(it)) { it ^= HEAVY_BIT; skip = !listed; }   // listed: traced as one of the first work items
                        if (skip) {}
                        else if (im == DEAD_RAY) { L.rayIndex = idx; L.cost = 0; C.sfound = 0; L.mfound = 0; L.state = ST_FINISH; }
                        else lane_begin(L, C, S, o, d, im, it, idx, M, A.meshId);
                    }
                    batchNext += take;
                }
            }
            if (exhausted && idle == ~0ull) {
#ifdef XRT_WAVE_TIMES
                if (A.debugTimes && lane == 0) {
                    unsigned long long *o = A.debugTimes + 3 * (size_t)((int)blockIdx.x * 4 + wave);
                    o[0] = tStart; o[1] = tDry ? tDry : tStart; o[2] = wall_clock64();
                }
#endif
                break;
            }
        }
        // while-while: lanes gather in the same phase before the wave pays for that phase's code
        if (M == MODE_SCENE) {
            while (__any(L.state == ST_SCENE)) {
                if (L.state == ST_SCENE) advance_scene(L, C, S, st);
            }
        }
        {   // the NaN-free box test is valid for the whole wave unless some live lane has a parallel axis or a non-finite ray
            const bool fast = !__any(L.state != ST_IDLE && (L.r.par != 0 || L.weird != 0));
            for (int it = 0; it < A.nodeBurst && __any(L.state == ST_NODE); it++) {
                if (L.state == ST_NODE) advance_node(L, S, st, M, fast);
            }
        }
        for (int it = 0; it < A.leafBurst; it++) {
            const unsigned long long inLeaf = __ballot(L.state == ST_LEAF);
            if (inLeaf == 0ull) break;
            // many lanes in a leaf: two references per lane and step; few: their lists are dealt to the whole wave
            if (T >= 40 || __popcll(inLeaf) > A.coopMax) { if (L.state == ST_LEAF) advance_leaf(L, S); }
            else coop_leaf_step(L, S, &coopOwn[wave * 64 * COOP_ROUNDS]);
        }
        if (L.state == ST_FINISH) {
            const HitOut h = lane_result(L, C, S, M);
            const bool seg2 = L.rayIndex < 0;
            const int ati = seg2 ? ~L.rayIndex : L.rayIndex;
            const int *const sc = seg2 ? A.scatter2 : A.scatter;
            const int at = sc ? sc[ati] : ati;
            int *const fl = seg2 ? A.flags2 : A.flags;
            if (fl) fl[at] = h.hit;
            if (!fl || h.hit || A.missRecords) store_hit((seg2 ? A.hits2 : A.hits) + at, h);
            L.state = ST_IDLE;
        }
    }
    stamp_end(A.stamps);
}

int intersect_stack_capacity(int needed) {
    const int caps[] = {8, 12, 16, 24, 40};
    for (int c : caps) if (needed <= c) return c;
    return -1;
}
template <int T, int M> static int bpc() {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_intersect<T, M>, 256, 0) != hipSuccess || nb < 1) nb = 1;
    return nb > 8 ? 8 : nb;
}
template <int M> static int bpc_mode(int cap) {
    switch (cap) {
        case 8: return bpc<8, M>();
        case 12: return bpc<12, M>();
        case 16: return bpc<16, M>();
        case 24: return bpc<24, M>();
        default: return bpc<40, M>();
    }
}
int intersect_blocks_per_cu(int stackNeeded, int mode) {
    const int cap = intersect_stack_capacity(stackNeeded);
    if (mode == MODE_SINGLE) return bpc_mode<MODE_SINGLE>(cap);
    if (mode == MODE_MESH) return bpc_mode<MODE_MESH>(cap);
    return bpc_mode<MODE_SCENE>(cap);
}
// The start/stop events ride on the kernel's own dispatch packet (hipExtLaunchKernelGGL): per-launch timing
// without the ~3 us barrier packets hipEventRecord would put on both sides of every launch.
template <int M> static void launch_mode(int cap, dim3 g, const SceneView &S, const IntersectArgs &A, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    dim3 b(256);
    switch (cap) {
        case 8: hipExtLaunchKernelGGL((k_intersect<8, M>), g, b, 0, st, e0, e1, 0, S, A); break;
        case 12: hipExtLaunchKernelGGL((k_intersect<12, M>), g, b, 0, st, e0, e1, 0, S, A); break;
        case 16: hipExtLaunchKernelGGL((k_intersect<16, M>), g, b, 0, st, e0, e1, 0, S, A); break;
        case 24: hipExtLaunchKernelGGL((k_intersect<24, M>), g, b, 0, st, e0, e1, 0, S, A); break;
        default: hipExtLaunchKernelGGL((k_intersect<40, M>), g, b, 0, st, e0, e1, 0, S, A); break;
    }
}
void launch_intersect(const SceneView &S, const IntersectArgs &A, int stackNeeded, int gridBlocks, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    const int cap = intersect_stack_capacity(stackNeeded);
    dim3 g((unsigned)gridBlocks);
    if (A.mode == MODE_SINGLE) launch_mode<MODE_SINGLE>(cap, g, S, A, st, e0, e1);
    else if (A.mode == MODE_MESH) launch_mode<MODE_MESH>(cap, g, S, A, st, e0, e1);
    else launch_mode<MODE_SCENE>(cap, g, S, A, st, e0, e1);
}

// ---- reference work counters (untimed) ---------------------------------------------------------------------------
// Counts what the C# code path does for each ray: every slab test of OSM:460 / MO:331, every
// body visit, mesh box test and mesh query of the scene buckets up to and including the first one with a
// hit (OSM:334), and every leaf-list entry / triangle test of the mesh buckets up to and including the
// first with a hit (MO:281).  The winning keys come from the same state machine the hot kernel runs.
struct LocalStack {
    unsigned w[160];
    __device__ __forceinline__ unsigned get(int i) const { return w[i]; }
    __device__ __forceinline__ void set(int i, unsigned v) { w[i] = v; }
};

__device__ void count_mesh(const SceneView &S, const RayPre &r, int mesh, int ignoreId, bool mfound, float mKey,
                           unsigned long long *c, LocalStack &stk) {
    const MeshRec &mr = S.meshes[mesh];
    float key;
    c[C_NODES]++;   // the root's own box (MO:331)
    if (!slab(r, mr.rmin[0], mr.rmin[1], mr.rmin[2], mr.rmax[0], mr.rmax[1], mr.rmax[2], key)) return;
    auto count_leaf = [&](int start, int cnt, float k) {
        if (!mfound || k <= mKey) {
            c[C_REFS] += (unsigned long long)cnt;
            int ign = 0;
            if (ignoreId >= 0) for (int i = 0; i < cnt; i++) ign += (f2i(S.refN[start + i].w) == ignoreId) ? 1 : 0;
            c[C_TRIS] += (unsigned long long)(cnt - ign);
        }
    };
    if (mr.rootBlock < 0) { count_leaf(mr.rootRef, mr.rootCount, key); return; }
    // (how many of the queries the reference walks below are answered by the mesh's normal box on this library: every triangle faces away)
    if (all_back_facing(f4{mr.nbMin[0], mr.nbMin[1], mr.nbMin[2], mr.nbMin[3]}, f4{mr.nbMax[0], mr.nbMax[1], mr.nbMax[2], mr.nbMax[3]}, r.d)) c[C_MESH_AWAY]++;
    // explicit DFS over every child of every hit interior node (the reference does not prune): stack of
    // (block, pending mask) words plus the parent box per level
    struct Frame { int blk, mask; v3 bmin, half; };
    Frame fr[24];
    int sp = 0;
    fr[0].blk = mr.rootBlock; fr[0].mask = 0xff;
    fr[0].bmin = mk(mr.rmin[0], mr.rmin[1], mr.rmin[2]); fr[0].half = half_of(fr[0].bmin, mk(mr.rmax[0], mr.rmax[1], mr.rmax[2]));
    (void)stk;
    while (sp >= 0) {
        Frame &f = fr[sp];
        if (f.mask == 0) { sp--; continue; }
        int ch = ctz32((unsigned)f.mask);
        f.mask &= f.mask - 1;
        f4 lo = S.blocks[2 * (size_t)f.blk], hi = S.blocks[2 * (size_t)f.blk + 1];
        int d0 = f2i(lo.x), d1 = f2i(lo.y), d2 = f2i(lo.z), d3 = f2i(lo.w);
        int offw[4] = {f2i(hi.x), f2i(hi.y), f2i(hi.z), f2i(hi.w)};
        v3 cmin, cmax;
        child_box(f.bmin, f.half, ch, cmin, cmax);
        c[C_NODES]++;
        if (!slab(r, cmin.x, cmin.y, cmin.z, cmax.x, cmax.y, cmax.z, key)) continue;
        if ((d2 >> ch) & 1) {
            int nb = d0 + __builtin_popcount((unsigned)(d2 & 0xff) & ((1u << ch) - 1u));
            sp++;
            fr[sp].blk = nb; fr[sp].mask = 0xff; fr[sp].bmin = cmin; fr[sp].half = half_of(cmin, cmax);
        } else {
            auto off = [&](int q) { int w = offw[q >> 1]; return (q & 1) ? (int)((unsigned)w >> 16) : (w & 0xffff); };
            int start = d1 + off(ch), end = d1 + (ch == 7 ? d3 : off(ch + 1));
            count_leaf(start, end - start, key);
        }
    }
}

__device__ void run_query(Lane &L, const SceneView &S, LocalStack &stk, int mode) {
    while (L.state != ST_FINISH) {
        if (L.state == ST_SCENE) advance_scene(L, S, stk);
        else if (L.state == ST_NODE) advance_node(L, S, stk, mode, false);
        else advance_leaf(L, S);
    }
}

__global__ __launch_bounds__(256) void k_count(SceneView S, IntersectArgs A, unsigned long long *counters) {
    int n = A.nDev ? (*A.nDev) * A.nMul : A.n;
    if (A.nCap > 0 && n > A.nCap) n = A.nCap;
    unsigned long long c[C_COUNT];
    for (int i = 0; i < C_COUNT; i++) c[i] = 0;
    LocalStack stk, stk2;
    for (int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x); idx < n; idx += (int)(gridDim.x * blockDim.x)) {
        v3 o, d; int im, it;
        load_ray(A.rays + (A.index ? A.index[idx] : idx), o, d, im, it);
        if (A.nHeavy && heavy_marked(it)) it ^= HEAVY_BIT;
        if (im == DEAD_RAY) continue;
        c[C_RAYS]++;
        Lane L;
        const int cmode = (A.mode == MODE_MESH) ? MODE_MESH : MODE_SCENE;   // the counting pass always walks the general machine
        lane_begin(L, S, o, d, im, it, idx, cmode, A.meshId);
        run_query(L, S, stk, cmode);
        if (lane_found(L, cmode)) c[C_HITS]++;
        if (A.mode == MODE_MESH) {
            c[C_MESH_QUERIES]++;
            count_mesh(S, L.sc.w, A.meshId, L.ignoreId, L.mfound != 0, L.mKey, c, stk);
            continue;
        }
        const bool sfound = L.sc.sfound != 0;
        const float sbKey = L.sc.sbKey;
        const int ignoreId = L.ignoreId;
        const RayPre w = L.sc.w;
        int sp = 0, blk = 0, mask = 1;
        for (;;) {
            if (mask == 0) {
                if (sp == 0) break;
                unsigned wv = stk2.get(--sp);
                blk = (int)(wv >> 8); mask = (int)(wv & 0xffu);
                continue;
            }
            int ch = ctz32((unsigned)mask);
            mask &= mask - 1;
            int node = blk * 8 + ch;
            f4 lo = S.snodes[2 * node], hi = S.snodes[2 * node + 1];
            int a = f2i(lo.w), b = f2i(hi.w);
            c[C_SCENE_NODES]++;
            float key;
            if (!slab(w, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, key)) continue;
            if (b >= 0) {
                if (mask) stk2.set(sp++, ((unsigned)blk << 8) | (unsigned)mask);
                blk = a >> 3; mask = 0xff;
                continue;
            }
            int cnt = b & 0x0fffffff;
            if (sfound && key > sbKey) continue;
            for (int oi = 0; oi < cnt; oi++) {
                const ObjRec &ob = S.objects[S.srefs[a + oi]];
                c[C_INSTANCES]++;
                v3 v1 = transform(w.o, ob.invWorld);
                v3 v2 = transform(add(w.o, w.d), ob.invWorld);
                RayPre r = make_ray(v1, normalize(sub(v2, v1)));
                for (int mi = 0; mi < ob.meshCount; mi++) {
                    int m = S.objMesh[ob.meshStart + mi];
                    const MeshRec &mr = S.meshes[m];
                    c[C_MESH_AABB]++;
                    float k;
                    if (!slab(r, mr.bmin[0], mr.bmin[1], mr.bmin[2], mr.bmax[0], mr.bmax[1], mr.bmax[2], k)) continue;
                    c[C_MESH_QUERIES]++;
                    Lane Q;
                    lane_begin(Q, S, r.o, r.d, -1, -1, idx, MODE_MESH, m);
                    Q.ignoreId = ignoreId;
                    run_query(Q, S, stk, MODE_MESH);
                    count_mesh(S, r, m, ignoreId, Q.mfound != 0, Q.mKey, c, stk);
                }
            }
        }
    }
    for (int i = 0; i < C_COUNT; i++) {
        unsigned long long v = c[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane_id() == 0 && v) atomicAdd(&counters[i], v);
    }
}

void launch_count(const SceneView &S, const IntersectArgs &A, unsigned long long *counters, hipStream_t st) {
    hipLaunchKernelGGL(k_count, dim3(1024), dim3(256), 0, st, S, A, counters);
}

// ---- ray generation (RT:410-421 / RT:224-232) ----------------------------------------------------------------------
// Path p of a frame: pixel slot = p / samples (64x8 tiles, row-major inside a tile, tiles dealt round-robin
// to shards), sample = p % samples.
__device__ __forceinline__ bool path_pixel(const RayGenParams &g, long long pix, int &x, int &y) {
    long long slot = pix >> 9;
    int within = (int)(pix & 511);
    long long t = g.tileOfSlot ? (long long)g.tileOfSlot[slot] : shard_tile(slot, g.shardRank, g.shardCount, g.tilesX);
    if (t < 0 || t >= (long long)g.tilesX * g.tilesY) return false;
    const unsigned ti = (unsigned)t;   // tilesX * tilesY < 2^31: one 32-bit division instead of two 64-bit ones
    const int ty = (int)(ti / (unsigned)g.tilesX), tx = (int)(ti - (unsigned)ty * (unsigned)g.tilesX);
    int wx, wy;
    tile_slot_xy(within, wx, wy);
    x = tx * XRT_TILE_W + wx;
    y = ty * XRT_TILE_H + wy;
    return x < g.width && y < g.height;
}

// Rays that miss the scene octree's root box are answered here (OSM:318-320: no cuboid collected -> return
// false) and the others are appended, wave by wave, to a compact index list for the traversal kernel.
// One global atomic per list covers RG_ROUNDS x 1024 consecutive paths of a block (2025 atomics on one word were 20 of
// the kernel's 21 us on a 1080p frame): the rays are written first, the list slots afterwards.
constexpr int RG_ROUNDS = 4;
// With a live list (`index`, inside frames) the rays are COMPACT: the ray of the j-th live path sits at rays[j] and index[j] names its path
// -- what the traversal kernels read is one contiguous run, and every per-ray array of generation 0 needs room for the live rays only
// (at most the paths inside the root box's screen rectangle, known on the host: `liveCap`), not for every path of the frame.  Without
// a list (xrt_generate_primary_rays) ray p sits at rays[p].
__global__ __launch_bounds__(APPEND_BLOCK) void k_raygen(RayGenParams g, SceneView S, xrt_ray *rays, f4 *lvlB0, int *index, int *count, int Phost,
                                                         long long pathBase, HeavyArgs H, int liveCap) {
    __shared__ int ldsLive[RG_ROUNDS * 16], ldsHeavy[RG_ROUNDS * 16];
    const int P = pass_paths(g, Phost);
    const f4 rlo = S.snodes[0], rhi = S.snodes[1];
    const int span = RG_ROUNDS * APPEND_BLOCK;
    const int groups = (P + span - 1) / span;
    const int wave = (int)(threadIdx.x >> 6), lane = lane_id();
    for (int grp = (int)blockIdx.x; grp < groups; grp += (int)gridDim.x) {
        int liveAt[RG_ROUNDS], heavyAt[RG_ROUNDS];   // rank among the flagged lanes of the wave, -1: not flagged
        v3 keepO[RG_ROUNDS], keepD[RG_ROUNDS];       // the live rays of this thread, written once their places in the list are known
#pragma unroll
        for (int r = 0; r < RG_ROUNDS; r++) {
            const int p = grp * span + r * APPEND_BLOCK + (int)threadIdx.x;
            bool live = false, heavy = false, record = true;
            keepO[r] = mk(0, 0, 0); keepD[r] = mk(0, 0, 0);
            if (p < P) {
                long long gp = pathBase + p;
                const int sshift = g.samples == 16 ? 4 : (g.samples == 4 ? 2 : 0);   // samples is 1, 4 or 16
                int s = (int)(gp & (long long)(g.samples - 1));
                int x = 0, y = 0;
                const bool listed = g.quadLevel > 0;   // quadrant centres come from a list, every entry is valid
                if (!listed && !path_pixel(g, gp >> sshift, x, y)) {
                    if (!index) store_ray(rays + p, mk(0, 0, 0), mk(0, 0, 0), DEAD_RAY, -1);
                    record = g.cullSkipsRecord == 0;   // (a path without a pixel -- the rim of an edge tile -- is treated like a pixel outside the rectangle: k_compose reads no record of it)
                } else if (index && !listed && (x < g.cullX0 || x > g.cullX1 || y < g.cullY0 || y > g.cullY1)) {
                    record = g.cullSkipsRecord == 0;   // cannot reach the root box (RayGenParams): live stays false
                } else {
                    float sx = (float)x, sy = (float)y;
                    if (g.quadLevel >= 0) {   // RT:218-276: four rays at centre -+ size/4, order UL, UR, LL, LR
                        if (listed) { sx = g.quadCx[gp >> 2]; sy = g.quadCy[gp >> 2]; }
                        const float quarter = g.quadSize * 0.25f;
                        sx = (s & 1) ? sx + quarter : sx - quarter;
                        sy = (s & 2) ? sy + quarter : sy - quarter;
                    } else if (g.samples == 16) {   // XRT_MS_FIXED16: corner q = s/4 at +-0.25, sub-sample s%4 at +-0.125 (RT:218-305)
                        int q = s >> 2, rr = s & 3;
                        sx = (sx + ((q & 1) ? 0.25f : -0.25f)) + ((rr & 1) ? 0.125f : -0.125f);
                        sy = (sy + ((q & 2) ? 0.25f : -0.25f)) + ((rr & 2) ? 0.125f : -0.125f);
                    }
                    v3 nearP = unproject(g, sx, sy, 0.0f);   // RT:415
                    v3 farP = unproject(g, sx, sy, 1.0f);    // RT:419
                    v3 dir = normalize(sub(farP, nearP));    // RT:420-421
                    if (index) {
                        RayPre w = make_ray(nearP, dir);
                        float key;
                        live = slab(w, rlo.x, rlo.y, rlo.z, rhi.x, rhi.y, rhi.z, key);   // OSM:460 on the root
                    }
                    heavy = live && H.list && long_ray(S, H, p, nearP, dir);
                    if (!index) store_ray(rays + p, nearP, dir, -1, -1);
                    keepO[r] = nearP; keepD[r] = dir;   // (a culled ray is never read: it gets no place)
                }
                if (index && !live && record) lvlB0[lvl_at(g.lvl, p)] = f4{0, 0, 0, i2f(FLAG_MISS)};   // generation 0 ends here (RT:729-733)
            }
            const unsigned long long ml = __ballot(live), mh = __ballot(heavy);
            liveAt[r] = live ? lanes_below(ml) : -1;
            heavyAt[r] = heavy ? lanes_below(mh) : -1;
            if (lane == 0) { ldsLive[r * 16 + wave] = (int)__popcll(ml); ldsHeavy[r * 16 + wave] = (int)__popcll(mh); }
        }
        if (!index) continue;   // grid-uniform: no lists
        __syncthreads();
        if (wave < 2 && (wave == 0 || H.list)) {   // wave 0 reserves the block's range of the live list, wave 1 of the long-ray list
            int *cells = wave == 0 ? ldsLive : ldsHeavy;
            const int c = cells[lane];   // entry r * 16 + w: ascending path order
            const int incl = wave_scan_add(c);
            const int total = __builtin_amdgcn_readlane(incl, 63);
            int base = 0;
            if (lane == 0 && total) base = atomicAdd(wave == 0 ? count : H.count, total);
            base = __builtin_amdgcn_readfirstlane(base);
            cells[lane] = base + incl - c;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RG_ROUNDS; r++) {
            const int p = grp * span + r * APPEND_BLOCK + (int)threadIdx.x;
            if (liveAt[r] >= 0) {
                const int slot = ldsLive[r * 16 + wave] + liveAt[r];
                if (slot < liveCap) {   // (liveCap bounds the live rays by construction: the guard is against a wrong bound, not a code path)
                    index[slot] = p;
                    store_ray(rays + slot, keepO[r], keepD[r], -1, heavyAt[r] >= 0 ? (-1 ^ HEAVY_BIT) : -1);
                    if (heavyAt[r] >= 0) H.list[ldsHeavy[r * 16 + wave] + heavyAt[r]] = slot;
                }
            }
        }
        __syncthreads();   // the cells are reused by the next group
    }
}
void launch_raygen(const RayGenParams &g, const SceneView &S, xrt_ray *rays, f4 *lvlB0, int *index, int *count, int P, long long pathBase,
                   const HeavyArgs &H, hipStream_t st, hipEvent_t startEvent, int liveCap) {
    static_assert(RG_ROUNDS * 16 == 64, "one wave scans the block's cells");
    int blocks = (P + RG_ROUNDS * APPEND_BLOCK - 1) / (RG_ROUNDS * APPEND_BLOCK);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipExtLaunchKernelGGL(k_raygen, dim3(blocks), dim3(APPEND_BLOCK), 0, st, startEvent, nullptr, 0, g, S, rays, lvlB0, index, count, P, pathBase, H, liveCap);
}

// ---- shading ----------------------------------------------------------------------------------------------------------
struct ShadePoint {
    v3 normal, world;
    int gtri, mat;
};
// (`flag`: the ray's hit / miss word where the launch wrote one -- a miss has no record then, IntersectArgs::flags)
__device__ __forceinline__ void load_hit(const xrt_hit *src, int &hit, int &object, int &mesh, int &tri, float &u, float &v, float &d, v3 &w,
                                         const int *flag = nullptr) {
    const Hit16 *p = reinterpret_cast<const Hit16 *>(src);
    Hit16 a = Hit16{0, -1, -1, -1}, b = Hit16{0, 0, 0, 0}, c = b;
    if (!flag || *flag != 0) { a = p[0]; b = p[1]; c = p[2]; }
    hit = a.i0; object = a.i1; mesh = a.i2; tri = a.i3;
    u = i2f(b.i1); v = i2f(b.i2); d = i2f(b.i3);
    w = mk(i2f(c.i0), i2f(c.i1), i2f(c.i2));
}
// RT:520-531 fragment normal
__device__ __forceinline__ v3 fragment_normal(const ShadeView &V, int gtri, int matFlags, float u, float v) {
    const f4 *s = V.shade + (size_t)gtri * 6;
    if (matFlags & MAT_INTERP) {
        f4 s0 = s[0], s1 = s[1], s2 = s[2];
        v3 n1 = mk(s0.x, s0.y, s0.z), n2 = mk(s1.x, s1.y, s1.z), n3 = mk(s2.x, s2.y, s2.z);
        v3 a = sub(n2, n1), b = sub(n3, n1);
        return normalize(add(add(n1, scale(a, u)), scale(b, v)));
    }
    f4 s5 = s[5];
    return mk(s5.x, s5.y, s5.z);
}
// RT:469-479 direction and distance towards a light
__device__ __forceinline__ void light_dir(const LightRec &L, v3 world, v3 &dir, float &dist) {
    if (L.kind == 0) {
        v3 t = sub(mk(L.px, L.py, L.pz), world);
        dist = length(t);
        dir = normalize(t);
    } else {
        dir = neg(mk(L.dx, L.dy, L.dz));
        dist = FLT_MAX;
    }
}

// MAT:71-160 LookupUV: address mode + point sample
__device__ __forceinline__ v3 lookup_uv(const ShadeView &V, const MaterialRec &M, float ux, float uy) {
    if (V.addressMode == XRT_ADDRESS_WRAP) {   // MAT:125-136
        if (ux > 1.0f) ux = fmod1(ux);
        if (uy > 1.0f) uy = fmod1(uy);
        if (ux < 0.0f) ux = 1.0f + fmod1(ux);
        if (uy < 0.0f) uy = 1.0f + fmod1(uy);
    } else if (V.addressMode == XRT_ADDRESS_CLAMP) {   // MAT:138-143
        ux = (ux > 1.0f) ? 1.0f : ux; ux = (ux < 0.0f) ? 0.0f : ux;
        uy = (uy > 1.0f) ? 1.0f : uy; uy = (uy < 0.0f) ? 0.0f : uy;
    } else {   // MAT:102-123
        float ox = ux, oy = uy;
        if (ux > 1.0f) ux = fmod1(ux);
        if (uy > 1.0f) uy = fmod1(uy);
        if (ux < 0.0f) ux = 1.0f + fmod1(ux);
        if (uy < 0.0f) uy = 1.0f + fmod1(uy);
        if (((int)(ox - ux)) % 2 == 0) ux = 1.0f - ux;
        if (((int)(oy - uy)) % 2 == 0) uy = 1.0f - uy;
    }
    if (V.filtering == XRT_FILTER_BILINEAR) {   // MAT:162-232
        const float tdx = 1.0f / (float)M.texWidth, tdy = 1.0f / (float)M.texHeight;   // MAT:67
        const double remX = remainder((double)ux, (double)tdx), remY = remainder((double)uy, (double)tdy);   // Math.IEEERemainder, exact
        ux -= (float)remX;
        uy -= (float)remY;
        const int bx = (int)(ux * (float)(M.texWidth - 1)), by = (int)(uy * (float)(M.texHeight - 1));
        const int bx2 = (int)((ux + tdx) * (float)(M.texWidth - 1)), by2 = (int)((uy + tdy) * (float)(M.texHeight - 1));
        auto texel = [&](int xx, int yy) {
            long long idx = (long long)M.texWidth * yy + xx;
            if (idx < 0 || idx >= (long long)M.texWidth * M.texHeight) idx = 0;
            uint32_t w = V.texels[M.texOffsetP + idx];   // MAT:186-189: Texture.ColorData
            return mk((float)((w >> 16) & 0xffu), (float)((w >> 8) & 0xffu), (float)(w & 0xffu));
        };
        const v3 c00 = texel(bx, by), c10 = texel(bx2, by), c01 = texel(bx, by2), c11 = texel(bx2, by2);
        const float dx = (float)(remX * (double)M.texWidth) + 0.5f, dy = (float)(remY * (double)M.texHeight) + 0.5f;
        const float ix = 1.0f - dx, iy = 1.0f - dy;
        v3 sum = add(add(add(scale(scale(c00, ix), iy), scale(scale(c01, ix), dy)), scale(scale(c10, dx), iy)), scale(scale(c11, dx), dy));
        return scale(sum, 1.0f / 255.0f);
    }
    int x = (int)(ux * (float)(M.texWidth - 1));    // MAT:147
    int y = (int)(uy * (float)(M.texHeight - 1));   // MAT:148
    long long idx = (long long)M.texWidth * y + x;
    if (idx < 0 || idx >= (long long)M.texWidth * M.texHeight) idx = 0;   // the C# reads through a raw pointer; guard NaN uv
    uint32_t argb = V.texels[M.texOffset + idx];
    const float BYTE_RECIPROCAL = 1.0f / 255.0f;   // MAT:27
    return mk((float)((argb >> 16) & 0xffu) * BYTE_RECIPROCAL, (float)((argb >> 8) & 0xffu) * BYTE_RECIPROCAL, (float)(argb & 0xffu) * BYTE_RECIPROCAL);
}

// CastRay's shading for one step of the wavefront.  After intersect launch #k two independent pieces of work exist and
// ShadeArgs::ae: would the one body's one mesh answer this world-space ray "no intersection" because all its triangles face away from it?  The
// object-space direction is formed exactly as the traversal kernels form it (traverse.h lane_begin, OSM:358-364), the test is theirs
// (all_back_facing on the mesh's normal box, with its margin; a NaN anywhere makes it false: such a ray is traced).
__device__ __forceinline__ bool faces_away_single(const SceneView &S, v3 o, v3 d) {
    const ObjRec &ob = S.objects[0];
    const MeshRec &mr = S.meshes[0];
    const v3 v1 = transform(o, ob.invWorld);
    const v3 v2 = transform(add(o, d), ob.invWorld);
    const v3 dir = normalize(sub(v2, v1));
    return all_back_facing(f4{mr.nbMin[0], mr.nbMin[1], mr.nbMin[2], mr.nbMin[3]}, f4{mr.nbMax[0], mr.nbMax[1], mr.nbMax[2], mr.nbMax[3]}, dir);
}

// one kernel does both:
//   part A, generation k   : the closest-hit answers.  Misses end their path (RT:729-733); every hit takes a slot, emits one
//                            shadow ray per light (RT:535-537 -> RT:482-485) and -- the reflected / refracted directions
//                            depend on the hit alone, not on the lighting -- the rays of generation k+1 (RT:545-559,
//                            RT:656-698); and everything else that depends on the hit alone goes into the level record
//                            now: fragment normal, surface colour (RT:568-581 / 711-724), Reflectiveness.  Launch #k+1 traces
//                            both ray sets together.
//   part B, generation k-1 : the shadow answers of launch #k: light accumulation (RT:534-542) into the level record the return
//                            path (k_compose) needs.  It reads the hit's 32-byte slot record (world position, path, normal, Reflectiveness:
//                            left by part A) where it used to read the 48-byte hit, the material and the shading record a second time.
__global__ __launch_bounds__(APPEND_BLOCK) void k_shade(SceneView S, ShadeView V, ShadeArgs X) {
    __shared__ int ldsCounts[17];
    const int stride = (int)(gridDim.x * blockDim.x);
    const int tid = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const size_t P = (size_t)X.P;
    if (X.doB) {
        // Generation k-1: everything that depends on the hit alone (fragment normal, surface colour, Reflectiveness) was computed by
        // part A of the previous step; what is left is the light sum, which needs the shadow answers: one dense 32-byte slot record in
        // (kernels.h SlotRec) and one 16-byte level record out, instead of the 48-byte hit, the material and the shading record again.
        int n = *X.scntPrev;
        if (n > X.shadowCap) n = X.shadowCap;
        for (int s = tid; s < n; s += stride) {
            const SlotRec rec = X.slotPrev[s];   // left by part A of the previous step
            const int node = X.heap ? X.slotNodePrev[s] : X.level - 1;
            const size_t at = (size_t)node * P + lvl_at(X.lvl, rec.path);
            const v3 normal = mk(rec.nx, rec.ny, rec.nz), w = mk(rec.wx, rec.wy, rec.wz);
            v3 lightResult = mk(0, 0, 0);
            for (int l = 0; l < V.nLights; l++) {
                const LightRec &Lt = V.lights[l];
                v3 dir; float dist;
                light_dir(Lt, w, dir, dist);
                int sh, sobj, smesh, stri; float su, sv, sd; v3 sw;
                load_hit(X.shadowHits + (size_t)s * V.nLights + l, sh, sobj, smesh, stri, su, sv, sd, sw,
                         X.shadowFlags ? X.shadowFlags + (size_t)s * V.nLights + l : nullptr);
                float lightAmount = 0.0f;   // RT:485-501
                if (sh && sd < dist) {
                    const MaterialRec SM = V.materials[V.meshes[smesh].material];
                    if (SM.flags & MAT_TRANSPARENT) lightAmount = V.shade[(size_t)(V.meshes[smesh].triBase + stri) * 6 + 3].w;
                    else lightAmount = 1.0f;
                }
                if (lightAmount != 1.0f) lightResult = add(lightResult, scale(light_for_fragment(Lt, w, normal), 1.0f - lightAmount));   // RT:538-541
            }
            X.lvlA[at] = f4{lightResult.x, lightResult.y, lightResult.z, rec.refl};
        }
    }
    if (!X.doA) return;
    int n = X.nDev ? *X.nDev : X.nHost;
    if (n > X.cap) n = X.cap;
    const bool emitNext = X.level < X.maxReflections;   // grid-uniform
    const int rounds = (n + stride - 1) / stride;
    for (int it = 0; it < rounds; it++) {
        const int j = it * stride + tid;
        int i = j, hit = 0, object, mesh = 0, tri = 0, p = 0, node = X.level;
        float u = 0, v = 0, d = 0, curRef = 1.0f, n2 = 1.0f;
        v3 w = mk(0, 0, 0), rdir = mk(0, 0, 0), tdir = mk(0, 0, 0);
        bool refracts = false;
        if (j < n) {
            if (X.index) i = X.index[j];   // generation 0: only the rays that reached the scene's root box were traced
            load_hit(X.hits + i, hit, object, mesh, tri, u, v, d, w, X.hitFlags ? X.hitFlags + i : nullptr);
            p = X.rayPath ? X.rayPath[i] : i;
            if (X.costOut) {   // (launches whose cost words are read write a record for every ray: IntersectArgs::missRecords)
                const int c = reinterpret_cast<const Hit16 *>(X.hits + i)[2].i3;
                X.costOut[p] = (X.epoch << 16) | (unsigned)(c > 0xffff ? 0xffff : (c < 0 ? 0 : c));
            }
            if (X.heap) { node = X.rayNode ? X.rayNode[i] : 0; curRef = X.rayRef ? X.rayRef[i] : 1.0f; }   // generation 0: root, in vacuum (RT:424)
            if (!hit) X.lvlB[(size_t)node * P + lvl_at(X.lvl, p)] = f4{0, 0, 0, i2f(FLAG_MISS)};
        }
        const int slot = block_append(X.scnt, hit != 0, ldsCounts);
        if (hit && slot >= X.shadowCap) { *X.overflow = 1; hit = 0; }   // more rays than the chunk's buffers hold: the host retries with fewer paths
        if (X.ae) {   // (grid-uniform) shadow rays: answered here where the whole mesh faces away, else appended to the compact list
            for (int l = 0; l < V.nLights; l++) {
                v3 dir = mk(0, 0, 0); float dist;
                bool emit = false;
                if (hit) {
                    light_dir(V.lights[l], w, dir, dist);
                    emit = !faces_away_single(S, w, dir);
                    if (!emit) X.shadowFlagsOut[(size_t)slot * V.nLights + l] = 0;   // IsLightPathObstructed's query finds nothing (RT:482-485)
                }
                const int pos = block_append(X.shadowCnt, emit, ldsCounts);
                if (emit) {
                    store_ray(X.shadowRays + pos, w, dir, mesh, tri);   // ignore = shaded triangle (RT:485)
                    X.shadowOut[pos] = slot * V.nLights + l;
                }
            }
        }
        if (hit) {
            for (int l = 0; l < V.nLights && !X.ae; l++) {
                v3 dir; float dist;
                light_dir(V.lights[l], w, dir, dist);
                store_ray(X.shadowRays + (size_t)slot * V.nLights + l, w, dir, mesh, tri);   // ignore = shaded triangle (RT:485)
            }
            // the hit's own share of RT:516-581: fragment normal (RT:520-531), surface colour (RT:568-581 / 711-724), Reflectiveness -- the colour final
            // in lvlB, (normal, Reflectiveness) in the slot record until part B of the next step has the shadow answers for the light sum
            const int gtri = V.meshes[mesh].triBase + tri;
            const MaterialRec M = V.materials[V.meshes[mesh].material];
            const v3 normal = fragment_normal(V, gtri, M.flags, u, v);
            {
                v3 surf;
                const f4 *sr = V.shade + (size_t)gtri * 6;
                if (M.flags & MAT_TEXTURE) {   // RT:568-575
                    f4 s0 = sr[0], s1 = sr[1], s2 = sr[2], s4 = sr[4];
                    float uv1x = s0.w, uv1y = s1.w, uv2x = s2.w, uv2y = s4.x, uv3x = s4.y, uv3y = s4.z;
                    float ax = uv2x - uv1x, ay = uv2y - uv1y, bx = uv3x - uv1x, by = uv3y - uv1y;
                    float ix = (uv1x + ax * u) + bx * v, iy = (uv1y + ay * u) + by * v;
                    surf = lookup_uv(V, M, ix, iy);
                } else {
                    f4 c = sr[3];
                    surf = mk(c.x, c.y, c.z);
                }
                const bool transparent = (M.flags & MAT_TRANSPARENT) != 0;
                const size_t at = (size_t)node * P + lvl_at(X.lvl, p);
                X.slotOut[slot] = SlotRec{w.x, w.y, w.z, p, normal.x, normal.y, normal.z, M.reflectiveness};
                if (X.heap) X.slotNodeOut[slot] = node;
                X.lvlB[at] = f4{surf.x, surf.y, surf.z, i2f(FLAG_HIT | (transparent ? FLAG_TRANSPARENT : 0))};
                if (X.heap) X.lvlAlpha[at] = sr[3].w;   // triangle.color.W (RT:699)
            }
            if (emitNext) {
                v3 o, dd; int im, itri;
                load_ray(X.rays + i, o, dd, im, itri);
                rdir = normalize(reflect(dd, normal));   // RT:549-550
                if (X.heap && (M.flags & MAT_TRANSPARENT)) {   // RT:656-694: Snell refraction, the System.Math calls in double
                    float n1;
                    if (curRef == M.refractionIndex) { n1 = 1.0f; n2 = curRef; }
                    else { n1 = M.refractionIndex; n2 = 1.0f; }
                    const float cos1 = dot(normal, neg(dd));
                    const double ratio = (double)(n1 / n2), c1 = (double)cos1;
                    const float cos2 = (float)sqrt(1 - (ratio * ratio) * (1 - (c1 * c1)));   // Math.Pow(x, 2.0) == x*x exactly here
                    const float q = n1 / n2;
                    const v3 a = scale(dd, q), b = scale(normal, q * cos1 - cos2);
                    tdir = normalize(cos1 >= 0 ? add(a, b) : sub(a, b));
                    refracts = true;
                }
            }
        }
        if (!emitNext) continue;
        if (!X.heap && X.ae) {   // chain of reflections, answered at emission: a reflection the whole mesh faces away from ends its path here
            const bool emit = hit && !faces_away_single(S, w, rdir);
            if (hit && !emit) X.lvlB[(size_t)(X.level + 1) * P + lvl_at(X.lvl, p)] = f4{0, 0, 0, i2f(FLAG_MISS)};   // what part A of the next step writes for a miss (RT:729-733)
            const int pos = block_append(X.nextCnt, emit, ldsCounts);
            const bool heavy = emit && X.heavy.list && pos < X.nextCap && long_ray(S, X.heavy, p, w, rdir);
            if (emit && pos < X.nextCap) {   // (pos <= the parent's slot < cap: a guard, not a code path)
                store_ray(X.nextRays + pos, w, rdir, mesh, heavy ? (tri ^ HEAVY_BIT) : tri);   // origin = result.triangle (RT:559)
                X.nextPath[pos] = p;
            }
            if (X.heavy.list) {
                const int hs = block_append(X.heavy.count, heavy, ldsCounts);
                if (heavy) X.heavy.list[hs] = pos;
            }
            continue;
        }
        if (!X.heap) {   // chain of reflections: the ray of generation k+1 sits at its parent's slot
            const bool heavy = hit && X.heavy.list && long_ray(S, X.heavy, p, w, rdir);
            if (hit) {
                store_ray(X.nextRays + slot, w, rdir, mesh, heavy ? (tri ^ HEAVY_BIT) : tri);   // origin = result.triangle (RT:559)
                X.nextPath[slot] = p;
            }
            if (X.heavy.list) {
                const int hs = block_append(X.heavy.count, heavy, ldsCounts);
                if (heavy) X.heavy.list[hs] = slot;
            }
            continue;
        }
        const int slot1 = block_append(X.nextCnt, hit != 0, ldsCounts);
        if (hit && slot1 >= X.nextCap) *X.overflow = 1;
        else if (hit) {
            store_ray(X.nextRays + slot1, w, rdir, mesh, tri);
            X.nextPath[slot1] = p;
            X.nextNode[slot1] = 2 * node + 1; X.nextRef[slot1] = curRef;
        }
        const int slot2 = block_append(X.nextCnt, refracts, ldsCounts);   // the refracted ray of RT:698 continues in the medium with index n2
        if (refracts && slot2 >= X.nextCap) *X.overflow = 1;
        else if (refracts) {
            store_ray(X.nextRays + slot2, w, tdir, mesh, tri);
            X.nextPath[slot2] = p;
            X.nextNode[slot2] = 2 * node + 2; X.nextRef[slot2] = n2;
        }
    }
}
void launch_shade(const SceneView &S, const ShadeView &V, const ShadeArgs &X, hipStream_t st, int blocks, int threads) {
    if (threads != 256) threads = APPEND_BLOCK;   // 256-thread blocks spread a small generation over the CUs (block_append takes any block of whole waves)
    const int cap = threads == 256 ? 4096 : 1024;
    hipLaunchKernelGGL(k_shade, dim3(blocks < 1 ? 1 : (blocks > cap ? cap : blocks)), dim3(threads), 0, st, S, V, X);
}

// Frame epilogue (kernels.h FrameEpilogue): every kernel that counts rays has finished -- hand the counters to the host, clear them
__device__ __forceinline__ void frame_epilogue(const FrameEpilogue &E) {
    if (E.cntSrc && blockIdx.x == 0) {
        for (int i = (int)threadIdx.x; i < E.zeroWords; i += (int)blockDim.x) {
            const int v = E.cntSrc[i];
            if (i < E.cntWords) E.hostCnt[i] = v;
            if (i >= E.zeroFrom) E.cntSrc[i] = 0;
        }
        if (E.flagSrc && threadIdx.x == 0) { E.hostCnt[E.cntWords] = *E.flagSrc; *E.flagSrc = 0; }
        __threadfence_system();
    }
}
// Frame epilogue: (start, latest wave end) of traversal launches row0 .. row1-1 go to host-visible memory, one block per launch
// where the grid has them (block 0 is busy with the counters).  All threads of every block call it.
__device__ __forceinline__ void fold_stamps(const StampFold &F) {
    const int rows = F.row1 - F.row0;
    if (rows <= 0) return;
    __shared__ unsigned long long latest[4];
    const bool spread = (int)gridDim.x > rows;
    const int j0 = spread ? (int)blockIdx.x - 1 : (blockIdx.x == 0 ? 0 : rows), j1 = spread ? min(j0 + 1, rows) : rows;
    for (int jj = j0; jj >= 0 && jj < j1; jj++) {
        const int j = F.row0 + jj;
        const unsigned long long *row = F.src + (size_t)j * STAMP_STRIDE;
        const int nb = min((int)row[1], STAMP_SLOTS);
        unsigned long long m = 0;
        for (int b = (int)threadIdx.x; b < nb; b += (int)blockDim.x) { const unsigned long long t = row[STAMP_HEADER + b]; m = t > m ? t : m; }
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(m, o); m = t > m ? t : m; }
        if (lane_id() == 0) latest[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < (int)(blockDim.x >> 6) && w < 4; w++) m = latest[w] > m ? latest[w] : m;
            F.host[2 * j] = row[0];
            F.host[2 * j + 1] = m;
            __threadfence_system();
        }
        __syncthreads();
    }
}

// The return path of the CastRay recursion: deepest generation first, one RGBA8 quantisation per level.
__global__ __launch_bounds__(256) void k_compose(const f4 *lvlA, const f4 *lvlB, int countHost, int P, int maxReflections, uint32_t *sampleColor, float *sampleF32,
                                                 ResolveArgs RA) {
    const int count = pass_paths(RA.g, countHost);   // (read before the epilogue below may clear the word it comes from)
    {
        FrameEpilogue E;
        E.cntSrc = RA.cntSrc; E.hostCnt = RA.hostCnt; E.cntWords = RA.cntWords; E.zeroWords = RA.zeroWords; E.zeroFrom = RA.zeroFrom;
        frame_epilogue(E);
    }
    fold_stamps(RA.stamps);
    const int sshift = RA.g.samples == 16 ? 4 : (RA.g.samples == 4 ? 2 : 0);
    for (int p = (int)(blockIdx.x * blockDim.x + threadIdx.x); p < count; p += (int)(gridDim.x * blockDim.x)) {
        int kd = 0;
        int flag = FLAG_MISS;
        bool culled = false;
        if (RA.g.cullSkipsRecord && RA.g.quadLevel <= 0) {   // k_raygen left no record for pixels that cannot reach the root box
            int cx, cy;
            if (path_pixel(RA.g, (RA.pixelBase + p) >> sshift, cx, cy))
                culled = cx < RA.g.cullX0 || cx > RA.g.cullX1 || cy < RA.g.cullY0 || cy > RA.g.cullY1;
            else culled = true;   // (a path without a pixel: no record, and nobody looks at its colour)
        }
        const size_t lp = culled ? 0 : lvl_at(RA.g.lvl, p);   // (a culled path has no records)
        while (!culled) {
            flag = f2i(lvlB[(size_t)kd * P + lp].w);
            if (!(flag & FLAG_HIT) || kd == maxReflections) break;
            kd++;
        }
        v3 cv = mk(0, 0, 0);
        uint32_t col;
        if (!(flag & FLAG_HIT)) col = pack_color(mk(0, 0, 0));   // RT:732
        else {   // RT:708-727: generation MaxReflections has no reflection term
            f4 a = lvlA[(size_t)kd * P + lp], b = lvlB[(size_t)kd * P + lp];
            cv = mul(mk(a.x, a.y, a.z), mk(b.x, b.y, b.z));
            col = pack_color(cv);
        }
        for (int k = kd - 1; k >= 0; k--) {   // RT:584 + RT:705
            f4 a = lvlA[(size_t)k * P + lp], b = lvlB[(size_t)k * P + lp];
            cv = mul(lerp(unpack_color(col), mk(b.x, b.y, b.z), 1.0f - a.w), mk(a.x, a.y, a.z));
            col = pack_color(cv);
        }
        if (RA.fused) {   // one sample per pixel: write the framebuffer directly (RT:425), no sample buffer round trip
            long long pix = RA.pixelBase + p;
            int x, y;
            bool ok = path_pixel(RA.g, pix, x, y);
            if (RA.g.shardCount > 1) RA.out[pix] = ok ? col : 0u;
            else if (ok) {
                size_t o = (size_t)y * RA.g.width + x;
                RA.out[o] = col;
                if (RA.outF32) { RA.outF32[3 * o] = cv.x; RA.outF32[3 * o + 1] = cv.y; RA.outF32[3 * o + 2] = cv.z; }
            }
            continue;
        }
        sampleColor[p] = col;
        if (sampleF32) { sampleF32[3 * (size_t)p] = cv.x; sampleF32[3 * (size_t)p + 1] = cv.y; sampleF32[3 * (size_t)p + 2] = cv.z; }
    }
}
// The same return path over the binary ray tree of a scene with Transparent materials (RT:586-702): node i has
// its reflection at 2i+1 and its refraction at 2i+2; evaluated depth first like the recursion itself.
__global__ __launch_bounds__(256) void k_compose_tree(const f4 *lvlA, const f4 *lvlB, const float *lvlAlpha, int count, int P, int maxReflections,
                                                      uint32_t *sampleColor, float *sampleF32, StampFold stamps, FrameEpilogue epilogue) {
    constexpr int MAXD = 14;
    frame_epilogue(epilogue);
    fold_stamps(stamps);
    for (int p = (int)(blockIdx.x * blockDim.x + threadIdx.x); p < count; p += (int)(gridDim.x * blockDim.x)) {
        int stNode[MAXD], stPhase[MAXD];
        uint32_t stRefl[MAXD];
        int sp = 0;
        stNode[0] = 0; stPhase[0] = 0; stRefl[0] = 0;
        uint32_t ret = 0;
        v3 retCv = mk(0, 0, 0);
        while (sp >= 0) {
            const int node = stNode[sp];
            const f4 b = lvlB[(size_t)node * P + p];
            const int flag = f2i(b.w);
            if (stPhase[sp] == 0) {
                if (!(flag & FLAG_HIT)) { ret = pack_color(mk(0, 0, 0)); retCv = mk(0, 0, 0); sp--; continue; }   // RT:732
                if (sp == maxReflections) {   // RT:708-727
                    const f4 a = lvlA[(size_t)node * P + p];
                    retCv = mul(mk(a.x, a.y, a.z), mk(b.x, b.y, b.z));
                    ret = pack_color(retCv);
                    sp--;
                    continue;
                }
                stPhase[sp] = 1;   // RT:556-559: reflection first
                sp++;
                stNode[sp] = 2 * node + 1; stPhase[sp] = 0;
                continue;
            }
            const f4 a = lvlA[(size_t)node * P + p];
            if (stPhase[sp] == 1) {
                stRefl[sp] = ret;
                if (flag & FLAG_TRANSPARENT) {   // RT:698
                    stPhase[sp] = 2;
                    sp++;
                    stNode[sp] = 2 * node + 2; stPhase[sp] = 0;
                    continue;
                }
                retCv = mul(lerp(unpack_color(stRefl[sp]), mk(b.x, b.y, b.z), 1.0f - a.w), mk(a.x, a.y, a.z));   // RT:584
                ret = pack_color(retCv);   // RT:705
                sp--;
                continue;
            }
            v3 cv = mul(lerp(unpack_color(stRefl[sp]), mk(b.x, b.y, b.z), 1.0f - a.w), mk(a.x, a.y, a.z));   // RT:584
            retCv = lerp(unpack_color(ret), cv, lvlAlpha[(size_t)node * P + p]);                               // RT:699
            ret = pack_color(retCv);
            sp--;
        }
        sampleColor[p] = ret;
        if (sampleF32) { sampleF32[3 * (size_t)p] = retCv.x; sampleF32[3 * (size_t)p + 1] = retCv.y; sampleF32[3 * (size_t)p + 2] = retCv.z; }
    }
}
void launch_compose_tree(const f4 *lvlA, const f4 *lvlB, const float *lvlAlpha, int count, int P, int maxReflections, uint32_t *sampleColor,
                         float *sampleF32, const StampFold &stamps, const FrameEpilogue &epilogue, hipStream_t st) {
    int blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_compose_tree, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, st, lvlA, lvlB, lvlAlpha, count, P, maxReflections, sampleColor, sampleF32,
                       stamps, epilogue);
}
void launch_compose(const f4 *lvlA, const f4 *lvlB, int count, int P, int maxReflections, uint32_t *sampleColor, float *sampleF32,
                    const ResolveArgs &RA, hipStream_t st, hipEvent_t stopEvent) {
    int blocks = (count + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipExtLaunchKernelGGL(k_compose, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, st, nullptr, stopEvent, 0, lvlA, lvlB, count, P, maxReflections,
                          sampleColor, sampleF32, RA);
}

// Supersample averaging (RT:309: mean of four quantised colours, re-quantised, twice for 16 samples) and the
// framebuffer write renderTargetData[y*W + x] = color (RT:425) — or the shard's tile-contiguous buffer.
__global__ __launch_bounds__(256) void k_resolve(RayGenParams g, const uint32_t *sampleColor, const float *sampleF32, int pixels, long long pixelBase,
                                                 uint32_t *out, float *outF32, int *zeroPtr, int zeroN) {
    // (an adaptive frame in flight: the quadrant-level counts its fold kernels read are cleared here, by the frame's last kernel)
    if (zeroPtr && blockIdx.x == 0) for (int i = (int)threadIdx.x; i < zeroN; i += (int)blockDim.x) zeroPtr[i] = 0;
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < pixels; i += (int)(gridDim.x * blockDim.x)) {
        long long pix = pixelBase + i;
        int x, y;
        bool ok = path_pixel(g, pix, x, y);
        uint32_t col;
        v3 cv;
        if (g.quadLevel == 0) {   // adaptive: the pixel's level-0 quadrant after all folds (RT:309)
            const uint32_t *s = sampleColor + (size_t)i * 4;
            v3 sum = add(add(add(unpack_color(s[0]), unpack_color(s[1])), unpack_color(s[2])), unpack_color(s[3]));
            col = pack_color(divf(sum, 4.0f));
            cv = unpack_color(col);
        } else if (g.samples == 1) {
            col = sampleColor[i];
            cv = sampleF32 ? mk(sampleF32[3 * (size_t)i], sampleF32[3 * (size_t)i + 1], sampleF32[3 * (size_t)i + 2]) : unpack_color(col);
        } else {
            uint32_t corner[4];
            for (int q = 0; q < 4; q++) {
                const uint32_t *s = sampleColor + (size_t)i * 16 + q * 4;
                v3 sum = add(add(add(unpack_color(s[0]), unpack_color(s[1])), unpack_color(s[2])), unpack_color(s[3]));
                corner[q] = pack_color(divf(sum, 4.0f));
            }
            v3 sum = add(add(add(unpack_color(corner[0]), unpack_color(corner[1])), unpack_color(corner[2])), unpack_color(corner[3]));
            col = pack_color(divf(sum, 4.0f));
            cv = unpack_color(col);
        }
        if (g.shardCount > 1) {
            out[pix] = ok ? col : 0u;
        } else if (ok) {
            size_t o = (size_t)y * g.width + x;
            out[o] = col;
            if (outF32) { outF32[3 * o] = cv.x; outF32[3 * o + 1] = cv.y; outF32[3 * o + 2] = cv.z; }
        }
    }
}
void launch_resolve(const RayGenParams &g, const uint32_t *sampleColor, const float *sampleF32, int pixels, long long pixelBase, uint32_t *out,
                    float *outF32, hipStream_t st, int *zeroPtr, int zeroN) {
    int blocks = (pixels + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_resolve, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, st, g, sampleColor, sampleF32, pixels, pixelBase, out, outF32, zeroPtr, zeroN);
}

// ---- adaptive supersampling (RT:170-311) ------------------------------------------------------------------------
// Level l holds a list of quadrants; each got four CastRay colours (quadColor[4q..4q+3]).  k_ms_decide applies
// RT:279-306: a corner whose colour-vector length differs from the length of the 4-mean by more than 0.5
// (RT:340) is subdivided: its child quadrant (centre = the corner's sample position, half the size) is appended
// to the next level.  Children of one quadrant are appended together, in corner order.
__global__ __launch_bounds__(APPEND_BLOCK) void k_ms_decide(RayGenParams g, const uint32_t *quadColor, const int *nQuadsDev, int nQuadsHost,
                                                            long long pixelBase, int *childBase, int *childMask, float *nextCx, float *nextCy,
                                                            int *nextCount, int nextCap, int *overflow) {
    __shared__ int ldsCounts[17];
    int n = nQuadsDev ? *nQuadsDev : nQuadsHost;
    if (nQuadsDev && n > nQuadsHost) n = nQuadsHost;   // (nQuadsHost: this level's capacity when the count is the device's)
    const int stride = (int)(gridDim.x * blockDim.x);
    const int rounds = (n + stride - 1) / stride;
    for (int it = 0; it < rounds; it++) {
        const int q = it * stride + (int)(blockIdx.x * blockDim.x + threadIdx.x);
        int mask = 0;
        float cx = 0, cy = 0;
        if (q < n) {
            bool ok = true;
            if (g.quadLevel == 0) { int x, y; ok = path_pixel(g, pixelBase + q, x, y); cx = (float)x; cy = (float)y; }
            else { cx = g.quadCx[q]; cy = g.quadCy[q]; }
            if (ok) {
                v3 c0 = unpack_color(quadColor[4 * (size_t)q]), c1 = unpack_color(quadColor[4 * (size_t)q + 1]);
                v3 c2 = unpack_color(quadColor[4 * (size_t)q + 2]), c3 = unpack_color(quadColor[4 * (size_t)q + 3]);
                v3 average = divf(add(add(add(c0, c1), c2), c3), 4.0f);   // RT:285
                float al = length(average);                               // RT:286
                const float TRESHOLD = 0.5f;                              // RT:340
                if (fabsf(al - length(c0)) > TRESHOLD) mask |= 1;         // RT:288
                if (fabsf(al - length(c1)) > TRESHOLD) mask |= 2;         // RT:293
                if (fabsf(al - length(c2)) > TRESHOLD) mask |= 4;         // RT:298
                if (fabsf(al - length(c3)) > TRESHOLD) mask |= 8;         // RT:303
            }
        }
        // variable-count append: one block_append per corner keeps a quadrant's children in corner order only if
        // they are contiguous, so reserve them with a single call on the count
        const int cnt = __builtin_popcount((unsigned)mask);
        // exclusive scan of cnt over the block
        int incl = cnt;
        for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(incl, off); if (lane_id() >= off) incl += t; }
        const int wave = (int)(threadIdx.x >> 6), nw = (int)(blockDim.x >> 6);
        if (lane_id() == 63) ldsCounts[wave] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            int total = 0;
            for (int w = 0; w < nw; w++) { int c = ldsCounts[w]; ldsCounts[w] = total; total += c; }
            ldsCounts[16] = total ? atomicAdd(nextCount, total) : 0;
        }
        __syncthreads();
        const int base = ldsCounts[16] + ldsCounts[wave] + (incl - cnt);
        __syncthreads();
        if (q < n && nextCap > 0 && base + cnt > nextCap) {   // the next level's buffers are sized optimistically: the host renders the frame again the careful way
            *overflow = 1;
            childBase[q] = 0; childMask[q] = 0;
        } else if (q < n) {
            childBase[q] = base;
            childMask[q] = mask;
            const float quarter = g.quadSize * 0.25f;
            int o = base;
            for (int j = 0; j < 4; j++)
                if (mask & (1 << j)) {
                    nextCx[o] = (j & 1) ? cx + quarter : cx - quarter;   // RT:290-305: centre of the child = the corner's sample position
                    nextCy[o] = (j & 2) ? cy + quarter : cy - quarter;
                    o++;
                }
        }
    }
}
void launch_ms_decide(const RayGenParams &g, const uint32_t *quadColor, const int *nQuadsDev, int nQuadsHost, long long pixelBase, int *childBase,
                      int *childMask, float *nextCx, float *nextCy, int *nextCount, hipStream_t st, int nextCap, int *overflow) {
    hipLaunchKernelGGL(k_ms_decide, dim3(1024), dim3(APPEND_BLOCK), 0, st, g, quadColor, nQuadsDev, nQuadsHost, pixelBase, childBase, childMask, nextCx,
                       nextCy, nextCount, nextCap, overflow);
}
// Fold the results of level l+1 into level l, in the order the recursion assigns them (RT:288-306): UL, UR, LL
// replace their own corner; the LOWER-RIGHT recursion writes `out urColor` (RT:305), i.e. slot 1, after the
// upper-right one, and the lower-right corner keeps its first-pass colour.
__global__ __launch_bounds__(256) void k_ms_fold(uint32_t *quadColor, const uint32_t *childColor, const int *childBase, const int *childMask, int nHost,
                                                 const int *nDev) {
    int n = nHost;   // (with nDev: the level's capacity)
    if (nDev) { n = *nDev; if (n > nHost) n = nHost; }
    for (int q = (int)(blockIdx.x * blockDim.x + threadIdx.x); q < n; q += (int)(gridDim.x * blockDim.x)) {
        const int mask = childMask[q];
        if (!mask) continue;
        int o = childBase[q];
        for (int j = 0; j < 4; j++)
            if (mask & (1 << j)) {
                const uint32_t *s = childColor + 4 * (size_t)o;
                v3 sum = add(add(add(unpack_color(s[0]), unpack_color(s[1])), unpack_color(s[2])), unpack_color(s[3]));
                const uint32_t col = pack_color(divf(sum, 4.0f));   // RT:309 of the child call
                quadColor[4 * (size_t)q + (j == 3 ? 1 : j)] = col;
                o++;
            }
    }
}
void launch_ms_fold(uint32_t *quadColor, const uint32_t *childColor, const int *childBase, const int *childMask, int n, hipStream_t st, const int *nDev) {
    int blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_ms_fold, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, st, quadColor, childColor, childBase, childMask, n, nDev);
}

// rank-major gathered tiles -> W*H frame (rank 0 after the RCCL gather)
// (table: the tile of every (rank, slot) under an installed tile table, xrt.h xrt_scene_set_tile_table; null: round-robin)
__global__ __launch_bounds__(256) void k_detile(int width, int height, int shardCount, int tilesPerRank, int tilesX, int tilesY,
                                                const uint32_t *gathered, long long rankStride, uint32_t *out, const int *table) {
    const long long total = (long long)shardCount * tilesPerRank * 512;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int rank;
        long long rem;
        if (total <= 0x7fffffffLL) {   // every real frame: one 32-bit division
            const unsigned per = (unsigned)tilesPerRank * 512u, q = (unsigned)i / per;
            rank = (int)q; rem = (long long)((unsigned)i - q * per);
        } else {
            rank = (int)(i / ((long long)tilesPerRank * 512));
            rem = i % ((long long)tilesPerRank * 512);
        }
        long long slot = rem >> 9;
        int within = (int)(rem & 511);
        long long t = table ? (long long)table[(long long)rank * tilesPerRank + slot] : shard_tile(slot, rank, shardCount, tilesX);
        if (t < 0 || t >= (long long)tilesX * tilesY) continue;
        const unsigned ti = (unsigned)t, tyq = ti / (unsigned)tilesX;
        int wx, wy;
        tile_slot_xy(within, wx, wy);
        int x = (int)(ti - tyq * (unsigned)tilesX) * XRT_TILE_W + wx, y = (int)tyq * XRT_TILE_H + wy;
        if (x < width && y < height) out[(size_t)y * width + x] = gathered[(long long)rank * rankStride + rem];
    }
}
void launch_detile(int width, int height, int shardCount, int tilesPerRank, const uint32_t *gathered, long long rankStride, uint32_t *out, hipStream_t st,
                   const int *table) {
    int tilesX = (width + XRT_TILE_W - 1) / XRT_TILE_W, tilesY = (height + XRT_TILE_H - 1) / XRT_TILE_H;
    hipLaunchKernelGGL(k_detile, dim3(2048), dim3(256), 0, st, width, height, shardCount, tilesPerRank, tilesX, tilesY, gathered, rankStride, out, table);
}

}  // namespace xrt
