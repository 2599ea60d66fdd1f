// packet.hip — k_packet: wave-packet form of ISpatialManager.GetRayIntersection (OSM:312-455) -> MeshOctree.GetRayIntersection
// (MO:259-353) for COHERENT ray populations (primary rays of neighbouring pixels / sub-samples, shadow rays towards one light, the
// first reflections off flat surfaces): gfx950, wave64.
//
// k_intersect gives every lane its own walk: its own stack, block descriptor, pending masks, parent box.  For rays that
// go the same way that is ~50 registers of state per lane repeating what the neighbour holds, lanes of one wave wait for
// each other in different phases (utilisation 0.4-0.6), every step of a lane's walk is a dependent vector load (block descriptor,
// body record, leaf records, triangles: a chain of 100-500 cycle round trips that four waves per SIMD cannot hide -- C3's waves
// spent 52 % of their life parked on memory), and four waves per SIMD are all the registers allow.  Here ONE wavefront walks the
// octrees once for its 64 rays:
//   * wave-uniform state (SGPRs / one LDS frame per level): current block, its descriptor, the parent box, the lanes that
//     are inside it, the position in the child order; everything the walk reads is the same for all lanes and arrives through the
//     scalar cache (s_load), requested ahead of its use;
//   * per lane only the ray and its best answer so far in registers, and in LDS one byte per stacked level (which children of that
//     level's block the ray's own box tests accepted);
//   * a child is visited when ANY lane of the current set passed its box test (MO:331 -- hit8_*_children evaluates the
//     reference's test for all eight children of a block); lanes that did not are masked off for that subtree, as are
//     lanes the bucket rule lets prune it (key above the lane's best key on a `safe` interior node or a leaf);
//   * in a leaf every remaining lane tests every triangle (RE:42-75): the triangle is the same for all lanes, so its
//     16 + 36 bytes arrive through the scalar cache and the vector units only do arithmetic.
// Each lane therefore performs exactly the (leaf, triangle) tests its own walk in k_intersect could perform, minus ones
// that cannot win, and keeps the lexicographic arg-min (leaf entry key, distance, leaf DFS index, list position) -- the
// order-independent form of MO:281-301: bit-identical answers, in any visiting order.
// Lanes with a parallel axis or a non-finite component take the literal box test (`slab`) inside the same walk.
//
// MODE_SCENE (two-level scenes, OSM:312-455): the scene octree is walked wave-uniformly too, in the reference's DFS order (the
// scene-level rule is the streaming one, traverse.h merge_mesh_result: the order of the visits matters there and every lane sees
// its own visits in the order its own walk would make them).  A scene node is visited by the lanes whose own box test (OSM:460)
// accepted it and all its ancestors; a body of a leaf by the lanes of that leaf the world-space pre-cull (traverse.h precull_hit)
// does not exclude -- its record is one scalar load for the whole wave (SceneView::scull), where the per-lane kernel pays two
// dependent vector loads per lane and body -- ; those lanes transform their ray into the body's space (OSM:349-364), test the mesh boxes (MESH:34-39) and
// share one walk of the mesh octree; a lane's answer for the body is merged into its scene-level best as OSM:370-378 does.
#include "device_util.h"
#include "kernels.h"

#include <hip/hip_ext.h>

namespace xrt {

constexpr int PK_LEVELS = 24;        // deeper octrees than this fall back to k_intersect (scene_build limits depth to 20)
constexpr int PK_FRAME_WORDS = 4;    // blk, pending children (the first lane's order is found again from the lanes), lanes (2): one 16-byte LDS access
// a wave's stack: PK_LEVELS frames, then per level one byte per lane -- which children of that level's block the lane's own box tests accepted
constexpr int PK_SPLIT_AT = PK_LEVELS * (PK_FRAME_WORDS + 16);   // ... and behind them the wave's split-walk words (pk_walk): record, item, budget, budget clock, packet
// make variant NAME=bundle DEFS=-DXRT_PK_BUNDLE: the bundle prefilter of big leaves (below).  Not in the shipped kernel: it needs 106 vector registers where the walk has 80
// (four waves per SIMD instead of six: +19 %), and at equal occupancy it gains 5 % (profiles/r04/bundle_prefilter.txt).
#ifdef XRT_PK_BUNDLE
constexpr bool PK_BUNDLE = true;
#else
constexpr bool PK_BUNDLE = false;
#endif
constexpr int PK_BUNDLE_AT = PK_SPLIT_AT + 8;    // ... and the packet's ray bundle (xrt_core.h RayBundle: 20 words) + [20] "the bundle may be used"
// make variant NAME=pf DEFS=-DXRT_PK_PREFETCH: the prefetches of small launches (pk_prefetch below; PacketArgs::prefetch switches them per launch).  Not in the shipped kernel: merely
// compiled in -- and switched off -- they change the register allocation of the walk (89-92 spilled scalars instead of 84-87) and cost whole C5 frames 2 % (profiles/r04/split_walks.txt).
#ifdef XRT_PK_PREFETCH
constexpr bool PK_PREFETCH = true, PK_FORCE6 = true;    // (compiled for six waves per SIMD explicitly: left to itself the allocator takes 82-83 registers)
#else
constexpr bool PK_PREFETCH = false, PK_FORCE6 = false;
#endif
constexpr int PK_STACK_WORDS = PK_BUNDLE_AT + 24;
constexpr int PK_SLEVELS = 12;       // scene octree levels a packet can stack (deeper scene trees: k_intersect)
constexpr int PK_SFRAME_WORDS = 4;   // scene block, pending children, lanes (2)
#ifndef XRT_PK_QUEUES
#define XRT_PK_QUEUES 8
#endif
constexpr int PK_QUEUES = XRT_PK_QUEUES;   // interleaved heads of the packet queue (PacketArgs::queue points at PACKET_QUEUE_WORDS zeroed words)
static_assert(PK_QUEUES >= 1 && PK_QUEUES <= PACKET_QUEUE_HEADS, "the host zeroes PACKET_QUEUE_WORDS heads per packet launch");
#ifndef PK_SINGLE_WAVES
#define PK_SINGLE_WAVES 6             // waves per SIMD the one-body variants are compiled for (7: make variant DEFS=-DPK_SINGLE_WAVES=7 -- 96 SGPRs, 72 VGPRs)
#endif
constexpr int PK_SGPRS = PK_SINGLE_WAVES >= 7 ? 96 : 112;   // SGPR allocation the kernel may reach (checked against the ISA in tests/test_numerics_contract.py)

// make variant NAME=cnt DEFS=-DXRT_PK_COUNTERS: event counts of the shared walk (development aid; tools/pk_counters.py reads them through
// xrt_debug_packet_counters, which exists only in such a build)
#if defined(XRT_PK_COUNTERS) && !defined(XRT_PK_TICKS)
#define XRT_PK_TICKS   // (make variant NAME=ticks DEFS=-DXRT_PK_TICKS: only the packets' durations -- histogram and the longest one -- at the product build's speed; tools/pk_ticks.py)
#endif
#ifdef XRT_PK_TICKS
__device__ unsigned long long g_pkWorst[16];   // the packet that took longest: ticks, work item, segment, its walks / blocks / child visits / triangle steps / run tests, valid rays, packets of its launch
__device__ unsigned g_pkDump[2 * 65536];         // per packet of the launch's first 65536: ticks, block entries (the last launch that had more than 4096 packets wins)
__device__ unsigned long long g_pkTicks[32];   // packets by duration: bucket b counts packets of 2^b .. 2^(b+1) - 1 ticks of the 100 MHz device clock (xrt_debug_packet_ticks)
#endif
#ifdef XRT_PK_COUNTERS
__device__ unsigned long long g_pkCounters[16];
__device__ unsigned g_pkCur[8 * 65536];         // (per resident wave: this packet's running totals; indexed by a wave id below 65536)
#define PKC(i) (pkc[i]++)
#else
#define PKC(i) ((void)0)
#endif

__device__ unsigned long long g_splitStats[4];   // split walks (pk_walk): subtrees handed over, taken, packets split, packets whose results a taker wrote (xrt_split_stats)
struct alignas(4) TriWords { float w[16]; };   // a 13-word record of refT and the first three words of the next one
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
// min / max of v over the 64 lanes (DPP row shifts + the two row broadcasts, as wave_or; a lane without a source keeps its own value): wave-uniform
__device__ __forceinline__ float wave_fmin(float v) {
    const int inf = 0x7f800000;
    v = fminf(v, i2f(__builtin_amdgcn_update_dpp(inf, f2i(v), 0x111, 0xf, 0xf, false)));
    v = fminf(v, i2f(__builtin_amdgcn_update_dpp(inf, f2i(v), 0x112, 0xf, 0xf, false)));
    v = fminf(v, i2f(__builtin_amdgcn_update_dpp(inf, f2i(v), 0x114, 0xf, 0xf, false)));
    v = fminf(v, i2f(__builtin_amdgcn_update_dpp(inf, f2i(v), 0x118, 0xf, 0xf, false)));
    v = fminf(v, i2f(__builtin_amdgcn_update_dpp(inf, f2i(v), 0x142, 0xa, 0xf, false)));
    v = fminf(v, i2f(__builtin_amdgcn_update_dpp(inf, f2i(v), 0x143, 0xc, 0xf, false)));
    return i2f(__builtin_amdgcn_readlane(f2i(v), 63));
}
__device__ __forceinline__ float wave_fmax(float v) { return -wave_fmin(-v); }
// The packet's ray bundle (xrt_core.h RayBundle) over the lanes `part`: bounds of the origins and directions by wave reductions; the inverse directions, |D|_2 and
// the slack follow from the direction bounds (1f / d is correctly rounded, hence monotone; so are the sums of squares and of absolute values, evaluated in
// make_ray_cull's association).  Usable when every lane of `part` may use the fast box test and takes part in the tight-box tests (RayCull::d2 > 0) and every
// axis has one sign for all of them.  Lane 0 leaves it in the wave's LDS words bw[0..19], bw[20] = usable.
__device__ __forceinline__ void pk_bundle(unsigned *bw, int lane, bool part, bool okL, const RayPre &r) {
    const float inf = i2f(0x7f800000);
    const bool allOk = !__any(part && !okL) && __any(part);
    float v[12];
    v[0] = wave_fmin(part ? r.o.x : inf); v[1] = wave_fmin(part ? r.o.y : inf); v[2] = wave_fmin(part ? r.o.z : inf);
    v[3] = wave_fmax(part ? r.o.x : -inf); v[4] = wave_fmax(part ? r.o.y : -inf); v[5] = wave_fmax(part ? r.o.z : -inf);
    v[6] = wave_fmin(part ? r.d.x : inf); v[7] = wave_fmin(part ? r.d.y : inf); v[8] = wave_fmin(part ? r.d.z : inf);
    v[9] = wave_fmax(part ? r.d.x : -inf); v[10] = wave_fmax(part ? r.d.y : -inf); v[11] = wave_fmax(part ? r.d.z : -inf);
    if (lane == 0) {
        bool ok = allOk;
        for (int k = 0; k < 3; k++) ok = ok && ((v[6 + k] > 0.0f) || (v[9 + k] < 0.0f));   // one sign per axis, no zero (NaN: false)
        for (int k = 0; k < 12; k++) bw[k] = (unsigned)f2i(v[k]);
        for (int k = 0; k < 3; k++) {   // inverse directions: 1 / d falls as d grows
            bw[12 + k] = (unsigned)f2i(1.0f / v[9 + k]); bw[15 + k] = (unsigned)f2i(1.0f / v[6 + k]);
        }
        const float mx = fmaxf(fabsf(v[6]), fabsf(v[9])), my = fmaxf(fabsf(v[7]), fabsf(v[10])), mz = fmaxf(fabsf(v[8]), fabsf(v[11]));
        bw[18] = (unsigned)f2i(sqrtf((mx * mx + my * my) + mz * mz) * 1.000001f);   // >= RayCull::d2 of every lane
        bw[19] = (unsigned)f2i(((mx + my) + mz) * 4.7683716e-7f);                  // >= RayCull::slack
        bw[20] = ok ? 1u : 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// xrt_core.h bundle_certainly_missed, the same arithmetic, shaped for the register budget of k_packet (80 vector registers): the bundle's 20 words sit in the
// lanes of ONE vector register (lane k holds word k: a v_readlane hands a word to the scalar operand of the instruction that needs it), the record's four
// quarters are loaded where they are used, and the three phases -- rho, then one axis at a time -- are kept apart for the scheduler.
__device__ __forceinline__ float bword(int vB, int k) { return i2f(__builtin_amdgcn_readlane(vB, k)); }
__device__ __forceinline__ bool pk_bundle_missed(int vB, const f4 *__restrict__ t) {
    auto far1 = [](float o0, float o1, float lo, float hi) { return fmaxf(fmaxf(fabsf(o0 - lo), fabsf(o1 - lo)), fmaxf(fabsf(o0 - hi), fabsf(o1 - hi))); };
    const f4 a = t[0], b = t[1];
    float rho, ok;
    {
        const float fx = far1(bword(vB, 0), bword(vB, 3), a.x, b.x), fy = far1(bword(vB, 1), bword(vB, 4), a.y, b.y), fz = far1(bword(vB, 2), bword(vB, 5), a.z, b.z);
        const float tmax = sqrtf((fx * fx + fy * fy) + fz * fz) * 1.000001f;
        const f4 nl = t[2], nh = t[3];
        float lo, hi;
        {
            const float d0 = bword(vB, 6), d1 = bword(vB, 9);
            const float x0 = d0 * nl.x, x1 = d0 * nh.x, x2 = d1 * nl.x, x3 = d1 * nh.x;
            lo = min4(x0, x1, x2, x3); hi = max4(x0, x1, x2, x3);
        }
        {
            const float d0 = bword(vB, 7), d1 = bword(vB, 10);
            const float y0 = d0 * nl.y, y1 = d0 * nh.y, y2 = d1 * nl.y, y3 = d1 * nh.y;
            lo = lo + min4(y0, y1, y2, y3); hi = hi + max4(y0, y1, y2, y3);
        }
        {
            const float d0 = bword(vB, 8), d1 = bword(vB, 11);
            const float z0 = d0 * nl.z, z1 = d0 * nh.z, z2 = d1 * nl.z, z3 = d1 * nh.z;
            lo = lo + min4(z0, z1, z2, z3); hi = hi + max4(z0, z1, z2, z3);
        }
        const float cmin = fmaxf(lo, -hi) - bword(vB, 19);
        rho = (((a.w * (b.w + tmax)) * bword(vB, 18)) * (__builtin_amdgcn_rcpf(cmin) * 1.000001f)) * 1.00001f;
        ok = (nl.w > 0.0f && bword(vB, 18) > 0.0f && cmin > 0.0f && rho < 1.0e15f) ? 1.0f : 0.0f;
    }
    __builtin_amdgcn_sched_barrier(0);
    float tn = 0.0f, tf = FLT_MAX;
    auto axis = [&](float alo, float bhi, int k) {
        const float o0 = bword(vB, k), o1 = bword(vB, 3 + k), i0 = bword(vB, 12 + k), i1 = bword(vB, 15 + k);
        const float v1lo = (alo - rho) - o1, v1hi = alo - o0, v2lo = bhi - o1, v2hi = (bhi + rho) - o0;
        const float p0 = v1lo * i0, p1 = v1lo * i1, p2 = v1hi * i0, p3 = v1hi * i1;
        const float q0 = v2lo * i0, q1 = v2lo * i1, q2 = v2hi * i0, q3 = v2hi * i1;
        tn = fmaxf(tn, fminf(min4(p0, p1, p2, p3), min4(q0, q1, q2, q3)));
        tf = fminf(tf, fmaxf(max4(p0, p1, p2, p3), max4(q0, q1, q2, q3)));
        __builtin_amdgcn_sched_barrier(0);
    };
    axis(a.x, b.x, 0); axis(a.y, b.y, 1); axis(a.z, b.z, 2);
    return ok != 0.0f && tn > tf;
}

// A prefetch: an ordinary vector load whose value nobody looks at -- what it is for is the line's trip from memory into the L2 while the walk goes on.  ONE register
// carries it: the load's destination, which the NEXT prefetch folds into its own address as an opaque zero (so that the only wait for a prefetch is at the next one,
// long after it has arrived) and the walk's caller retires at the end.  (A load straight into LDS, global_load_lds, needs no register at all and upset the register
// allocation of the whole kernel: 140 vector registers.)
__device__ __forceinline__ void pk_prefetch(const char *base /* wave-uniform */, int laneOff, bool pred, int &pend) {
    int z = pend;
    asm volatile("v_and_b32 %0, 0, %0" : "+v"(z));
    if (pred) pend = *reinterpret_cast<const int *>(base + (unsigned)(laneOff + z));
}

// Data that waves hand each other inside a launch (split walks): every word of it is written with an agent-scope store and read with an agent-scope load (sc1: past the
// CU's L1 and whatever the XCD's L2 holds), ordered by waiting for the stores -- no cache-wide operation.  Measured on the way here (profiles/r04/split_walks.txt):
// __threadfence() (write the whole L2 back, invalidate it) costs every CU of the XCD ~1 us per call; a workgroup-scope (sc0) load may be served by the CU's L1 for ever;
// a workgroup-scope release fence is no instruction at all outside threadgroup-split mode, and two stores of one wave to two L2 channels do overtake each other.
__device__ __forceinline__ unsigned ld_ag(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_ag(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stores_done() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void loads_after() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }
// the XCD this wave runs on (HW_REG_XCC_ID = 20, bits 3:0; gfx950: eight XCDs with an L2 each)
__device__ __forceinline__ unsigned xcc_id() { return (unsigned)__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 7u; }
__device__ __forceinline__ float rflf(float v) { return i2f(__builtin_amdgcn_readfirstlane(f2i(v))); }

struct PkUniform {   // wave-uniform cursor
    int blk, p, dm0;
    unsigned long long lanes;
};
// One block of the walk as it arrives through the scalar cache (SceneView::pblocks, traverse.h PBLOCK_*): the descriptor and the planes of
// the eight child boxes.  No box is computed in the walk: a wave-uniform float lives in a vector register on gfx950, and deriving the child
// boxes from the parent's (what the per-lane kernel does instead of loading them) cost a packet ~25 vector instructions per block entered and
// ~15 per child visited -- all 64 lanes computing the same number.
struct alignas(16) PkBlockWords { float w[PBLOCK_WORDS]; };
struct alignas(16) Frame4 { unsigned a, b, c, d; };
// hit8_fast_children (traverse.h) on the block's stored planes: the reference's test MO:331 for all eight children, bit c <-> child c.
// Near / far plane of a slab by min / max of its two products -- BoundingBox.Intersects' own `if (t1 > t2) swap` -- instead of a select on
// the direction's sign (the same two numbers for a ray without NaNs: rounding is monotone), which needed three lane masks in scalar registers.
__device__ __forceinline__ int pk_hit8_fast(const RayPre &r, const PkBlockWords &B) {
    const float tx0 = (B.w[8] - r.o.x) * r.inv.x, tx1 = (B.w[9] - r.o.x) * r.inv.x, tx2 = (B.w[10] - r.o.x) * r.inv.x;
    const float ty0 = (B.w[12] - r.o.y) * r.inv.y, ty1 = (B.w[13] - r.o.y) * r.inv.y, ty2 = (B.w[14] - r.o.y) * r.inv.y;
    const float tz0 = (B.w[16] - r.o.z) * r.inv.z, tz1 = (B.w[17] - r.o.z) * r.inv.z, tz2 = (B.w[18] - r.o.z) * r.inv.z;
    const float nearX[2] = {fminf(tx0, tx1), fminf(tx1, tx2)}, farX[2] = {fmaxf(tx0, tx1), fmaxf(tx1, tx2)};
    const float nearY[2] = {fminf(ty0, ty1), fminf(ty1, ty2)}, farY[2] = {fmaxf(ty0, ty1), fmaxf(ty1, ty2)};
    const float nearZ[2] = {fminf(tz0, tz1), fminf(tz1, tz2)}, farZ[2] = {fmaxf(tz0, tz1), fmaxf(tz1, tz2)};
    int m = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int i = (c >> 2) & 1, j = (c >> 1) & 1, k = c & 1;
        const float num = fmaxf(fmaxf(fmaxf(nearX[i], nearY[j]), 0.0f), nearZ[k]);
        const float num2 = fminf(fminf(fminf(farX[i], farY[j]), FLT_MAX), farZ[k]);
        if (!(num > num2)) m |= 1 << c;
    }
    return m;
}
// The box of child c as child_box (traverse.h) forms it, read from the planes (c is wave-uniform: scalar selects).
__device__ __forceinline__ void pk_child_box(const PkBlockWords &B, int c, v3 &cmin, v3 &cmax) {
    // (selected as integers: a select between two wave-uniform floats is compiled as a vector select)
    auto pick = [](int bit, float hi, float lo) { return i2f(rfl(bit ? f2i(hi) : f2i(lo))); };
    cmin = mk(pick(c & 4, B.w[9], B.w[8]), pick(c & 2, B.w[13], B.w[12]), pick(c & 1, B.w[17], B.w[16]));
    cmax = mk(pick(c & 4, B.w[10], B.w[11]), pick(c & 2, B.w[14], B.w[15]), pick(c & 1, B.w[18], B.w[19]));
}
// The same decisions with the literal box test (a ray with a parallel axis or a non-finite component, MO:331).
__device__ __forceinline__ int pk_hit8_slow(const RayPre &r, const PkBlockWords &B) {
    int m = 0;
    for (int c = 0; c < 8; c++) {
        v3 cmin, cmax;
        pk_child_box(B, c, cmin, cmax);
        float key;
        if (slab(r, cmin.x, cmin.y, cmin.z, cmax.x, cmax.y, cmax.z, key)) m |= 1 << c;
    }
    return m;
}


// The triangle test of a packet, shaped for the SCALAR unit: every `if` on a per-lane condition costs the wave two or three scalar
// instructions and a branch whether or not a lane takes it (the nested early rejections of xrt_core.h tri_stage_a / tri_stage_b and
// traverse.h leaf_candidate were 53 scalar instructions and 13 branches per triangle that some lane hits; this form is 17 and 4), and the
// scalar unit is the busier of this kernel's two.  RE:42-75 as TWO predicates: the certainly_negative() tests become sign-bit arithmetic
// on the operands' bit patterns (vector instructions) and are OR-ed into one word with the ignored-triangle test; the lanes that pass
// both divide exactly as RE:66-74 does.  (All of it as ONE predicate -- both cross products for every lane -- was measured too: C5 1 %
// slower than this, C3 / C4 3 % slower than the nested form; profiles/r03/packet_triangle_test_forms.txt.)
//   cn_bits(a, det): sign bit set iff certainly_negative(a, det) -- opposite signs, |a| >= 2^-60, |det| <= 2^60.  (A NaN `a` counts as large
//   here and as small there: either way the triangle is rejected, there by the comparisons after the division, which are false for NaNs.)
__device__ __forceinline__ int cn_bits(float a, int detBits, int detSmall) {
    const int ia = f2i(a);
    return (ia ^ detBits) & ~((ia & 0x7fffffff) - 0x21800000 /* 2^-60 */) & detSmall;
}
// Two predicates: (1) front-facing, not the ignored triangle, u not certainly negative -- the first cross product and two dot products,
// which reject most triangles for every lane of a packet; (2) for the lanes that are left, everything else at once.
struct PkTriMid { v3 T; float det, row2; };
__device__ __forceinline__ bool pk_tri_pre(const TriWords &q, const Lane &L, PkTriMid &m) {
    const v3 O = L.r.o, D = L.r.d;
    const v3 v1 = mk(q.w[4], q.w[5], q.w[6]), E1 = mk(q.w[7], q.w[8], q.w[9]), E2 = mk(q.w[10], q.w[11], q.w[12]);
    const float fc = facing(mk(q.w[0], q.w[1], q.w[2]), D);   // RE:48-51
    m.T = mk(O.x - v1.x, O.y - v1.y, O.z - v1.z);            // RE:46
    const v3 P = cross(D, E2);                                // RE:58
    m.det = dot(P, E1);                                       // RE:66 (the divisor)
    m.row2 = dot(P, m.T);                                     // RE:63
    const int db = f2i(m.det), detSmall = ~(0x5d800000 /* 2^60 */ - (db & 0x7fffffff));
    const int x = f2i(q.w[3]) ^ L.ignoreId;                   // MO:290: (x - 1) & ~x has its sign bit set iff x == 0, the ignored triangle
    int bad = cn_bits(m.row2, db, detSmall) | ((x - 1) & ~x); // u < 0, certainly
    bad = fc > 0.0f ? -1 : bad;                               // RE:50
    return bad >= 0;
}
// RE:59-74 and MO:293-294 for a lane that passed pk_tri_pre: v < 0 or distance < 0 certainly (sign bits), else the reference's division and
// u >= 0, v >= 0, distance >= 0, u + v <= 1, distance < float.MaxValue.  A NaN among the three fails u + v <= 1 or distance < MaxValue (as it
// fails the reference's comparisons), so the three sign tests may be one minimum, whatever it makes of a NaN.
__device__ __forceinline__ bool pk_tri_hit(const TriWords &q, const Lane &L, const PkTriMid &m, float &u, float &v, float &t) {
    const v3 D = L.r.d, E1 = mk(q.w[7], q.w[8], q.w[9]), E2 = mk(q.w[10], q.w[11], q.w[12]);
    const v3 Q = cross(m.T, E1);                              // RE:59
    const float row3 = dot(Q, D);                             // RE:64
    const float row1 = dot(Q, E2);                            // RE:62
    const int db = f2i(m.det), detSmall = ~(0x5d800000 - (db & 0x7fffffff));
    const int bad = cn_bits(row3, db, detSmall) | cn_bits(row1, db, detSmall);
    const float inv = 1.0f / m.det;                           // RE:66
    t = row1 * inv; u = m.row2 * inv; v = row3 * inv;
    const float s0 = bad >= 0 ? fminf(fminf(u, v), t) : -1.0f;
    const float s = s0 >= 0.0f ? u + v : 2.0f;
    const float d = s <= 1.0f ? t : FLT_MAX;
    return d < FLT_MAX;
}
// leaf_candidate (traverse.h) for a lane whose triangle passed, without nested branches: the bucket rule's comparison is evaluated as one
// predicate; only an exact tie of key AND distance (two leaves holding the same triangle) goes to memory for the leaves' DFS numbers.
__device__ __forceinline__ void pk_candidate(Lane &L, const SceneView &S, int r, float u, float v, float t) {
    const bool eq = L.leafKey == L.mKey;
    bool better = (L.mfound == 0) | (L.leafKey < L.mKey) | (eq & (t < L.mDist));
    if (eq & (t == L.mDist)) {   // (rare)
        if ((L.mfound != 0) & (L.leafNode != L.mLeaf)) better = node_dfs(S, L.leafNode) < node_dfs(S, L.mLeaf);
        else if (L.mfound != 0) better = f2i(S.refN[r].w) < f2i(S.refN[L.mRef].w);   // one leaf: the earlier in the reference's list = the smaller triangle index (scene_host.cpp spatial_runs)
    }
    L.mfound = 1;   // (every lane here has a candidate now, its old one or this)
    L.mKey = better ? L.leafKey : L.mKey; L.mDist = better ? t : L.mDist; L.mU = better ? u : L.mU; L.mV = better ? v : L.mV;
    L.mRef = better ? r : L.mRef; L.mLeaf = better ? L.leafNode : L.mLeaf;
}

// MO:288-304 for the lanes selected by the caller (exec), over the references r0 .. r1-1 of one leaf.  The triangle is the same
// for every lane: normals and geometry come through the scalar cache.  Two register sets take turns, and a triangle's 52 bytes are
// requested before the previous one's arithmetic starts (the records of a leaf are back to back; the arrays end in two dummy
// records, so asking one past the leaf is safe).  One record of 13 words (refT: normal, id, v1, E1, E2) = one s_load_dwordx16
// (the three words past it belong to the next record; the array ends in padding): the scalar unit is this kernel's scarce
// resource, and a triangle used to cost three loads from two streams.  `keyed`: L.leafKey holds the leaf's entry key already;
// otherwise entry_key() is evaluated by the lanes that find a candidate.
template <class KeyFn>
__device__ __forceinline__ void pk_scan_leaf(const float *__restrict__ refT, int r0, int r1, Lane &L, const SceneView &S, bool keyed, KeyFn entry_key) {
    const char *pt = reinterpret_cast<const char *>(refT) + (size_t)r0 * TRI_REC_BYTES;
    auto test = [&](const TriWords &q, int r) {
        float u, v, t;
        PkTriMid m;
        if (pk_tri_pre(q, L, m)) {
            if (pk_tri_hit(q, L, m, u, v, t)) {
                if (!keyed) L.leafKey = entry_key();
                pk_candidate(L, S, r, u, v, t);
            }
        }
    };
    TriWords qA = *reinterpret_cast<const TriWords *>(pt);
    int r = r0;
    for (;;) {
        const TriWords qB = *reinterpret_cast<const TriWords *>(pt + TRI_REC_BYTES);
        test(qA, r);
        if (r + 1 >= r1) break;
        qA = *reinterpret_cast<const TriWords *>(pt + 2 * TRI_REC_BYTES);
        test(qB, r + 1);
        r += 2; pt += 2 * TRI_REC_BYTES;
        if (r >= r1) break;
    }
}

// Arguments a packet needs once -- its ray / hit arrays, the queue, the tile-cost words, the arrays lane_result reads -- are RE-READ from the
// kernel-argument segment where they are used (scalar loads from constant memory, the pointer passed through an empty asm per packet so
// that the loads stay inside the loop) instead of living in scalar registers across the walk: the allocator kept them in SGPRs spilled to
// vector-register lanes (v_writelane / v_readlane) around every walk.  Offsets: the eight pointer parameters, then SceneView, then PacketArgs,
// each at its natural alignment (checked against the code object's metadata in tests/test_numerics_contract.py).
constexpr unsigned PK_KERNARG_SCENE = 8 * 8, PK_KERNARG_ARGS = PK_KERNARG_SCENE + (unsigned)sizeof(SceneView);
static_assert(sizeof(SceneView) % 8 == 0 && alignof(PacketArgs) == 8, "kernel-argument offsets of k_packet");
typedef const __attribute__((address_space(4))) PacketArgs *PkArgsK;
typedef const __attribute__((address_space(4))) SceneView *PkSceneK;
struct PkKernarg {
    unsigned long long base;
    __device__ __forceinline__ PkKernarg() : base((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr()) {}
    // (the loads that follow cannot be hoisted above this point: the pointer's halves go through an empty asm; the readfirstlane in front
    // of it guarantees the asm's scalar-register operand whatever register class the allocator keeps `base` in)
    __device__ __forceinline__ void fresh() {
        unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32));
        asm volatile("" : "+s"(lo), "+s"(hi));
        base = (unsigned long long)lo | ((unsigned long long)hi << 32);
    }
    __device__ __forceinline__ PkArgsK args() const { return (PkArgsK)(base + PK_KERNARG_ARGS); }
    __device__ __forceinline__ PkSceneK scene() const { return (PkSceneK)(base + PK_KERNARG_SCENE); }
    // what lane_result (traverse.h) reads of the scene
    __device__ __forceinline__ SceneView result_view() const {
        const PkSceneK k = scene();
        SceneView v = {};
        v.refG = k->refG; v.refN = k->refN; v.meshes = k->meshes; v.childDfs = k->childDfs; v.objects = k->objects;
        return v;
    }
};

// One shared walk of the mesh octree whose root block is `rootBlock` for the lanes `lanes0`: the lanes whose ray passed the root's own
// box test (MO:265 / MO:331).  On return every lane's L.mfound / mKey / mDist / mU / mV / mRef / mLeaf hold its answer of
// MeshOctree.GetRayIntersection (the caller cleared mfound).  Everything the walk reads comes from three arrays -- pblocks (descriptor +
// child planes per block), lrec (per node: normal box, tight box, first run; the run records behind them), refT (triangles) --: three base
// pointers in scalar registers where round 3 held six, no parent box, no child box arithmetic.
//
// Split walks (`budget` != 0, PacketArgs::splitCtl).  A launch ends when its longest packet does, and packets are heavy-tailed: rays that skim a terrain near the
// horizon walk tens of times the median (a 1/8 tile shard of C5: median ~50 us, the longest 722 us -- longer than the whole launch should take).  The walk is one
// instruction stream, but the arg-min it computes does not care who visits what: a walk that has outlasted `budget` ticks of the 100 MHz device clock hands the
// PENDING children of its stacked levels (the later siblings of the path it is on) to other waves -- per level one ITEM (block, pending children, lanes, the lanes'
// accepted children, the lanes' best answers so far as a pruning hint) in this launch's item queue -- and walks on with what is left.  A wave that takes an item
// sets the packet's rays up again, seeds level 0 of its stack with the item and starts by "coming back" to it (`resume`); it may split again.  Every participant
// leaves its lanes' answers in the arena; the one that finishes last (PacketArgs::splitRecs [0], a count of units outstanding: nobody ever waits for anybody)
// merges them with the rule of DESIGN.md §3 and writes the packet's results.
template <bool SPLIT, bool BUNDLE = PK_BUNDLE>
__device__ __forceinline__ void pk_walk(const float *__restrict__ pblocks, const float *__restrict__ refT, const float *__restrict__ lrec,
                                        const SceneView &S, int cullMin, unsigned *stk, int lane, Lane &L,
                                        const RayCull &RC, bool fastL, int rootBlock, unsigned long long lanes0, bool nodeCull,
                                        bool resume, bool prefetch, int &pfAcc) {
#ifdef XRT_PK_COUNTERS
    unsigned pkc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    pkc[0] = 1;
    pkc[12] = (unsigned)__popcll(lanes0);
#endif
    PkUniform U;
    U.blk = rootBlock;
    U.lanes = lanes0;
    U.p = 0; U.dm0 = 0;
    unsigned char *const cbOf = reinterpret_cast<unsigned char *>(stk + PK_LEVELS * PK_FRAME_WORDS) + lane;   // [level * 64]: this lane's byte
    int sp = resume ? 1 : 0;   // (resume: level 0 holds the item, the walk starts by coming back to it)
    bool entering = !resume, anyFound = resume && __any(L.mfound != 0);   // anyFound: some lane of the wave has a candidate
    PkBlockWords B;
    int cb = 0;
    unsigned *const sst = stk + PK_SPLIT_AT;   // [0] the packet's record (-1: none), [1] the item being traced (-1: the packet itself), [2] budget (0: off), [3] the clock it counts from, [4] packet
    // Hand the pending children of levels 0 .. sp-1 to other waves of this XCD (wave-uniform, rare).  What they read of it has arrived (stores_done) before the item's
    // `ready` word is written; the counters -- queue, units outstanding, items of a record -- are only ever touched by atomics and agent-scope loads.
    auto spill = [&]() {
        PkKernarg K;
        K.fresh();
        const PkArgsK a = K.args();
        unsigned *const ctl = a->splitCtl, *const itemsB = a->splitItems, *const recsB = a->splitRecs;
        const int NR = a->splitNR;
        const unsigned serial = a->splitSerial, xcc = xcc_id(), NIx = (unsigned)a->splitNI / 8u;
        const unsigned takerBudget = (unsigned)a->splitBudgetItem, takerEvery = (unsigned)rfl((int)sst[7]);   // (an eagerly split packet stays one)
        int rec = rfl((int)sst[0]);
        for (int lvl = 0; lvl < sp; lvl++) {
            const Frame4 f = *reinterpret_cast<const Frame4 *>(stk + lvl * PK_FRAME_WORDS);
            const unsigned p = (unsigned)rfl((int)f.b);
            if (p == 0u) continue;
            if (rec < 0) {   // the packet's record: one unit outstanding (whoever traces the packet itself), no items yet
                unsigned r = 0u;
                if (lane == 0) r = atomicAdd(ctl + 256, 1u);
                rec = rfl((int)r);
                if (rec >= NR) { rec = -1; break; }
                if (lane == 0) {
                    unsigned *const R0 = recsB + (size_t)rec * SPLIT_REC_WORDS;
                    (void)atomicExch(R0, 1u); (void)atomicExch(R0 + 1, 0u);
                    sst[0] = (unsigned)rec;
                    atomicAdd(&g_splitStats[2], 1ull);
                }
            }
            unsigned *const R = recsB + (size_t)rec * SPLIT_REC_WORDS;
            unsigned j = 0u, idx = 0u;
            if (lane == 0) j = atomicAdd(R + 1, 1u);
            j = (unsigned)rfl((int)j);
            if (j >= (unsigned)SPLIT_REC_ITEMS) break;   // (readers clamp the count)
            if (lane == 0) idx = atomicAdd(ctl + 32u * xcc, 1u);
            idx = (unsigned)rfl((int)idx);
            if (idx >= NIx) { if (lane == 0) st_ag(R + 32 + j, 0xffffffffu); break; }   // (takers clamp the tail)
            idx += xcc * NIx;
            unsigned *const I = itemsB + (size_t)idx * SPLIT_ITEM_WORDS;
            if (lane == 0) {
                st_ag(I + 1, sst[4]); st_ag(I + 2, (unsigned)rec); st_ag(I + 3, (unsigned)rfl((int)f.a)); st_ag(I + 4, p); st_ag(I + 5, (unsigned)rfl((int)f.c)); st_ag(I + 6, (unsigned)rfl((int)f.d)); st_ag(I + 7, takerBudget); st_ag(I + 8, takerEvery);
                st_ag(R + 32 + j, idx);
                (void)atomicAdd(R, 1u);                  // one more unit outstanding, before anybody can take it
                atomicAdd(&g_splitStats[0], 1ull);
                stk[lvl * PK_FRAME_WORDS + 1] = 0u;      // these children are no longer this walk's
            }
            st_ag(I + SPLIT_HEAD_WORDS + lane, (unsigned)cbOf[lvl * 64]);
            unsigned *const P = I + SPLIT_HEAD_WORDS + 64 + lane;
            st_ag(P, (unsigned)L.mfound); st_ag(P + 64, (unsigned)f2i(L.mKey)); st_ag(P + 128, (unsigned)f2i(L.mDist)); st_ag(P + 192, (unsigned)f2i(L.mU)); st_ag(P + 256, (unsigned)f2i(L.mV));
            st_ag(P + 320, (unsigned)L.mRef); st_ag(P + 384, (unsigned)L.mLeaf);
            stores_done();
            if (lane == 0) st_ag(I, serial);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // The scalar unit is shared by the CU's four SIMDs, so scalar instructions are the scarce resource of this kernel
    // (measured: 3,900 per packet against 4,000 vector ones made it scalar-bound): the pending children are a bit mask
    // walked with ctz, the lanes of a leaf are selected once for the whole leaf, a triangle costs one wave-level branch.
    while (U.lanes != 0ull) {
        const bool in = ((U.lanes >> lane) & 1ull) != 0;
        if (entering) {
            PKC(1);
#ifdef XRT_PK_TICKS
            if (lane == 0) stk[PK_SPLIT_AT + 6] += 1u;   // (development: block entries of this packet, tools/pk_ticks.py --dump)
#endif
#ifdef XRT_PK_PRIO
            {   // the longer a walk has been going on, the higher its wave's priority at the instruction arbiter: a launch ends when its longest packet does, and a packet
                // of 60-80 block entries (the median makes 15) shares its SIMD with five waves that would not miss the issue slots
                const unsigned c = (unsigned)rfl((int)stk[PK_SPLIT_AT + 5]) + 1u;
                if (lane == 0) stk[PK_SPLIT_AT + 5] = c;
                if (c == (unsigned)XRT_PK_PRIO) __builtin_amdgcn_s_setprio(1);
                else if (c == 2u * XRT_PK_PRIO) __builtin_amdgcn_s_setprio(2);
                else if (c == 3u * XRT_PK_PRIO) __builtin_amdgcn_s_setprio(3);
            }
#endif
            if constexpr (SPLIT) {   // (wave-uniform) the walk's cost in block entries (what the same packet of the next frame is judged by); has it been going on for long,
                                     // or is it a packet that was long last time (every sst[7] block entries)?  Then let others have what is pending above
                const unsigned cnt = (unsigned)rfl((int)sst[6]) + 1u, every = (unsigned)rfl((int)sst[7]);
                if (lane == 0) sst[6] = cnt;
                if (sp > 0) {
                    const unsigned budget = (unsigned)rfl((int)sst[2]);
                    const bool late = budget != 0u && (unsigned)wall_clock64() - (unsigned)rfl((int)sst[3]) > budget;
                    if (late || (every != 0u && cnt % every == 0u)) {
                        spill();
                        if (lane == 0) sst[3] = (unsigned)wall_clock64();
                    }
                }
            }
            B = *reinterpret_cast<const PkBlockWords *>(pblocks + (size_t)U.blk * PBLOCK_WORDS);
            // One level ahead: the eight children's node records (1 KB in a row) and their block records are asked for with ONE vector load, so that a walk through
            // parts of the tree no other wave has touched lately pays one trip to memory per level instead of one per record (the loads' values are never looked at:
            // they are OR-ed into a word the walk's caller throws away).
            if (PK_PREFETCH && prefetch) {   // (wave-uniform)
                const int cb0 = f2i(B.w[0]);
                pk_prefetch(reinterpret_cast<const char *>(lrec + (size_t)U.blk * 8 * LREC_WORDS), lane * 64, lane < 16, pfAcc);
                pk_prefetch(reinterpret_cast<const char *>(pblocks + (size_t)cb0 * PBLOCK_WORDS), lane * 64, lane < 10, pfAcc);
            }
            cb = 0;
            if (in) cb = fastL ? pk_hit8_fast(L.r, B) : pk_hit8_slow(L.r, B);
            cb &= 0xff & ~((f2i(B.w[2]) >> 8) & 0xff);   // empty leaves can never hit (Q4)
            U.dm0 = __builtin_amdgcn_readlane(L.dmask, (int)__builtin_ctzll(U.lanes));   // front-to-back order of the first lane
            // children some lane entered, in that order: bit p <-> child (p ^ dm0).  The xor moves a bit by 4, 2 and 1 places; each lane
            // moves its own bits (vector shifts by wave-uniform amounts: 0 leaves the byte as it is) before the wave's OR, which costs the
            // scalar unit three instructions where permuting the OR's result cost it twenty-four.
            const int a4 = U.dm0 & 4, a2 = U.dm0 & 2, a1 = U.dm0 & 1;
            int q = cb;
            q = ((q << a4) | (q >> a4)) & 0xff;
            q = ((q & 0x33) << a2) | ((q & 0xcc) >> a2);
            q = ((q & 0x55) << a1) | ((q & 0xaa) >> a1);
            U.p = wave_or(q);
            entering = false;
        }
        if (U.p == 0) {   // block exhausted: back to the level above
            if (sp == 0) break;
            sp--;
            PKC(10);
            const Frame4 f = *reinterpret_cast<const Frame4 *>(stk + sp * PK_FRAME_WORDS);
            U.blk = rfl((int)f.a); U.p = rfl((int)f.b);
            U.lanes = (unsigned long long)(unsigned)rfl((int)f.c) | ((unsigned long long)(unsigned)rfl((int)f.d) << 32);
            U.dm0 = __builtin_amdgcn_readlane(L.dmask, (int)__builtin_ctzll(U.lanes));   // (the order the pending bits of this level were made in)
            B = *reinterpret_cast<const PkBlockWords *>(pblocks + (size_t)U.blk * PBLOCK_WORDS);
            cb = (int)cbOf[sp * 64];
            continue;
        }
        const int c = (int)__builtin_ctz((unsigned)U.p) ^ U.dm0;
        U.p &= U.p - 1;
        PKC(2);
        const bool inC = in && ((cb >> c) & 1);
        const int d0 = f2i(B.w[0]), d1 = f2i(B.w[1]), d2 = f2i(B.w[2]), d3 = f2i(B.w[3]);
        // The child's own test gives the entry key (its outcome is known: hit).  The bucket rule compares keys only once a lane
        // has a candidate; until some lane of the wave has one (most of a packet's walk) the key of a leaf is computed by the
        // lanes that find a candidate in it, and nobody computes the key of an interior child.
        auto entry_key = [&]() {
            v3 cmin, cmax;
            pk_child_box(B, c, cmin, cmax);
            float k = 0.0f;
            if (fastL) (void)slab_fast(L.r, L.dmask, cmin, cmax, k);
            else (void)slab(L.r, cmin.x, cmin.y, cmin.z, cmax.x, cmax.y, cmax.z, k);
            return k;
        };
        const bool keyed = anyFound;   // wave-uniform
        float key = 0.0f;
        if (keyed && inC) key = entry_key();
#ifdef XRT_PK_COUNTERS
        if (keyed) pkc[11]++;
        pkc[13] += (unsigned)__popcll(__ballot(inC));
#endif
        const int node = U.blk * 8 + c;
        if (!((d2 >> c) & 1)) {   // ---- leaf (non-empty): MO:288-304 for the lanes the bucket rule lets in ----
            PKC(3);
            const f4 *const nr = reinterpret_cast<const f4 *>(lrec + (size_t)node * LREC_WORDS);
            const f4 nlo = nr[0], nhi = nr[1];
            bool go = inC & !all_back_facing(nlo, nhi, L.r.d);
            if (keyed) go = go & !((L.mfound != 0) & (key > L.mKey));   // (wave-uniform branch: nobody has a candidate before `keyed`)
            if (!__any(go)) continue;
            PKC(4);
            const unsigned long long offLo = (unsigned long long)(unsigned)f2i(B.w[4]) | ((unsigned long long)(unsigned)f2i(B.w[5]) << 32);
            const unsigned long long offHi = (unsigned long long)(unsigned)f2i(B.w[6]) | ((unsigned long long)(unsigned)f2i(B.w[7]) << 32);
            const int r0 = d1 + child_ref_offset(offLo, offHi, c);
            const int r1 = d1 + ((c == 7) ? d3 : child_ref_offset(offLo, offHi, c + 1));
            if (r1 - r0 >= cullMin) {   // lanes whose ray cannot reach any triangle of the leaf (xrt_core.h leaf_certainly_missed) stay out of it
                go = go && !leaf_certainly_missed(L.r, RC, nr[2], nr[3], nr[4], nr[5]);
                if (!__any(go)) continue;
            }
            PKC(5);
            if (PK_PREFETCH && prefetch) {   // the leaf's triangle records (52 bytes each, back to back) in one vector load; see above
                const int bytes = (r1 - r0) * TRI_REC_BYTES;
                pk_prefetch(reinterpret_cast<const char *>(refT) + (size_t)r0 * TRI_REC_BYTES, lane * 64, lane * 64 < bytes, pfAcc);
            }
            // A leaf of LEAF_RUN_MIN references or more is scanned run by run (LEAF_RUN references each): every run has a
            // tight box of its own and a run no lane can reach is passed over -- the octree stops splitting at 50 triangles (MO:42) and a
            // coherent packet comes near only a few of them.  Smaller leaves are one run.
            // Bundle prefilter (xrt_core.h bundle_certainly_missed): 64 lanes look at 64 TRIANGLES of the leaf at once -- can any ray of the packet reach this one? -- and
            // only the triangles some ray may reach get the rays' own tests (in list order: MO:293-294's strict '<' holds).  A leaf's triangles fill a thin sheet of its
            // cube and the 64 rays of a packet are a narrow beam: of C5's 56 triangle steps per primary packet most were made for triangles no lane comes near.
            if (BUNDLE && r1 - r0 >= cullMin && rfl((int)stk[PK_BUNDLE_AT + 20]) != 0) {
                const int vB = (int)stk[PK_BUNDLE_AT + (lane & 31)];   // lane k < 20: word k of the bundle
                PkKernarg K;
                K.fresh();
                const f4 *const triTB = K.scene()->triTB;
                L.leafKey = key; L.leafNode = node;
                for (int base = r0; base < r1; base += 64) {
                    const int rr = base + lane;
                    bool maybe = false;
                    if (rr < r1) {
                        const f4 *const t = triTB + 4 * (size_t)rr;
                        maybe = !pk_bundle_missed(vB, t);
                    }
                    unsigned long long M = __ballot(maybe);
                    PKC(6);
                    if (M == 0ull) continue;
                    PKC(7);
                    auto test = [&](const TriWords &q, int r) {
                        float u, v, t;
                        PkTriMid m;
                        if (go && pk_tri_pre(q, L, m)) {
                            if (pk_tri_hit(q, L, m, u, v, t)) {
                                if (!keyed) L.leafKey = entry_key();
                                pk_candidate(L, S, r, u, v, t);
                            }
                        }
                    };
                    int j = (int)__builtin_ctzll(M);
                    M &= M - 1ull;
                    TriWords qA = *reinterpret_cast<const TriWords *>(reinterpret_cast<const char *>(refT) + (size_t)(base + j) * TRI_REC_BYTES);
                    for (;;) {
                        const bool more = M != 0ull;
                        int j2 = j;
                        if (more) { j2 = (int)__builtin_ctzll(M); M &= M - 1ull; }
                        const TriWords qB = *reinterpret_cast<const TriWords *>(reinterpret_cast<const char *>(refT) + (size_t)(base + j2) * TRI_REC_BYTES);   // (the next one, requested before this one's arithmetic)
#ifdef XRT_PK_COUNTERS
                        pkc[8]++; pkc[14] += (unsigned)__popcll(__ballot(go));
#endif
                        test(qA, base + j);
                        if (!more) break;
                        qA = qB; j = j2;
                    }
                }
                if (!keyed) anyFound = __any(L.mfound != 0);
                continue;
            }
            int nRuns = 1, rb = -1;
            if (r1 - r0 >= LEAF_RUN_MIN) { rb = f2i(lrec[(size_t)node * LREC_WORDS + 24]); if (rb >= 0) nRuns = (r1 - r0 + LEAF_RUN - 1) / LEAF_RUN; }
            for (int jr = 0; jr < nRuns; jr++) {
                int ra = r0, rz = r1;
                bool goR = go;
                if (rb >= 0) {
                    ra = r0 + LEAF_RUN * jr; rz = min(ra + LEAF_RUN, r1);
                    const f4 *tb = reinterpret_cast<const f4 *>(lrec + (size_t)rb + (size_t)jr * RUN_WORDS);
                    PKC(6);
                    goR = go && !leaf_certainly_missed(L.r, RC, tb[0], tb[1], tb[2], tb[3]);
                    if (!__any(goR)) continue;
                }
                PKC(7);
#ifdef XRT_PK_COUNTERS
                pkc[8] += (unsigned)(rz - ra);
                pkc[14] += (unsigned)(rz - ra) * (unsigned)__popcll(__ballot(goR));
#endif
                if (goR) {   // the lanes of this run, selected once for all its triangles
                    L.leafKey = key; L.leafNode = node;
                    // (kept inline: as a function of its own the same loop costs 13 more VGPRs, i.e. the sixth wave per SIMD)
                    const char *pt = reinterpret_cast<const char *>(refT) + (size_t)ra * TRI_REC_BYTES;
                    auto test = [&](const TriWords &q, int r) {
                        float u, v, t;
                        PkTriMid m;
                        if (pk_tri_pre(q, L, m)) {   // one wave-level branch per triangle (s_cbranch_execz): most are rejected here for every lane
                            if (pk_tri_hit(q, L, m, u, v, t)) {
                                if (!keyed) L.leafKey = entry_key();
                                pk_candidate(L, S, r, u, v, t);
                            }
                        }
                    };
                    TriWords qA = *reinterpret_cast<const TriWords *>(pt);
                    int r = ra;
                    for (;;) {
                        const TriWords qB = *reinterpret_cast<const TriWords *>(pt + TRI_REC_BYTES);
                        test(qA, r);
                        if (r + 1 >= rz) break;
                        qA = *reinterpret_cast<const TriWords *>(pt + 2 * TRI_REC_BYTES);
                        test(qB, r + 1);
                        r += 2; pt += 2 * TRI_REC_BYTES;
                        if (r >= rz) break;
                    }
                }
            }
            if (!keyed) anyFound = __any(L.mfound != 0);
            continue;
        }
        // ---- interior child: a lane prunes it by key only where that is a proven lower bound (safe bit) ----
        bool go = inC;
        if (keyed && ((d2 >> (16 + c)) & 1)) go = go & !((L.mfound != 0) & (key > L.mKey));   // (wave-uniform branch)
        if (nodeCull && ((d2 >> (24 + c)) & 1)) {   // (wave-uniform) a subtree whose triangles all face away from a lane's ray (RE:48-51 rejects each one) is not entered by that lane
            const f4 *const nr = reinterpret_cast<const f4 *>(lrec + (size_t)node * LREC_WORDS);
            go = go && !all_back_facing(nr[0], nr[1], L.r.d);
            PKC(9);
        }
        const unsigned long long LL = __ballot(go);
        if (LL == 0ull) continue;
        if (sp >= PK_LEVELS - 1) continue;   // (cannot happen: packet_supported checks the depth)
        if (U.p != 0) {   // something is left to do at this level: come back
            cbOf[sp * 64] = (unsigned char)cb;   // (every lane: its own accepted children of this level's block)
            if (lane == 0) *reinterpret_cast<Frame4 *>(stk + sp * PK_FRAME_WORDS) = Frame4{(unsigned)U.blk, (unsigned)U.p, (unsigned)U.lanes, (unsigned)(U.lanes >> 32)};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            sp++;
        }
        U.blk = d0 + __builtin_popcount((unsigned)(d2 & 0xff) & ((1u << c) - 1u));
        U.lanes = LL;
        entering = true;
    }
#ifdef XRT_PK_COUNTERS
    if (lane == 0) for (int i = 0; i < 16; i++) if (pkc[i]) atomicAdd(&g_pkCounters[i], (unsigned long long)pkc[i]);
    if (lane == 0) {
        unsigned *cur = g_pkCur + 8 * ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 65535);
        cur[0] += pkc[0]; cur[1] += pkc[1]; cur[2] += pkc[2]; cur[3] += pkc[8]; cur[4] += pkc[6];
    }
#endif
}

// Scene mode: what a lane keeps for the whole query but touches only between mesh walks -- the world ray and its scene-level best
// answer (OSM:370-378) -- lives in LDS ([word][lane], device_util.h LdsField): the mesh walk then has the registers of the
// one-body variants.
constexpr int PK_PARK_WORDS = 19;
struct PkScene {
    LdsRay w;                       // world ray (9 words)
    LdsField<float> sKey;           // entry key of the scene leaf being scanned (OSM:334 bucket key)
    LdsField<int> sfound;
    LdsField<float> sbKey, sbD, sbU, sbV;
    LdsField<int> sbRef, sbLeaf, sbObj, sbMesh;
    int obj;                        // (wave-uniform: the body being visited)
    __device__ __forceinline__ explicit PkScene(unsigned *b)   // b = &park[wave][0][lane]
        : w{b}, sKey{b + 9 * 64}, sfound{b + 10 * 64}, sbKey{b + 11 * 64}, sbD{b + 12 * 64}, sbU{b + 13 * 64}, sbV{b + 14 * 64},
          sbRef{b + 15 * 64}, sbLeaf{b + 16 * 64}, sbObj{b + 17 * 64}, sbMesh{b + 18 * 64}, obj(-1) {}
};
template <int M> __device__ __forceinline__ unsigned *scene_park() {
    if constexpr (M == MODE_SCENE) { __shared__ unsigned mem[4 * PK_PARK_WORDS * 64]; return mem; }
    else return nullptr;
}
template <int M> __device__ __forceinline__ unsigned *scene_frames() {
    if constexpr (M == MODE_SCENE) { __shared__ unsigned mem[4 * PK_SLEVELS * PK_SFRAME_WORDS]; return mem; }
    else return nullptr;
}

#ifndef PK_SCENE_WAVES
#define PK_SCENE_WAVES 5   // waves per SIMD the scene variant is compiled for: 96 VGPRs + 44 bytes of scratch per lane, touched per packet (not per step): C3 -5 %, C4 -7.5 % against 4 waves at 106 VGPRs (profiles/r03/packet_scene_five_waves.txt; at 111 VGPRs the same switch lost)
#endif
// SP: the split-walk variant (pk_walk: PacketArgs::splitCtl != nullptr selects it at launch).  The variant without it is the kernel of round 3 / 4 instruction for
// instruction: the one-body kernel sits at the edge of its register budget (80 vector registers for six waves per SIMD, ~80 spilled scalars), and code that merely
// EXISTS beside the walk moved spills into its loops (measured: a frame of twice the length with the switch off).
template <int M, bool SP>
__global__ __launch_bounds__(256, (M == MODE_SCENE) ? PK_SCENE_WAVES : (PK_SINGLE_WAVES >= 7 ? 7 : ((SP || PK_FORCE6) ? 6 : 1))) __attribute__((amdgpu_num_sgpr(PK_SGPRS))) void k_packet(const float *__restrict__ pblocks, const float *__restrict__ refT,
                                                const float *__restrict__ lrec, const MeshRec *__restrict__ meshes,
                                                const f4 *__restrict__ snodes, const f4 *__restrict__ scull, const ObjRec *__restrict__ objects,
                                                const int *__restrict__ objMesh, SceneView S, PacketArgs A) {
    __shared__ unsigned frames[4 * PK_STACK_WORDS];
    unsigned *const sframesAll = scene_frames<M>();
    unsigned *const parkAll = scene_park<M>();
    stamp_begin(A.stamps);
    PkKernarg KA;
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    unsigned *const stk = &frames[wave * PK_STACK_WORDS];
    int n = A.nDev ? (*A.nDev) * A.nMul : A.n;
    if (A.nCap > 0 && n > A.nCap) n = A.nCap;
    int n2 = A.nDev2 ? (*A.nDev2) * A.nMul2 : 0;   // second segment (PacketArgs::rays2)
    if (n2 > A.nCap2) n2 = A.nCap2;
    const int nPk1 = (n + 63) >> 6, nPk = nPk1 + ((n2 + 63) >> 6);   // (a packet never mixes the two populations)
    // Work distribution.  A quarter (PacketArgs::staticDiv) of the packets are dealt statically and strided -- wave w takes packets w, w + nWaves, .. -- so that
    // every wave sees a fair sample of the image (rays skimming the surface near the horizon cost tens of times the average) while
    // neighbouring waves work on neighbouring packets at the same time (their leaves are in cache); the rest comes from a
    // queue, guided: 1/(2 * waves) of what is left per atomic, at most PacketArgs::grabMax packets, at least one.  (Measured against the
    // alternatives on the 16-sub-ray frame: one ticket per packet from eight sharded queue words -- equal for primary rays, 40 %
    // slower for shadow and reflection packets; runs of 8 packets scattered over the image -- 20 % slower, the cache locality
    // between neighbouring waves is worth more than the balance.)  The grid is sized to be resident at once (packet_blocks_per_cu).
    const int nWaves = (int)gridDim.x * 4, waveId = (int)blockIdx.x * 4 + wave;
    // (a launch of few packets per wave -- a tile shard of a frame, a later generation -- still hands every wave its FIRST packet without
    // an atomic: 6144 waves asking one queue word at once are served at ~12 ns each, the last of them 70 us into a 350 us launch)
    int staticPer = A.staticDiv > 0 ? nPk / (nWaves * A.staticDiv) : 0;
    if (staticPer == 0) staticPer = 1;
    const int qBase = min(nWaves * staticPer, nPk);
    // The dynamic packets are handed out by PK_QUEUES counters, INTERLEAVED: head h owns the packets qBase + PK_QUEUES * t + h
    // (t = 0, 1, ..), a wave draws from head waveId % PK_QUEUES and, when that has run out, from the next ones.  All heads advance
    // together, so the chip still works through ONE narrow window of neighbouring packets (contiguous shares per head -- eight windows
    // -- were 5-40 % slower), but a head is asked by an eighth of the waves: after the static packets of a small launch 6144 waves no
    // longer queue at one word for ~12 ns each (an eighth tile shard of C5: 0.96-1.22 -> 0.81-0.97 ms per frame, two in flight; whole
    // frames unchanged).  Every head sits on a 256-byte line of its own (PACKET_HEAD_STRIDE): atomics on ONE line serialise whatever
    // the word -- with the heads in adjacent words the same scheme was 4-19 % SLOWER than one head.  Only the value an atomic RETURNS
    // says that a head has run out (a load may be served by this XCD's L2, which other XCDs' atomics do not update: a version that
    // looked before asking kept retrying heads long empty).  Measurements: profiles/r03/packet_queue_heads.txt.
    const int dynTotal = nPk - qBase;
    unsigned dead = 0u;   // heads this wave knows to have run out
    for (int h = 0; h < PK_QUEUES; h++) if (dynTotal - h <= 0) dead |= 1u << h;
    int cur = waveId % PK_QUEUES, seen = 0, curR = cur;   // the head this wave draws from, the last ticket it saw there; curR: head of the range in hand
    int sNext = 0, dNext = 0, dEnd = 0;   // dNext .. dEnd: tickets of head curR
    // Split walks (pk_walk): the launch's items come first -- they are what a long packet's critical path is made of -- and a wave leaves only when it
    // has seen the item queue empty AFTER its own last walk (whatever a walk hands over late is taken by someone who is still there, at the latest by
    // the giver itself).  Nobody waits: taking is a compare-and-swap on the count of items taken.
    constexpr bool splitOn = SP && M != MODE_SCENE;
    unsigned *const sst = stk + PK_SPLIT_AT;
    if (splitOn && lane == 0) sst[2] = 0u;   // (no budget: the walks of this wave are not split until a packet says so)
    bool mainDone = false;
    for (;;) {
        int pk = -1, item = -1;
        bool contended = false;   // an item was there but another wave took it
        if (splitOn) {
            KA.fresh();
            const unsigned xcc = xcc_id(), NIx = (unsigned)KA.args()->splitNI / 8u;
            // this XCD's queue.  Control words live on the memory side: atomics execute there, and these loads (agent scope: past the L2, which
            // keeps whatever it read before) look there; one lane asks -- 64 lanes asking one address are served one after the other.  ONE attempt:
            // hundreds of waves retrying at once are hundreds of atomics on one address, ~12 ns each, for every item (measured: a frame of six times the length).
            unsigned *const ctl = KA.args()->splitCtl + 32u * xcc;
            unsigned long long th = 0ull;
            if (lane == 0) th = __hip_atomic_load(reinterpret_cast<unsigned long long *>(ctl), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned t = (unsigned)rfl((int)(unsigned)th), h = (unsigned)rfl((int)(unsigned)(th >> 32));
            if (h < (t < NIx ? t : NIx)) {
                unsigned old = 0u;
                if (lane == 0) old = atomicCAS(ctl + 1, h, h + 1u);
                if ((unsigned)rfl((int)old) == h) item = (int)(xcc * NIx + h);
                else contended = true;
            }
        }
        if (item < 0) {
            if (mainDone) {   // on the way out: only a queue SEEN empty lets a wave leave
                if (!contended) break;
                __builtin_amdgcn_s_sleep(8);
                continue;
            }
            if (sNext < staticPer) {
                pk = sNext * nWaves + waveId; sNext++;
                if (pk >= nPk) {   // (only where the static share is the one packet per wave above)
                    if (qBase >= nPk) { if constexpr (!splitOn) break; else mainDone = true; }
                    continue;
                }
            }
            else {
                if (qBase >= nPk) { if constexpr (!splitOn) break; else { mainDone = true; continue; } }
                if (dNext >= dEnd) {
                    bool got = false;
                    while (dead != (1u << PK_QUEUES) - 1u) {   // at most PK_QUEUES failed requests per wave and launch
                        if ((dead >> cur) & 1u) {
                            const unsigned alive = ~dead & ((1u << PK_QUEUES) - 1u);
                            const unsigned rot = ((alive >> cur) | (alive << (PK_QUEUES - cur))) & ((1u << PK_QUEUES) - 1u);
                            cur = (cur + (int)__builtin_ctz(rot)) % PK_QUEUES;
                            seen = 0;
                        }
                        const int len = (dynTotal - cur + PK_QUEUES - 1) / PK_QUEUES;   // tickets of this head
                        int want = (len - seen) / (nWaves / PK_QUEUES * 2 + 1);   // guided: a share of what is left of this head's tickets
                        KA.fresh();
                        const int grabMax = KA.args()->grabMax;
                        want = want < 1 ? 1 : (want > grabMax ? grabMax : want);
                        unsigned g = 0;
                        if (lane == 0) g = atomicAdd(KA.args()->queue + cur * PACKET_HEAD_STRIDE, (unsigned)want);
                        const int head = rfl((int)g);
                        if (head >= 0 && head < len) { dNext = head; dEnd = min(head + want, len); seen = dEnd; curR = cur; got = true; break; }
                        dead |= 1u << cur;
                    }
                    if (!got) { if constexpr (!splitOn) break; else { mainDone = true; continue; } }
                }
                pk = qBase + PK_QUEUES * dNext + curR;
                dNext++;
            }
        }
        // an item: wait for its giver to have finished writing it (it is between reserving the index and this store, and waits for nobody), then its header
        unsigned *itemW = nullptr;
        if (item >= 0) {
            KA.fresh();
            itemW = KA.args()->splitItems + (size_t)item * SPLIT_ITEM_WORDS;
            const unsigned serial = KA.args()->splitSerial;
            for (;;) {
                unsigned rdy = 0u;
                if (lane == 0) rdy = ld_ag(itemW);
                if ((unsigned)rfl((int)rdy) == serial) break;
                __builtin_amdgcn_s_sleep(2);
            }
            loads_after();
            pk = rfl((int)ld_ag(itemW + 1));
            if (lane == 0) atomicAdd(&g_splitStats[1], 1ull);
        }
        if (splitOn && lane == 0) {
            sst[0] = item >= 0 ? ld_ag(itemW + 2) : 0xffffffffu;
            unsigned budget = (unsigned)KA.args()->splitBudget, pred = 0u, every = 0u;
            if (item >= 0) { budget = ld_ag(itemW + 7); every = ld_ag(itemW + 8); }
            else if (KA.args()->splitCost) { pred = KA.args()->splitCost[pk]; if (pred > (unsigned)KA.args()->splitLong) every = (unsigned)KA.args()->splitBudgetLong; }   // (the block entries this packet made in the context's last frame)
            sst[1] = (unsigned)item; sst[2] = budget; sst[3] = (unsigned)wall_clock64(); sst[4] = (unsigned)pk; sst[5] = pred; sst[6] = 0u; sst[7] = every;
        }
        if (splitOn) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#ifdef XRT_PK_PRIO
        __builtin_amdgcn_s_setprio(0);
        if (lane == 0) stk[PK_SPLIT_AT + 5] = 0u;
#endif
        // ---- the packet's 64 rays ------------------------------------------------------------------------------------
        KA.fresh();
#ifdef XRT_PK_TICKS
        const bool costed = true;
#else
        const bool costed = KA.args()->tileCost != nullptr || (splitOn && KA.args()->splitCost != nullptr);
#endif
        const unsigned long long tPacket = costed ? wall_clock64() : 0ull;
        const bool seg2 = pk >= nPk1;   // wave-uniform
        const int w = (seg2 ? pk - nPk1 : pk) * 64 + lane;
        const bool valid = w < (seg2 ? n2 : n);
        const xrt_ray *const raysS = seg2 ? KA.args()->rays2 : KA.args()->rays;
        Lane L;
        L.state = ST_FINISH; L.mfound = 0; L.cost = 0; L.rayIndex = 0; L.mesh = 0; L.weird = 0; L.dmask = 0; L.ignoreId = -1; L.mask = 0;
        L.r = make_ray(mk(0, 0, 0), mk(1, 1, 1));
        int idx = 0;
        v3 o = mk(0, 0, 0), d = mk(0, 0, 0);
        int im = DEAD_RAY, it = -1;
        if (valid) {
            const int *const indexL = KA.args()->index;
            idx = (indexL && !seg2) ? indexL[w] : w;
            load_ray(raysS + idx, o, d, im, it);
            if (KA.args()->unmark && heavy_marked(it)) it ^= HEAVY_BIT;
        }
        if constexpr (M != MODE_SCENE) {
            SceneLane C;
            C.sfound = 0; C.obj = 0;
            if (valid && im != DEAD_RAY) lane_begin(L, C, S, o, d, im, it, idx, M, A.meshId, false);
            const RayCull RC = make_ray_cull(L.r.o, L.r.d);   // tight leaf boxes (xrt_core.h): what depends on the ray alone
            const int mesh = (M == MODE_MESH) ? A.meshId : 0;
            const MeshRec &mr = meshes[mesh];
            const bool fastL = L.r.par == 0 && L.weird == 0;
            // the lanes inside the root box of an interior root (MO:265; lane_begin left them in ST_NODE with mask 1 -- a root that is a
            // leaf is k_intersect's business: packet_supported)
            // (S.nodeCull: 1 = packets of rays that leave a surface -- the shadow rays and reflections of a frame --, 2 = every packet)
            const bool nodeCull = S.nodeCull == 2 || (S.nodeCull == 1 && __any(valid && L.ignoreId >= 0));
            const bool meshAway = nodeCull && all_back_facing(f4{mr.nbMin[0], mr.nbMin[1], mr.nbMin[2], mr.nbMin[3]}, f4{mr.nbMax[0], mr.nbMax[1], mr.nbMax[2], mr.nbMax[3]}, L.r.d);
            unsigned long long lanes0 = mr.rootBlock < 0 ? 0ull : __ballot(valid && L.state == ST_NODE && L.mask != 0 && !meshAway);
            if (splitOn) {
                if (item >= 0) {   // level 0 of the stack := the item; the lanes' answers so far := what its giver had (a pruning hint, and a candidate like any other)
                    const unsigned *const P = itemW + SPLIT_HEAD_WORDS + 64 + lane;
                    L.mfound = (int)ld_ag(P); L.mKey = i2f((int)ld_ag(P + 64)); L.mDist = i2f((int)ld_ag(P + 128)); L.mU = i2f((int)ld_ag(P + 192)); L.mV = i2f((int)ld_ag(P + 256));
                    L.mRef = (int)ld_ag(P + 320); L.mLeaf = (int)ld_ag(P + 384);
                    reinterpret_cast<unsigned char *>(stk + PK_LEVELS * PK_FRAME_WORDS)[lane] = (unsigned char)ld_ag(itemW + SPLIT_HEAD_WORDS + lane);
                    const unsigned hw = lane < 4 ? ld_ag(itemW + 3 + lane) : 0u;   // block, pending children, lanes (2)
                    if (lane < 4) stk[lane] = hw;
                    lanes0 = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hw, 2) | ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hw, 3) << 32);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
            if constexpr (PK_BUNDLE) {
                if (A.bundle) pk_bundle(stk + PK_BUNDLE_AT, lane, ((lanes0 >> lane) & 1ull) != 0ull, fastL && RC.d2 > 0.0f, L.r);   // (wave-uniform switch)
                else if (lane == 0) stk[PK_BUNDLE_AT + 20] = 0u;
            }
            int pf = 0;
            pk_walk<splitOn>(pblocks, refT, lrec, S, A.cullMin, stk, lane, L, RC, fastL, mr.rootBlock, lanes0, nodeCull, splitOn && item >= 0, PK_PREFETCH && A.prefetch != 0, pf);
            if constexpr (PK_PREFETCH) asm volatile("" :: "v"(pf));   // (the last prefetch is retired here)
            L.mesh = mesh;
            KA.fresh();
            bool mine = true;   // this wave writes the packet's results
            if (splitOn) {
                const int rec = rfl((int)sst[0]);
                if (rec >= 0) {   // the packet was split: leave this participant's answers, and if it is the last one merge them all
                    const int itemA = rfl((int)sst[1]);
                    unsigned *const R = KA.args()->splitRecs + (size_t)rec * SPLIT_REC_WORDS;
                    unsigned *const P = (itemA >= 0 ? KA.args()->splitItems + (size_t)itemA * SPLIT_ITEM_WORDS + SPLIT_HEAD_WORDS + 64 : R + SPLIT_REC_HEAD) + lane;
                    st_ag(P, (unsigned)L.mfound); st_ag(P + 64, (unsigned)f2i(L.mKey)); st_ag(P + 128, (unsigned)f2i(L.mDist)); st_ag(P + 192, (unsigned)f2i(L.mU)); st_ag(P + 256, (unsigned)f2i(L.mV));
                    st_ag(P + 320, (unsigned)L.mRef); st_ag(P + 384, (unsigned)L.mLeaf);
                    stores_done();
                    unsigned left = 0u;
                    if (lane == 0) left = atomicSub(R, 1u);
                    mine = rfl((int)left) == 1;
                    if (mine) {
                        loads_after();
                        if (itemA >= 0 && lane == 0) atomicAdd(&g_splitStats[3], 1ull);
                        const SceneView Sr = KA.result_view();
                        int nIt = rfl((int)__hip_atomic_load(R + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        if (nIt > SPLIT_REC_ITEMS) nIt = SPLIT_REC_ITEMS;
                        for (int j = -1; j < nIt; j++) {   // (-1: the packet's own partial answers)
                            const unsigned *Q = R + SPLIT_REC_HEAD + lane;
                            if (j >= 0) {
                                const unsigned ix = (unsigned)rfl((int)ld_ag(R + 32 + j));
                                if (ix == 0xffffffffu) continue;
                                Q = KA.args()->splitItems + (size_t)ix * SPLIT_ITEM_WORDS + SPLIT_HEAD_WORDS + 64 + lane;
                            }
                            const int qf = (int)ld_ag(Q), qRef = (int)ld_ag(Q + 320), qLeaf = (int)ld_ag(Q + 384);
                            const float qKey = i2f((int)ld_ag(Q + 64)), qD = i2f((int)ld_ag(Q + 128)), qU = i2f((int)ld_ag(Q + 192)), qV = i2f((int)ld_ag(Q + 256));
                            // the rule of pk_candidate / leaf_candidate between two candidates of different participants: key, distance, then the leaves' DFS numbers
                            // (a leaf belongs to one participant; the same candidate may arrive twice: a taker starts from its giver's)
                            if (qf != 0) {   // (the words of a lane without a candidate mean nothing)
                                bool better = (L.mfound == 0) | (qKey < L.mKey) | ((qKey == L.mKey) & (qD < L.mDist));
                                if ((L.mfound != 0) & (qKey == L.mKey) & (qD == L.mDist))
                                    better = qLeaf != L.mLeaf ? node_dfs(Sr, qLeaf) < node_dfs(Sr, L.mLeaf) : f2i(Sr.refN[qRef].w) < f2i(Sr.refN[L.mRef].w);
                                if (better) { L.mfound = 1; L.mKey = qKey; L.mDist = qD; L.mU = qU; L.mV = qV; L.mRef = qRef; L.mLeaf = qLeaf; }
                            }
                        }
                    }
                }
            }
            if (valid && mine) {
                const SceneView Sr = KA.result_view();
                const HitOut h = lane_result(L, C, Sr, M);
                xrt_hit *const hitsS = seg2 ? KA.args()->hits2 : KA.args()->hits;
                int *const flagsS = seg2 ? KA.args()->flags2 : KA.args()->flags;
                const int *const scat = seg2 ? KA.args()->scatter2 : KA.args()->scatter;
                const int at = scat ? scat[idx] : idx;   // (a compact list of rays whose answers belong elsewhere: ShadeArgs::ae)
                if (flagsS) flagsS[at] = h.hit;
                if (!flagsS || h.hit) store_hit(hitsS + at, h);
            }
        } else {
            // ---- OSM:312-455, wave-uniform: scene octree in DFS order, bodies and meshes in list order ---------------------------
            PkScene C(parkAll + wave * PK_PARK_WORDS * 64 + lane);
            int pfScene = 0;
            unsigned *const sfr = sframesAll + wave * PK_SLEVELS * PK_SFRAME_WORDS;
            // start of the query (traverse.h lane_begin): ignoreTriangle identity (MO:290), non-finite rays take the literal box
            // tests, a NaN component means "no intersection" at once (no triangle can be accepted)
            L.weird = (is_finite(o.x) && is_finite(o.y) && is_finite(o.z) && is_finite(d.x) && is_finite(d.y) && is_finite(d.z)) ? 0 : 1;
            if (it >= 0 && im >= 0 && im < S.nMeshes && it < meshes[im].ntri) L.ignoreId = meshes[im].triBase + it;
            const bool nan = is_nan(o.x) || is_nan(o.y) || is_nan(o.z) || is_nan(d.x) || is_nan(d.y) || is_nan(d.z);
            C.w = make_ray(o, d);
            C.sfound = 0;
            const bool nodeCull = S.nodeCull == 2 || (S.nodeCull == 1 && __any(valid && L.ignoreId >= 0));
            int sblk = 0, smask = 1, ssp = 0;   // root = slot 0 of block 0
            unsigned long long slanes = __ballot(valid && im != DEAD_RAY && !nan);
            while (slanes != 0ull) {
                if (smask == 0) {
                    if (ssp == 0) break;
                    ssp--;
                    const unsigned *f = sfr + ssp * PK_SFRAME_WORDS;
                    sblk = rfl((int)f[0]); smask = rfl((int)f[1]);
                    slanes = (unsigned long long)(unsigned)rfl((int)f[2]) | ((unsigned long long)(unsigned)rfl((int)f[3]) << 32);
                    continue;
                }
                const int c = (int)__builtin_ctz((unsigned)smask);
                smask &= smask - 1;
                const int node = sblk * 8 + c;
                const f4 lo = snodes[2 * (size_t)node], hi = snodes[2 * (size_t)node + 1];
                const bool in = ((slanes >> lane) & 1ull) != 0;
                float key = 0.0f;
                bool hit = false;
                if (in) { const RayPre wr = C.w; hit = slab(wr, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, key); }   // OSM:460
                const unsigned long long hm = __ballot(hit);
                if (hm == 0ull) continue;
                const int a_ = rfl(f2i(lo.w)), b_ = rfl(f2i(hi.w));
                if (b_ >= 0) {   // interior
                    if (ssp >= PK_SLEVELS - 1) continue;   // (cannot happen: packet_supported checks the depth)
                    if (smask) {
                        if (lane == 0) {
                            unsigned *f = sfr + ssp * PK_SFRAME_WORDS;
                            f[0] = (unsigned)sblk; f[1] = (unsigned)smask; f[2] = (unsigned)slanes; f[3] = (unsigned)(slanes >> 32);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        ssp++;
                    }
                    sblk = a_ >> 3; smask = 0xff; slanes = hm;
                    continue;
                }
                const int cnt = b_ & 0x0fffffff;
                if (cnt == 0) continue;
                const bool go = hit && !((int)C.sfound && key > (float)C.sbKey);   // a later bucket than the one that already has a hit (OSM:334)
                if (!__any(go)) continue;
                if (go) C.sKey = key;
                // OSM:341-364: the bodies of the leaf, in list order, 32 at a time.  First every lane of the leaf decides for each body
                // whether the world-space pre-cull excludes it: one 48-byte record per body through the scalar cache
                // (SceneView::scull, in leaf order: no srefs -> objects chain), the next one requested before this one is used, the
                // lane's world ray in registers for the whole loop; then the bodies some lane kept are visited.
                for (int base = 0; base < cnt; base += 32) {
                    const int nb = min(32, cnt - base);
                    const f4 *const rec = scull + 4 * (size_t)(a_ + base);
                    unsigned pass = 0u;   // bit j: body base + j is not excluded for this lane
                    if (go) {
                        const RayPre wr = C.w;
                        const bool weird = L.weird != 0;
                        for (int j = 0; j < nb; j++) {
                            const f4 c0 = rec[4 * j], c1 = rec[4 * j + 1], c2 = rec[4 * j + 2];
                            const bool p = (f2i(c2.y) == 0) | weird | precull_box(wr, c0, c1, c2.x);
                            pass |= (p ? 1u : 0u) << j;
                        }
                    }
                    unsigned any = (unsigned)wave_or((int)pass);
                    while (any != 0u) {
                        const int j = (int)__builtin_ctz(any);
                        any &= any - 1u;
                        const bool inB = ((pass >> j) & 1u) != 0u;
                        const f4 h2 = rec[4 * j + 2], h3 = rec[4 * j + 3];
                        const int o = rfl(f2i(h2.z)), m0 = rfl(f2i(h2.w)), m1 = m0 + rfl(f2i(h3.x));
                        if (inB) {   // world -> object space
                            const f4 *const oq = reinterpret_cast<const f4 *>(objects + o);   // InverseWorld, four wide scalar loads
                            const f4 q0 = oq[0], q1 = oq[1], q2 = oq[2], q3 = oq[3];
                            const float iw[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
                            const RayPre wr = C.w;
                            const v3 rayDirPosition = add(wr.o, wr.d);                  // OSM:358
                            const v3 v1 = transform(wr.o, iw);                          // OSM:360
                            const v3 v2 = transform(rayDirPosition, iw);                // OSM:361
                            const v3 dir = normalize(sub(v2, v1));                      // OSM:362-364
                            L.r = make_ray(v1, dir);
                            L.dmask = dir_mask(dir);
                            if (!(is_finite(v1.x) && is_finite(v1.y) && is_finite(v1.z) && is_finite(dir.x) && is_finite(dir.y) && is_finite(dir.z))) L.weird = 1;
                        }
                        const RayCull RC = make_ray_cull(L.r.o, L.r.d);
                        const bool fastL = L.r.par == 0 && L.weird == 0;
                        for (int mi = m0; mi < m1; mi++) {   // OSM:366-368: its meshes
                            const int m = rfl(objMesh[mi]);
                            const MeshRec &mr = meshes[m];
                            float k;
                            const bool inM = inB && slab(L.r, mr.bmin[0], mr.bmin[1], mr.bmin[2], mr.bmax[0], mr.bmax[1], mr.bmax[2], k);   // MESH:34-39
                            if (!__any(inM)) continue;
                            float rkey = 0.0f;
                            const bool inRoot = inM && slab(L.r, mr.rmin[0], mr.rmin[1], mr.rmin[2], mr.rmax[0], mr.rmax[1], mr.rmax[2], rkey);   // MO:265 on the root (MO:331)
                            L.mfound = 0;
                            const int rootBlock = rfl(mr.rootBlock);
                            if (rootBlock < 0) {   // the root is a leaf: one bucket
                                const int r0 = rfl(mr.rootRef), rc = rfl(mr.rootCount);
                                if (rc > 0 && __any(inRoot)) {
                                    if (inRoot) {
                                        L.leafKey = rkey; L.leafNode = ROOT_NODE;
                                        pk_scan_leaf(refT, r0, r0 + rc, L, S, true, [&]() { return rkey; });
                                    }
                                }
                            } else {
                                const bool meshAway = nodeCull && all_back_facing(f4{mr.nbMin[0], mr.nbMin[1], mr.nbMin[2], mr.nbMin[3]}, f4{mr.nbMax[0], mr.nbMax[1], mr.nbMax[2], mr.nbMax[3]}, L.r.d);
                                const unsigned long long lanes0 = __ballot(inRoot && !meshAway);
                                if (lanes0 != 0ull)
                                    pk_walk<false, false>(pblocks, refT, lrec, S, A.cullMin, stk, lane, L, RC, fastL, rootBlock, lanes0, nodeCull, false, false, pfScene);   // (no prefetches in scene mode: measured +-0 on C3 / C4 and their shards, and they cost the kernel 28 bytes of scratch per lane)
                            }
                            if (L.mfound) {   // OSM:370-378
                                L.mesh = m; C.obj = o;
                                merge_mesh_result(L, C);
                                L.mfound = 0;
                            }
                        }
                    }
                }
            }
            KA.fresh();
            if (valid) {
                L.mfound = 0;
                const SceneView Sr = KA.result_view();
                const HitOut h = lane_result(L, C, Sr, M);
                xrt_hit *const hitsS = seg2 ? KA.args()->hits2 : KA.args()->hits;
                int *const flagsS = seg2 ? KA.args()->flags2 : KA.args()->flags;
                const int *const scat = seg2 ? KA.args()->scatter2 : KA.args()->scatter;
                const int at = scat ? scat[idx] : idx;   // (a compact list of rays whose answers belong elsewhere: ShadeArgs::ae)
                if (flagsS) flagsS[at] = h.hit;
                if (!flagsS || h.hit) store_hit(hitsS + at, h);
            }
        }
        if (costed) {   // the tile this packet's first ray belongs to pays for the packet (scheduling feedback for the next frame's tile table)
            const unsigned dt = (unsigned)(wall_clock64() - tPacket);
#ifdef XRT_PK_TICKS
            if (lane == 0) atomicAdd(&g_pkTicks[dt ? 31 - __builtin_clz(dt) : 0], 1ull);
            if (lane == 0 && pk < 65536 && nPk > 4096 && !seg2 && g_pkDump[2 * pk] == 0u) { g_pkDump[2 * pk] = dt; g_pkDump[2 * pk + 1] = stk[PK_SPLIT_AT + 6]; }   // (the first big launch since the last dump)
            if (lane == 0) stk[PK_SPLIT_AT + 6] = 0u;
            {
                const unsigned long long nv = __popcll(__ballot(valid));
                const unsigned long long nSlow = __popcll(__ballot(valid && (L.r.par != 0 || L.weird != 0)));   // lanes that take the literal box test
                if (lane == 0 && nSlow) { atomicAdd(&g_pkWorst[11], 1ull); atomicAdd(&g_pkWorst[12], (unsigned long long)dt); }   // packets with such lanes, their ticks
                if (lane == 0) {
#ifdef XRT_PK_COUNTERS
                    unsigned *cur = g_pkCur + 8 * ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 65535);
#else
                    unsigned cur[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
                    if ((unsigned long long)dt > atomicMax(&g_pkWorst[0], (unsigned long long)dt)) {
                        g_pkWorst[1] = (unsigned long long)pk; g_pkWorst[2] = seg2 ? 1ull : 0ull; g_pkWorst[3] = cur[0]; g_pkWorst[4] = cur[1]; g_pkWorst[5] = cur[2];
                        g_pkWorst[6] = cur[3]; g_pkWorst[7] = cur[4]; g_pkWorst[8] = nv; g_pkWorst[9] = (unsigned long long)nPk; g_pkWorst[10] = nSlow;
                    }
                    for (int i = 0; i < 8; i++) cur[i] = 0;
                }
            }
            if (lane == 0 && KA.args()->tileCost) {
#else
            if constexpr (splitOn) {   // what the packet cost, for the same packet of the context's next frame (kernels.h PacketArgs::splitCost): items do not write, a split packet stays long
                if (splitOn && lane == 0 && KA.args()->splitCost && (int)sst[1] < 0) {
                    const unsigned pred = sst[5], mine = sst[6];
                    KA.args()->splitCost[pk] = ((int)sst[0] >= 0 && pred > mine) ? pred : mine;
                }
            }
            if (lane == 0 && KA.args()->tileCost) {
#endif
                const PkArgsK a = KA.args();
                const int first = (seg2 ? pk - nPk1 : pk) * 64;
                const SlotRec *const sl = seg2 ? a->slotOf2 : a->slotOf1;
                const int *const po = a->pathOf1;
                const int *const scat = seg2 ? a->scatter2 : a->scatter;
                const int firstAt = (sl && scat) ? scat[first] : first;   // (shadow rays in a compact list: where the first one's answer goes = slot * lights + light)
                const int path = sl ? sl[firstAt / (seg2 ? a->nL2 : a->nL1)].path : (po ? po[first] : first);
                atomicAdd(a->tileCost + ((a->tileBase + path) >> a->tileShift), dt);
            }
        }
    }
    KA.fresh();
    stamp_end(KA.args()->stamps);
}

int packet_split_stats(unsigned long long out[4], bool reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_splitStats), 4 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_splitStats), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}

bool packet_supported(int mode, int meshDepth, int sceneDepth) {
    if (mode == MODE_SCENE) return meshDepth + 1 < PK_LEVELS && sceneDepth + 1 < PK_SLEVELS;   // (meshes whose root is a leaf are handled here)
    return (mode == MODE_SINGLE || mode == MODE_MESH) && meshDepth > 0 && meshDepth + 1 < PK_LEVELS;
}

int packet_blocks_per_cu(int mode) {
    int nb = 0;
    hipError_t e = mode == MODE_MESH ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_packet<MODE_MESH, false>, 256, 0)
                   : (mode == MODE_SCENE ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_packet<MODE_SCENE, false>, 256, 0)
                                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_packet<MODE_SINGLE, false>, 256, 0));
    if (e != hipSuccess || nb < 1) nb = 1;
    // 800 SGPRs per SIMD, allocated in sixteens plus sixteen per wave: the API does not account for it in the 81-112 range
    const int bySgpr = 800 / (((PK_SGPRS + 15) / 16) * 16 + 16);
    if (nb > bySgpr) nb = bySgpr;
    return nb > 8 ? 8 : nb;
}

#ifdef XRT_PK_TICKS
extern "C" int xrt_debug_packet_worst(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_pkWorst), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_pkWorst), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
extern "C" int xrt_debug_packet_dump(unsigned *out, int n) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pkDump), (size_t)n * sizeof(unsigned)) != hipSuccess) return -1;
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_pkDump)) != hipSuccess || hipMemset(p, 0, sizeof(g_pkDump)) != hipSuccess) return -1;
    return 0;
}
extern "C" int xrt_debug_packet_ticks(unsigned long long *out32, int reset) {
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_pkTicks), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_pkTicks), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
#ifdef XRT_PK_COUNTERS
extern "C" int xrt_debug_packet_counters(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_pkCounters), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_pkCounters), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif

void launch_packet(const SceneView &S, const PacketArgs &A, int gridBlocks, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    dim3 g((unsigned)gridBlocks), b(256);
    const bool sp = A.splitCtl != nullptr;   // (one-body scenes only: xrt_api.cpp split_arena)
    if (A.mode == MODE_MESH) {
        if (sp) hipExtLaunchKernelGGL((k_packet<MODE_MESH, true>), g, b, 0, st, e0, e1, 0, S.pblocks, S.refT, S.lrec, S.meshes, S.snodes, S.scull, S.objects, S.objMesh, S, A);
        else hipExtLaunchKernelGGL((k_packet<MODE_MESH, false>), g, b, 0, st, e0, e1, 0, S.pblocks, S.refT, S.lrec, S.meshes, S.snodes, S.scull, S.objects, S.objMesh, S, A);
    } else if (A.mode == MODE_SCENE)
        hipExtLaunchKernelGGL((k_packet<MODE_SCENE, false>), g, b, 0, st, e0, e1, 0, S.pblocks, S.refT, S.lrec, S.meshes, S.snodes, S.scull, S.objects, S.objMesh, S, A);
    else {
        if (sp) hipExtLaunchKernelGGL((k_packet<MODE_SINGLE, true>), g, b, 0, st, e0, e1, 0, S.pblocks, S.refT, S.lrec, S.meshes, S.snodes, S.scull, S.objects, S.objMesh, S, A);
        else hipExtLaunchKernelGGL((k_packet<MODE_SINGLE, false>), g, b, 0, st, e0, e1, 0, S.pblocks, S.refT, S.lrec, S.meshes, S.snodes, S.scull, S.objects, S.objMesh, S, A);
    }
}

}  // namespace xrt
