// packet.hip — k_packet: wave-packet form of MeshOctree.GetRayIntersection (MO:259-353) for COHERENT ray populations
// (primary rays of neighbouring pixels / sub-samples, shadow rays towards one light): gfx950, wave64.
//
// k_intersect gives every lane its own walk: its own stack, block descriptor, pending masks, parent box.  For rays that
// go the same way that is ~50 registers of state per lane repeating what the neighbour holds, lanes of one wave wait for
// each other in different phases (utilisation 0.4-0.6), and four waves per SIMD are all the registers allow.  Here ONE
// wavefront walks the octree once for its 64 rays:
//   * wave-uniform state (SGPRs / one LDS frame per level): current block, its descriptor, the parent box, the lanes that
//     are inside it, the position in the child order;
//   * per lane only the ray, its best answer so far and one byte per level (which children of that level's block the
//     ray's own box tests accepted);
//   * a child is visited when ANY lane of the current set passed its box test (MO:331 -- hit8_*_children evaluates the
//     reference's test for all eight children of a block); lanes that did not are masked off for that subtree, as are
//     lanes the bucket rule lets prune it (key above the lane's best key on a `safe` interior node or a leaf);
//   * in a leaf every remaining lane tests every triangle (RE:42-75): the triangle is the same for all lanes, so its
//     16 + 36 bytes arrive through the scalar cache (s_load) and the vector units only do arithmetic.
// Each lane therefore performs exactly the (leaf, triangle) tests its own walk in k_intersect could perform, minus ones
// that cannot win, and keeps the lexicographic arg-min (leaf entry key, distance, leaf DFS index, list position) -- the
// order-independent form of MO:281-301: bit-identical answers, in any visiting order.
// Lanes with a parallel axis or a non-finite component take the literal box test (`slab`) inside the same walk.
#include "device_util.h"
#include "kernels.h"

#include <hip/hip_ext.h>

namespace xrt {

constexpr int PK_LEVELS = 24;        // deeper octrees than this fall back to k_intersect (scene_build limits depth to 20)
constexpr int PK_FRAME_WORDS = 12;   // blk, next child position, order mask, lanes (2), parent box min (3), half (3), pad
constexpr int PK_SGPRS = 112;        // SGPR allocation the kernel may reach (checked against the ISA in tests/test_numerics_contract.py)

struct PkUniform {   // wave-uniform cursor
    int blk, p, dm0;
    unsigned long long lanes;
    v3 bmin, half;
};

struct alignas(4) TriWords { float w[16]; };   // a 13-word record of refT and the first three words of the next one
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float rflf(float v) { return i2f(__builtin_amdgcn_readfirstlane(f2i(v))); }

template <int M>
__global__ __launch_bounds__(256) void k_packet(const f4 *__restrict__ blocks, const float *__restrict__ refT,
                                                const f4 *__restrict__ leafNB, const f4 *__restrict__ leafTB, const MeshRec *__restrict__ meshes, SceneView S,
                                                PacketArgs A) {
    __shared__ unsigned frames[4 * PK_LEVELS * PK_FRAME_WORDS];
    stamp_begin(A.stamps);
    const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
    unsigned *const stk = &frames[wave * PK_LEVELS * PK_FRAME_WORDS];
    int n = A.nDev ? (*A.nDev) * A.nMul : A.n;
    if (A.nCap > 0 && n > A.nCap) n = A.nCap;
    const int nPk = (n + 63) >> 6;
    // Work distribution.  A quarter (PacketArgs::staticDiv) of the packets are dealt statically and strided -- wave w takes packets w, w + nWaves, .. -- so that
    // every wave sees a fair sample of the image (rays skimming the surface near the horizon cost tens of times the average) while
    // neighbouring waves work on neighbouring packets at the same time (their leaves are in cache); the rest comes from a
    // queue, guided: 1/(2 * waves) of what is left per atomic, at most PacketArgs::grabMax packets, at least one.  (Measured against the
    // alternatives on the 16-sub-ray frame: one ticket per packet from eight sharded queue words -- equal for primary rays, 40 %
    // slower for shadow and reflection packets; runs of 8 packets scattered over the image -- 20 % slower, the cache locality
    // between neighbouring waves is worth more than the balance.)  The grid is sized to be resident at once (packet_blocks_per_cu).
    const int nWaves = (int)gridDim.x * 4, waveId = (int)blockIdx.x * 4 + wave;
    const int staticPer = A.staticDiv > 0 ? nPk / (nWaves * A.staticDiv) : 0, qBase = nWaves * staticPer;
    int sNext = 0, dNext = 0, dEnd = 0, left = nPk - qBase;
    for (;;) {
        int pk;
        if (sNext < staticPer) { pk = sNext * nWaves + waveId; sNext++; }
        else {
            if (dNext >= dEnd) {
                int want = left / (nWaves * 2);
                want = want < 1 ? 1 : (want > A.grabMax ? A.grabMax : want);
                unsigned g = 0;
                if (lane == 0) g = atomicAdd(A.queue, (unsigned)want);
                dNext = qBase + rfl((int)g);
                if (dNext >= nPk || dNext < qBase) break;
                dEnd = min(dNext + want, nPk);
                left = nPk - dEnd;
            }
            pk = dNext++;
        }
        // ---- the packet's 64 rays ------------------------------------------------------------------------------------
        const int w = pk * 64 + lane;
        const bool valid = w < n;
        Lane L;
        SceneLane C;
        L.state = ST_FINISH; L.mfound = 0; L.cost = 0; L.rayIndex = 0; L.mesh = 0; L.weird = 0; L.dmask = 0; L.ignoreId = -1;
        L.r = make_ray(mk(0, 0, 0), mk(1, 1, 1));
        C.sfound = 0; C.obj = 0;
        int idx = 0;
        if (valid) {
            idx = A.index ? A.index[w] : w;
            v3 o, d; int im, it;
            load_ray(A.rays + idx, o, d, im, it);
            if (A.unmark && heavy_marked(it)) it ^= HEAVY_BIT;
            if (im != DEAD_RAY) lane_begin(L, C, S, o, d, im, it, idx, M, A.meshId, false);
        }
        const RayCull RC = make_ray_cull(L.r.o, L.r.d);   // tight leaf boxes (xrt_core.h): what depends on the ray alone
        const int mesh = (M == MODE_MESH) ? A.meshId : 0;
        const MeshRec &mr = meshes[mesh];
        const bool fastL = L.r.par == 0 && L.weird == 0;
        PkUniform U;
        U.blk = mr.rootBlock;
        U.bmin = mk(mr.rmin[0], mr.rmin[1], mr.rmin[2]);
        U.half = half_of(U.bmin, mk(mr.rmax[0], mr.rmax[1], mr.rmax[2]));
        U.lanes = __ballot(valid && L.state == ST_NODE);   // inside the root box of an interior root (MO:265)
        U.p = 0; U.dm0 = 0;
        // per lane: for every level of the shared stack, which children of that level's block the lane's own box tests accepted
        unsigned long long cbLo = 0, cbMid = 0, cbHi = 0;
        int sp = 0;
        bool entering = true, anyFound = false;   // anyFound: some lane of the wave has a candidate
        int d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        unsigned long long offLo = 0, offHi = 0;
        int cb = 0;
        if (U.blk < 0) U.lanes = 0ull;   // (a root that is a leaf is k_intersect's business: packet_supported)
        // The scalar unit is shared by the CU's four SIMDs, so scalar instructions are the scarce resource of this kernel
        // (measured: 3,900 per packet against 4,000 vector ones made it scalar-bound): the pending children are a bit mask
        // walked with ctz, the lanes of a leaf are selected once for the whole leaf, a triangle costs one wave-level branch.
        while (U.lanes != 0ull) {
            const bool in = ((U.lanes >> lane) & 1ull) != 0;
            if (entering) {
                const f4 lo = blocks[2 * (size_t)U.blk], hi = blocks[2 * (size_t)U.blk + 1];
                d0 = f2i(lo.x); d1 = f2i(lo.y); d2 = f2i(lo.z); d3 = f2i(lo.w);
                offLo = (unsigned long long)(unsigned)f2i(hi.x) | ((unsigned long long)(unsigned)f2i(hi.y) << 32);
                offHi = (unsigned long long)(unsigned)f2i(hi.z) | ((unsigned long long)(unsigned)f2i(hi.w) << 32);
                cb = 0;
                if (in) cb = fastL ? hit8_fast_children(L.r, L.dmask, U.bmin, U.half) : hit8_slow_children(L.r, U.bmin, U.half);
                cb &= 0xff & ~((d2 >> 8) & 0xff);   // empty leaves can never hit (Q4)
                {   // remember it for the return to this level
                    const int sh = (sp & 7) * 8;
                    const unsigned long long m = ~(0xffull << sh), v = (unsigned long long)(unsigned)cb << sh;
                    if (sp < 8) cbLo = (cbLo & m) | v; else if (sp < 16) cbMid = (cbMid & m) | v; else cbHi = (cbHi & m) | v;
                }
                U.dm0 = __builtin_amdgcn_readlane(L.dmask, (int)__builtin_ctzll(U.lanes));   // front-to-back order of the first lane
                // children some lane entered, in that order: bit p <-> child (p ^ dm0)
                int un = wave_or(cb);
                if (U.dm0 & 4) un = ((un & 0xf0) >> 4) | ((un & 0x0f) << 4);
                if (U.dm0 & 2) un = ((un & 0xcc) >> 2) | ((un & 0x33) << 2);
                if (U.dm0 & 1) un = ((un & 0xaa) >> 1) | ((un & 0x55) << 1);
                U.p = un;
                entering = false;
            }
            if (U.p == 0) {   // block exhausted: back to the level above
                if (sp == 0) break;
                sp--;
                const unsigned *f = stk + sp * PK_FRAME_WORDS;
                U.blk = rfl((int)f[0]); U.p = rfl((int)f[1]); U.dm0 = rfl((int)f[2]);
                U.lanes = (unsigned long long)(unsigned)rfl((int)f[3]) | ((unsigned long long)(unsigned)rfl((int)f[4]) << 32);
                U.bmin = mk(rflf(i2f((int)f[5])), rflf(i2f((int)f[6])), rflf(i2f((int)f[7])));
                U.half = mk(rflf(i2f((int)f[8])), rflf(i2f((int)f[9])), rflf(i2f((int)f[10])));
                const f4 lo = blocks[2 * (size_t)U.blk], hi = blocks[2 * (size_t)U.blk + 1];
                d0 = f2i(lo.x); d1 = f2i(lo.y); d2 = f2i(lo.z); d3 = f2i(lo.w);
                offLo = (unsigned long long)(unsigned)f2i(hi.x) | ((unsigned long long)(unsigned)f2i(hi.y) << 32);
                offHi = (unsigned long long)(unsigned)f2i(hi.z) | ((unsigned long long)(unsigned)f2i(hi.w) << 32);
                const int sh = (sp & 7) * 8;
                cb = (int)(((sp < 8 ? cbLo : (sp < 16 ? cbMid : cbHi)) >> sh) & 0xffull);
                continue;
            }
            const int c = (int)__builtin_ctz((unsigned)U.p) ^ U.dm0;
            U.p &= U.p - 1;
            const bool inC = in && ((cb >> c) & 1);
            v3 cmin, cmax;
            child_box(U.bmin, U.half, c, cmin, cmax);
            // The child's own test gives the entry key (its outcome is known: hit).  The bucket rule compares keys only once a lane
            // has a candidate; until some lane of the wave has one (most of a packet's walk) the key of a leaf is computed by the
            // lanes that find a candidate in it, and nobody computes the key of an interior child.
            auto entry_key = [&]() {
                float k = 0.0f;
                if (fastL) (void)slab_fast(L.r, L.dmask, cmin, cmax, k);
                else (void)slab(L.r, cmin.x, cmin.y, cmin.z, cmax.x, cmax.y, cmax.z, k);
                return k;
            };
            const bool keyed = anyFound;   // wave-uniform
            float key = 0.0f;
            if (keyed && inC) key = entry_key();
            const int node = U.blk * 8 + c;
            if (!((d2 >> c) & 1)) {   // ---- leaf (non-empty): MO:288-304 for the lanes the bucket rule lets in ----
                const f4 nlo = leafNB[2 * (size_t)node], nhi = leafNB[2 * (size_t)node + 1];
                bool go = inC && !(L.mfound && key > L.mKey) && !all_back_facing(nlo, nhi, L.r.d);
                if (!__any(go)) continue;
                const int r0 = d1 + child_ref_offset(offLo, offHi, c);
                const int r1 = d1 + ((c == 7) ? d3 : child_ref_offset(offLo, offHi, c + 1));
                if (r1 - r0 >= A.cullMin) {   // lanes whose ray cannot reach any triangle of the leaf (xrt_core.h leaf_certainly_missed) stay out of it
                    const f4 *tb = leafTB + 4 * (size_t)node;
                    go = go && !leaf_certainly_missed(L.r, RC, tb[0], tb[1], tb[2], tb[3]);
                    if (!__any(go)) continue;
                }
                if (go) {   // the lanes of this leaf, selected once for all its triangles
                    L.leafKey = key; L.leafNode = node;
                    // The triangle is the same for every lane: normals and geometry come through the scalar cache.  Two register
                    // sets take turns, and a triangle's 52 bytes are requested before the previous one's arithmetic starts (the
                    // records of a leaf are back to back; the arrays end in two dummy records, so asking one past the leaf is safe).
                    // One record of 13 words (refT: normal, id, v1, E1, E2) = one s_load_dwordx16 (the three words past it belong to
                    // the next record; the array ends in padding): the scalar unit is this kernel's scarce resource, and a
                    // triangle used to cost three loads from two streams.
                    const char *pt = reinterpret_cast<const char *>(refT) + (size_t)r0 * TRI_REC_BYTES;
                    auto test = [&](const TriWords &q, int r) {
                        const bool f = !(facing(mk(q.w[0], q.w[1], q.w[2]), L.r.d) > 0.0f) & (f2i(q.w[3]) != L.ignoreId);   // RE:48-51, MO:290
                        v3 T; float det, row2;
                        const v3 gb = mk(q.w[7], q.w[8], q.w[9]), gc = mk(q.w[10], q.w[11], q.w[12]);
                        const bool sA = tri_stage_a(L.r.o, L.r.d, mk(q.w[4], q.w[5], q.w[6]), gb, gc, T, det, row2) & f;
                        if (sA) {   // one wave-level branch per triangle (s_cbranch_execz): most are rejected by the sign of u for every lane
                            float u, v, t;
                            if (tri_stage_b(L.r.d, gb, gc, T, det, row2, u, v, t)) {
                                if (!keyed) L.leafKey = entry_key();
                                leaf_candidate(L, S, r, -2, true, u, v, t);   // (the ignored triangle was filtered above: -2 matches no id)
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
                if (!keyed) anyFound = __any(L.mfound != 0);
                continue;
            }
            // ---- interior child: a lane prunes it by key only where that is a proven lower bound (safe bit) ----
            const bool go = inC && !(L.mfound && ((d2 >> (16 + c)) & 1) && key > L.mKey);
            const unsigned long long LL = __ballot(go);
            if (LL == 0ull) continue;
            if (sp >= PK_LEVELS - 1) continue;   // (cannot happen: packet_supported checks the depth)
            if (U.p != 0) {   // something is left to do at this level: come back
                if (lane == 0) {
                    unsigned *f = stk + sp * PK_FRAME_WORDS;
                    f[0] = (unsigned)U.blk; f[1] = (unsigned)U.p; f[2] = (unsigned)U.dm0;
                    f[3] = (unsigned)U.lanes; f[4] = (unsigned)(U.lanes >> 32);
                    f[5] = (unsigned)f2i(U.bmin.x); f[6] = (unsigned)f2i(U.bmin.y); f[7] = (unsigned)f2i(U.bmin.z);
                    f[8] = (unsigned)f2i(U.half.x); f[9] = (unsigned)f2i(U.half.y); f[10] = (unsigned)f2i(U.half.z);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                sp++;
            }
            U.blk = d0 + __builtin_popcount((unsigned)(d2 & 0xff) & ((1u << c) - 1u));
            U.bmin = cmin; U.half = half_of(cmin, cmax);
            U.lanes = LL;
            entering = true;
        }
        if (valid) {
            L.mesh = mesh;
            const HitOut h = lane_result(L, C, S, M);
            if (A.flags) A.flags[idx] = h.hit;
            if (!A.flags || h.hit) store_hit(A.hits + idx, h);
        }
    }
    stamp_end(A.stamps);
}

bool packet_supported(int mode, int meshDepth) { return (mode == MODE_SINGLE || mode == MODE_MESH) && meshDepth > 0 && meshDepth + 1 < PK_LEVELS; }

int packet_blocks_per_cu(int mode) {
    int nb = 0;
    hipError_t e = mode == MODE_MESH ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_packet<MODE_MESH>, 256, 0)
                                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_packet<MODE_SINGLE>, 256, 0);
    if (e != hipSuccess || nb < 1) nb = 1;
    // 800 SGPRs per SIMD, allocated in sixteens plus sixteen per wave: the API does not account for it in the 81-112 range
    const int bySgpr = 800 / (((PK_SGPRS + 15) / 16) * 16 + 16);
    if (nb > bySgpr) nb = bySgpr;
    return nb > 8 ? 8 : nb;
}

void launch_packet(const SceneView &S, const PacketArgs &A, int gridBlocks, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
    dim3 g((unsigned)gridBlocks), b(256);
    if (A.mode == MODE_MESH) hipExtLaunchKernelGGL((k_packet<MODE_MESH>), g, b, 0, st, e0, e1, 0, S.blocks, S.refT, S.leafNB, S.leafTB, S.meshes, S, A);
    else hipExtLaunchKernelGGL((k_packet<MODE_SINGLE>), g, b, 0, st, e0, e1, 0, S.blocks, S.refT, S.leafNB, S.leafTB, S.meshes, S, A);
}

}  // namespace xrt
