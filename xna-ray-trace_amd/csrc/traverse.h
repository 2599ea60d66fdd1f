// traverse.h — the per-ray state machine of ISpatialManager.GetRayIntersection (OSM:312-455) →
// MeshOctree.GetRayIntersection (MO:259-326) → IntersectsTriangleBackfaceCulling (RE:42-75).
//
// The reference collects every leaf the ray's box tests reach into buckets sorted by entry distance and
// scans buckets in ascending order until one produced a hit.  That answer equals the lexicographic
// arg-min over all accepted (leaf, triangle) tests of
//        (leaf entry key, distance, leaf DFS index, position in the leaf list)
// (MO:281-301: buckets ascending, leaves in DFS order inside a bucket, strict '<' on distance), which no
// longer depends on the visiting order.  So the mesh octree is walked front to back and every node
// whose key lower bound exceeds the best key so far is skipped; nothing else is approximated.
// Scene level (few nodes) is walked in the reference's DFS order with the streaming form of the same rule.
//
// One `advance_*` call does one unit of work for one lane.  The HIP kernel (kernels.hip) drives 64
// lanes per wavefront with the stack in LDS; tests/emul drives a single lane on the CPU.
#pragma once
#include "xrt_core.h"

namespace xrt {

struct SceneView {
    const f4 *blocks;       // mesh octrees: 2 per block descriptor (implicit boxes, xrt_core.h)
    const int *childDfs;    // 8 per block: DFS pre-order index of child c
    const f4 *leafNB;       // 2 per node: component-wise min / max of the surface normals of the leaf / of every leaf below the interior node
    const f4 *leafTB;       // 4 per node: the leaf's tight box (xrt_core.h leaf_certainly_missed)
    const int *runBase;     // per node: first run record of a leaf of >= LEAF_RUN_MIN references (index into runTB / 4), else -1
    const f4 *runTB;        // 4 per run of LEAF_RUN consecutive references: the run's tight box (same form as leafTB)
    const f4 *refN;         // per leaf reference: (surfaceNormal.xyz, global triangle id)
    const g3 *refG;         // 3 per leaf reference: v1, E1, E2
    const f4 *triTB;        // 4 per leaf reference: the tight-box record of that ONE triangle (same form as leafTB; packet.hip: the bundle prefilter of big leaves)
    const float *refT;      // the same two streams as one record of TRI_REC_WORDS words per reference (k_packet's scalar loads)
    const float *pblocks;   // k_packet: PBLOCK_WORDS per block -- the descriptor and the child planes (below)
    const float *lrec;      // k_packet: LREC_WORDS per node (leafNB | leafTB | first run's word offset), the run records behind them
    const MeshRec *meshes;
    const f4 *snodes;       // scene octree, 2 per record
    const int *srefs;       // object id per scene leaf reference
    const f4 *scull;        // 4 per scene leaf reference: the body's pre-cull record where the packet kernel's scalar loads find it without the
                            // srefs -> objects chain: (cullMin.xyz, k0) (cullMax.xyz, k1) (k2, cullOk, object id, meshStart) (meshCount, -, -, -)
    const ObjRec *objects;
    const int *objMesh;
    int nMeshes, nObjects;
    int sceneDepth, meshDepth;   // stack capacities needed
    int nodeCull;                // interior nodes (and whole meshes) whose triangles all face away from the ray are not entered (all_back_facing on the node's
                                 // normal box): 0 off, 1 for rays that lately met back faces only / packets of rays leaving a surface, 2 always
};

// k_packet's block record (SceneView::pblocks), PBLOCK_WORDS floats per block:
//   [0..7]   the block descriptor (the two f4 of `blocks`)
//   [8..11]  x0, x1, x2, hx0   [12..15] y0, y1, y2, hy0   [16..19] z0, z1, z2, hz0
// the planes of the eight children of the node whose own box is (bmin, half), evaluated on the host with the builder's binary32
// operations (MO:207, 217-218): p0 = bmin + half * 0f, p1 = bmin + half * 1f, p2 = p1 + half (what hit8_* forms: child bit b spans planes
// b .. b+1) and hp0 = p0 + half, the far plane of child bit 0 as child_box forms it (== p1 for finite boxes).  With them a block costs the
// walk no box arithmetic at all: where the per-lane kernel recomputes a child's box from its parent's (ALU instead of memory, one lane
// one walk), the packet's box is wave-uniform, and a wave-uniform float lives in a VECTOR register on gfx950 (no scalar float ALU): the
// implicit boxes cost a packet ~25 vector instructions per block entered and ~15 per child visited, all 64 lanes computing one number.
constexpr int PBLOCK_WORDS = 20;
// k_packet's node record (SceneView::lrec), LREC_WORDS floats per node (= block * 8 + child):
//   [0..7] leafNB (2 f4)   [8..23] leafTB (4 f4)   [24] word offset of the leaf's first run record in lrec, or -1   [25..31] -
// and behind the node records RUN_WORDS floats per run (runTB), then one run of padding.
constexpr int LREC_WORDS = 32, RUN_WORDS = 16;

enum : int { ST_IDLE = 0, ST_SCENE = 1, ST_NODE = 2, ST_LEAF = 3, ST_FINISH = 4 };
// MODE_SINGLE: a scene of one SceneObject with one Mesh (C1, C2, C5): the scene-level walk collapses into
// the query's prologue and its registers disappear from the traversal loop.
enum : int { MODE_SCENE = 0, MODE_MESH = 1, MODE_SINGLE = 2 };

// Scene-level half of a query: what only the scene cursor (advance_scene) and the final answer need.  The hot kernel
// keeps it in LDS in scene mode (kernels.hip, ParkedScene) so that the mesh loops fit the register budget; every
// function that touches it is a template over its type.
struct SceneLane {
    RayPre w;            // world ray
    int sblk, smask, ssp;
    int sRef, sRefEnd;
    float sKey;
    int obj, mPtr, mEnd;
    // scene best
    int sfound;
    float sbKey, sbD, sbU, sbV;
    int sbRef, sbLeaf, sbObj, sbMesh;
};

struct Lane {
    RayPre r;            // object-space ray
    int ignoreId;        // global triangle id, -1 = none
    int weird;           // a non-finite ray component: only the NaN-exact box test may be used
    int rayIndex;
    int state;
    // mesh query: current block (children of the node whose own box is bmin..bmax)
    int mesh, dmask;
    int blk, mask, sp;       // mask: children still to test, bit p <-> child (p ^ dmask), front to back
    unsigned long long path; // child index taken at each level (3 bits per level) to recompute boxes on pop
    v3 bmin, half;           // own box of the current block's parent: min corner and (max - min) / 2 (MO:207)
    int d0, d1, d2, d3;      // block descriptor words 0..3
    unsigned long long offLo, offHi;   // 16-bit reference offsets of children 0..3 / 4..7
    int mfound;
    float mKey, mDist, mU, mV;
    int mRef, mLeaf;
    // leaf scan
    int ref, refEnd, leafNode;
    float leafKey;
    int spec;            // the last leaf step met a triangle facing the ray: fetch geometry together with the normals
    int cost;            // rounds of the kernel's outer loop this ray has been in flight (scheduling feedback, not a result)
    SceneLane sc;        // plain-register home of the scene-level half (unused where the kernel parks it in LDS)
};

XRT_HD int ctz32(unsigned x) { return __builtin_ctz(x); }
XRT_HD int node_dfs(const SceneView &S, int node) { return node < 0 ? 0 : S.childDfs[node]; }

XRT_HD void finish_mesh_query(Lane &L, int mode);
XRT_HD bool all_back_facing(f4 nmin, f4 nmax, v3 d);

// The eight children of block `blk`: descriptor into registers, every non-empty child pending.
XRT_HD void load_block(Lane &L, const SceneView &S, int blk) {
    f4 lo = S.blocks[2 * (size_t)blk], hi = S.blocks[2 * (size_t)blk + 1];
    L.blk = blk;
    L.d0 = f2i(lo.x); L.d1 = f2i(lo.y); L.d2 = f2i(lo.z); L.d3 = f2i(lo.w);
    L.offLo = (unsigned long long)(unsigned)f2i(hi.x) | ((unsigned long long)(unsigned)f2i(hi.y) << 32);
    L.offHi = (unsigned long long)(unsigned)f2i(hi.z) | ((unsigned long long)(unsigned)f2i(hi.w) << 32);
}
// pending-children mask in front-to-back bit order for the children set `childSet` (bit c = child c)
XRT_HD int permute_mask(int childSet, int dmask) {
    int m = 0;
#pragma unroll
    for (int p = 0; p < 8; p++) m |= ((childSet >> (p ^ dmask)) & 1) << p;
    return m;
}
// MO:217-218 (cubePosition = parent.Min + cubeSize * (i,j,k); box = [cubePosition, cubePosition + cubeSize])
XRT_HD v3 half_of(v3 pmin, v3 pmax) {   // cubeSize = (Max - Min) / 2f = (Max - Min) * (1f / 2f)   (MO:207)
    return mk((pmax.x - pmin.x) * 0.5f, (pmax.y - pmin.y) * 0.5f, (pmax.z - pmin.z) * 0.5f);
}
XRT_HD void child_box(v3 pmin, v3 half, int c, v3 &cmin, v3 &cmax) {
    float fi = (c & 4) ? 1.0f : 0.0f, fj = (c & 2) ? 1.0f : 0.0f, fk = (c & 1) ? 1.0f : 0.0f;
    cmin = mk(pmin.x + half.x * fi, pmin.y + half.y * fj, pmin.z + half.z * fk);
    cmax = mk(cmin.x + half.x, cmin.y + half.y, cmin.z + half.z);
}
// The same box test for an octree child when the ray has no parallel axis and no non-finite component
// (`fast`): the swap of BoundingBox.Intersects is decided by the sign of the direction, Math.Max/Min
// reduce to max/min (no NaN can occur), and the per-axis early-outs collapse into the final comparison
// because tmin only grows and tmax only shrinks.  Same key, same hit/miss (DESIGN.md §box test).
XRT_HD bool slab_fast(const RayPre &r, int dmask, v3 cmin, v3 cmax, float &key) {
    float nx = (dmask & 4) ? cmax.x : cmin.x, fx = (dmask & 4) ? cmin.x : cmax.x;
    float ny = (dmask & 2) ? cmax.y : cmin.y, fy = (dmask & 2) ? cmin.y : cmax.y;
    float nz = (dmask & 1) ? cmax.z : cmin.z, fz = (dmask & 1) ? cmin.z : cmax.z;
    float t1x = (nx - r.o.x) * r.inv.x, t2x = (fx - r.o.x) * r.inv.x;
    float t1y = (ny - r.o.y) * r.inv.y, t2y = (fy - r.o.y) * r.inv.y;
    float t1z = (nz - r.o.z) * r.inv.z, t2z = (fz - r.o.z) * r.inv.z;
    float num = fmaxf(fmaxf(fmaxf(t1x, 0.0f), t1y), t1z);
    float num2 = fminf(fminf(fminf(t2x, FLT_MAX), t2y), t2z);
    key = num;
    return !(num > num2);
}
// slab_fast for all eight children of the block whose parent box is (pmin, half): each axis has three planes
// (p0 = pmin + half*0, p1 = pmin + half*1, p2 = p1 + half -- the very sums child_box forms), so nine products serve
// the eight tests.  Returns the children the ray enters, bit p <-> child (p ^ dmask) like Lane::mask.  The visit of a
// child repeats its own test (it needs the entry key anyway), so this only removes visits that would end at MO:331.
XRT_HD int hit8_fast(const RayPre &r, int dmask, v3 pmin, v3 half) {
    const float x0 = pmin.x + half.x * 0.0f, x1 = pmin.x + half.x * 1.0f, x2 = x1 + half.x;
    const float y0 = pmin.y + half.y * 0.0f, y1 = pmin.y + half.y * 1.0f, y2 = y1 + half.y;
    const float z0 = pmin.z + half.z * 0.0f, z1 = pmin.z + half.z * 1.0f, z2 = z1 + half.z;
    const float tx0 = (x0 - r.o.x) * r.inv.x, tx1 = (x1 - r.o.x) * r.inv.x, tx2 = (x2 - r.o.x) * r.inv.x;
    const float ty0 = (y0 - r.o.y) * r.inv.y, ty1 = (y1 - r.o.y) * r.inv.y, ty2 = (y2 - r.o.y) * r.inv.y;
    const float tz0 = (z0 - r.o.z) * r.inv.z, tz1 = (z1 - r.o.z) * r.inv.z, tz2 = (z2 - r.o.z) * r.inv.z;
    // child bit b on an axis spans planes b..b+1; the near plane is the upper one when the direction is negative
    const bool nx = (dmask & 4) != 0, ny = (dmask & 2) != 0, nz = (dmask & 1) != 0;
    const float nearX[2] = {nx ? tx1 : tx0, nx ? tx2 : tx1}, farX[2] = {nx ? tx0 : tx1, nx ? tx1 : tx2};
    const float nearY[2] = {ny ? ty1 : ty0, ny ? ty2 : ty1}, farY[2] = {ny ? ty0 : ty1, ny ? ty1 : ty2};
    const float nearZ[2] = {nz ? tz1 : tz0, nz ? tz2 : tz1}, farZ[2] = {nz ? tz0 : tz1, nz ? tz1 : tz2};
    int m = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int i = (c >> 2) & 1, j = (c >> 1) & 1, k = c & 1;
        const float num = fmaxf(fmaxf(fmaxf(nearX[i], 0.0f), nearY[j]), nearZ[k]);
        const float num2 = fminf(fminf(fminf(farX[i], FLT_MAX), farY[j]), farZ[k]);
        if (!(num > num2)) m |= 1 << (c ^ dmask);
    }
    return m;
}
// hit8_fast with the answer indexed by child: bit c <-> child c (the wave-packet kernel shares one visiting order
// among lanes whose front-to-back orders differ).
XRT_HD int hit8_fast_children(const RayPre &r, int dmask, v3 pmin, v3 half) {
    const float x0 = pmin.x + half.x * 0.0f, x1 = pmin.x + half.x * 1.0f, x2 = x1 + half.x;
    const float y0 = pmin.y + half.y * 0.0f, y1 = pmin.y + half.y * 1.0f, y2 = y1 + half.y;
    const float z0 = pmin.z + half.z * 0.0f, z1 = pmin.z + half.z * 1.0f, z2 = z1 + half.z;
    const float tx0 = (x0 - r.o.x) * r.inv.x, tx1 = (x1 - r.o.x) * r.inv.x, tx2 = (x2 - r.o.x) * r.inv.x;
    const float ty0 = (y0 - r.o.y) * r.inv.y, ty1 = (y1 - r.o.y) * r.inv.y, ty2 = (y2 - r.o.y) * r.inv.y;
    const float tz0 = (z0 - r.o.z) * r.inv.z, tz1 = (z1 - r.o.z) * r.inv.z, tz2 = (z2 - r.o.z) * r.inv.z;
    const bool nx = (dmask & 4) != 0, ny = (dmask & 2) != 0, nz = (dmask & 1) != 0;
    const float nearX[2] = {nx ? tx1 : tx0, nx ? tx2 : tx1}, farX[2] = {nx ? tx0 : tx1, nx ? tx1 : tx2};
    const float nearY[2] = {ny ? ty1 : ty0, ny ? ty2 : ty1}, farY[2] = {ny ? ty0 : ty1, ny ? ty1 : ty2};
    const float nearZ[2] = {nz ? tz1 : tz0, nz ? tz2 : tz1}, farZ[2] = {nz ? tz0 : tz1, nz ? tz1 : tz2};
    int m = 0;
#pragma unroll
    for (int c = 0; c < 8; c++) {
        const int i = (c >> 2) & 1, j = (c >> 1) & 1, k = c & 1;
        const float num = fmaxf(fmaxf(fmaxf(nearX[i], 0.0f), nearY[j]), nearZ[k]);
        const float num2 = fminf(fminf(fminf(farX[i], FLT_MAX), farY[j]), farZ[k]);
        if (!(num > num2)) m |= 1 << c;
    }
    return m;
}
// The same decisions with the literal box test (a ray with a parallel axis or a non-finite component, MO:331).
XRT_HD int hit8_slow_children(const RayPre &r, v3 pmin, v3 half) {
    int m = 0;
    for (int c = 0; c < 8; c++) {
        v3 cmin, cmax;
        child_box(pmin, half, c, cmin, cmax);
        float key;
        if (slab(r, cmin.x, cmin.y, cmin.z, cmax.x, cmax.y, cmax.z, key)) m |= 1 << c;
    }
    return m;
}
XRT_HD bool is_finite(float x) { return fabsf(x) <= FLT_MAX; }
XRT_HD int child_ref_offset(unsigned long long offLo, unsigned long long offHi, int c) {
    unsigned long long w = (c & 4) ? offHi : offLo;
    return (int)((w >> (16 * (c & 3))) & 0xffffull);
}

// enter == false (the wave-packet kernel, packet.hip): stop after the root's own box test -- state ST_NODE with mask != 0 then says
// "the ray is inside the root box of an interior root"; the packet enters the root block itself.
XRT_HD void begin_mesh_query(Lane &L, const SceneView &S, int mesh, bool enter = true) {
    const MeshRec &mr = S.meshes[mesh];
    L.mesh = mesh;
    L.sp = 0;
    L.mfound = 0;
    L.path = 0ull;
    L.mask = 0;
    L.state = ST_NODE;
    float key;
    if (!slab(L.r, mr.rmin[0], mr.rmin[1], mr.rmin[2], mr.rmax[0], mr.rmax[1], mr.rmax[2], key)) return;   // MO:265 on the root (MO:331)
    if (mr.rootBlock < 0) {   // the root is a leaf: one bucket
        if (mr.rootCount > 0) {
            L.ref = mr.rootRef; L.refEnd = mr.rootRef + mr.rootCount; L.leafKey = key; L.leafNode = ROOT_NODE;
            L.state = ST_LEAF;
        }
        return;
    }
    if (!enter) { L.mask = 1; return; }   // (mask 1 tells the packet "inside the root box": a ray that misses it ends here with mask 0, MO:265)
    // every triangle of the mesh faces away from the ray (RE:48-51 would reject each one): the answer of MO:259 is "no intersection"
    if (S.nodeCull && (S.nodeCull == 2 || !L.spec) &&
        all_back_facing(f4{mr.nbMin[0], mr.nbMin[1], mr.nbMin[2], mr.nbMin[3]}, f4{mr.nbMax[0], mr.nbMax[1], mr.nbMax[2], mr.nbMax[3]}, L.r.d)) return;
    L.bmin = mk(mr.rmin[0], mr.rmin[1], mr.rmin[2]);
    L.half = half_of(L.bmin, mk(mr.rmax[0], mr.rmax[1], mr.rmax[2]));
    load_block(L, S, mr.rootBlock);
    L.mask = permute_mask(0xff & ~((L.d2 >> 8) & 0xff), L.dmask);
    if (L.r.par == 0 && !L.weird) L.mask &= hit8_fast(L.r, L.dmask, L.bmin, L.half);
}

// Start of a query.  ignore (mesh, tri) is the `ignoreTriangle` identity (MO:290).
template <class SC>
XRT_HD void lane_begin(Lane &L, SC &C, const SceneView &S, v3 o, v3 d, int ignoreMesh, int ignoreTri, int rayIndex, int mode, int meshId, bool enter = true) {
    L.rayIndex = rayIndex;
    L.cost = 0;
    L.weird = (is_finite(o.x) && is_finite(o.y) && is_finite(o.z) && is_finite(d.x) && is_finite(d.y) && is_finite(d.z)) ? 0 : 1;
    L.ignoreId = -1;
    if (ignoreTri >= 0 && ignoreMesh >= 0 && ignoreMesh < S.nMeshes && ignoreTri < S.meshes[ignoreMesh].ntri)
        L.ignoreId = S.meshes[ignoreMesh].triBase + ignoreTri;
    C.sfound = 0; L.mfound = 0;
    C.sbKey = 0; C.sbD = 0; C.sbU = 0; C.sbV = 0; C.sbRef = 0; C.sbLeaf = 0; C.sbObj = -1; C.sbMesh = -1;
    C.sRef = 0; C.sRefEnd = 0; C.mPtr = 0; C.mEnd = 0; C.ssp = 0; C.sKey = 0; C.obj = -1;
    if (is_nan(o.x) || is_nan(o.y) || is_nan(o.z) || is_nan(d.x) || is_nan(d.y) || is_nan(d.z)) {
        // A NaN component (e.g. the refracted direction of a total internal reflection, RT:676-694) poisons every
        // determinant of RE:42-75 (NaN * 0 is NaN): no triangle can be accepted, while the NaN-propagating box test
        // of the reference accepts EVERY box — the reference walks the whole scene to return "no intersection".
        L.r = make_ray(o, d);
        C.w = L.r;   // (the counting pass reproduces that walk from the world ray)
        L.mfound = 0; L.mKey = 0;
        L.state = ST_FINISH;
        return;
    }
    L.spec = (L.ignoreId < 0) ? 1 : 0;   // rays leaving a surface (RT:485, RT:559) start among back faces
    if (mode == MODE_SCENE) {
        C.w = make_ray(o, d);
        C.sblk = 0; C.smask = 1;   // root = slot 0 of block 0
        L.state = ST_SCENE;
    } else if (mode == MODE_SINGLE) {
        // OSM:318 root box (a leaf holding the one body), OSM:349-364 ray transform, MESH:34-39, then MO:259
        RayPre w = make_ray(o, d);
        L.state = ST_FINISH;
        f4 lo = S.snodes[0], hi = S.snodes[1];
        float key;
        if (!slab(w, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, key)) return;
        if ((f2i(hi.w) & 0x0fffffff) == 0) return;
        const ObjRec &ob = S.objects[0];
        v3 v1 = transform(o, ob.invWorld);
        v3 v2 = transform(add(o, d), ob.invWorld);
        v3 dir = normalize(sub(v2, v1));
        L.r = make_ray(v1, dir);
        L.dmask = dir_mask(dir);
        if (!(is_finite(v1.x) && is_finite(v1.y) && is_finite(v1.z) && is_finite(dir.x) && is_finite(dir.y) && is_finite(dir.z))) L.weird = 1;
        C.obj = 0;
        C.sKey = key;
        const MeshRec &mr = S.meshes[0];
        float k;
        if (!slab(L.r, mr.bmin[0], mr.bmin[1], mr.bmin[2], mr.bmax[0], mr.bmax[1], mr.bmax[2], k)) return;
        begin_mesh_query(L, S, 0, enter);
    } else {
        L.r = make_ray(o, d);
        C.w = L.r;
        L.dmask = dir_mask(d);
        C.sblk = 0; C.smask = 0;
        C.obj = -1;
        begin_mesh_query(L, S, meshId, enter);
    }
}

XRT_HD void lane_begin(Lane &L, const SceneView &S, v3 o, v3 d, int ignoreMesh, int ignoreTri, int rayIndex, int mode, int meshId) {
    lane_begin(L, L.sc, S, o, d, ignoreMesh, ignoreTri, rayIndex, mode, meshId);
}

// ---- scene level: one step of OSM:318 (node collection) / OSM:334-433 (bucket scan), DFS order ---------
// OSM:370-378: the answer of the mesh query that just ended competes with the scene's best so far, strict '<' on
// the object-space distance inside one bucket.
template <class SC>
XRT_HD void merge_mesh_result(Lane &L, SC &C) {
    bool accept = !C.sfound || C.sKey < C.sbKey || (C.sKey == C.sbKey && L.mDist < C.sbD);
    if (accept) {
        C.sfound = 1;
        C.sbKey = C.sKey; C.sbD = L.mDist; C.sbU = L.mU; C.sbV = L.mV;
        C.sbRef = L.mRef; C.sbLeaf = L.mLeaf; C.sbObj = C.obj; C.sbMesh = L.mesh;
    }
}

// World-space object pre-cull (ObjRec).  BoundingBox.Intersects of the world ray against the object's world
// hull enlarged by m(|o|), one axis at a time so that few registers are live: the margin grows with the square of the
// origin's distance from the world origin, which is how the cancellation in OSM:358-364 (Transform(o + d) - Transform(o))
// bends the reference's object-space ray.  Same decisions as `slab` for a ray without NaNs (the early-outs collapse into
// the final comparison: tmin only grows, tmax only shrinks).  W = RayPre or its LDS proxy (component access by axis).
XRT_HD void ray_axis(const RayPre &w, int k, float &o, float &d, float &inv) {
    o = k == 0 ? w.o.x : (k == 1 ? w.o.y : w.o.z); d = k == 0 ? w.d.x : (k == 1 ? w.d.y : w.d.z); inv = k == 0 ? w.inv.x : (k == 1 ? w.inv.y : w.inv.z);
}
// (cmn = cullMin[0..3], cmx = cullMax[0..3], k2 = cullK2 of an ObjRec: the packet kernel reads them from SceneView::scull)
template <class W>
XRT_HD bool precull_box(const W &w, const f4 &cmn, const f4 &cmx, float k2) {
    float ox, oy, oz, dd, ii;
    ray_axis(w, 0, ox, dd, ii); ray_axis(w, 1, oy, dd, ii); ray_axis(w, 2, oz, dd, ii);
    const float r = (fabsf(ox) + fabsf(oy)) + fabsf(oz);   // >= |o|
    const float m = cmn.w + (cmx.w + k2 * r) * r;          // m(r) of ObjRec (xrt_core.h cull_margin)
    if (!(m <= 1.0e30f)) return true;   // an overflowing or NaN margin: no cull
    const float mn[3] = {cmn.x, cmn.y, cmn.z}, mx[3] = {cmx.x, cmx.y, cmx.z};
    float tmin = 0.0f, tmax = FLT_MAX;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float o, d, inv;
        ray_axis(w, k, o, d, inv);
        const float lo = mn[k] - m, hi = mx[k] + m;
        if (fabsf(d) < 1e-06f) ok = ok && !(o < lo || o > hi);
        else {
            const float t1 = ((d < 0.0f ? hi : lo) - o) * inv, t2 = ((d < 0.0f ? lo : hi) - o) * inv;
            tmin = fmaxf(tmin, t1); tmax = fminf(tmax, t2);
        }
    }
    return ok && !(tmin > tmax);
}
template <class W>
XRT_HD bool precull_hit(const W &w, const ObjRec &ob) {
    return precull_box(w, f4{ob.cullMin[0], ob.cullMin[1], ob.cullMin[2], ob.cullMin[3]}, f4{ob.cullMax[0], ob.cullMax[1], ob.cullMax[2], ob.cullMax[3]}, ob.cullK2);
}

template <class Stack, class SC>
XRT_HD void advance_scene(Lane &L, SC &C, const SceneView &S, Stack &stk) {
    if (L.mfound) {   // a mesh query ended with a hit since the last scene step (finish_mesh_query defers the merge to here)
        merge_mesh_result(L, C);
        L.mfound = 0;
    }
    // One step runs the scene cursor, in the reference's order, up to the next mesh query (or the end of the scene
    // walk, or a pop): meshes of the current body, then the next bodies of the current leaf, then the next children of
    // the current block.  Everything that is rejected on the way -- a mesh box, a pre-culled body, a child box the ray
    // misses, an empty or later-bucket leaf -- is passed inside the step.
    for (;;) {
        while (C.mPtr < C.mEnd) {   // OSM:366-368: next mesh of the current SceneObject
            const int m = S.objMesh[C.mPtr++];
            const MeshRec &mr = S.meshes[m];
            float k;
            if (slab(L.r, mr.bmin[0], mr.bmin[1], mr.bmin[2], mr.bmax[0], mr.bmax[1], mr.bmax[2], k)) {   // MESH:34-39
                begin_mesh_query(L, S, m);
                return;
            }
        }
        while (C.sRef < C.sRefEnd) {   // OSM:341-364: next body of the current leaf, world -> object space
            const int o = S.srefs[C.sRef++];
            const ObjRec &ob = S.objects[o];
            const RayPre w = C.w;
            if (ob.cullOk && !L.weird && !precull_hit(w, ob)) continue;   // conservative world-space reject: the visit would end at MESH:34-39 for every mesh
            C.obj = o;
            v3 rayDirPosition = add(w.o, w.d);                      // OSM:358
            v3 v1 = transform(w.o, ob.invWorld);                    // OSM:360
            v3 v2 = transform(rayDirPosition, ob.invWorld);         // OSM:361
            v3 dir = normalize(sub(v2, v1));                        // OSM:362-364
            L.r = make_ray(v1, dir);
            L.dmask = dir_mask(dir);
            if (!(is_finite(v1.x) && is_finite(v1.y) && is_finite(v1.z) && is_finite(dir.x) && is_finite(dir.y) && is_finite(dir.z))) L.weird = 1;
            C.mPtr = ob.meshStart;
            C.mEnd = ob.meshStart + ob.meshCount;
            return;   // its meshes on the next step
        }
        if (C.smask == 0) {   // block exhausted
            if (C.ssp == 0) { L.state = ST_FINISH; return; }
            unsigned wv = stk.get(--C.ssp);
            C.sblk = (int)(wv >> 8);
            C.smask = (int)(wv & 0xffu);
            return;
        }
        int sm = C.smask;
        const int c = ctz32((unsigned)sm);
        sm &= sm - 1;
        C.smask = sm;
        const int node = C.sblk * 8 + c;
        const f4 lo = S.snodes[2 * node], hi = S.snodes[2 * node + 1];
        float key;
        const RayPre w = C.w;
        if (!slab(w, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, key)) continue;   // OSM:460
        const int a_ = f2i(lo.w), b_ = f2i(hi.w);
        if (b_ < 0) {   // leaf
            const int cnt = b_ & 0x0fffffff;
            if (cnt == 0) continue;
            if (C.sfound && key > C.sbKey) continue;   // a later bucket than the one that already has a hit (OSM:334)
            C.sRef = a_; C.sRefEnd = a_ + cnt; C.sKey = key;
        } else {
            if (sm) stk.set(C.ssp++, ((unsigned)C.sblk << 8) | (unsigned)sm);
            C.sblk = a_ >> 3;
            C.smask = 0xff;
        }
    }
}

template <class Stack>
XRT_HD void advance_scene(Lane &L, const SceneView &S, Stack &stk) { advance_scene(L, L.sc, S, stk); }

// End of one MeshOctree.GetRayIntersection.  Scene mode: back to the scene cursor, which merges the answer on its next
// step -- the mesh phases never touch the scene-level half of the query.  MODE_MESH / MODE_SINGLE: the one mesh query
// was the whole query and lane_result reads its answer directly.
XRT_HD void finish_mesh_query(Lane &L, int mode) {
    L.state = (mode == MODE_SCENE) ? ST_SCENE : ST_FINISH;
}

// Every triangle of a leaf is rejected by the back-face test (RE:48-51: N.D > 0 in binary32) when a lower bound
// of N.D over the box [nmin, nmax] of the leaf's normals exceeds the rounding error of the evaluated dot product:
// |fl(N.D) - N.D| <= 3u * sum|N_k D_k| with u = 2^-24, and the bound itself is evaluated with the same error, so
// a margin of 1e-5 * sum max|N_k| |D_k| (> 50 x 6u) is safe.  NaNs make the comparison false (no skip).
XRT_HD bool all_back_facing(f4 nmin, f4 nmax, v3 d) {
    float lx = fminf(nmin.x * d.x, nmax.x * d.x), ly = fminf(nmin.y * d.y, nmax.y * d.y), lz = fminf(nmin.z * d.z, nmax.z * d.z);
    float sx = fmaxf(fabsf(nmin.x), fabsf(nmax.x)) * fabsf(d.x), sy = fmaxf(fabsf(nmin.y), fabsf(nmax.y)) * fabsf(d.y),
          sz = fmaxf(fabsf(nmin.z), fabsf(nmax.z)) * fabsf(d.z);
    float lower = (lx + ly) + lz, scale_ = (sx + sy) + sz;
    return nmin.w == 0.0f && lower > 1e-5f * scale_ && scale_ < 3.0e38f && lower == lower && d.x == d.x && d.y == d.y && d.z == d.z;
}

// ---- mesh level: one child of the current block (MO:328-353), front to back, with key pruning ------------------
// Memory is touched only when a block is entered (descend) or re-entered (pop): 32 bytes of descriptor.
template <class Stack>
XRT_HD void advance_node(Lane &L, const SceneView &S, Stack &stk, int mode, bool fast) {
    if (L.mask == 0) {   // block exhausted: pop the deepest level that still has pending children
        int m = 0;
        unsigned wv = 0;
        while (L.sp > 0) {
            wv = stk.get(S.sceneDepth + (--L.sp));
            m = (int)(wv & 0xffu);
            if (m) break;
        }
        if (m == 0) { finish_mesh_query(L, mode); return; }
        // re-enter block wv>>8 whose parent node sits at depth L.sp: recompute that node's own box from the
        // root with the child indices recorded in `path` (same binary32 operations as the builder, MO:207-218)
        const MeshRec &mr = S.meshes[L.mesh];
        v3 bmin = mk(mr.rmin[0], mr.rmin[1], mr.rmin[2]);
        v3 half = half_of(bmin, mk(mr.rmax[0], mr.rmax[1], mr.rmax[2]));
        for (int i = 0; i < L.sp; i++) {
            v3 cmin, cmax;
            child_box(bmin, half, (int)((L.path >> (3 * i)) & 7ull), cmin, cmax);
            bmin = cmin; half = half_of(cmin, cmax);
        }
        L.bmin = bmin; L.half = half;
        load_block(L, S, (int)(wv >> 8));
        L.mask = m;
        return;
    }
    int p = ctz32((unsigned)L.mask);
    L.mask &= L.mask - 1;
    int c = p ^ L.dmask;
    v3 cmin, cmax;
    child_box(L.bmin, L.half, c, cmin, cmax);
    float key;
    bool hitBox = fast ? slab_fast(L.r, L.dmask, cmin, cmax, key)
                       : slab(L.r, cmin.x, cmin.y, cmin.z, cmax.x, cmax.y, cmax.z, key);   // MO:331
    if (!hitBox) return;
    if (!((L.d2 >> c) & 1)) {   // leaf child (non-empty by construction of the mask): its entry key is the bucket key
        if (L.mfound && key > L.mKey) return;
        if (!L.spec) {   // the lane has lately met back faces only: can this whole leaf be rejected by RE:48-51 without reading it?
            const f4 nlo = S.leafNB[2 * (size_t)(L.blk * 8 + c)], nhi = S.leafNB[2 * (size_t)(L.blk * 8 + c) + 1];
            if (all_back_facing(nlo, nhi, L.r.d)) return;
        }
        {   // can the ray reach any triangle of the leaf at all (xrt_core.h leaf_certainly_missed)?
            const f4 *tb = S.leafTB + 4 * (size_t)(L.blk * 8 + c);
            if (leaf_certainly_missed(L.r, make_ray_cull(L.r.o, L.r.d), tb[0], tb[1], tb[2], tb[3])) return;
        }
        // offs[] is a running total over all eight children, so the list ends where the next child's starts
        const unsigned long long offLo = L.offLo, offHi = L.offHi;
        L.ref = L.d1 + child_ref_offset(offLo, offHi, c);
        L.refEnd = L.d1 + ((c == 7) ? L.d3 : child_ref_offset(offLo, offHi, c + 1));
        L.leafKey = key; L.leafNode = L.blk * 8 + c;
        L.state = ST_LEAF;
        return;
    }
    // interior child: prune by its own entry key only when that is a proven lower bound (safe bit)
    if (L.mfound && ((L.d2 >> (16 + c)) & 1) && key > L.mKey) return;
    if (S.nodeCull && ((L.d2 >> (24 + c)) & 1) && (S.nodeCull == 2 || !L.spec)) {   // ... and do not enter a subtree whose triangles all face away from the ray (the normal box of an interior node)
        const f4 nlo = S.leafNB[2 * (size_t)(L.blk * 8 + c)], nhi = S.leafNB[2 * (size_t)(L.blk * 8 + c) + 1];
        if (all_back_facing(nlo, nhi, L.r.d)) return;
    }
    stk.set(S.sceneDepth + L.sp, ((unsigned)L.blk << 8) | (unsigned)L.mask);
    L.path = (L.path & ~(7ull << (3 * L.sp))) | ((unsigned long long)c << (3 * L.sp));
    L.sp++;
    int nb = L.d0 + __builtin_popcount((unsigned)(L.d2 & 0xff) & ((1u << c) - 1u));
    L.bmin = cmin; L.half = half_of(cmin, cmax);
    load_block(L, S, nb);
    L.mask = permute_mask(0xff & ~((L.d2 >> 8) & 0xff), L.dmask);
    if (L.r.par == 0 && !L.weird) L.mask &= hit8_fast(L.r, L.dmask, L.bmin, L.half);
}

// ---- leaf: triangles of MO:288-304 --------------------------------------------------------------------------------
XRT_HD void leaf_candidate(Lane &L, const SceneView &S, int r, int triId, bool pass, float u, float v, float dist) {
    if (pass && dist < FLT_MAX) {   // MO:293-294 (minDistance starts at float.MaxValue)
        if (triId != L.ignoreId) {   // MO:290
            bool better;
            if (!L.mfound || L.leafKey < L.mKey) better = true;
            else if (L.leafKey == L.mKey) {
                if (dist < L.mDist) better = true;
                // (two leaves: the earlier in DFS order; one leaf: the earlier in the reference's list, which is ascending in the triangle index -- the
                // storage order of a big leaf is not the list order, scene_host.cpp spatial_runs)
                else if (dist == L.mDist) better = (L.leafNode != L.mLeaf) ? (node_dfs(S, L.leafNode) < node_dfs(S, L.mLeaf)) : (f2i(S.refN[r].w) < f2i(S.refN[L.mRef].w));
                else better = false;
            } else better = false;
            if (better) {
                L.mfound = 1;
                L.mKey = L.leafKey; L.mDist = dist; L.mU = u; L.mV = v; L.mRef = r; L.mLeaf = L.leafNode;
            }
        }
    }
}
// Two references per step.  The 16-byte normal records decide the back-face test; the 36-byte geometry is
// fetched only for front-facing references — together with the normals when the lane's previous step met a
// front-facing triangle (`spec`), one round trip later otherwise.  Rays leaving a surface see almost only
// back faces around their origin, so most of their references cost 16 bytes and five multiply-adds.
XRT_HD void advance_leaf(Lane &L, const SceneView &S) {
    const int r0 = L.ref;
    const bool two = (r0 + 1) < L.refEnd;
    const int r1 = two ? r0 + 1 : r0;
    const f4 n0 = S.refN[r0], n1 = S.refN[r1];
    g3 a0 = {0, 0, 0}, b0 = {0, 0, 0}, c0 = {0, 0, 0}, a1 = {0, 0, 0}, b1 = {0, 0, 0}, c1 = {0, 0, 0};
    const bool spec = L.spec != 0;
    if (spec) {
        a0 = S.refG[3 * (size_t)r0]; b0 = S.refG[3 * (size_t)r0 + 1]; c0 = S.refG[3 * (size_t)r0 + 2];
        a1 = S.refG[3 * (size_t)r1]; b1 = S.refG[3 * (size_t)r1 + 1]; c1 = S.refG[3 * (size_t)r1 + 2];
    }
    const bool f0 = !(facing(mk(n0.x, n0.y, n0.z), L.r.d) > 0.0f);            // RE:48-51
    const bool f1 = two && !(facing(mk(n1.x, n1.y, n1.z), L.r.d) > 0.0f);
    if ((f0 || f1) && !spec) {
        a0 = S.refG[3 * (size_t)r0]; b0 = S.refG[3 * (size_t)r0 + 1]; c0 = S.refG[3 * (size_t)r0 + 2];
        a1 = S.refG[3 * (size_t)r1]; b1 = S.refG[3 * (size_t)r1 + 1]; c1 = S.refG[3 * (size_t)r1 + 2];
    }
    if (f0 || f1) {
        float u0, v0, t0, u1, v1, t1;
        bool p0 = tri_test_front(L.r.o, L.r.d, mk(a0.x, a0.y, a0.z), mk(b0.x, b0.y, b0.z), mk(c0.x, c0.y, c0.z), u0, v0, t0);
        bool p1 = tri_test_front(L.r.o, L.r.d, mk(a1.x, a1.y, a1.z), mk(b1.x, b1.y, b1.z), mk(c1.x, c1.y, c1.z), u1, v1, t1);
        leaf_candidate(L, S, r0, f2i(n0.w), f0 && p0, u0, v0, t0);
        leaf_candidate(L, S, r1, f2i(n1.w), f1 && p1, u1, v1, t1);
    }
    L.spec = (f0 || f1) ? 1 : 0;
    L.ref = r0 + 2;
    if (L.ref >= L.refEnd) L.state = ST_NODE;
}

// ---- result: MO:308-323 interpolated position, OSM:438-452 world position ----------------------------------------
struct HitOut {
    int hit, object, mesh, tri, leaf;
    float u, v, d, wx, wy, wz;
    int cost;
};
template <class SC>
XRT_HD HitOut lane_result(const Lane &L, const SC &C, const SceneView &S, int mode) {
    HitOut h;
    h.hit = 0; h.object = -1; h.mesh = -1; h.tri = -1; h.leaf = -1;
    h.u = 0; h.v = 0; h.d = 0; h.wx = 0; h.wy = 0; h.wz = 0;
    h.cost = L.cost;
    const bool sc = mode == MODE_SCENE;
    if (!(sc ? (int)C.sfound : L.mfound)) return h;
    const int ref = sc ? (int)C.sbRef : L.mRef, mesh = sc ? (int)C.sbMesh : L.mesh, obj = sc ? (int)C.sbObj : 0;   // MODE_SINGLE: body 0
    const float u = sc ? (float)C.sbU : L.mU, v = sc ? (float)C.sbV : L.mV;
    g3 a = S.refG[3 * (size_t)ref], b = S.refG[3 * (size_t)ref + 1], c = S.refG[3 * (size_t)ref + 2];
    v3 v1 = mk(a.x, a.y, a.z), p1 = mk(b.x, b.y, b.z), p2 = mk(c.x, c.y, c.z);
    v3 pos = add(add(v1, scale(p1, u)), scale(p2, v));   // MO:310-312
    h.hit = 1;
    h.mesh = mesh;
    h.tri = f2i(S.refN[ref].w) - S.meshes[mesh].triBase;
    h.leaf = node_dfs(S, sc ? (int)C.sbLeaf : L.mLeaf);
    h.u = u; h.v = v; h.d = sc ? (float)C.sbD : L.mDist;
    if (mode != MODE_MESH) {
        h.object = obj;
        pos = transform(pos, S.objects[obj].world);   // OSM:441-443
    }
    h.wx = pos.x; h.wy = pos.y; h.wz = pos.z;
    return h;
}
XRT_HD HitOut lane_result(const Lane &L, const SceneView &S, int mode) { return lane_result(L, L.sc, S, mode); }
XRT_HD bool lane_found(const Lane &L, int mode) { return (mode == MODE_SCENE ? L.sc.sfound : L.mfound) != 0; }

}  // namespace xrt
