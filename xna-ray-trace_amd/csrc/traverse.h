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
    const f4 *nodes;        // mesh octrees, 2 per record
    const f4 *ownBox;       // 2 per interior record (reference's own box)
    const int *nodeDfs;     // per record
    const f4 *triRec;       // 3 per leaf reference
    const int *refTri;      // global triangle id per leaf reference
    const MeshRec *meshes;
    const f4 *snodes;       // scene octree, 2 per record
    const int *srefs;       // object id per scene leaf reference
    const ObjRec *objects;
    const int *objMesh;
    int nMeshes, nObjects;
    int sceneDepth, meshDepth;   // stack capacities needed
};

enum : int { ST_IDLE = 0, ST_SCENE = 1, ST_NODE = 2, ST_LEAF = 3, ST_FINISH = 4 };
enum : int { MODE_SCENE = 0, MODE_MESH = 1 };

struct Lane {
    RayPre w;            // world ray (scene mode)
    RayPre r;            // object-space ray
    int ignoreId;        // global triangle id, -1 = none
    int rayIndex;
    int state;
    // scene cursor
    int sblk, smask, ssp;
    int sRef, sRefEnd;
    float sKey;
    int obj, mPtr, mEnd;
    // mesh query
    int mesh, dmask;
    int blk, mask, sp;
    int mfound;
    float mKey, mDist, mU, mV;
    int mRef, mLeaf;
    // leaf scan
    int ref, refEnd, leafNode;
    float leafKey;
    // scene best
    int sfound;
    float sbKey, sbD, sbU, sbV;
    int sbRef, sbLeaf, sbObj, sbMesh;
};

XRT_HD int ctz32(unsigned x) { return __builtin_ctz(x); }

XRT_HD void begin_mesh_query(Lane &L, const SceneView &S, int mesh) {
    L.mesh = mesh;
    L.blk = S.meshes[mesh].rootNode >> 3;
    L.mask = 1 << L.dmask;   // only slot 0 (the root) of the root block: p ^ dmask == 0
    L.sp = 0;
    L.mfound = 0;
    L.state = ST_NODE;
}

// Start of a query.  ignore (mesh, tri) is the `ignoreTriangle` identity (MO:290).
XRT_HD void lane_begin(Lane &L, const SceneView &S, v3 o, v3 d, int ignoreMesh, int ignoreTri, int rayIndex, int mode, int meshId) {
    L.rayIndex = rayIndex;
    L.ignoreId = -1;
    if (ignoreTri >= 0 && ignoreMesh >= 0 && ignoreMesh < S.nMeshes && ignoreTri < S.meshes[ignoreMesh].ntri)
        L.ignoreId = S.meshes[ignoreMesh].triBase + ignoreTri;
    L.sfound = 0;
    L.sbKey = 0; L.sbD = 0; L.sbU = 0; L.sbV = 0; L.sbRef = 0; L.sbLeaf = 0; L.sbObj = -1; L.sbMesh = -1;
    L.sRef = 0; L.sRefEnd = 0; L.mPtr = 0; L.mEnd = 0; L.ssp = 0; L.sKey = 0; L.obj = -1;
    if (mode == MODE_SCENE) {
        L.w = make_ray(o, d);
        L.sblk = 0; L.smask = 1;   // root = slot 0 of block 0
        L.state = ST_SCENE;
    } else {
        L.w = make_ray(o, d);
        L.r = L.w;
        L.dmask = dir_mask(d);
        L.sblk = 0; L.smask = 0;
        begin_mesh_query(L, S, meshId);
    }
}

// ---- scene level: one step of OSM:318 (node collection) / OSM:334-433 (bucket scan), DFS order ---------
template <class Stack>
XRT_HD void advance_scene(Lane &L, const SceneView &S, Stack &stk) {
    if (L.mPtr < L.mEnd) {   // OSM:366-368: next mesh of the current SceneObject
        int m = S.objMesh[L.mPtr++];
        const MeshRec &mr = S.meshes[m];
        float k;
        if (slab(L.r, mr.bmin[0], mr.bmin[1], mr.bmin[2], mr.bmax[0], mr.bmax[1], mr.bmax[2], k))   // MESH:34-39
            begin_mesh_query(L, S, m);
        return;
    }
    if (L.sRef < L.sRefEnd) {   // OSM:341-364: next body of the current leaf, world -> object space
        int o = S.srefs[L.sRef++];
        L.obj = o;
        const ObjRec &ob = S.objects[o];
        v3 rayDirPosition = add(L.w.o, L.w.d);                 // OSM:358
        v3 v1 = transform(L.w.o, ob.invWorld);                  // OSM:360
        v3 v2 = transform(rayDirPosition, ob.invWorld);         // OSM:361
        v3 dir = normalize(sub(v2, v1));                        // OSM:362-364
        L.r = make_ray(v1, dir);
        L.dmask = dir_mask(dir);
        L.mPtr = ob.meshStart;
        L.mEnd = ob.meshStart + ob.meshCount;
        return;
    }
    if (L.smask == 0) {
        if (L.ssp == 0) { L.state = ST_FINISH; return; }
        unsigned wv = stk.get(--L.ssp);
        L.sblk = (int)(wv >> 8);
        L.smask = (int)(wv & 0xffu);
        return;
    }
    int c = ctz32((unsigned)L.smask);
    L.smask &= L.smask - 1;
    int node = L.sblk * 8 + c;
    f4 lo = S.snodes[2 * node], hi = S.snodes[2 * node + 1];
    float key;
    if (!slab(L.w, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, key)) return;   // OSM:460
    int a = f2i(lo.w), b = f2i(hi.w);
    if (b < 0) {   // leaf
        int cnt = b & 0x0fffffff;
        if (cnt == 0) return;
        if (L.sfound && key > L.sbKey) return;   // a later bucket than the one that already has a hit (OSM:334)
        L.sRef = a; L.sRefEnd = a + cnt; L.sKey = key;
    } else {
        if (L.smask) stk.set(L.ssp++, ((unsigned)L.sblk << 8) | (unsigned)L.smask);
        L.sblk = a >> 3;
        L.smask = 0xff;
    }
}

// End of one MeshOctree.GetRayIntersection: OSM:370-378 accept with strict '<' on the object-space d.
XRT_HD void finish_mesh_query(Lane &L, int mode) {
    if (L.mfound) {
        bool accept = !L.sfound || L.sKey < L.sbKey || (L.sKey == L.sbKey && L.mDist < L.sbD);
        if (accept) {
            L.sfound = 1;
            L.sbKey = L.sKey; L.sbD = L.mDist; L.sbU = L.mU; L.sbV = L.mV;
            L.sbRef = L.mRef; L.sbLeaf = L.mLeaf; L.sbObj = L.obj; L.sbMesh = L.mesh;
        }
    }
    L.state = (mode == MODE_SCENE) ? ST_SCENE : ST_FINISH;
}

// ---- mesh level: pop one octree node (MO:328-353), front to back, with key pruning --------------------------
template <class Stack>
XRT_HD void advance_node(Lane &L, const SceneView &S, Stack &stk, int mode) {
    if (L.mask == 0) {
        if (L.sp == 0) { finish_mesh_query(L, mode); return; }
        unsigned wv = stk.get(S.sceneDepth + (--L.sp));
        L.blk = (int)(wv >> 8);
        L.mask = (int)(wv & 0xffu);
        return;
    }
    int p = ctz32((unsigned)L.mask);
    L.mask &= L.mask - 1;
    int node = L.blk * 8 + (p ^ L.dmask);
    f4 lo = S.nodes[2 * node], hi = S.nodes[2 * node + 1];
    int a = f2i(lo.w), b = f2i(hi.w);
    if (b < 0) {   // leaf: the record holds the reference's own box, whose entry distance is the bucket key
        int cnt = b & 0x0fffffff;
        if (cnt == 0) return;   // empty leaves are bucketed by the reference but cannot produce a hit (Q4)
        float key;
        if (!slab(L.r, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, key)) return;
        if (L.mfound && key > L.mKey) return;
        L.ref = a; L.refEnd = a + cnt; L.leafKey = key; L.leafNode = node;
        L.state = ST_LEAF;
        return;
    }
    if (b & NODE_EMPTY) return;
    float key;   // lower bound of every bucket key below this node (union of its non-empty leaf boxes)
    if (!slab(L.r, lo.x, lo.y, lo.z, hi.x, hi.y, hi.z, key)) return;
    if (L.mfound && key > L.mKey) return;
    if (b & NODE_OWN_TEST) {   // a descendant protrudes from the own box: the reference's own test decides (MO:331-334)
        int side = b & NODE_SIDE_MASK;
        f4 olo = S.ownBox[2 * side], ohi = S.ownBox[2 * side + 1];
        float k2;
        if (!slab(L.r, olo.x, olo.y, olo.z, ohi.x, ohi.y, ohi.z, k2)) return;
    }
    if (L.mask) stk.set(S.sceneDepth + (L.sp++), ((unsigned)L.blk << 8) | (unsigned)L.mask);
    L.blk = a >> 3;
    L.mask = 0xff;
}

// ---- leaf: one triangle of MO:288-304 -------------------------------------------------------------------------
XRT_HD void advance_leaf(Lane &L, const SceneView &S) {
    int r = L.ref++;
    f4 a = S.triRec[3 * r], b = S.triRec[3 * r + 1], c = S.triRec[3 * r + 2];
    float u, v, dist;
    if (tri_test(L.r.o, L.r.d, a, b, c, u, v, dist) && dist < FLT_MAX) {   // MO:293-294 (minDistance starts at float.MaxValue)
        if (S.refTri[r] != L.ignoreId) {   // MO:290
            bool better;
            if (!L.mfound || L.leafKey < L.mKey) better = true;
            else if (L.leafKey == L.mKey) {
                if (dist < L.mDist) better = true;
                else if (dist == L.mDist) better = (L.leafNode != L.mLeaf) && (S.nodeDfs[L.leafNode] < S.nodeDfs[L.mLeaf]);
                else better = false;
            } else better = false;
            if (better) {
                L.mfound = 1;
                L.mKey = L.leafKey; L.mDist = dist; L.mU = u; L.mV = v; L.mRef = r; L.mLeaf = L.leafNode;
            }
        }
    }
    if (L.ref >= L.refEnd) L.state = ST_NODE;
}

// ---- result: MO:308-323 interpolated position, OSM:438-452 world position ----------------------------------------
struct HitOut {
    int hit, object, mesh, tri, leaf;
    float u, v, d, wx, wy, wz;
};
XRT_HD HitOut lane_result(const Lane &L, const SceneView &S, int mode) {
    HitOut h;
    h.hit = 0; h.object = -1; h.mesh = -1; h.tri = -1; h.leaf = -1;
    h.u = 0; h.v = 0; h.d = 0; h.wx = 0; h.wy = 0; h.wz = 0;
    if (!L.sfound) return h;
    f4 a = S.triRec[3 * L.sbRef], b = S.triRec[3 * L.sbRef + 1], c = S.triRec[3 * L.sbRef + 2];
    v3 v1 = mk(a.x, a.y, a.z), p1 = mk(b.x, b.y, b.z), p2 = mk(c.x, c.y, c.z);
    v3 pos = add(add(v1, scale(p1, L.sbU)), scale(p2, L.sbV));   // MO:310-312
    h.hit = 1;
    h.mesh = L.sbMesh;
    h.tri = S.refTri[L.sbRef] - S.meshes[L.sbMesh].triBase;
    h.leaf = S.nodeDfs[L.sbLeaf];
    h.u = L.sbU; h.v = L.sbV; h.d = L.sbD;
    if (mode == MODE_SCENE) {
        h.object = L.sbObj;
        pos = transform(pos, S.objects[L.sbObj].world);   // OSM:441-443
    }
    h.wx = pos.x; h.wy = pos.y; h.wz = pos.z;
    return h;
}

}  // namespace xrt
