// scene_build.cpp — see scene_build.h.  The tree SHAPE, the node order and the leaf contents are
// those of the reference (they define bucket keys, DFS order and tie-breaks); only
// the storage differs: children contiguous in blocks of 8, leaves as contiguous reference runs, and
// per interior node the union of its non-empty descendant leaf boxes for exact key pruning.
#include "scene_build.h"

#include <cstring>

namespace xrt {
namespace {

struct Box { float mn[3], mx[3]; };

inline bool contains_point(const Box &b, const float *p) {   // BoundingBox.Contains(Vector3) != Disjoint
    return b.mn[0] <= p[0] && p[0] <= b.mx[0] && b.mn[1] <= p[1] && p[1] <= b.mx[1] && b.mn[2] <= p[2] && p[2] <= b.mx[2];
}
inline bool boxes_intersect(const float *a, const Box &b) {   // a.Intersects(b), a = {min xyz, max xyz} (OSM:240)
    if (a[3] < b.mn[0] || a[0] > b.mx[0]) return false;
    if (a[4] < b.mn[1] || a[1] > b.mx[1]) return false;
    return a[5] >= b.mn[2] && a[2] <= b.mx[2];
}
inline void child_box(const Box &parent, int i, int j, int k, Box &c) {   // MO:207,217-218 / OSM:221,231-232
    float half[3];
    for (int a = 0; a < 3; a++) half[a] = (parent.mx[a] - parent.mn[a]) * (1.0f / 2.0f);   // Vector3 / 2f
    const int ijk[3] = {i, j, k};
    for (int a = 0; a < 3; a++) {
        float pos = parent.mn[a] + half[a] * (float)ijk[a];
        c.mn[a] = pos;
        c.mx[a] = pos + half[a];
    }
}
inline void put_node(FlatTree &t, int rec, const Box &b, int a, int bb) {
    t.nodes[2 * rec] = f4{b.mn[0], b.mn[1], b.mn[2], i2f(a)};
    t.nodes[2 * rec + 1] = f4{b.mx[0], b.mx[1], b.mx[2], i2f(bb)};
}
inline int alloc_block(FlatTree &t) {
    int first = (int)(t.nodes.size() / 2);
    t.nodes.resize(t.nodes.size() + 16, f4{0, 0, 0, 0});
    t.nodeDfs.resize(t.nodeDfs.size() + 8, -1);
    return first;
}
inline void push_info(FlatTree &t, const Box &b, bool leaf, int count, int dfs, int depth, int firstRef) {
    xrt_node_info o;
    std::memset(&o, 0, sizeof(o));
    for (int a = 0; a < 3; a++) { o.bmin[a] = b.mn[a]; o.bmax[a] = b.mx[a]; }
    o.is_leaf = leaf ? 1 : 0; o.count = count; o.dfs_index = dfs; o.depth = depth; o.first_ref = firstRef;
    t.info.push_back(o);
}

struct MeshBuilder {
    const HostMesh &m;
    int threshold;
    FlatTree &t;
    std::string &err;
    int dfs = 0;
    static constexpr int kMaxLevel = 20;          // 3 path bits per level in one 64-bit word (traverse.h)
    static constexpr int kMaxNodes = 1 << 24;

    bool tri_in(const Box &b, int tri) const {   // MO:226-228: any of the three vertices inside-or-on
        const float *p = &m.v[(size_t)tri * 9];
        return contains_point(b, p) || contains_point(b, p + 3) || contains_point(b, p + 6);
    }
    int alloc_blocks(int n) {
        int first = (int)(t.blocks.size() / 2);
        t.blocks.resize(t.blocks.size() + 2 * (size_t)n, f4{0, 0, 0, 0});
        t.childDfs.resize(t.childDfs.size() + 8 * (size_t)n, -1);
        return first;
    }

    // Fills block `blk` = the eight children of the interior node (box, list) at depth `level`
    // (MO:204-236 SplitCuboid, then MO:91-94 recursion in index order).  `myDfs` is the node's own DFS index,
    // already pushed to `info`.  uni/uniValid: union of the boxes of the non-empty leaves below the node.
    bool build_interior(int blk, const Box &box, const std::vector<int> &list, int level, Box &uni, bool &uniValid) {
        if (level + 1 > kMaxLevel || t.nodeCount > kMaxNodes) {
            err = "MeshOctree.BuildTree would not terminate or is deeper than 20 levels: more than the item threshold triangles share a vertex (MO:84-96 has no depth limit)";
            return false;
        }
        if (level + 1 > t.maxDepth) t.maxDepth = level + 1;
        t.interiors++;
        Box cb[8];
        std::vector<int> cl[8];
        int index = 0;
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int k = 0; k < 2; k++) {
                    child_box(box, i, j, k, cb[index]);
                    std::vector<int> &l = cl[index];
                    for (int tri : list)
                        if (tri_in(cb[index], tri)) l.push_back(tri);
                    index++;
                }
        unsigned interiorMask = 0, emptyMask = 0, safeMask = 0;
        int nInterior = 0;
        for (int c = 0; c < 8; c++)
            if ((int)cl[c].size() > threshold) { interiorMask |= 1u << c; nInterior++; }   // MO:86
            else if (cl[c].empty()) emptyMask |= 1u << c;
        const int childBlockBase = nInterior ? alloc_blocks(nInterior) : 0;
        const int refBase = (int)t.leafRefs.size();
        unsigned offs[8];
        unsigned total = 0;
        for (int c = 0; c < 8; c++) {   // this block's leaf lists, contiguous, child order
            offs[c] = total;
            if (!(interiorMask & (1u << c))) {
                t.leafRefs.insert(t.leafRefs.end(), cl[c].begin(), cl[c].end());
                total += (unsigned)cl[c].size();
            }
        }
        if (total > 0xffffu) { err = "more than 65535 triangle references in the leaves of one octree node (item threshold too large for the block descriptor)"; return false; }
        bool any = false;
        Box u{};
        auto merge = [&](const Box &cu) {
            if (!any) { u = cu; any = true; }
            else for (int a = 0; a < 3; a++) { if (cu.mn[a] < u.mn[a]) u.mn[a] = cu.mn[a]; if (cu.mx[a] > u.mx[a]) u.mx[a] = cu.mx[a]; }
        };
        int rank = 0;
        for (int c = 0; c < 8; c++) {   // DFS pre-order: child c and its whole subtree before child c+1
            const int myDfs = dfs++;
            t.childDfs[(size_t)blk * 8 + c] = myDfs;
            t.nodeCount++;
            if (interiorMask & (1u << c)) {
                push_info(t, cb[c], false, (int)cl[c].size(), myDfs, level + 1, -1);
                Box cu; bool cv = false;
                if (!build_interior(childBlockBase + rank, cb[c], cl[c], level + 1, cu, cv)) return false;
                rank++;
                bool inside = true;
                if (cv) {
                    merge(cu);
                    for (int a = 0; a < 3; a++) inside = inside && cu.mn[a] >= cb[c].mn[a] && cu.mx[a] <= cb[c].mx[a];
                }
                if (inside) safeMask |= 1u << c; else t.unsafeNodes++;
            } else {
                push_info(t, cb[c], true, (int)cl[c].size(), myDfs, level + 1, (int)t.infoRefs.size());
                t.infoRefs.insert(t.infoRefs.end(), cl[c].begin(), cl[c].end());
                t.leafCount++;
                if (cl[c].empty()) t.emptyLeaves++; else merge(cb[c]);
            }
            std::vector<int>().swap(cl[c]);
        }
        int w[8];
        w[0] = childBlockBase; w[1] = refBase;
        w[2] = (int)(interiorMask | (emptyMask << 8) | (safeMask << 16));
        w[3] = (int)total;
        for (int q = 0; q < 4; q++) w[4 + q] = (int)(offs[2 * q] | (offs[2 * q + 1] << 16));
        t.blocks[2 * (size_t)blk] = f4{i2f(w[0]), i2f(w[1]), i2f(w[2]), i2f(w[3])};
        t.blocks[2 * (size_t)blk + 1] = f4{i2f(w[4]), i2f(w[5]), i2f(w[6]), i2f(w[7])};
        uni = u;
        uniValid = any;
        return true;
    }
};

}  // namespace

bool build_mesh_tree(const HostMesh &m, int threshold, FlatTree &t, std::string &err) {
    t = FlatTree();
    if (threshold > 8000) { err = "mesh item threshold above 8000 is not supported by the block descriptor"; return false; }
    // MO:56-82: root box from all vertices, starting at (+MaxValue, -MaxValue); Vector3.Min/Max.
    Box root;
    for (int a = 0; a < 3; a++) { root.mn[a] = FLT_MAX; root.mx[a] = -FLT_MAX; }
    for (int i = 0; i < m.ntri; i++)
        for (int vtx = 0; vtx < 3; vtx++) {
            const float *p = &m.v[(size_t)i * 9 + vtx * 3];
            for (int a = 0; a < 3; a++) {
                root.mn[a] = (root.mn[a] < p[a]) ? root.mn[a] : p[a];
                root.mx[a] = (root.mx[a] > p[a]) ? root.mx[a] : p[a];
            }
        }
    for (int a = 0; a < 3; a++) { t.rootBox[a] = root.mn[a]; t.rootBox[3 + a] = root.mx[a]; }
    std::vector<int> all(m.ntri);
    for (int i = 0; i < m.ntri; i++) all[i] = i;
    MeshBuilder b{m, threshold, t, err};
    b.dfs = 1;   // the root is DFS index 0
    t.nodeCount = 1;
    if (m.ntri <= threshold) {   // MO:86: the root is a leaf
        push_info(t, root, true, m.ntri, 0, 0, 0);
        t.rootIsLeaf = true;
        t.rootCount = m.ntri;
        t.leafRefs = all;
        t.infoRefs = all;
        t.leafCount = 1;
        if (m.ntri == 0) t.emptyLeaves = 1;
        return true;
    }
    t.rootIsLeaf = false;
    push_info(t, root, false, m.ntri, 0, 0, -1);
    const int blk = b.alloc_blocks(1);
    Box u; bool uv = false;
    return b.build_interior(blk, root, all, 0, u, uv);
}

namespace {
struct SceneBuilder {
    const std::vector<HostObject> &objs;
    int threshold;
    FlatTree &t;
    std::string &err;
    int dfs = 0;
    bool build(int rec, const Box &box, const std::vector<int> &list, int level) {
        const int myDfs = dfs++;
        t.nodeDfs[rec] = myDfs;
        t.nodeCount++;
        if (level > t.maxDepth) t.maxDepth = level;
        if ((int)list.size() <= threshold) {   // OSM:103
            int start = (int)t.leafRefs.size();
            push_info(t, box, true, (int)list.size(), myDfs, level, (int)t.infoRefs.size());
            t.leafRefs.insert(t.leafRefs.end(), list.begin(), list.end());
            t.infoRefs.insert(t.infoRefs.end(), list.begin(), list.end());
            put_node(t, rec, box, start, NODE_LEAF | (int)list.size());
            t.leafCount++;
            return true;
        }
        if (level >= 24 || t.nodeCount > (1 << 20)) { err = "OctreeSpatialManager.BuildTree would not terminate (OSM:101-113 has no depth limit)"; return false; }
        push_info(t, box, false, (int)list.size(), myDfs, level, -1);
        t.interiors++;
        const int first = alloc_block(t);
        Box cb[8];
        std::vector<int> cl[8];
        int index = 0;
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int k = 0; k < 2; k++) {
                    child_box(box, i, j, k, cb[index]);
                    for (int o : list)
                        if (boxes_intersect(objs[o].worldBbox, cb[index])) cl[index].push_back(o);   // OSM:240
                    index++;
                }
        put_node(t, rec, box, first, 0);
        for (int c = 0; c < 8; c++)
            if (!build(first + c, cb[c], cl[c], level + 1)) return false;
        return true;
    }
};
}  // namespace

bool build_scene_tree(const std::vector<HostObject> &objs, int threshold, FlatTree &t, std::string &err) {
    t = FlatTree();
    // OSM:64-99: root = merge of per-body boxes, each = sorted {Min*W, Max*W} (two corners only).
    Box root;
    for (int a = 0; a < 3; a++) root.mn[a] = root.mx[a] = 0.0f;
    for (size_t i = 0; i < objs.size(); i++) {
        v3 tmn = transform(mk(objs[i].bbox[0], objs[i].bbox[1], objs[i].bbox[2]), objs[i].world);
        v3 tmx = transform(mk(objs[i].bbox[3], objs[i].bbox[4], objs[i].bbox[5]), objs[i].world);
        const float a0[3] = {tmn.x, tmn.y, tmn.z}, a1[3] = {tmx.x, tmx.y, tmx.z};
        Box ob;
        for (int a = 0; a < 3; a++) { ob.mn[a] = (a0[a] < a1[a]) ? a0[a] : a1[a]; ob.mx[a] = (a0[a] > a1[a]) ? a0[a] : a1[a]; }
        if (i == 0) root = ob;
        else for (int a = 0; a < 3; a++) { root.mn[a] = (root.mn[a] < ob.mn[a]) ? root.mn[a] : ob.mn[a]; root.mx[a] = (root.mx[a] > ob.mx[a]) ? root.mx[a] : ob.mx[a]; }
    }
    std::vector<int> all(objs.size());
    for (size_t i = 0; i < objs.size(); i++) all[i] = (int)i;
    alloc_block(t);
    SceneBuilder b{objs, threshold, t, err};
    return b.build(0, root, all, 0);
}

}  // namespace xrt
