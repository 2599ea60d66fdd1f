// scene_build.h — host side of xrt_scene_build: MeshOctree.Build (MO:56-96, 204-236) and
// OctreeSpatialManager.Build (OSM:64-113, 218-248) restated for flat, HBM-friendly arrays.
// Plain C++ (no HIP): the arrays are uploaded by xrt_api.cpp.
#pragma once
#include <string>
#include <vector>

#include "../../include/xrt.h"
#include "xrt_core.h"

namespace xrt {

struct HostMesh {
    std::vector<float> v, n, uv, sn, color;   // 9,9,6,3,4 floats per triangle
    int ntri = 0;
    float bbox[6] = {0, 0, 0, 0, 0, 0};
    float reflectiveness = 0, refractionIndex = 0;
    bool transparent = false, interpolateNormals = false, useTexture = false;
    int texW = 0, texH = 0;
    std::vector<uint32_t> texels;    // the Format32bppArgb lock (MAT:65)
    std::vector<uint32_t> texelsP;   // Texture.ColorData, the premultiplied copy (TEX:24-33); empty: same as texels
};

struct HostObject {
    std::vector<int> meshes;
    float world[16], invWorld[16], bbox[6], worldBbox[6];
};

// One flattened tree, indices local to the tree.  Mesh trees use the implicit-box block layout of
// xrt_core.h (`blocks`, `childDfs`, root fields); the scene tree uses explicit records (`nodes`, root =
// record 0 of block 0).  `info`/`infoRefs` are the DFS pre-order inspection view (xrt_scene_get_tree).
struct FlatTree {
    // mesh trees
    std::vector<f4> blocks;         // 2 per block descriptor
    std::vector<int> childDfs;      // 8 per block: DFS pre-order index of child c
    float rootBox[6] = {0, 0, 0, 0, 0, 0};
    bool rootIsLeaf = true;
    int rootCount = 0;
    // scene tree
    std::vector<f4> nodes;          // 2 per record
    std::vector<int> nodeDfs;       // per record: DFS pre-order index (-1 for padding records)
    // both
    std::vector<int> leafRefs;      // storage order: local triangle index (mesh trees) or object id (scene tree)
    std::vector<xrt_node_info> info;   // DFS order
    std::vector<int> infoRefs;         // DFS order
    int nodeCount = 0, leafCount = 0, emptyLeaves = 0, maxDepth = 0, unsafeNodes = 0, interiors = 0;
};

// Returns false (and sets err) when the reference's recursion would not terminate.
bool build_mesh_tree(const HostMesh &m, int threshold, FlatTree &out, std::string &err);
bool build_scene_tree(const std::vector<HostObject> &objs, int threshold, FlatTree &out, std::string &err);

}  // namespace xrt
