// scene_host.h — host representation of one xrt_scene: the uploaded Mesh / SceneObject graph
// (xrt_scene_add_mesh / xrt_scene_add_object) and, after build(), the flat arrays that go to HBM.
// Plain C++ (no HIP).
#pragma once
#include <string>
#include <vector>

#include "scene_build.h"
#include "traverse.h"

namespace xrt {

constexpr int SHADE_F4 = 6;   // f4 per triangle shading record:
//   s0 = (n1.xyz, uv1.x) s1 = (n2.xyz, uv1.y) s2 = (n3.xyz, uv2.x) s3 = color.xyzw
//   s4 = (uv2.y, uv3.x, uv3.y, as_float(material))  s5 = (surfaceNormal.xyz, as_float(mesh))

struct SceneArrays {
    std::vector<f4> blocks, refN, snodes, shade, leafNB;   // leafNB: 2 per node: component-wise min / max of the leaf's surface normals
    std::vector<float> refT;                               // refN + refG as one 13-word record per reference, 16 words of padding at the end
    std::vector<f4> triTB;                                 // 4 per leaf reference: the tight box of that one triangle (xrt_core.h bundle_certainly_missed)
    std::vector<f4> runTB;                                 // 4 per run of LEAF_RUN references of a big leaf: the run's tight box (same record as leafTB)
    std::vector<int> runBase;                              // per node: index of the leaf's first run in runTB / 4, or -1 (small leaf, interior, empty)
    std::vector<f4> scull;                                 // 4 per scene leaf reference (traverse.h SceneView::scull)
    std::vector<f4> leafTB;                                // 4 per node: the leaf's tight box (xrt_core.h leaf_certainly_missed)
    std::vector<g3> refG;
    // The wave-packet kernel's copies (packet.hip: everything it walks arrives through scalar loads from THREE arrays -- pblocks, lrec, refT):
    std::vector<float> pblocks;                            // PBLOCK_WORDS per block: descriptor (8 words of `blocks`) + the planes of its eight children (traverse.h PBLOCK_*)
    std::vector<float> lrec;                               // LREC_WORDS per node (leafNB, leafTB, first run), then RUN_WORDS per run (runTB): traverse.h LREC_*
    std::vector<int> childDfs, srefs, objMesh;
    std::vector<MeshRec> meshes;
    std::vector<ObjRec> objects;
    std::vector<MaterialRec> materials;
    std::vector<uint32_t> texels;
    int sceneDepth = 0, meshDepth = 0;
    int totalTris = 0;
    bool anyTransparent = false, anyTexture = false;
    size_t bytes() const;
};

struct HostScene {
    std::vector<HostMesh> meshes;
    std::vector<HostObject> objects;
    std::vector<FlatTree> meshTrees;
    FlatTree sceneTree;
    SceneArrays arrays;
    bool built = false;
#ifdef XRT_PK_BUNDLE
    bool buildTriTB = true;        // the per-reference tight boxes of the bundle prefilter (packet.hip, a build variant): 64 bytes per leaf reference
#else
    bool buildTriTB = false;
#endif
    bool spatialRuns = true;       // the references of a big leaf are stored in runs of neighbouring triangles (scene_host.cpp spatial_runs; XRT_LEAF_ORDER=0: in list order)
    double leafCullSafety = 1.0;   // factor on the tight-leaf-box margin (xrt_core.h LEAF_CULL_C): 0 switches the skip off, below 1 the bound is no longer proven (tests)
    double cullSafety = 2.0;   // factor S of the object pre-cull margin (scene_host.cpp); tests lower it to see the bound bite

    int add_mesh(const float *v, const float *n, const float *uv, const float *sn, const float *color, int ntri,
                 const xrt_material *m, const float bbox[6], std::string &err);
    int add_object(const int *meshIds, int n, const float *world, const float *invWorld, const float *bbox,
                   const float *worldBbox, std::string &err);
    bool build(int meshThreshold, int sceneThreshold, std::string &err);
    // Scene file (xrt_scene_save / xrt_scene_load): the meshes, materials, texels and bodies exactly as they were added --
    // what the reference keeps in .xnb files (Model.Tag, TMP:113-117) -- little-endian, no pointers.  The trees are rebuilt on load.
    bool save(const char *path, std::string &err) const;
    bool load(const char *path, std::string &err);
    SceneView host_view() const;   // pointers into `arrays` (CPU single-stepping in tests/emul only)
};

}  // namespace xrt
