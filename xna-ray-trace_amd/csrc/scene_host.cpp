// scene_host.cpp — see scene_host.h.
#include "scene_host.h"

#include <cmath>
#include <cstring>

namespace xrt {

size_t SceneArrays::bytes() const {
    return (blocks.size() + refN.size() + snodes.size() + shade.size() + leafNB.size()) * sizeof(f4) + refG.size() * sizeof(g3) +
           (childDfs.size() + srefs.size() + objMesh.size()) * sizeof(int) + meshes.size() * sizeof(MeshRec) +
           objects.size() * sizeof(ObjRec) + materials.size() * sizeof(MaterialRec) + texels.size() * sizeof(uint32_t);
}

int HostScene::add_mesh(const float *v, const float *n, const float *uv, const float *sn, const float *color, int ntri,
                        const xrt_material *m, const float bbox[6], std::string &err) {
    if (!v || !sn || !m || !bbox || ntri < 0) { err = "xrt_scene_add_mesh: null argument"; return -1; }
    if (ntri >= (1 << 28)) { err = "xrt_scene_add_mesh: too many triangles"; return -1; }
    HostMesh hm;
    hm.ntri = ntri;
    hm.v.assign(v, v + (size_t)ntri * 9);
    if (n) hm.n.assign(n, n + (size_t)ntri * 9); else hm.n.assign((size_t)ntri * 9, 0.0f);
    if (uv) hm.uv.assign(uv, uv + (size_t)ntri * 6); else hm.uv.assign((size_t)ntri * 6, 0.0f);
    hm.sn.assign(sn, sn + (size_t)ntri * 3);
    if (color) hm.color.assign(color, color + (size_t)ntri * 4); else hm.color.assign((size_t)ntri * 4, 1.0f);
    std::memcpy(hm.bbox, bbox, sizeof(hm.bbox));
    hm.reflectiveness = m->reflectiveness;
    hm.refractionIndex = m->refraction_index;
    hm.transparent = m->transparent != 0;
    hm.interpolateNormals = m->interpolate_normals != 0;
    hm.useTexture = m->use_texture != 0;
    if (hm.useTexture) {
        if (!m->tex_argb || m->tex_width <= 0 || m->tex_height <= 0) { err = "xrt_scene_add_mesh: UseTexture without texels (Bitmap.FromFile would throw, MAT:63)"; return -1; }
        hm.texW = m->tex_width; hm.texH = m->tex_height;
        hm.texels.assign(m->tex_argb, m->tex_argb + (size_t)m->tex_width * m->tex_height);
    }
    meshes.push_back(std::move(hm));
    built = false;
    return (int)meshes.size() - 1;
}

int HostScene::add_object(const int *meshIds, int n, const float *world, const float *invWorld, const float *bbox,
                          const float *worldBbox, std::string &err) {
    if (!meshIds || n < 0 || !world || !invWorld || !bbox || !worldBbox) { err = "xrt_scene_add_object: null argument"; return -1; }
    HostObject o;
    for (int i = 0; i < n; i++) {
        if (meshIds[i] < 0 || meshIds[i] >= (int)meshes.size()) { err = "xrt_scene_add_object: unknown mesh id"; return -1; }
        o.meshes.push_back(meshIds[i]);
    }
    std::memcpy(o.world, world, 64); std::memcpy(o.invWorld, invWorld, 64);
    std::memcpy(o.bbox, bbox, 24); std::memcpy(o.worldBbox, worldBbox, 24);
    objects.push_back(std::move(o));
    built = false;
    return (int)objects.size() - 1;
}

// World-space pre-cull box of one SceneObject.  The reference transforms the ray into object space with
// InverseWorld and tests each mesh AABB there (OSM:349-368).  The affine image of a mesh AABB under
// (InverseWorld)^-1 lies inside the axis-aligned box of its eight transformed corners, so a world ray that misses
// that box misses the AABB in object space in exact arithmetic; enlarging the box by 1e-3 of its size leaves
// three to four orders of magnitude over the binary32 rounding of the reference's own transform and slab test as
// long as the transform is well conditioned (checked; otherwise the object is never pre-culled).
static void object_cull_box(const HostObject &o, const std::vector<HostMesh> &meshes, ObjRec &r) {
    r.cullOk = 0;
    for (int a = 0; a < 4; a++) { r.cullMin[a] = 0; r.cullMax[a] = 0; }
    double A[3][3], t[3];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) A[i][j] = (double)o.invWorld[4 * i + j]; t[i] = (double)o.invWorld[12 + i]; }
    for (int i = 0; i < 16; i++) if (!(std::fabs((double)o.invWorld[i]) <= 1e30)) return;
    // row-vector convention: p' = p * A + t  =>  p = (p' - t) * A^-1
    double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                 A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    if (!(std::fabs(det) > 1e-30)) return;
    double B[3][3];
    B[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det; B[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det; B[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det;
    B[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det; B[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det; B[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
    B[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det; B[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det; B[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
    double na = 0, nb = 0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { na += A[i][j] * A[i][j]; nb += B[i][j] * B[i][j]; }
    if (!(std::sqrt(na) * std::sqrt(nb) <= 300.0)) return;   // Frobenius condition estimate (3 for a rotation)
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    if (o.meshes.empty()) return;
    for (int mi : o.meshes) {
        const float *bb = meshes[(size_t)mi].bbox;
        for (int c = 0; c < 8; c++) {
            double p[3] = {(double)bb[(c & 1) ? 3 : 0] - t[0], (double)bb[(c & 2) ? 4 : 1] - t[1], (double)bb[(c & 4) ? 5 : 2] - t[2]};
            for (int j = 0; j < 3; j++) {
                double w = p[0] * B[0][j] + p[1] * B[1][j] + p[2] * B[2][j];
                if (!(std::fabs(w) <= 1e30)) return;
                if (w < mn[j]) mn[j] = w;
                if (w > mx[j]) mx[j] = w;
            }
        }
    }
    double diag = 0, big = 0;
    for (int j = 0; j < 3; j++) { diag += (mx[j] - mn[j]) * (mx[j] - mn[j]); big = std::fmax(big, std::fmax(std::fabs(mn[j]), std::fabs(mx[j]))); }
    const double delta = 1e-3 * (std::sqrt(diag) + big) + 1e-30;
    for (int j = 0; j < 3; j++) { r.cullMin[j] = (float)(mn[j] - delta); r.cullMax[j] = (float)(mx[j] + delta); }
    // float conversion may round inwards by half an ulp: far inside the margin
    r.cullOk = 1;
}

bool HostScene::build(int meshThreshold, int sceneThreshold, std::string &err) {
    if (meshThreshold <= 0) meshThreshold = 50;    // MO:42
    if (sceneThreshold <= 0) sceneThreshold = 20;  // OSM:50
    built = false;
    arrays = SceneArrays();
    meshTrees.assign(meshes.size(), FlatTree());
    SceneArrays &A = arrays;
    int triBase = 0;
    for (size_t mi = 0; mi < meshes.size(); mi++) {
        const HostMesh &m = meshes[mi];
        FlatTree &t = meshTrees[mi];
        if (!build_mesh_tree(m, meshThreshold, t, err)) return false;   // Mesh.Init (MESH:27-32)
        const int blockBase = (int)(A.blocks.size() / 2);
        const int refBase = (int)A.refN.size();
        for (size_t bi = 0; bi < t.blocks.size() / 2; bi++) {   // local -> global indices
            f4 lo = t.blocks[2 * bi], hi = t.blocks[2 * bi + 1];
            lo.x = i2f(f2i(lo.x) + blockBase);
            lo.y = i2f(f2i(lo.y) + refBase);
            A.blocks.push_back(lo); A.blocks.push_back(hi);
        }
        A.childDfs.insert(A.childDfs.end(), t.childDfs.begin(), t.childDfs.end());
        // per leaf: bounds of its triangles' surface normals, so a leaf whose triangles all face away from a ray
        // (RE:48-51 would reject every one of them) can be skipped without reading its references
        for (size_t bi = 0; bi < t.blocks.size() / 2; bi++) {
            const f4 lo = t.blocks[2 * bi], hi = t.blocks[2 * bi + 1];
            const int lref = f2i(lo.y), masks = f2i(lo.z), total = f2i(lo.w);
            const int offw[4] = {f2i(hi.x), f2i(hi.y), f2i(hi.z), f2i(hi.w)};
            auto off = [&](int q) { int w = offw[q >> 1]; return (q & 1) ? (int)((unsigned)w >> 16) : (w & 0xffff); };
            for (int c = 0; c < 8; c++) {
                f4 mn{0, 0, 0, 0}, mx{0, 0, 0, 0};
                if (!((masks >> c) & 1)) {
                    const int b0 = lref + off(c), b1 = lref + (c == 7 ? total : off(c + 1));
                    for (int r = b0; r < b1; r++) {
                        const float *sn = &m.sn[(size_t)t.leafRefs[r] * 3];
                        if (r == b0) { mn = f4{sn[0], sn[1], sn[2], 0}; mx = mn; }
                        else {
                            mn.x = sn[0] < mn.x ? sn[0] : mn.x; mn.y = sn[1] < mn.y ? sn[1] : mn.y; mn.z = sn[2] < mn.z ? sn[2] : mn.z;
                            mx.x = sn[0] > mx.x ? sn[0] : mx.x; mx.y = sn[1] > mx.y ? sn[1] : mx.y; mx.z = sn[2] > mx.z ? sn[2] : mx.z;
                        }
                        if (!(sn[0] == sn[0] && sn[1] == sn[1] && sn[2] == sn[2])) { mn.w = 1.0f; }   // a NaN normal: never skip this leaf
                    }
                }
                A.leafNB.push_back(mn); A.leafNB.push_back(mx);
            }
        }
        for (int tri : t.leafRefs) {   // leaf references in leaf order: normal stream + geometry stream
            const float *p = &m.v[(size_t)tri * 9];
            const float *sn = &m.sn[(size_t)tri * 3];
            A.refN.push_back(f4{sn[0], sn[1], sn[2], i2f(triBase + tri)});
            A.refG.push_back(g3{p[0], p[1], p[2]});
            A.refG.push_back(g3{p[3] - p[0], p[4] - p[1], p[5] - p[2]});   // Edge1 = v2 - v1 (RE:54)
            A.refG.push_back(g3{p[6] - p[0], p[7] - p[1], p[8] - p[2]});   // Edge2 = v3 - v1 (RE:55)
        }
        MaterialRec mat;
        std::memset(&mat, 0, sizeof(mat));
        mat.reflectiveness = m.reflectiveness;
        mat.refractionIndex = m.refractionIndex;
        mat.flags = (m.transparent ? MAT_TRANSPARENT : 0) | (m.interpolateNormals ? MAT_INTERP : 0) | (m.useTexture ? MAT_TEXTURE : 0);
        mat.texOffset = (int)A.texels.size();
        mat.texWidth = m.texW; mat.texHeight = m.texH;
        A.texels.insert(A.texels.end(), m.texels.begin(), m.texels.end());
        A.materials.push_back(mat);
        A.anyTransparent = A.anyTransparent || m.transparent;
        A.anyTexture = A.anyTexture || m.useTexture;
        for (int i = 0; i < m.ntri; i++) {
            const float *n = &m.n[(size_t)i * 9], *uv = &m.uv[(size_t)i * 6], *c = &m.color[(size_t)i * 4], *sn = &m.sn[(size_t)i * 3];
            A.shade.push_back(f4{n[0], n[1], n[2], uv[0]});
            A.shade.push_back(f4{n[3], n[4], n[5], uv[1]});
            A.shade.push_back(f4{n[6], n[7], n[8], uv[2]});
            A.shade.push_back(f4{c[0], c[1], c[2], c[3]});
            A.shade.push_back(f4{uv[3], uv[4], uv[5], i2f((int)mi)});
            A.shade.push_back(f4{sn[0], sn[1], sn[2], i2f((int)mi)});
        }
        MeshRec mr;
        std::memset(&mr, 0, sizeof(mr));
        for (int a = 0; a < 3; a++) { mr.bmin[a] = m.bbox[a]; mr.bmax[a] = m.bbox[3 + a]; }
        for (int a = 0; a < 3; a++) { mr.rmin[a] = t.rootBox[a]; mr.rmax[a] = t.rootBox[3 + a]; }
        mr.rootBlock = t.rootIsLeaf ? -1 : blockBase;
        mr.rootRef = refBase; mr.rootCount = t.rootIsLeaf ? t.rootCount : 0;
        mr.triBase = triBase; mr.ntri = m.ntri; mr.material = (int)mi; mr.maxDepth = t.maxDepth; mr.dfsBase = blockBase * 8;
        A.meshes.push_back(mr);
        if (t.maxDepth > A.meshDepth) A.meshDepth = t.maxDepth;
        triBase += m.ntri;
    }
    A.totalTris = triBase;
    if (!build_scene_tree(objects, sceneThreshold, sceneTree, err)) return false;   // OSM:64-99
    A.snodes = sceneTree.nodes;
    A.srefs = sceneTree.leafRefs;
    A.sceneDepth = sceneTree.maxDepth;
    for (const HostObject &o : objects) {
        ObjRec r;
        std::memset(&r, 0, sizeof(r));
        std::memcpy(r.invWorld, o.invWorld, 64); std::memcpy(r.world, o.world, 64);
        r.meshStart = (int)A.objMesh.size(); r.meshCount = (int)o.meshes.size();
        object_cull_box(o, meshes, r);
        A.objMesh.insert(A.objMesh.end(), o.meshes.begin(), o.meshes.end());
        A.objects.push_back(r);
    }
    // never hand out empty arrays (a zero-size allocation has no address)
    if (A.refN.empty()) A.refN.assign(1, f4{0, 0, 0, i2f(-1)});
    if (A.refG.empty()) A.refG.assign(3, g3{0, 0, 0});
    if (A.srefs.empty()) A.srefs.assign(1, -1);
    if (A.objMesh.empty()) A.objMesh.assign(1, -1);
    if (A.shade.empty()) A.shade.assign(SHADE_F4, f4{0, 0, 0, 0});
    if (A.texels.empty()) A.texels.assign(1, 0u);
    if (A.meshes.empty()) { MeshRec z; std::memset(&z, 0, sizeof(z)); A.meshes.push_back(z); }
    if (A.objects.empty()) { ObjRec z; std::memset(&z, 0, sizeof(z)); A.objects.push_back(z); }
    if (A.materials.empty()) { MaterialRec z; std::memset(&z, 0, sizeof(z)); A.materials.push_back(z); }
    if (A.blocks.empty()) A.blocks.assign(2, f4{0, 0, 0, 0});
    if (A.leafNB.empty()) A.leafNB.assign(16, f4{0, 0, 0, 0});
    if (A.childDfs.empty()) A.childDfs.assign(8, -1);
    built = true;
    return true;
}

SceneView HostScene::host_view() const {
    SceneView S;
    const SceneArrays &A = arrays;
    S.blocks = A.blocks.data(); S.childDfs = A.childDfs.data(); S.leafNB = A.leafNB.data();
    S.refN = A.refN.data(); S.refG = A.refG.data(); S.meshes = A.meshes.data();
    S.snodes = A.snodes.data(); S.srefs = A.srefs.data(); S.objects = A.objects.data(); S.objMesh = A.objMesh.data();
    S.nMeshes = (int)meshes.size(); S.nObjects = (int)objects.size();
    S.sceneDepth = A.sceneDepth + 1; S.meshDepth = A.meshDepth + 1;
    return S;
}

}  // namespace xrt
