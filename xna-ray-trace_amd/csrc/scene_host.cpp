// scene_host.cpp — see scene_host.h.
#include "scene_host.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace xrt {

size_t SceneArrays::bytes() const {
    return (blocks.size() + refN.size() + snodes.size() + shade.size() + leafNB.size() + leafTB.size() + scull.size() + runTB.size()) * sizeof(f4) + (refT.size() + pblocks.size() + lrec.size()) * sizeof(float) + refG.size() * sizeof(g3) +
           (childDfs.size() + srefs.size() + objMesh.size() + runBase.size()) * sizeof(int) + meshes.size() * sizeof(MeshRec) +
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
        if (m->tex_pargb) hm.texelsP.assign(m->tex_pargb, m->tex_pargb + (size_t)m->tex_width * m->tex_height);
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

// Largest eigenvalue of the symmetric 3x3 matrix M (closed form, double): the square of a 2-norm.
static double sym3_max_eig(const double M[3][3]) {
    const double p1 = M[0][1] * M[0][1] + M[0][2] * M[0][2] + M[1][2] * M[1][2];
    if (p1 == 0.0) return std::fmax(M[0][0], std::fmax(M[1][1], M[2][2]));
    const double q = (M[0][0] + M[1][1] + M[2][2]) / 3.0;
    const double p2 = (M[0][0] - q) * (M[0][0] - q) + (M[1][1] - q) * (M[1][1] - q) + (M[2][2] - q) * (M[2][2] - q) + 2.0 * p1;
    const double p = std::sqrt(p2 / 6.0);
    double Bm[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Bm[i][j] = (M[i][j] - (i == j ? q : 0.0)) / p;
    double r = (Bm[0][0] * (Bm[1][1] * Bm[2][2] - Bm[1][2] * Bm[2][1]) - Bm[0][1] * (Bm[1][0] * Bm[2][2] - Bm[1][2] * Bm[2][0]) +
                Bm[0][2] * (Bm[1][0] * Bm[2][1] - Bm[1][1] * Bm[2][0])) / 2.0;
    r = r < -1.0 ? -1.0 : (r > 1.0 ? 1.0 : r);
    return q + 2.0 * p * std::cos(std::acos(r) / 3.0);
}
static double norm2_3x3(const double A[3][3]) {   // ||A||_2 = sqrt(lambda_max(A^T A)), rounded up
    double M[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[i][j] = A[0][i] * A[0][j] + A[1][i] * A[1][j] + A[2][i] * A[2][j];
    const double l = sym3_max_eig(M);
    return std::sqrt(l > 0.0 ? l : 0.0) * (1.0 + 1e-9);
}

// World-space pre-cull record of one SceneObject (ObjRec::cullMin/cullMax/cullK2; DESIGN.md §3 has the derivation).
//
// The reference transforms the world ray (o, d) into object space in binary32 (OSM:358-364):
//     q = fl(o + d);  v1 = fl(o*A + t);  v2 = fl(q*A + t);  w = fl(v2 - v1);  dir = fl(normalize(w))
// (A, t = linear part and translation row of InverseWorld) and tests every mesh AABB with BoundingBox.Intersects
// (MESH:34-39).  With u = 2^-24, r = |o|, a = ||A||_2, alpha = ||A||_F, beta = ||A^-1||_2, tau = |t|, B = the largest
// corner norm of the object's mesh AABBs and V = a r + tau + B (>= |v1| + B, the reach of the object-space ray up to
// the box):
//   * |v1 - (oA + t)|        <= 4u (alpha r + tau)                                   (three products, three sums per component)
//   * |w' - dA|              <= 10u (alpha (r + 2) + tau), w' = the direction actually used (normalisation is a common
//                               factor; its last multiplication perturbs each component by u)
//   * a hit of the binary32 slab test means the point of the ray (v1, dir) at parameter tmin >= 0 lies inside the AABB
//     enlarged by 3.1u (|v1_k| + |b_k|) per axis, or, on an axis taken as parallel (|dir_k| < 1e-6), by 1e-6 * tmin:
//     together <= 29.1u V
//   so the exact image of the world ray, oA + t + s dA (s >= 0), passes within
//       D = 4u (alpha r + tau) + beta V 10u (alpha (r + 2) + tau) + 29.1u V
//   of the AABB, and the world ray itself within beta * D of the AABB's exact world image, which lies inside the
//   axis-aligned hull of its eight mapped corners (cullMin/cullMax, evaluated in double).
//   * the world-space test is binary32 too: 3.1u (|o_k| + |c_k|) per axis, 1e-6 * reach on a parallel axis: <= 23u (r + 2C),
//     C = the hull's largest coordinate.
// m(r) = S * [beta * D + 23u (r + 2C)] is a quadratic in r; S = 2 pays for rounding m's own evaluation, the 1-norm the
// kernel uses for r and the float conversion of the coefficients.  A ray that misses the hull enlarged by m(r) cannot be
// accepted by MESH:34-39 for any mesh of the object.
static void object_cull_box(const HostObject &o, const std::vector<HostMesh> &meshes, ObjRec &r, double safety) {
    r.cullOk = 0;
    r.cullK2 = 0;
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
    double fro = 0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) fro += A[i][j] * A[i][j];
    const double alpha = std::sqrt(fro) * (1.0 + 1e-9), a2 = norm2_3x3(A), beta = norm2_3x3(B);
    const double tau = std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    if (!(alpha <= 1e15 && beta <= 1e15 && tau <= 1e15)) return;
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300}, Bn = 0;
    if (o.meshes.empty()) return;
    for (int mi : o.meshes) {
        const float *bb = meshes[(size_t)mi].bbox;
        for (int c = 0; c < 8; c++) {
            const double q[3] = {(double)bb[(c & 1) ? 3 : 0], (double)bb[(c & 2) ? 4 : 1], (double)bb[(c & 4) ? 5 : 2]};
            Bn = std::fmax(Bn, std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]));
            const double p[3] = {q[0] - t[0], q[1] - t[1], q[2] - t[2]};
            for (int j = 0; j < 3; j++) {
                double w = p[0] * B[0][j] + p[1] * B[1][j] + p[2] * B[2][j];
                if (!(std::fabs(w) <= 1e30)) return;
                if (w < mn[j]) mn[j] = w;
                if (w > mx[j]) mx[j] = w;
            }
        }
    }
    double Cmax = 0;
    for (int j = 0; j < 3; j++) Cmax = std::fmax(Cmax, std::fmax(std::fabs(mn[j]), std::fabs(mx[j])));
    const double u = std::ldexp(1.0, -24), S = safety;
    const double k2 = S * u * 10.0 * beta * beta * a2 * alpha;
    const double k1 = S * u * (4.0 * beta * alpha + 10.0 * beta * beta * (a2 * (2.0 * alpha + tau) + alpha * (tau + Bn)) + 29.1 * beta * a2 + 23.0);
    const double k0 = S * u * (4.0 * beta * tau + 10.0 * beta * beta * (tau + Bn) * (2.0 * alpha + tau) + 29.1 * beta * (tau + Bn) + 46.0 * Cmax);
    if (!(k0 <= 1e30 && k1 <= 1e30 && k2 <= 1e30)) return;
    const double up = 1.0 + 1e-6;   // the float conversions below may round down by half an ulp
    for (int j = 0; j < 3; j++) { r.cullMin[j] = (float)(mn[j] - 1e-6 * std::fabs(mn[j])); r.cullMax[j] = (float)(mx[j] + 1e-6 * std::fabs(mx[j])); }
    r.cullMin[3] = (float)(k0 * up) + 1e-30f; r.cullMax[3] = (float)(k1 * up); r.cullK2 = (float)(k2 * up);
    r.cullOk = 1;
}

// Tight-box record of the triangles t.leafRefs[b0 .. b1) of mesh m (xrt_core.h leaf_certainly_missed; DESIGN.md §3 "Tight leaf boxes"):
// [0] = (vertex box min, K = C u0 max_t |E1||E2|/|E1 x E2| * safety), [1] = (vertex box max, longest edge), [2] = (min of the unit geometric
// normals, ok), [3] = (their max, -), all rounded outwards.  The bound behind it holds for ANY set of triangles -- a leaf, or a run of
// consecutive references of a leaf.  rec stays zero (ok = 0: never skipped) when a triangle is outside the magnitudes the analysis assumes.
static void tight_box_record(const HostMesh &m, const std::vector<int> &leafRefs, int b0, int b1, double safety, f4 rec[4]) {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, nlo[3] = {2, 2, 2}, nhi[3] = {-2, -2, -2};
    double aMax = 0, eMax = 0;
    bool ok = b1 > b0;
    for (int r = b0; r < b1 && ok; r++) {
        const float *p = &m.v[(size_t)leafRefs[r] * 9];
        const float e1f[3] = {p[3] - p[0], p[4] - p[1], p[5] - p[2]}, e2f[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};   // RE:54-55, as stored in refG
        const double e1[3] = {e1f[0], e1f[1], e1f[2]}, e2[3] = {e2f[0], e2f[1], e2f[2]};
        const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        const double l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
        const double ln = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        // magnitudes the error analysis assumes (no overflow, no subnormal products)
        if (!(l1 >= 1e-9 && l2 >= 1e-9 && l1 <= 1e12 && l2 <= 1e12 && ln > 0 && l1 * l2 / ln <= 1e6)) { ok = false; break; }
        aMax = std::fmax(aMax, l1 * l2 / ln);
        eMax = std::fmax(eMax, std::fmax(l1, l2));
        for (int k = 0; k < 3; k++) {
            const double v[3] = {(double)p[k], (double)p[k] + e1[k], (double)p[k] + e2[k]};
            for (double x : v) { if (!(std::fabs(x) <= 1e12)) ok = false; lo[k] = std::fmin(lo[k], x); hi[k] = std::fmax(hi[k], x); }
            nlo[k] = std::fmin(nlo[k], n[k] / ln); nhi[k] = std::fmax(nhi[k], n[k] / ln);
        }
    }
    if (ok && safety > 0.0) {
        auto down = [](double x) { return std::nextafterf((float)x, -INFINITY); };
        auto up = [](double x) { return std::nextafterf((float)x, INFINITY); };
        rec[0] = f4{down(lo[0]), down(lo[1]), down(lo[2]), up((double)LEAF_CULL_C * std::ldexp(1.0, -24) * aMax * safety)};
        rec[1] = f4{up(hi[0]), up(hi[1]), up(hi[2]), up(eMax)};
        rec[2] = f4{down(nlo[0] - 1e-6), down(nlo[1] - 1e-6), down(nlo[2] - 1e-6), 1.0f};
        rec[3] = f4{up(nhi[0] + 1e-6), up(nhi[1] + 1e-6), up(nhi[2] + 1e-6), 0.0f};
    }
}

// Storage order of a big leaf's references.  The walk tests a leaf of >= LEAF_RUN_MIN references run by run (LEAF_RUN consecutive references, each run with a tight box of
// its own), and a run of consecutive LIST entries is a strip -- the builder hands triangles down in index order (MO:225-233), a heightfield's leaf holds them row by row --
// that a ray crossing the leaf touches whatever its direction.  Here the references of such a leaf are stored in runs of NEIGHBOURS instead: the set is cut at the middle of
// its longest extent (centroids; the cut at a multiple of LEAF_RUN) until a part fits one run.  Order matters to the reference in one place only -- of two triangles of a leaf
// hit at exactly the same distance the EARLIER in the list wins (MO:293-294, strict '<') -- and the list of a leaf is ascending in the triangle index (checked here; a leaf
// that is not keeps its order), so the kernels settle such a tie by the smaller index (traverse.h leaf_candidate, packet.hip pk_candidate) and the answers do not depend on
// the storage order.  xrt_scene_get_tree still shows the reference's order.
// (a leaf keeps its list order where that already gives the smaller runs: the sum of the surface areas of the runs' vertex boxes decides -- the twelve triangles of a
// crate's face pairs lie better in the order they were modelled in)
static double run_boxes_area(const HostMesh &m, const std::vector<int> &lr, int b0, int b1) {
    double sum = 0;
    for (int ra = b0; ra < b1; ra += LEAF_RUN) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int r = ra; r < ra + LEAF_RUN && r < b1; r++) {
            const float *p = &m.v[(size_t)lr[(size_t)r] * 9];
            for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) { const double x = p[3 * j + k]; if (x == x) { lo[k] = std::fmin(lo[k], x); hi[k] = std::fmax(hi[k], x); } }
        }
        const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx >= 0 && dy >= 0 && dz >= 0) sum += dx * dy + dy * dz + dz * dx;
    }
    return sum;
}
static void spatial_runs_cut(const HostMesh &m, std::vector<int> &lr, int b0, int b1);
static void spatial_runs(const HostMesh &m, std::vector<int> &lr, int b0, int b1) {
    for (int r = b0 + 1; r < b1; r++) if (lr[(size_t)r] <= lr[(size_t)r - 1]) return;
    const std::vector<int> before(lr.begin() + b0, lr.begin() + b1);
    const double a0 = run_boxes_area(m, lr, b0, b1);
    spatial_runs_cut(m, lr, b0, b1);
    if (!(run_boxes_area(m, lr, b0, b1) < a0)) std::copy(before.begin(), before.end(), lr.begin() + b0);
}
static void spatial_runs_cut(const HostMesh &m, std::vector<int> &lr, int b0, int b1) {
    struct Part { int a, b; };
    std::vector<Part> todo{Part{b0, b1}};
    std::vector<std::pair<float, int>> key;
    while (!todo.empty()) {
        const Part q = todo.back();
        todo.pop_back();
        if (q.b - q.a <= LEAF_RUN) { std::sort(lr.begin() + q.a, lr.begin() + q.b); continue; }
        float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        auto centroid = [&](int tri, int k) { const float *p = &m.v[(size_t)tri * 9]; return (p[k] + p[3 + k]) + p[6 + k]; };
        for (int r = q.a; r < q.b; r++)
            for (int k = 0; k < 3; k++) { const float c = centroid(lr[(size_t)r], k); if (c == c) { lo[k] = std::fmin(lo[k], c); hi[k] = std::fmax(hi[k], c); } }
        int ax = 0;
        for (int k = 1; k < 3; k++) if (hi[k] - lo[k] > hi[ax] - lo[ax]) ax = k;
        key.clear();
        for (int r = q.a; r < q.b; r++) { const float c = centroid(lr[(size_t)r], ax); key.push_back({c == c ? c : FLT_MAX, lr[(size_t)r]}); }
        std::sort(key.begin(), key.end());   // (ties by triangle index: deterministic)
        for (int r = q.a; r < q.b; r++) lr[(size_t)r] = key[(size_t)(r - q.a)].second;
        const int n = q.b - q.a, runs = (n + LEAF_RUN - 1) / LEAF_RUN, left = (runs / 2) * LEAF_RUN;   // (runs >= 2 here)
        todo.push_back(Part{q.a, q.a + left});
        todo.push_back(Part{q.a + left, q.b});
    }
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
        std::vector<int> lr = t.leafRefs;   // the references in STORAGE order (spatial_runs)
        for (size_t bi = 0; bi < t.blocks.size() / 2; bi++) {   // what the kernels' tie rule relies on: every leaf's list is ascending in the triangle index (MO:225-233 hands lists down in order)
            const f4 lo = t.blocks[2 * bi], hi = t.blocks[2 * bi + 1];
            const int lref = f2i(lo.y), masks = f2i(lo.z), total = f2i(lo.w);
            const int offw[4] = {f2i(hi.x), f2i(hi.y), f2i(hi.z), f2i(hi.w)};
            auto off = [&](int q) { int w = offw[q >> 1]; return (q & 1) ? (int)((unsigned)w >> 16) : (w & 0xffff); };
            for (int c = 0; c < 8; c++) {
                if ((masks >> c) & 1) continue;
                const int b0 = lref + off(c), b1 = lref + (c == 7 ? total : off(c + 1));
                for (int r = b0 + 1; r < b1; r++) if (lr[(size_t)r] <= lr[(size_t)r - 1]) { err = "internal: a leaf's triangle list is not ascending"; return false; }
            }
        }
        if (t.rootIsLeaf) for (int r = 1; r < t.rootCount; r++) if (lr[(size_t)r] <= lr[(size_t)r - 1]) { err = "internal: a leaf's triangle list is not ascending"; return false; }
        if (spatialRuns)
            for (size_t bi = 0; bi < t.blocks.size() / 2; bi++) {
                const f4 lo = t.blocks[2 * bi], hi = t.blocks[2 * bi + 1];
                const int lref = f2i(lo.y), masks = f2i(lo.z), total = f2i(lo.w);
                const int offw[4] = {f2i(hi.x), f2i(hi.y), f2i(hi.z), f2i(hi.w)};
                auto off = [&](int q) { int w = offw[q >> 1]; return (q & 1) ? (int)((unsigned)w >> 16) : (w & 0xffff); };
                for (int c = 0; c < 8; c++) {
                    if ((masks >> c) & 1) continue;
                    const int b0 = lref + off(c), b1 = lref + (c == 7 ? total : off(c + 1));
                    if (b1 - b0 >= LEAF_RUN_MIN) spatial_runs(m, lr, b0, b1);
                }
            }
        const int blockBase = (int)(A.blocks.size() / 2);
        const int refBase = (int)A.refN.size();
        for (size_t bi = 0; bi < t.blocks.size() / 2; bi++) {   // local -> global indices
            f4 lo = t.blocks[2 * bi], hi = t.blocks[2 * bi + 1];
            lo.x = i2f(f2i(lo.x) + blockBase);
            lo.y = i2f(f2i(lo.y) + refBase);
            A.blocks.push_back(lo); A.blocks.push_back(hi);
        }
        A.childDfs.insert(A.childDfs.end(), t.childDfs.begin(), t.childDfs.end());
        // the packet kernel's block records: descriptor + child planes (traverse.h PBLOCK_*), boxes derived from the root box exactly as the
        // builder and the per-lane kernel derive them (child_box / half_of: the same inline functions, compiled without contraction)
        A.pblocks.resize((size_t)(blockBase + t.blocks.size() / 2) * PBLOCK_WORDS, 0.0f);
        if (!t.blocks.empty()) {
            struct Todo { int lb; v3 bmin, half; };
            std::vector<Todo> todo;
            const v3 rmn = mk(t.rootBox[0], t.rootBox[1], t.rootBox[2]);
            todo.push_back(Todo{0, rmn, half_of(rmn, mk(t.rootBox[3], t.rootBox[4], t.rootBox[5]))});   // (local block 0 = the root's children)
            while (!todo.empty()) {
                const Todo w = todo.back();
                todo.pop_back();
                const f4 lo = A.blocks[2 * (size_t)(blockBase + w.lb)], hi = A.blocks[2 * (size_t)(blockBase + w.lb) + 1];
                float *q = &A.pblocks[(size_t)(blockBase + w.lb) * PBLOCK_WORDS];
                q[0] = lo.x; q[1] = lo.y; q[2] = lo.z; q[3] = lo.w; q[4] = hi.x; q[5] = hi.y; q[6] = hi.z; q[7] = hi.w;
                const float bm[3] = {w.bmin.x, w.bmin.y, w.bmin.z}, hf[3] = {w.half.x, w.half.y, w.half.z};
                for (int ax = 0; ax < 3; ax++) {
                    const float p0 = bm[ax] + hf[ax] * 0.0f, p1 = bm[ax] + hf[ax] * 1.0f, p2 = p1 + hf[ax], hp0 = p0 + hf[ax];
                    q[8 + 4 * ax] = p0; q[9 + 4 * ax] = p1; q[10 + 4 * ax] = p2; q[11 + 4 * ax] = hp0;
                }
                const int childBase = f2i(t.blocks[2 * (size_t)w.lb].x), interior = f2i(lo.z) & 0xff;
                for (int c = 0; c < 8; c++)
                    if ((interior >> c) & 1) {
                        v3 cmin, cmax;
                        child_box(w.bmin, w.half, c, cmin, cmax);
                        todo.push_back(Todo{childBase + __builtin_popcount((unsigned)interior & ((1u << c) - 1u)), cmin, half_of(cmin, cmax)});
                    }
            }
        }
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
                        const float *sn = &m.sn[(size_t)lr[(size_t)r] * 3];
                        if (r == b0) { mn = f4{sn[0], sn[1], sn[2], 0}; mx = mn; }
                        else {
                            mn.x = sn[0] < mn.x ? sn[0] : mn.x; mn.y = sn[1] < mn.y ? sn[1] : mn.y; mn.z = sn[2] < mn.z ? sn[2] : mn.z;
                            mx.x = sn[0] > mx.x ? sn[0] : mx.x; mx.y = sn[1] > mx.y ? sn[1] : mx.y; mx.z = sn[2] > mx.z ? sn[2] : mx.z;
                        }
                        if (!(sn[0] == sn[0] && sn[1] == sn[1] && sn[2] == sn[2])) { mn.w = 1.0f; }   // a NaN normal: never skip this leaf
                    }
                }
                A.leafNB.push_back(mn); A.leafNB.push_back(mx);
                // ... and the tight box of the leaf (xrt_core.h leaf_certainly_missed): vertex box, box of the unit geometric normals,
                // K = C u0 max |E1||E2|/|E1 x E2| and the longest edge, all rounded outwards
                f4 rec[4] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
                int runFirst = -1;
                if (!((masks >> c) & 1)) {
                    const int b0 = lref + off(c), b1 = lref + (c == 7 ? total : off(c + 1));
                    tight_box_record(m, lr, b0, b1, leafCullSafety, rec);
                    // ... and the same record for every run of LEAF_RUN consecutive references of a leaf of at least LEAF_RUN_MIN (the
                    // bound holds for any set of triangles): a ray that reaches the leaf's box still tests only the runs it can reach
                    if (b1 - b0 >= LEAF_RUN_MIN) {
                        runFirst = (int)(A.runTB.size() / 4);
                        for (int ra = b0; ra < b1; ra += LEAF_RUN) {
                            f4 rr[4] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
                            tight_box_record(m, lr, ra, ra + LEAF_RUN < b1 ? ra + LEAF_RUN : b1, leafCullSafety, rr);
                            for (const f4 &q : rr) A.runTB.push_back(q);
                        }
                    }
                }
                for (const f4 &q : rec) A.leafTB.push_back(q);
                A.runBase.push_back(runFirst);
            }
        }
        // ... and for every INTERIOR node the box of all the surface normals below it (the union of its children's boxes, bottom-up: a child
        // block has a higher index than its parent's), so that a whole subtree RE:48-51 would reject triangle by triangle -- the terrain
        // under a ray that leaves it, the far side of a crate -- is never entered (traverse.h all_back_facing: the same test, the same margin).
        f4 meshNBlo{0, 0, 0, 0}, meshNBhi{0, 0, 0, 0};
        {
            bool meshAny = false;
            const size_t nb = t.blocks.size() / 2;
            for (size_t bi = nb; bi-- > 0;) {
                const int masks = f2i(t.blocks[2 * bi].z), childBase = f2i(t.blocks[2 * bi].x);
                for (int c = 0; c < 8; c++) {
                    if (!((masks >> c) & 1)) continue;   // (a leaf: done above)
                    const size_t cb = (size_t)childBase + (size_t)__builtin_popcount((unsigned)(masks & 0xff) & ((1u << c) - 1u));
                    const int cmasks = f2i(t.blocks[2 * cb].z);
                    f4 mn{0, 0, 0, 0}, mx{0, 0, 0, 0};
                    bool any = false;
                    for (int k = 0; k < 8; k++) {
                        if ((cmasks >> (8 + k)) & 1) continue;   // empty leaf
                        const f4 a = A.leafNB[2 * ((size_t)(blockBase + cb) * 8 + k)], b2 = A.leafNB[2 * ((size_t)(blockBase + cb) * 8 + k) + 1];
                        if (!any) { mn = a; mx = b2; any = true; }
                        else {
                            mn.x = a.x < mn.x ? a.x : mn.x; mn.y = a.y < mn.y ? a.y : mn.y; mn.z = a.z < mn.z ? a.z : mn.z;
                            mx.x = b2.x > mx.x ? b2.x : mx.x; mx.y = b2.y > mx.y ? b2.y : mx.y; mx.z = b2.z > mx.z ? b2.z : mx.z;
                            if (a.w != 0.0f) mn.w = 1.0f;
                        }
                    }
                    if (!any) mn.w = 1.0f;   // (cannot happen: an interior node holds triangles)
                    A.leafNB[2 * ((size_t)(blockBase + bi) * 8 + c)] = mn; A.leafNB[2 * ((size_t)(blockBase + bi) * 8 + c) + 1] = mx;
                    // all_back_facing needs sum_k min(nmin_k d_k, nmax_k d_k) > 0; an axis whose interval holds 0 contributes <= 0 for every d_k, so a box
                    // that holds the origin can never pass: only the other interior children are worth the test (descriptor bits 24..31)
                    const bool canPass = mn.w == 0.0f && (mn.x > 0.0f || mx.x < 0.0f || mn.y > 0.0f || mx.y < 0.0f || mn.z > 0.0f || mx.z < 0.0f);
                    if (canPass) {
                        f4 &lo = A.blocks[2 * (size_t)(blockBase + bi)];
                        lo.z = i2f(f2i(lo.z) | (1 << (24 + c)));
                        A.pblocks[(size_t)(blockBase + bi) * PBLOCK_WORDS + 2] = lo.z;
                    }
                }
            }
            if (nb) {
                const int masks = f2i(t.blocks[0].z);
                for (int k = 0; k < 8; k++) {
                    if ((masks >> (8 + k)) & 1) continue;
                    const f4 a = A.leafNB[2 * ((size_t)blockBase * 8 + k)], b2 = A.leafNB[2 * ((size_t)blockBase * 8 + k) + 1];
                    if (!meshAny) { meshNBlo = a; meshNBhi = b2; meshAny = true; }
                    else {
                        meshNBlo.x = a.x < meshNBlo.x ? a.x : meshNBlo.x; meshNBlo.y = a.y < meshNBlo.y ? a.y : meshNBlo.y; meshNBlo.z = a.z < meshNBlo.z ? a.z : meshNBlo.z;
                        meshNBhi.x = b2.x > meshNBhi.x ? b2.x : meshNBhi.x; meshNBhi.y = b2.y > meshNBhi.y ? b2.y : meshNBhi.y; meshNBhi.z = b2.z > meshNBhi.z ? b2.z : meshNBhi.z;
                        if (a.w != 0.0f) meshNBlo.w = 1.0f;
                    }
                }
            }
            if (!meshAny) meshNBlo.w = 1.0f;   // (a mesh whose root is a leaf, or without triangles: no record, never culled)
        }
        for (size_t li = 0; buildTriTB && li < lr.size(); li++) {   // ... and the tight box of every single reference (the bound holds for any set of triangles, also for one)
            f4 rr[4] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
            tight_box_record(m, lr, (int)li, (int)li + 1, leafCullSafety, rr);
            for (const f4 &q : rr) A.triTB.push_back(q);
        }
        for (int tri : lr) {   // leaf references in storage order: normal stream + geometry stream
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
        mat.texOffsetP = mat.texOffset;
        if (!m.texelsP.empty()) { mat.texOffsetP = (int)A.texels.size(); A.texels.insert(A.texels.end(), m.texelsP.begin(), m.texelsP.end()); }
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
        mr.nbMin[0] = meshNBlo.x; mr.nbMin[1] = meshNBlo.y; mr.nbMin[2] = meshNBlo.z; mr.nbMin[3] = meshNBlo.w;
        mr.nbMax[0] = meshNBhi.x; mr.nbMax[1] = meshNBhi.y; mr.nbMax[2] = meshNBhi.z; mr.nbMax[3] = meshNBhi.w;
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
        object_cull_box(o, meshes, r, cullSafety);
        A.objMesh.insert(A.objMesh.end(), o.meshes.begin(), o.meshes.end());
        A.objects.push_back(r);
    }
    for (int o : A.srefs) {   // the pre-cull records in scene-leaf order (one 64-byte scalar load per body for the packet kernel)
        const ObjRec &r = A.objects[(size_t)o];
        A.scull.push_back(f4{r.cullMin[0], r.cullMin[1], r.cullMin[2], r.cullMin[3]});
        A.scull.push_back(f4{r.cullMax[0], r.cullMax[1], r.cullMax[2], r.cullMax[3]});
        A.scull.push_back(f4{r.cullK2, i2f(r.cullOk), i2f(o), i2f(r.meshStart)});
        A.scull.push_back(f4{i2f(r.meshCount), 0, 0, 0});
    }
    for (int k = 0; k < 4; k++) A.scull.push_back(f4{0, 0, 0, 0});   // one record of padding: the packet kernel requests the next record a body ahead
    // never hand out empty arrays (a zero-size allocation has no address)
    // two dummy references at the end: the wave-packet kernel requests the next triangle's record before it knows the leaf has ended
    for (int k = 0; k < 2; k++) { A.refN.push_back(f4{0, 0, 0, i2f(-1)}); for (int j = 0; j < 3; j++) A.refG.push_back(g3{0, 0, 0}); }
    if (buildTriTB) A.triTB.resize(A.refN.size() * 4 + 64 * 4, f4{0, 0, 0, 0});
    else A.triTB.assign(4, f4{0, 0, 0, 0});   // (the dummy references, and a wave's worth of padding: 64 lanes read 64 consecutive records)
    A.refT.resize(A.refN.size() * TRI_REC_WORDS + 16, 0.0f);
    for (size_t r = 0; r < A.refN.size(); r++) {
        float *q = &A.refT[r * TRI_REC_WORDS];
        q[0] = A.refN[r].x; q[1] = A.refN[r].y; q[2] = A.refN[r].z; q[3] = A.refN[r].w;
        for (int j = 0; j < 3; j++) { q[4 + 3 * j] = A.refG[3 * r + j].x; q[5 + 3 * j] = A.refG[3 * r + j].y; q[6 + 3 * j] = A.refG[3 * r + j].z; }
    }
    if (A.srefs.empty()) A.srefs.assign(1, -1);
    if (A.objMesh.empty()) A.objMesh.assign(1, -1);
    if (A.shade.empty()) A.shade.assign(SHADE_F4, f4{0, 0, 0, 0});
    if (A.texels.empty()) A.texels.assign(1, 0u);
    if (A.meshes.empty()) { MeshRec z; std::memset(&z, 0, sizeof(z)); A.meshes.push_back(z); }
    if (A.objects.empty()) { ObjRec z; std::memset(&z, 0, sizeof(z)); A.objects.push_back(z); }
    if (A.materials.empty()) { MaterialRec z; std::memset(&z, 0, sizeof(z)); A.materials.push_back(z); }
    if (A.blocks.empty()) A.blocks.assign(2, f4{0, 0, 0, 0});
    if (A.leafNB.empty()) A.leafNB.assign(16, f4{0, 0, 0, 0});
    if (A.leafTB.empty()) A.leafTB.assign(32, f4{0, 0, 0, 0});
    if (A.runBase.empty()) A.runBase.assign(8, -1);
    for (int k = 0; k < 8; k++) A.runTB.push_back(f4{0, 0, 0, 0});   // (padding: the packet kernel may request the record after a leaf's last run)
    if (A.childDfs.empty()) A.childDfs.assign(8, -1);
    if (A.pblocks.empty()) A.pblocks.assign(PBLOCK_WORDS, 0.0f);
    {   // the packet kernel's node records (traverse.h LREC_*): leafNB | leafTB | first run, the run records behind them
        const size_t nNodes = A.runBase.size(), nRuns = A.runTB.size() / 4;   // (runTB ends in two records of padding)
        A.lrec.assign(nNodes * LREC_WORDS + nRuns * RUN_WORDS, 0.0f);
        for (size_t nd = 0; nd < nNodes; nd++) {
            float *q = &A.lrec[nd * LREC_WORDS];
            std::memcpy(q, &A.leafNB[2 * nd], 8 * sizeof(float));
            std::memcpy(q + 8, &A.leafTB[4 * nd], 16 * sizeof(float));
            q[24] = i2f(A.runBase[nd] < 0 ? -1 : (int)(nNodes * LREC_WORDS + (size_t)A.runBase[nd] * RUN_WORDS));
        }
        if (nRuns) std::memcpy(&A.lrec[nNodes * LREC_WORDS], A.runTB.data(), nRuns * RUN_WORDS * sizeof(float));
    }
    built = true;
    return true;
}

// ---- scene file -----------------------------------------------------------------------------------------------------------
// "XRTSCENE" u32 version=1 u32 nMeshes u32 nObjects, then per mesh: i32 ntri, f32 bbox[6], f32 reflectiveness, f32 refractionIndex,
// i32 flags (1 transparent, 2 interpolateNormals, 4 useTexture, 8 has premultiplied texels), i32 texW, i32 texH, f32 v[9n] n[9n]
// uv[6n] sn[3n] color[4n], u32 texels[texW*texH] (if 4), u32 texelsP[texW*texH] (if 8); per object: i32 nMeshes, i32 ids[],
// f32 world[16] invWorld[16] bbox[6] worldBbox[6].
namespace {
struct FileW {
    FILE *f; bool ok = true;
    template <class T> void put(const T *p, size_t n) { if (ok && n && fwrite(p, sizeof(T), n, f) != n) ok = false; }
    template <class T> void one(T v) { put(&v, 1); }
};
struct FileR {
    FILE *f; bool ok = true;
    long long left = 0;   // bytes of the file not read yet: no array is sized for more than the file can hold
    template <class T> void get(T *p, size_t n) {
        if (!ok || !n) return;
        if ((unsigned long long)n > (unsigned long long)left / sizeof(T) || fread(p, sizeof(T), n, f) != n) { ok = false; return; }
        left -= (long long)(n * sizeof(T));
    }
    template <class T> T one() { T v{}; get(&v, 1); return v; }
    template <class T> void vec(std::vector<T> &v, size_t n) {
        if (!ok) return;
        if ((unsigned long long)n > (unsigned long long)left / sizeof(T)) { ok = false; return; }   // a hostile header cannot ask for more than the file holds
        v.resize(n); get(v.data(), n);
    }
};
}  // namespace

bool HostScene::save(const char *path, std::string &err) const {
    FILE *f = path ? fopen(path, "wb") : nullptr;
    if (!f) { err = "xrt_scene_save: cannot open the file for writing"; return false; }
    FileW w{f};
    w.put("XRTSCENE", 8);
    w.one<uint32_t>(1); w.one<uint32_t>((uint32_t)meshes.size()); w.one<uint32_t>((uint32_t)objects.size());
    for (const HostMesh &m : meshes) {
        w.one<int32_t>(m.ntri); w.put(m.bbox, 6); w.one<float>(m.reflectiveness); w.one<float>(m.refractionIndex);
        w.one<int32_t>((m.transparent ? 1 : 0) | (m.interpolateNormals ? 2 : 0) | (m.useTexture ? 4 : 0) | (m.texelsP.empty() ? 0 : 8));
        w.one<int32_t>(m.texW); w.one<int32_t>(m.texH);
        w.put(m.v.data(), m.v.size()); w.put(m.n.data(), m.n.size()); w.put(m.uv.data(), m.uv.size()); w.put(m.sn.data(), m.sn.size());
        w.put(m.color.data(), m.color.size());
        if (m.useTexture) w.put(m.texels.data(), m.texels.size());
        if (!m.texelsP.empty()) w.put(m.texelsP.data(), m.texelsP.size());
    }
    for (const HostObject &o : objects) {
        w.one<int32_t>((int32_t)o.meshes.size()); w.put(o.meshes.data(), o.meshes.size());
        w.put(o.world, 16); w.put(o.invWorld, 16); w.put(o.bbox, 6); w.put(o.worldBbox, 6);
    }
    const bool ok = w.ok && fclose(f) == 0;
    if (!ok) err = "xrt_scene_save: write failed";
    return ok;
}

bool HostScene::load(const char *path, std::string &err) {
    FILE *f = path ? fopen(path, "rb") : nullptr;
    if (!f) { err = "xrt_scene_load: cannot open the file"; return false; }
    FileR r{f};
    if (fseek(f, 0, SEEK_END) == 0) { r.left = ftell(f); if (r.left < 0 || fseek(f, 0, SEEK_SET) != 0) r.ok = false; } else r.ok = false;
    char magic[8] = {0};
    r.get(magic, 8);
    const uint32_t version = r.one<uint32_t>(), nm = r.one<uint32_t>(), no = r.one<uint32_t>();
    if (!r.ok || std::memcmp(magic, "XRTSCENE", 8) != 0 || version != 1 || nm > (1u << 24) || no > (1u << 24)) { fclose(f); err = "xrt_scene_load: not a version-1 xrt scene file"; return false; }
    // The records are read into a scratch scene through add_mesh / add_object -- the very calls (and checks) the header promises a
    // load is equivalent to; the lists grow as records arrive, so the counts of a hostile header allocate nothing.
    HostScene tmp;
    std::string why;
    for (uint32_t i = 0; i < nm && r.ok; i++) {
        HostMesh m;
        m.ntri = r.one<int32_t>(); r.get(m.bbox, 6); m.reflectiveness = r.one<float>(); m.refractionIndex = r.one<float>();
        const int flags = r.one<int32_t>();
        m.texW = r.one<int32_t>(); m.texH = r.one<int32_t>();
        if (!r.ok || m.ntri < 0 || m.ntri >= (1 << 28) || m.texW < 0 || m.texH < 0 || (long long)m.texW * m.texH > (1LL << 30) || (flags & ~15)) { r.ok = false; break; }
        const bool useTexture = (flags & 4) != 0, hasP = (flags & 8) != 0;
        if (hasP && !useTexture) { r.ok = false; why = "premultiplied texels without UseTexture"; break; }
        if (!useTexture && (m.texW != 0 || m.texH != 0)) { r.ok = false; why = "texture size without UseTexture"; break; }
        const size_t n = (size_t)m.ntri;
        r.vec(m.v, n * 9); r.vec(m.n, n * 9); r.vec(m.uv, n * 6); r.vec(m.sn, n * 3); r.vec(m.color, n * 4);
        if (useTexture) r.vec(m.texels, (size_t)m.texW * m.texH);
        if (hasP) r.vec(m.texelsP, (size_t)m.texW * m.texH);
        if (!r.ok) break;
        xrt_material xm;
        std::memset(&xm, 0, sizeof(xm));
        xm.reflectiveness = m.reflectiveness; xm.refraction_index = m.refractionIndex;
        xm.transparent = (flags & 1) ? 1 : 0; xm.interpolate_normals = (flags & 2) ? 1 : 0; xm.use_texture = useTexture ? 1 : 0;
        xm.tex_argb = useTexture ? m.texels.data() : nullptr; xm.tex_pargb = hasP ? m.texelsP.data() : nullptr;
        xm.tex_width = m.texW; xm.tex_height = m.texH;
        static const float none[9] = {0};   // (an empty vector has no address; add_mesh rejects null arrays)
        if (tmp.add_mesh(n ? m.v.data() : none, n ? m.n.data() : none, n ? m.uv.data() : none, n ? m.sn.data() : none, n ? m.color.data() : none, m.ntri, &xm, m.bbox, why) < 0) { r.ok = false; break; }
    }
    for (uint32_t i = 0; i < no && r.ok; i++) {
        HostObject o;
        const int k = r.one<int32_t>();
        if (!r.ok || k < 0 || k > (1 << 20)) { r.ok = false; break; }
        r.vec(o.meshes, (size_t)k);
        r.get(o.world, 16); r.get(o.invWorld, 16); r.get(o.bbox, 6); r.get(o.worldBbox, 6);
        if (!r.ok) break;
        static const int noId[1] = {0};
        if (tmp.add_object(k ? o.meshes.data() : noId, k, o.world, o.invWorld, o.bbox, o.worldBbox, why) < 0) { r.ok = false; break; }
    }
    fclose(f);
    if (!r.ok) { err = "xrt_scene_load: truncated or corrupt scene file" + (why.empty() ? std::string() : " (" + why + ")"); return false; }
    meshes = std::move(tmp.meshes); objects = std::move(tmp.objects);
    built = false;
    return true;
}

SceneView HostScene::host_view() const {
    SceneView S;
    const SceneArrays &A = arrays;
    S.blocks = A.blocks.data(); S.childDfs = A.childDfs.data(); S.leafNB = A.leafNB.data(); S.leafTB = A.leafTB.data(); S.runBase = A.runBase.data(); S.runTB = A.runTB.data(); S.refT = A.refT.data(); S.pblocks = A.pblocks.data(); S.lrec = A.lrec.data();
    S.refN = A.refN.data(); S.refG = A.refG.data(); S.meshes = A.meshes.data();
    S.snodes = A.snodes.data(); S.srefs = A.srefs.data(); S.scull = A.scull.data(); S.objects = A.objects.data(); S.objMesh = A.objMesh.data();
    S.nMeshes = (int)meshes.size(); S.nObjects = (int)objects.size();
    S.sceneDepth = A.sceneDepth + 1; S.meshDepth = A.meshDepth + 1;
    S.nodeCull = 1;
    return S;
}

}  // namespace xrt
