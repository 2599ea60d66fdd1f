// xrt_core.h — records laid out in HBM and the strict-binary32 arithmetic of the hot path.
//
// Everything here is `__host__ __device__`: the HIP kernels (kernels.hip) are the product; the host
// side uses the same inline functions only to prepare per-frame constants (inverse view-projection)
// and, in tests/emul, to single-step the traversal state machine on the CPU for debugging.
//
// Arithmetic contract: IEEE-754 binary32, one rounding per operation, NO fma
// contraction (the library is built with -ffp-contract=off; division and sqrt are the correctly
// rounded expansions hipcc emits by default), `double` exactly where the C# calls System.Math.
// Each function cites the reference line it reproduces (aliases as in include/xrt.h).
#pragma once
#include <float.h>
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define XRT_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define XRT_HD inline
#endif

namespace xrt {

struct alignas(16) f4 { float x, y, z, w; };
struct v3 { float x, y, z; };

XRT_HD int   f2i(float f) { return __builtin_bit_cast(int, f); }
XRT_HD float i2f(int i)   { return __builtin_bit_cast(float, i); }

// ---- HBM records ------------------------------------------------------------------------------------
// MESH OCTREES use IMPLICIT boxes.  The reference derives the eight child boxes of a node from the
// parent box alone (MO:207,217-218: half = (max-min)/2; min_c = min + half*(i,j,k); max_c = min_c + half),
// so the traversal recomputes them with the same binary32 operations instead of loading them: a node
// costs ALU only, and the only memory touched per interior node is its 32-byte block descriptor.
//
// Block descriptor = the 8 children (c = 4i+2j+k) of one interior node, two f4:
//   w0 childBlockBase : interior child c owns block  childBlockBase + popcount(interiorMask & ((1<<c)-1))
//   w1 refBase        : first leaf reference of this block's leaf children (stored contiguously, child order)
//   w2 masks          : interiorMask | emptyMask << 8 | safeMask << 16 | facingMask << 24
//                       empty: leaf child without triangles (bucketed by the reference, can never hit, Q4)
//                       safe : interior child whose non-empty descendant leaf boxes all lie inside its own
//                              box, so its own entry key is a lower bound of every bucket key below it
//                              (slab monotonicity) and it may be pruned by key
//                       facing: interior child whose normal box (SceneView::leafNB) does not hold the origin -- only such a subtree
//                              can face away from a ray as a whole (traverse.h all_back_facing); set by scene_host.cpp
//   w3 total          : leaf references of the block
//   w4..w7 offs[8]    : 16-bit start offset of child c's references relative to refBase
// Node id (for leaf ids / ties) = block * 8 + c, root = -1; childDfs[node id] = DFS pre-order index.
//
// The SCENE OCTREE (few nodes, DFS order matters) keeps explicit 32-byte records:
//   lo = (box.min.xyz, a)  hi = (box.max.xyz, b);  leaf: a = first ref, b = NODE_LEAF | count;
//   interior: a = index of child 0 (multiple of 8), b = 0.
constexpr int NODE_LEAF = (int)0x80000000;
constexpr int ROOT_NODE = -1;

// Leaf reference, stored in leaf order so a leaf is one contiguous run, split in two streams:
//   refN[r] = (surfaceNormal.xyz, as_float(scene-global triangle id))      16 B — enough for the back-face
//             test RE:48-51 and the ignoreTriangle test MO:290; most references end here
//   refG[r] = v1.xyz, E1.xyz, E2.xyz                                        36 B — only for front-facing ones;
//             E1 = v2 - v1, E2 = v3 - v1 (RE:54-55, the same single binary32 subtraction the reference
//             performs per test)
struct g3 { float x, y, z; };

struct MeshRec {        // 112 B = seven f4
    float bmin[3];      // Mesh.MeshBoundingBox (MESH:14, TMP:244-307)
    int   rootBlock;    // block of the root's children, or -1 when the root is a leaf
    float bmax[3];
    int   triBase;      // global id of Triangles[0]
    float rmin[3];      // MeshOctree root box (MO:59-80)
    int   rootRef;      // root-is-leaf: first leaf reference
    float rmax[3];
    int   rootCount;    // root-is-leaf: triangle count
    int   ntri;
    int   material;
    int   maxDepth;
    int   dfsBase;      // offset of this mesh's entries in childDfs (= rootBlock * 8 for interior roots)
    float nbMin[4];     // component-wise min / max of ALL the mesh's surface normals (nbMin[3] != 0: a NaN normal, never cull) -- the
    float nbMax[4];     // root's record of the node normal boxes (SceneView::leafNB holds the other nodes', interior ones included)
};

struct ObjRec {         // 176 B
    float invWorld[16]; // SO:198
    float world[16];    // SO:193
    int   meshStart;    // into objMesh[]
    int   meshCount;
    int   cullOk;       // the pre-cull record below is valid (finite, invertible transform)
    float cullK2;
    // World-space pre-cull (DESIGN.md §3 "Object pre-cull": the bound and its proof).  cullMin/cullMax[0..2] = the
    // axis-aligned hull of the object's mesh AABBs mapped to world space (exact image, evaluated in double).  A world ray
    // with |origin| = r that misses this box enlarged by  m(r) = cullMin[3] + cullMax[3] * r + cullK2 * r * r  cannot pass
    // MESH:34-39 for any mesh of the object in the reference's binary32 arithmetic (OSM:358-364 transform + slab test).
    float cullMin[4];
    float cullMax[4];
};
// m(r) of ObjRec; r = Euclidean norm of the world-space ray origin (any upper bound of it is as good).
XRT_HD float cull_margin(const ObjRec &ob, float r) { return ob.cullMin[3] + (ob.cullMax[3] + ob.cullK2 * r) * r; }

struct MaterialRec {    // 32 B (MAT:234-268)
    float reflectiveness;
    float refractionIndex;
    int   flags;        // bit0 transparent, bit1 interpolateNormals, bit2 useTexture
    int   texOffset;    // into texels[]
    int   texWidth, texHeight;
    int   texOffsetP;   // Texture.ColorData (premultiplied copy, TEX:24-33) for the bilinear filter; == texOffset when the host gave none
    int   pad1;
};
constexpr int MAT_TRANSPARENT = 1, MAT_INTERP = 2, MAT_TEXTURE = 4;

// Per-triangle shading record, 80 B = five f4 (TRI:16-24 minus the intersection fields):
//   s0 = (n1.xyz, uv1.x) s1 = (n2.xyz, uv1.y) s2 = (n3.xyz, uv2.x) s3 = (color.xyzw) s4 = (uv2.y, uv3.x, uv3.y, material)
// plus the surface normal, read from the leaf reference.

struct LightRec {       // 64 B
    int    kind;
    float  px, py, pz;
    float  dx, dy, dz;
    float  cr, cg, cb;
    float  intensity;
    float  angleCosine; // SPOT:25
    double decayDenom;  // Math.Pow(1 - angleCosine, DecayExponent), SPOT:54
    int    pad0, pad1;
};

// Image-tile shards: the frame's 64x8-pixel tiles are numbered row-major and dealt to the ranks in GROUPS of consecutive
// tiles, group q -> rank q % N.  Slot s of rank r (its s-th tile, in tile order) is tile ((s / G) * N + r) * G + s % G.
// XRT_SHARD_GROUP: G; 0 = one group per tile row.  G = 1 (single tiles round-robin) is the ABI (xrt.h); groups of 4 or 15 tiles and whole
// tile rows measured the same per-shard frame period on C4 and C5 at eight shards (profiles/r03/shard_group_experiment.txt).
#ifndef XRT_SHARD_GROUP
#define XRT_SHARD_GROUP 1
#endif
XRT_HD int shard_group(int tilesX) { return XRT_SHARD_GROUP > 0 ? XRT_SHARD_GROUP : tilesX; }
XRT_HD long long shard_tile(long long slot, int rank, int count, int tilesX) {
    const int G = shard_group(tilesX);
    if (count <= 1 || G == 1) return slot * count + rank;
    const long long q = slot / G;
    return (q * count + rank) * G + (slot - q * G);
}
XRT_HD long long shard_tiles_per_rank(long long totalTiles, int count, int tilesX) {
    const int G = shard_group(tilesX);
    const long long groups = (totalTiles + G - 1) / G;
    return ((groups + count - 1) / count) * G;
}

// Position of path slot `within` (0..511) inside its 64x8-pixel tile.  A wavefront takes 64 consecutive slots; laid out as
// eight 8x8-pixel blocks side by side (Z-order inside a block) they cover a square patch of the image instead of a 64-pixel line, so the 64 rays of a
// wave stay close together in the scene (fewer distinct octree leaves per wave, for the per-lane and the wave-packet kernel
// alike).  Every kernel, the host's ray export and the tile-shard layout (xrt.h XRT_TILE_*, dist.py) use this one map.
XRT_HD void tile_slot_xy(int within, int &x, int &y) {
    const int blk = within >> 6, i = within & 63;
    // Z-order inside the 8x8 block: consecutive slots stay together at every scale (4 slots = a 2x2 quad, 16 = 4x4), so the
    // 4 pixels x 16 samples of a wave of a 16-sub-ray frame are a 2x2 quad (a 4x1 strip of one image row measured the same: C5
    // 5.83 against 5.86 ms per frame)
    x = blk * 8 + ((i & 1) | ((i >> 1) & 2) | ((i >> 2) & 4));
    y = ((i >> 1) & 1) | ((i >> 2) & 2) | ((i >> 3) & 4);
}

// ---- Vector3 (XNA definitions) ---------------------------------------------------------
XRT_HD v3 mk(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
XRT_HD v3 add(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
XRT_HD v3 sub(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
XRT_HD v3 neg(v3 a) { return mk(-a.x, -a.y, -a.z); }
XRT_HD v3 scale(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
XRT_HD v3 mul(v3 a, v3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
XRT_HD v3 divf(v3 a, float d) { float n = 1.0f / d; return mk(a.x * n, a.y * n, a.z * n); }
XRT_HD float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
XRT_HD v3 cross(v3 a, v3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
XRT_HD float sqrt_rn(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_sqrtf(x);   // correctly rounded expansion (checked in the ISA: v_sqrt + fma fix-up)
#else
    return sqrtf(x);
#endif
}
XRT_HD float length(v3 a) { return sqrt_rn((a.x * a.x + a.y * a.y) + a.z * a.z); }
XRT_HD v3 normalize(v3 a) {
    float n = (a.x * a.x + a.y * a.y) + a.z * a.z;
    float s = 1.0f / sqrt_rn(n);
    return mk(a.x * s, a.y * s, a.z * s);
}
XRT_HD v3 transform(v3 p, const float *m) {   // Vector3.Transform(position, matrix)
    return mk(((p.x * m[0] + p.y * m[4]) + p.z * m[8]) + m[12],
              ((p.x * m[1] + p.y * m[5]) + p.z * m[9]) + m[13],
              ((p.x * m[2] + p.y * m[6]) + p.z * m[10]) + m[14]);
}
XRT_HD v3 reflect(v3 v, v3 n) {   // RT:549
    float k = (v.x * n.x + v.y * n.y) + v.z * n.z;
    return mk(v.x - (2.0f * k) * n.x, v.y - (2.0f * k) * n.y, v.z - (2.0f * k) * n.z);
}
XRT_HD v3 lerp(v3 a, v3 b, float t) {   // RT:584
    return mk(a.x + (b.x - a.x) * t, a.y + (b.y - a.y) * t, a.z + (b.z - a.z) * t);
}
XRT_HD bool is_nan(float a) { return a != a; }
XRT_HD float math_max(float a, float b) { return a > b ? a : (is_nan(a) ? a : b); }   // System.Math.Max
XRT_HD float math_min(float a, float b) { return a < b ? a : (is_nan(a) ? a : b); }   // System.Math.Min

// ---- ray with the per-axis reciprocals BoundingBox.Intersects recomputes on every call -----------------
struct RayPre {
    v3 o, d, inv;
    int par;   // bit k set: fabs(d_k) < 1e-6f (the parallel branch)
};
XRT_HD RayPre make_ray(v3 o, v3 d) {
    RayPre r;
    r.o = o; r.d = d;
    r.par = (fabsf(d.x) < 1e-06f ? 1 : 0) | (fabsf(d.y) < 1e-06f ? 2 : 0) | (fabsf(d.z) < 1e-06f ? 4 : 0);
    r.inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    return r;
}
// Front-to-back child permutation: visit child (p ^ dirmask) for p = 0..7.
XRT_HD int dir_mask(v3 d) { return (d.x < 0.0f ? 4 : 0) | (d.y < 0.0f ? 2 : 0) | (d.z < 0.0f ? 1 : 0); }

// BoundingBox.Intersects(ref Ray, out float?) (MO:331, OSM:460, MESH:37): entry distance or miss.
XRT_HD bool slab(const RayPre &r, float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float &key) {
    float num = 0.0f, num2 = FLT_MAX;
    bool ok = true;
    if (r.par & 1) { ok = ok && !(r.o.x < mnx || r.o.x > mxx); }
    else {
        float t1 = (mnx - r.o.x) * r.inv.x, t2 = (mxx - r.o.x) * r.inv.x;
        if (t1 > t2) { float t = t1; t1 = t2; t2 = t; }
        num = math_max(t1, num); num2 = math_min(t2, num2);
        ok = ok && !(num > num2);
    }
    if (r.par & 2) { ok = ok && !(r.o.y < mny || r.o.y > mxy); }
    else {
        float t1 = (mny - r.o.y) * r.inv.y, t2 = (mxy - r.o.y) * r.inv.y;
        if (t1 > t2) { float t = t1; t1 = t2; t2 = t; }
        num = math_max(t1, num); num2 = math_min(t2, num2);
        ok = ok && !(num > num2);
    }
    if (r.par & 4) { ok = ok && !(r.o.z < mnz || r.o.z > mxz); }
    else {
        float t1 = (mnz - r.o.z) * r.inv.z, t2 = (mxz - r.o.z) * r.inv.z;
        if (t1 > t2) { float t = t1; t1 = t2; t2 = t; }
        num = math_max(t1, num); num2 = math_min(t2, num2);
        ok = ok && !(num > num2);
    }
    key = num;
    return ok;
}

// RayExtensions.IntersectsTriangleBackfaceCulling (RE:42-75), split at the back-face test.
XRT_HD float facing(v3 N, v3 D) { return (N.x * D.x + N.y * D.y) + N.z * D.z; }   // RE:49; culled when > 0 (RE:50)
// The reference divides first and tests afterwards: (distance, u, v) = inv * (row1, row2, row3), inv = 1f / det, accepted when all
// three are >= 0 and u + v <= 1 (RE:66-74).  A product fl(a * inv) is certainly negative -- so the triangle is certainly rejected --
// when a and det have opposite sign bits and the product can neither underflow to (-)0, which compares >= 0, nor be a NaN:
// sign(inv) == sign(det) always (1f / +-0 = +-inf), |det| <= 2^60 keeps |inv| >= 2^-61, |a| >= 2^-60 then keeps |a * inv| >= 2^-121,
// far above the smallest subnormal, and the comparisons are false for NaNs.  Most rejected triangles are rejected by the sign of
// u alone, before the second cross product, two dot products and the division; the accepted ones get the reference's arithmetic.
XRT_HD bool certainly_negative(float a, float det) {
    return ((f2i(a) ^ f2i(det)) < 0) && fabsf(a) >= 8.6736174e-19f /* 2^-60 */ && fabsf(det) <= 1.1529215e18f /* 2^60 */;
}
// Stage A: everything up to the sign of u.  Returns false when the triangle is certainly rejected (u < 0).
XRT_HD bool tri_stage_a(v3 O, v3 D, v3 v1, v3 E1, v3 E2, v3 &T, float &det, float &row2) {
    T = mk(O.x - v1.x, O.y - v1.y, O.z - v1.z);      // RE:46
    v3 P = cross(D, E2);                              // RE:58
    det = dot(P, E1);                                 // RE:66 (the divisor)
    row2 = dot(P, T);                                 // RE:63
    return !certainly_negative(row2, det);            // u < 0
}
// Stage B: the rest of RE:59-74 for a triangle that survived stage A.
XRT_HD bool tri_stage_b(v3 D, v3 E1, v3 E2, v3 T, float det, float row2, float &u, float &v, float &dist) {
    u = 0.0f; v = 0.0f; dist = 0.0f;
    v3 Q = cross(T, E1);                              // RE:59
    const float row3 = dot(Q, D);                     // RE:64
    const float row1 = dot(Q, E2);                    // RE:62
    if (certainly_negative(row3, det) || certainly_negative(row1, det)) return false;   // v < 0 or distance < 0
    float inv = 1.0f / det;                           // RE:66
    dist = row1 * inv; u = row2 * inv; v = row3 * inv;
    return u >= 0.0f && v >= 0.0f && dist >= 0.0f && (u + v) <= 1.0f;   // RE:71-74
}
XRT_HD bool tri_test_front(v3 O, v3 D, v3 v1, v3 E1, v3 E2, float &u, float &v, float &dist) {
    u = 0.0f; v = 0.0f; dist = 0.0f;
    v3 T; float det, row2;
    if (!tri_stage_a(O, D, v1, E1, E2, T, det, row2)) return false;
    return tri_stage_b(D, E1, E2, T, det, row2, u, v, dist);
}

// ---- tight leaf boxes (DESIGN.md §3 "Leaves whose triangles a ray cannot reach") -----------------------------------------
// A leaf's octree cell is a cube, its triangles usually fill a thin slab of it: a ray can cross the cell far from all of them,
// and MO:288-304 then runs RE:42-75 on every reference for nothing.  Per leaf the host stores the box of the triangles' vertices
// (v1, v1 + E1, v1 + E2 -- the very numbers RE:54-58 uses), the box of their unit geometric normals and two constants; a ray
// skips the leaf when it misses the vertex box grown by rho, where rho bounds how far from a triangle the exact ray can pass
// while the binary32 test RE:42-75 still answers "hit":
//     rho = LEAF_CULL_C u0 kappa (Tmax + Emax),  kappa <= max_t |E1||E2|/|E1 x E2| * |D| / min_t |D . n_t|,  u0 = 2^-24,
// Tmax >= |O - v1| (distance to the farthest corner of the vertex box), Emax = the leaf's longest edge E1 / E2.  The
// derivation (forward error of RE:42-75, no FMA: |X* - X_f| <= 84 u0 kappa (|T| + Emax), 42 u0 kappa |T| more for a hit reported
// slightly behind the origin, 3 u0 (Tmax + rho) for the slab test below) needs 130; 192 leaves room for the roundings of
// this evaluation itself.  kappa = inf (a ray in a triangle's plane, a degenerate triangle) grows the box over everything.
// Record: [0] = (lo.xyz, K = C u0 max_t |E1||E2|/|N|), [1] = (hi.xyz, Emax), [2] = (nlo.xyz, ok), [3] = (nhi.xyz, -).
constexpr float LEAF_CULL_C = 192.0f;
// Leaves of at least LEAF_RUN_MIN references also carry one such record per run of LEAF_RUN consecutive references (SceneView::runTB):
// the octree stops splitting at MO:42's 50 triangles, and a ray that reaches a leaf's box usually comes near only a few of them.
#ifndef XRT_LEAF_RUN
#define XRT_LEAF_RUN 8
#define XRT_LEAF_RUN_MIN 16
#endif
constexpr int LEAF_RUN = XRT_LEAF_RUN, LEAF_RUN_MIN = XRT_LEAF_RUN_MIN;
constexpr int TRI_REC_WORDS = 13, TRI_REC_BYTES = 52;   // refT: (surface normal, global triangle id, v1, E1, E2)
struct RayCull {
    float d2;    // |D|_2, or 0 when the ray takes no part (a component of D below 2^-40 |D| or non-finite, |O| above 2^40)
    float slack; // 2^-21 |D|_1: rounding of the three products of D . n below
};
XRT_HD RayCull make_ray_cull(v3 o, v3 d) {
    RayCull c;
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    const float d1 = (ax + ay) + az;
    const float q = (d.x * d.x + d.y * d.y) + d.z * d.z;
    const float lo = fminf(fminf(ax, ay), az), om = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z));
    const bool ok = d1 >= 9.5367432e-7f /* 2^-20 */ && d1 <= 1048576.0f && lo >= d1 * 9.094947e-13f /* 2^-40 */ && om <= 1.0995116e12f /* 2^40 */;
    c.d2 = ok ? sqrtf(q) * 1.000001f : 0.0f;   // (comparisons with a NaN are false: not ok)
    c.slack = d1 * 4.7683716e-7f;              // 2^-21
    return c;
}
XRT_HD bool leaf_certainly_missed(const RayPre &r, const RayCull &rc, const f4 &a, const f4 &b, const f4 &nl, const f4 &nh) {
    // the farthest corner of the vertex box from the origin: >= |O - v1| for every triangle of the leaf
    const float fx = fmaxf(fabsf(r.o.x - a.x), fabsf(r.o.x - b.x)), fy = fmaxf(fabsf(r.o.y - a.y), fabsf(r.o.y - b.y)),
                fz = fmaxf(fabsf(r.o.z - a.z), fabsf(r.o.z - b.z));
    const float tmax = sqrtf((fx * fx + fy * fy) + fz * fz) * 1.000001f;
    // a lower bound of |D . n| over the box of the leaf's unit normals
    const float px = r.d.x * nl.x, qx = r.d.x * nh.x, py = r.d.y * nl.y, qy = r.d.y * nh.y, pz = r.d.z * nl.z, qz = r.d.z * nh.z;
    const float lo = (fminf(px, qx) + fminf(py, qy)) + fminf(pz, qz), hi = (fmaxf(px, qx) + fmaxf(py, qy)) + fmaxf(pz, qz);
    const float cmin = fmaxf(lo, -hi) - rc.slack;
#if defined(__HIP_DEVICE_COMPILE__)
    const float rho = ((a.w * (b.w + tmax)) * rc.d2) * (__builtin_amdgcn_rcpf(cmin) * 1.000001f);   // (v_rcp_f32: 1 ulp)
#else
    const float rho = ((a.w * (b.w + tmax)) * rc.d2) * ((1.0f / cmin) * 1.000001f);
#endif
    // (every comparison below is false when something is not a finite positive number: no skip)
    if (!(nl.w > 0.0f && rc.d2 > 0.0f && cmin > 0.0f && rho < 1.0e15f)) return false;
    const float t1x = ((a.x - rho) - r.o.x) * r.inv.x, t2x = ((b.x + rho) - r.o.x) * r.inv.x;
    const float t1y = ((a.y - rho) - r.o.y) * r.inv.y, t2y = ((b.y + rho) - r.o.y) * r.inv.y;
    const float t1z = ((a.z - rho) - r.o.z) * r.inv.z, t2z = ((b.z + rho) - r.o.z) * r.inv.z;
    const float tn = fmaxf(fmaxf(fmaxf(fminf(t1x, t2x), 0.0f), fminf(t1y, t2y)), fminf(t1z, t2z));
    const float tf = fminf(fminf(fmaxf(t1x, t2x), fmaxf(t1y, t2y)), fmaxf(t1z, t2z));
    return tn > tf;
}

// The same decision for a whole BUNDLE of rays at once (packet.hip: 64 lanes test 64 triangles' records against the packet's rays): true only if
// leaf_certainly_missed is true for EVERY ray of the bundle.  The bundle is given by component-wise bounds of its rays' origins, directions and inverse
// directions (every axis: one sign for all rays, no zero), the largest |D|_2 (RayCull::d2) and slack; the same expressions are evaluated on interval end
// points.  Every binary32 operation involved is monotone in each argument, so the COMPUTED end points bound the values each ray computes for itself: the
// far-corner distance from above, the lower bound of |D . n| from below, hence rho from above (the hardware reciprocal is within an ulp of a monotone
// function: a factor 1.00001 pays for that), the entry parameters of the grown box from below and the exit parameters from above.  `tn > tf` for the bounds
// therefore implies `tn > tf` for every ray's own test.  (Interval arithmetic loses where the rays of a bundle diverge; for the 64 rays of a 2 x 2-pixel
// patch it loses next to nothing.)
struct RayBundle {
    v3 omin, omax, dmin, dmax, imin, imax;
    float d2, slack;   // largest RayCull::d2 / slack of the bundle's rays (d2 == 0: some ray takes no part -- no bundle)
};
XRT_HD float min4(float a, float b, float c, float d) { return fminf(fminf(a, b), fminf(c, d)); }
XRT_HD float max4(float a, float b, float c, float d) { return fmaxf(fmaxf(a, b), fmaxf(c, d)); }
XRT_HD bool bundle_certainly_missed(const RayBundle &q, const f4 &a, const f4 &b, const f4 &nl, const f4 &nh) {
    auto far1 = [](float o0, float o1, float lo, float hi) { return fmaxf(fmaxf(fabsf(o0 - lo), fabsf(o1 - lo)), fmaxf(fabsf(o0 - hi), fabsf(o1 - hi))); };
    const float fx = far1(q.omin.x, q.omax.x, a.x, b.x), fy = far1(q.omin.y, q.omax.y, a.y, b.y), fz = far1(q.omin.z, q.omax.z, a.z, b.z);
    const float tmax = sqrtf((fx * fx + fy * fy) + fz * fz) * 1.000001f;
    const float x0 = q.dmin.x * nl.x, x1 = q.dmin.x * nh.x, x2 = q.dmax.x * nl.x, x3 = q.dmax.x * nh.x;
    const float y0 = q.dmin.y * nl.y, y1 = q.dmin.y * nh.y, y2 = q.dmax.y * nl.y, y3 = q.dmax.y * nh.y;
    const float z0 = q.dmin.z * nl.z, z1 = q.dmin.z * nh.z, z2 = q.dmax.z * nl.z, z3 = q.dmax.z * nh.z;
    const float lo = (min4(x0, x1, x2, x3) + min4(y0, y1, y2, y3)) + min4(z0, z1, z2, z3), hi = (max4(x0, x1, x2, x3) + max4(y0, y1, y2, y3)) + max4(z0, z1, z2, z3);
    const float cmin = fmaxf(lo, -hi) - q.slack;
#if defined(__HIP_DEVICE_COMPILE__)
    const float rho = (((a.w * (b.w + tmax)) * q.d2) * (__builtin_amdgcn_rcpf(cmin) * 1.000001f)) * 1.00001f;
#else
    const float rho = (((a.w * (b.w + tmax)) * q.d2) * ((1.0f / cmin) * 1.000001f)) * 1.00001f;
#endif
    if (!(nl.w > 0.0f && q.d2 > 0.0f && cmin > 0.0f && rho < 1.0e15f)) return false;
    // per axis: the two plane parameters of every ray lie in [t1lo, t1hi], [t2lo, t2hi]; its entry parameter is >= the smaller lower end, its exit <= the larger upper end
    auto axis = [&](float alo, float bhi, float o0, float o1, float i0, float i1, float &nearLo, float &farHi) {
        const float v1lo = (alo - rho) - o1, v1hi = alo - o0, v2lo = bhi - o1, v2hi = (bhi + rho) - o0;
        const float t1lo = min4(v1lo * i0, v1lo * i1, v1hi * i0, v1hi * i1), t1hi = max4(v1lo * i0, v1lo * i1, v1hi * i0, v1hi * i1);
        const float t2lo = min4(v2lo * i0, v2lo * i1, v2hi * i0, v2hi * i1), t2hi = max4(v2lo * i0, v2lo * i1, v2hi * i0, v2hi * i1);
        nearLo = fminf(t1lo, t2lo); farHi = fmaxf(t1hi, t2hi);
    };
    float nx, fxh, ny, fyh, nz, fzh;
    axis(a.x, b.x, q.omin.x, q.omax.x, q.imin.x, q.imax.x, nx, fxh);
    axis(a.y, b.y, q.omin.y, q.omax.y, q.imin.y, q.imax.y, ny, fyh);
    axis(a.z, b.z, q.omin.z, q.omax.z, q.imin.z, q.imax.z, nz, fzh);
    const float tn = fmaxf(fmaxf(fmaxf(nx, 0.0f), ny), nz), tf = fminf(fminf(fxh, fyh), fzh);
    return tn > tf;
}

// ---- Color (RT:584,705,726,732) ---------------------------------------------------------------------------
XRT_HD uint32_t pack_unorm255(float v) {
    v = v * 255.0f;
    if (is_nan(v)) return 0u;
    if (v < 0.0f) return 0u;       // also -inf
    if (v > 255.0f) return 255u;   // also +inf
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__builtin_rintf(v);   // round half to even (Math.Round)
#else
    return (uint32_t)nearbyintf(v);
#endif
}
XRT_HD uint32_t pack_color(v3 c) { return pack_unorm255(c.x) | (pack_unorm255(c.y) << 8) | (pack_unorm255(c.z) << 16) | 0xff000000u; }
XRT_HD v3 unpack_color(uint32_t p) {
    return mk((float)(p & 0xffu) / 255.0f, (float)((p >> 8) & 0xffu) / 255.0f, (float)((p >> 16) & 0xffu) / 255.0f);
}

// ---- Matrix (host only: per-frame constants) -------------------------------------------------------------------
inline void mat_multiply(const float *a, const float *b, float *r) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r[4 * i + j] = ((a[4 * i] * b[j] + a[4 * i + 1] * b[4 + j]) + a[4 * i + 2] * b[8 + j]) + a[4 * i + 3] * b[12 + j];
}
inline void mat_invert(const float *m, float *r) {   // Matrix.Invert
    float n5 = m[0], n4 = m[1], n3 = m[2], n2 = m[3], n9 = m[4], n8 = m[5], n7 = m[6], n6 = m[7];
    float n17 = m[8], n16 = m[9], n15 = m[10], n14 = m[11], n13 = m[12], n12 = m[13], n11 = m[14], n10 = m[15];
    float n23 = n15 * n10 - n14 * n11, n22 = n16 * n10 - n14 * n12, n21 = n16 * n11 - n15 * n12;
    float n20 = n17 * n10 - n14 * n13, n19 = n17 * n11 - n15 * n13, n18 = n17 * n12 - n16 * n13;
    float n39 = (n8 * n23 - n7 * n22) + n6 * n21;
    float n38 = -((n9 * n23 - n7 * n20) + n6 * n19);
    float n37 = (n9 * n22 - n8 * n20) + n6 * n18;
    float n36 = -((n9 * n21 - n8 * n19) + n7 * n18);
    float num = 1.0f / (((n5 * n39 + n4 * n38) + n3 * n37) + n2 * n36);
    r[0] = n39 * num; r[4] = n38 * num; r[8] = n37 * num; r[12] = n36 * num;
    r[1] = -((n4 * n23 - n3 * n22) + n2 * n21) * num;
    r[5] = ((n5 * n23 - n3 * n20) + n2 * n19) * num;
    r[9] = -((n5 * n22 - n4 * n20) + n2 * n18) * num;
    r[13] = ((n5 * n21 - n4 * n19) + n3 * n18) * num;
    float n35 = n7 * n10 - n6 * n11, n34 = n8 * n10 - n6 * n12, n33 = n8 * n11 - n7 * n12;
    float n32 = n9 * n10 - n6 * n13, n31 = n9 * n11 - n7 * n13, n30 = n9 * n12 - n8 * n13;
    r[2] = ((n4 * n35 - n3 * n34) + n2 * n33) * num;
    r[6] = -((n5 * n35 - n3 * n32) + n2 * n31) * num;
    r[10] = ((n5 * n34 - n4 * n32) + n2 * n30) * num;
    r[14] = -((n5 * n33 - n4 * n31) + n3 * n30) * num;
    float n29 = n7 * n14 - n6 * n15, n28 = n8 * n14 - n6 * n16, n27 = n8 * n15 - n7 * n16;
    float n26 = n9 * n14 - n6 * n17, n25 = n9 * n15 - n7 * n17, n24 = n9 * n16 - n8 * n17;
    r[3] = -((n4 * n29 - n3 * n28) + n2 * n27) * num;
    r[7] = ((n5 * n29 - n3 * n26) + n2 * n25) * num;
    r[11] = -((n5 * n28 - n4 * n26) + n2 * n24) * num;
    r[15] = ((n5 * n27 - n4 * n25) + n3 * n24) * num;
}

// Per-frame ray-generation constants: Viewport.Unproject inverts world*view*proj on every call
// (RT:415,419) — the same matrix for every pixel, so it is computed once on the host with the same
// arithmetic and handed to the kernel.
// Where the level records of path p live (lvlA / lvlB [level * stride + lvl_at(p)]).  Plain unsharded frames keep records only for the TILES that overlap the screen rectangle of
// the scene's root box (RayGenParams::cull*): no other path ever writes or reads one (k_raygen answers those pixels, k_compose knows the rectangle), and on C5 the rectangle
// is half the image -- 1.5 GB of 3.2 per frame context.  Tiles are numbered row by row over the frame (tile = path >> shift), the kept ones row by row over their own grid.
struct LvlMap {
    int on;               // 0: lvl_at(p) = p
    int tilesX, tx0, ty0, rw, shift;
    unsigned inv;         // ceil(2^32 / tilesX): tile / tilesX == (tile * inv) >> 32 for every tile of the frame (checked on the host)
};
XRT_HD size_t lvl_at(const LvlMap &m, int p) {
    if (!m.on) return (size_t)p;
    const unsigned t = (unsigned)p >> m.shift, within = (unsigned)p & ((1u << m.shift) - 1u);
    const unsigned ty = (unsigned)(((unsigned long long)t * m.inv) >> 32), tx = t - ty * (unsigned)m.tilesX;
    return ((size_t)((ty - (unsigned)m.ty0) * (unsigned)m.rw + (tx - (unsigned)m.tx0)) << m.shift) + within;
}
struct RayGenParams {
    float m[16];          // Invert(Multiply(Multiply(Identity, view), proj))
    float vpX, vpY, vpW, vpH, minDepth, depthRange;
    int   width, height;
    int   tilesX, tilesY;
    int   shardRank, shardCount;   // path -> tile mapping
    const int *tileOfSlot;         // ... or this rank's row of an installed tile table (xrt_scene_set_tile_table): the tile of slot s, -1 = unused; null: round-robin
    int   samples;                 // 1, or 16 for XRT_MS_FIXED16, or 4 for one level of the adaptive quadrants
    int   quadLevel;               // adaptive supersampling (RT:215-311): -1 off; level 0 quadrants are the pixels,
    const float *quadCx, *quadCy;  // deeper ones are listed (centre per quadrant)
    float quadSize;                // 1, 0.5, 0.25 ... (RT:195, RT:290 size / 2.0f)
    // Pixels outside this rectangle cannot reach the scene's root box (its eight corners projected to the screen, two
    // pixels of margin; the whole viewport when a corner is not safely in front of the eye): k_raygen answers them
    // (OSM:318-320) without building their rays.
    int   cullX0, cullY0, cullX1, cullY1;
    int   cullSkipsRecord;         // k_compose knows the rectangle too: no generation-0 record is written or read for those pixels
    LvlMap lvl;                    // ... and the level records exist for the rectangle's tiles only (above)
    // A pass whose size only the device knows (the deeper quadrant levels of an adaptive frame in flight): the pass has
    // min(*pathsDev, pathsCap) * pathsMul paths; null: the host's count
    const int *pathsDev;
    int   pathsMul, pathsCap;
};
XRT_HD int pass_paths(const RayGenParams &g, int hostCount) {
    if (!g.pathsDev) return hostCount;
    int n = *g.pathsDev;
    if (n > g.pathsCap) n = g.pathsCap;
    return n < 0 ? 0 : n * g.pathsMul;
}

// One Viewport.Unproject (RT:415 / RT:419) given the hoisted inverse matrix.
XRT_HD v3 unproject(const RayGenParams &g, float sx, float sy, float sz) {
    float X = (((sx - g.vpX) / g.vpW) * 2.0f) - 1.0f;
    float Y = -((((sy - g.vpY) / g.vpH) * 2.0f) - 1.0f);
    float Z = (sz - g.minDepth) / g.depthRange;
    v3 vec = transform(mk(X, Y, Z), g.m);
    float a = (((X * g.m[3]) + (Y * g.m[7])) + (Z * g.m[11])) + g.m[15];
    float num = a - 1.0f;
    bool within = (-1.401298E-45f <= num) && (num <= 1.401298E-45f);
    if (!within) vec = divf(vec, a);
    return vec;
}

// SPOT:55 Math.Pow(surfaceDot, 12) as the double multiply chain of SURVEY Q14.
XRT_HD double pow12(double x) { double x2 = x * x; double x4 = x2 * x2; double x8 = x4 * x4; return x8 * x4; }

// ILight.GetLightForFragment (SPOT:37-62, DIR:23-30)
XRT_HD v3 light_for_fragment(const LightRec &L, v3 position, v3 normal) {
    if (L.kind == 0) {
        v3 dirToLight = normalize(sub(mk(L.px, L.py, L.pz), position));
        float surfaceDot = dot(dirToLight, normal);
        if (surfaceDot < 0.0f) return mk(0, 0, 0);
        float lightDot = dot(neg(dirToLight), mk(L.dx, L.dy, L.dz));
        if (lightDot > L.angleCosine) {
            float spotIntensity = L.intensity * (float)((double)(lightDot - L.angleCosine) / L.decayDenom);
            v3 c = scale(scale(mk(L.cr, L.cg, L.cb), spotIntensity), surfaceDot);
            float s12 = (float)pow12((double)surfaceDot);
            return add(c, scale(mk(1.0f, 1.0f, 1.0f), s12));
        }
        return mk(0, 0, 0);
    }
    float surfaceDot = dot(mk(L.dx, L.dy, L.dz), normal);
    if (surfaceDot < 0.0f) surfaceDot = 0.0f;
    return scale(scale(mk(L.cr, L.cg, L.cb), surfaceDot), L.intensity);
}

// float % 1.0f of C# (fmod, exact) for MAT:125-136.
XRT_HD float fmod1(float x) { return x - truncf(x); }

}  // namespace xrt
