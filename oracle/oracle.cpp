// TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may build, link or call anything under oracle/.  libxrt never links this file.
//
// oracle.cpp — CPU restatement (C++17, strict binary32, -ffp-contract=off) of the reference's
// per-pixel hot path, with the REFERENCE'S OWN data structures: pointer octrees, per-query sorted
// leaf buckets, recursive CastRay, per-pixel double Viewport.Unproject.
//
//                      *** PARITY UNPINNED ***
// The reference (C#/XNA 4.0, x86) cannot be compiled or run here (no .NET toolchain, closed-source
// Microsoft.Xna.Framework 4.0.0.0) and ships no tests, golden vectors or fixtures for this path
// (SURVEY §4, §8c).  This file is therefore the build's definition of truth, reviewed against the
// cited C# line by line; its L0 math is in xna_math.h.  Floating-point model: SURVEY §9 Q14.
//
// Citations: RT = RayTraceProject/RayTraceProject/RayTracer.cs, OSM = .../Spatial/OctreeSpatialManager.cs,
// SO = .../SceneObject.cs, MO = RayTracerTypeLibrary/MeshOctree.cs, RE = .../RayExtensions.cs,
// MESH = .../Mesh.cs, MAT = .../Material.cs, TRI = .../Triangle.cs, SPOT/DIR = .../SpotLight.cs,
// DirectionalLight.cs, TMP = RayTracePipeline/TracerModelProcessor.cs.
#include "../include/xrt.h"   // POD structs only (xrt_ray, xrt_hit, xrt_material, xrt_camera, xrt_light, ...)
#include "xna_math.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

using namespace xna;

namespace {

// ---- TRI:14-25 ---------------------------------------------------------------------------------
struct Triangle {
    int id;                 // position in Mesh.Triangles[]
    Vector3 v1, v2, v3;
    Vector2 uv1, uv2, uv3;
    Vector3 n1, n2, n3;
    Vector3 surfaceNormal;
    bool convexGeometry = false;   // never assigned in the reference (TRI:22)
    Vector4 color;
};

// ---- MAT:25-269 (shading inputs + point-sampled lookup) -------------------------------------------
struct Material {
    float Reflectiveness = 0;
    bool Transparent = false;
    float RefractionIndex = 0;
    bool InterpolateNormals = false;
    bool UseTexture = false;
    int Width = 0, Height = 0;
    std::vector<uint32_t> argb;   // BitmapData.Scan0 of the Format32bppArgb lock (MAT:65)
    std::vector<uint32_t> colorData;   // Texture.ColorData: the Format32bppPArgb copy (TEX:24-33) GetColorBilinear indexes (MAT:186-189)

    static constexpr float BYTE_RECIPROCAL = 1.0f / 255.0f;   // MAT:27

    static void WrapUV(Vector2 &uv) {   // MAT:125-136; C# float % == fmodf
        if (uv.X > 1.0f) uv.X = std::fmod(uv.X, 1.0f);
        if (uv.Y > 1.0f) uv.Y = std::fmod(uv.Y, 1.0f);
        if (uv.X < 0.0f) uv.X = 1.0f + std::fmod(uv.X, 1.0f);
        if (uv.Y < 0.0f) uv.Y = 1.0f + std::fmod(uv.Y, 1.0f);
    }
    static void MirrorUV(Vector2 &uv) {   // MAT:102-123
        Vector2 o = uv;
        if (uv.X > 1.0f) uv.X = std::fmod(uv.X, 1.0f);
        if (uv.Y > 1.0f) uv.Y = std::fmod(uv.Y, 1.0f);
        if (uv.X < 0.0f) uv.X = 1 + std::fmod(uv.X, 1.0f);
        if (uv.Y < 0.0f) uv.Y = 1 + std::fmod(uv.Y, 1.0f);
        if ((int)(o.X - uv.X) % 2 == 0) uv.X = 1.0f - uv.X;
        if ((int)(o.Y - uv.Y) % 2 == 0) uv.Y = 1.0f - uv.Y;
    }
    static void ClampUV(Vector2 &uv) {   // MAT:138-143, Vector2.Clamp
        float x = uv.X; x = (x > 1.0f) ? 1.0f : x; x = (x < 0.0f) ? 0.0f : x;
        float y = uv.Y; y = (y > 1.0f) ? 1.0f : y; y = (y < 0.0f) ? 0.0f : y;
        uv.X = x; uv.Y = y;
    }
    void GetColorPoint(const Vector2 &uv, Vector3 &color) const {   // MAT:145-160
        int x = (int)(uv.X * (float)(Width - 1));
        int y = (int)(uv.Y * (float)(Height - 1));
        // The C# reads through a raw pointer (no bounds check); indices are in range for finite uv in
        // [0,1].  Guard so a NaN uv cannot fault the test process (reference behaviour: undefined).
        int64_t idx = (int64_t)Width * y + x;
        if (idx < 0 || idx >= (int64_t)argb.size()) idx = 0;
        uint32_t baseArgb = argb[(size_t)idx];
        color = V3((float)((baseArgb >> 16) & 0xFF) * BYTE_RECIPROCAL,
                   (float)((baseArgb >> 8) & 0xFF) * BYTE_RECIPROCAL,
                   (float)(baseArgb & 0xFF) * BYTE_RECIPROCAL);
    }
    // MAT:162-232.  Texture.ColorData is the Format32bppPArgb copy (TEX:24-33); for opaque textures (every
    // fixture; a 24-bpp BMP has no alpha) it equals the Format32bppArgb words, which is what is stored here.
    void GetColorBilinear(Vector2 uv, Vector3 &color) const {
        const float tdx = 1.0f / (float)Width, tdy = 1.0f / (float)Height;   // texelDensity (MAT:67)
        double remainderX = std::remainder((double)uv.X, (double)tdx);         // Math.IEEERemainder (MAT:168-169)
        double remainderY = std::remainder((double)uv.Y, (double)tdy);
        uv.X -= (float)remainderX;
        uv.Y -= (float)remainderY;
        int x = (int)(uv.X * (float)(Width - 1));
        int y = (int)(uv.Y * (float)(Height - 1));
        int x2 = (int)((uv.X + tdx) * (float)(Width - 1));
        int y2 = (int)((uv.Y + tdy) * (float)(Height - 1));
        auto texel = [&](int xx, int yy) {   // the C# indexes a managed int[] (IndexOutOfRangeException when outside); guarded here
            int64_t idx = (int64_t)Width * yy + xx;
            if (idx < 0 || idx >= (int64_t)colorData.size()) idx = 0;
            uint32_t w = colorData[(size_t)idx];   // MAT:186-189: Texture.ColorData, not Scan0
            return V3((float)((w >> 16) & 0xFF), (float)((w >> 8) & 0xFF), (float)(w & 0xFF));
        };
        Vector3 baseColor = texel(x, y), blendXColor = texel(x2, y), blendYColor = texel(x, y2), blendXYColor = texel(x2, y2);
        float dx = (float)(remainderX * (double)Width) + 0.5f;
        float dy = (float)(remainderY * (double)Height) + 0.5f;
        float invertDx = 1.0f - dx, invertDy = 1.0f - dy;
        color = ((baseColor * invertDx * invertDy) + (blendYColor * invertDx * dy) + (blendXColor * dx * invertDy) + (blendXYColor * dx * dy)) * BYTE_RECIPROCAL;
    }
    // MAT:71-100.  Returns false for what the C# answers with ArgumentException.
    bool LookupUV(Vector2 uv, int addressMode, int filtering, Vector3 &color) const {
        switch (addressMode) {
            case XRT_ADDRESS_CLAMP: ClampUV(uv); break;
            case XRT_ADDRESS_WRAP: WrapUV(uv); break;
            case XRT_ADDRESS_MIRROR: MirrorUV(uv); break;
            default: return false;
        }
        if (filtering == XRT_FILTER_POINT) GetColorPoint(uv, color);
        else if (filtering == XRT_FILTER_BILINEAR) GetColorBilinear(uv, color);
        else return false;
        return true;
    }
};

struct alignas(128) Counters {   // one cache-line pair per render thread
    uint64_t rays_closest = 0, rays_shadow = 0, hits_closest = 0, hits_shadow = 0;
    uint64_t scene_node_tests = 0, instance_visits = 0, mesh_aabb_tests = 0, mesh_queries = 0;
    uint64_t node_tests = 0, leaf_refs = 0, tri_tests = 0, shaded_hits = 0;
    void add(const Counters &o) {
        rays_closest += o.rays_closest; rays_shadow += o.rays_shadow; hits_closest += o.hits_closest;
        hits_shadow += o.hits_shadow; scene_node_tests += o.scene_node_tests;
        instance_visits += o.instance_visits; mesh_aabb_tests += o.mesh_aabb_tests;
        mesh_queries += o.mesh_queries; node_tests += o.node_tests; leaf_refs += o.leaf_refs;
        tri_tests += o.tri_tests; shaded_hits += o.shaded_hits;
    }
};

// float.CompareTo ordering of SortedList<float, ...> keys (MO:263, OSM:316): NaN sorts first and
// equals NaN; -0 == +0.
static inline bool KeyLess(float a, float b) {
    if (std::isnan(a)) return !std::isnan(b);
    if (std::isnan(b)) return false;
    return a < b;
}

// SortedList<float, List<T>>: parallel sorted arrays, binary search, O(n) insert.
template <class T>
struct SortedBuckets {
    std::vector<float> keys;
    std::vector<std::vector<T>> values;
    int IndexOfKey(float k) const {
        int lo = 0, hi = (int)keys.size() - 1;
        while (lo <= hi) {
            int mid = lo + ((hi - lo) >> 1);
            if (KeyLess(keys[mid], k)) lo = mid + 1;
            else if (KeyLess(k, keys[mid])) hi = mid - 1;
            else return mid;
        }
        return ~lo;
    }
    void AddToBucket(float k, T v) {   // MO:343-350 / OSM:472-479
        int i = IndexOfKey(k);
        if (i >= 0) { values[i].push_back(v); return; }
        i = ~i;
        keys.insert(keys.begin() + i, k);
        values.insert(values.begin() + i, std::vector<T>{v});
    }
};

// RE:42-75 IntersectsTriangleBackfaceCulling
static inline bool IntersectsTriangleBackfaceCulling(const Ray &ray, const Triangle &t, float &u, float &v, float &distance) {
    u = v = distance = 0;
    Vector3 D = ray.Direction;
    Vector3 T = ray.Position - t.v1;
    float dot = Dot(t.surfaceNormal, D);
    if (dot > 0) return false;
    Vector3 Edge1 = t.v2 - t.v1;
    Vector3 Edge2 = t.v3 - t.v1;
    Vector3 P = Cross(D, Edge2);
    Vector3 Q = Cross(T, Edge1);
    float row1 = Dot(Q, Edge2);
    float row2 = Dot(P, T);
    float row3 = Dot(Q, D);
    Vector3 result = (1.0f / Dot(P, Edge1)) * V3(row1, row2, row3);
    distance = result.X;
    u = result.Y;
    v = result.Z;
    return u >= 0 && v >= 0 && distance >= 0 && u + v <= 1;
}

// ---- MO:9-355 ------------------------------------------------------------------------------------
struct MeshOctree {
    struct TriangleIntersectionResult {   // MO:11-30
        const Triangle *triangle = nullptr;
        float u = 0, v = 0, d = 0;
        Vector3 objectSpacePosition{0, 0, 0};
        int leaf_dfs = -1;   // SURVEY Q16 parity id (not a reference field)
    };
    struct CubeNode {   // MO:32-40
        uint32_t id = 0;
        BoundingBox bounds{};
        CubeNode *parent = nullptr;
        std::unique_ptr<CubeNode> children[8];
        bool hasChildren = false;
        std::vector<const Triangle *> containingObjects;
        int dfs = -1, depthLevel = 0;
    };
    std::unique_ptr<CubeNode> root;
    int itemTreshold = 50;   // MO:42
    std::vector<const Triangle *> objects;
    uint32_t depth = 0;
    int nodeCount = 0;
    int maxLevel = 0;
    bool overflow = false;

    void Build() {   // MO:56-82
        depth = 0;
        BoundingBox box{V3(FLT_MAX, FLT_MAX, FLT_MAX), V3(-FLT_MAX, -FLT_MAX, -FLT_MAX)};   // float.MinValue == -MaxValue
        root.reset(new CubeNode());
        root->id = 0;
        root->containingObjects = objects;
        for (size_t i = 0; i < objects.size(); i++) {
            box.Min = Min(box.Min, objects[i]->v1);
            box.Min = Min(box.Min, objects[i]->v2);
            box.Min = Min(box.Min, objects[i]->v3);
            box.Max = Max(box.Max, objects[i]->v1);
            box.Max = Max(box.Max, objects[i]->v2);
            box.Max = Max(box.Max, objects[i]->v3);
        }
        root->bounds = box;
        BuildTree(root.get(), 0);
        int counter = 0;
        Number(root.get(), counter, 0);
        nodeCount = counter;
    }
    int builtNodes = 1;
    void BuildTree(CubeNode *parent, int level) {   // MO:84-96
        if (overflow) return;
        if ((int)parent->containingObjects.size() > itemTreshold) {
            // The C# recursion has no depth limit and overflows the stack when more than
            // itemTreshold triangles share a vertex; the oracle stops at 64 levels / 2^24 nodes and flags it.
            builtNodes += 8;
            if (level >= 64 || builtNodes > (1 << 24)) { overflow = true; return; }
            depth++;
            SplitCuboid(parent);
            for (int i = 0; i < 8; i++) BuildTree(parent->children[i].get(), level + 1);
        }
    }
    void SplitCuboid(CubeNode *parent) {   // MO:204-236
        Vector3 cubeSize = (parent->bounds.Max - parent->bounds.Min) / 2.0f;
        parent->hasChildren = true;
        uint32_t index = 0;
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int k = 0; k < 2; k++) {
                    CubeNode *child = new CubeNode();
                    Vector3 cubePosition = parent->bounds.Min + V3(cubeSize.X * (float)i, cubeSize.Y * (float)j, cubeSize.Z * (float)k);
                    child->bounds = BoundingBox{cubePosition, cubePosition + cubeSize};
                    child->parent = parent;
                    child->id = parent->id + index + 2;
                    parent->children[index++].reset(child);
                    for (size_t o = 0; o < parent->containingObjects.size(); o++) {
                        const Triangle *t = parent->containingObjects[o];
                        if (ContainsPoint(child->bounds, t->v1) || ContainsPoint(child->bounds, t->v2) || ContainsPoint(child->bounds, t->v3))
                            child->containingObjects.push_back(t);
                    }
                }
    }
    void Number(CubeNode *n, int &counter, int level) {
        n->dfs = counter++;
        n->depthLevel = level;
        if (level > maxLevel) maxLevel = level;
        if (n->hasChildren) for (int i = 0; i < 8; i++) Number(n->children[i].get(), counter, level + 1);
    }

    // MO:328-353
    void GetRayCubeNodeIntersections(const Ray &ray, CubeNode *current, SortedBuckets<CubeNode *> &cuboids, Counters &c) const {
        float result;
        c.node_tests++;
        if (Intersects(current->bounds, ray, result)) {
            if (current->hasChildren) {
                for (int i = 0; i < 8; i++) GetRayCubeNodeIntersections(ray, current->children[i].get(), cuboids, c);
            } else {
                cuboids.AddToBucket(result, current);
            }
        }
    }
    // MO:259-326
    bool GetRayIntersection(const Ray &ray, TriangleIntersectionResult &result, const Triangle *ignoreTriangle, Counters &c) const {
        c.mesh_queries++;
        SortedBuckets<CubeNode *> cubeoids;
        GetRayCubeNodeIntersections(ray, root.get(), cubeoids, c);
        if (cubeoids.keys.empty()) return false;
        size_t cubeoidIndex = 0;
        float minDistance = FLT_MAX;
        float intersectionU = 0, intersectionV = 0;
        const Triangle *intersectedTriangle = nullptr;
        int leaf = -1;
        bool intersectionFound = false;
        while (!intersectionFound && cubeoidIndex < cubeoids.keys.size()) {
            const std::vector<CubeNode *> &cuboidGroup = cubeoids.values[cubeoidIndex++];
            for (size_t k = 0; k < cuboidGroup.size(); k++) {
                const std::vector<const Triangle *> &triangles = cuboidGroup[k]->containingObjects;
                for (size_t i = 0; i < triangles.size(); i++) {
                    c.leaf_refs++;
                    if (ignoreTriangle == nullptr || ignoreTriangle != triangles[i]) {
                        float currentU, currentV, distance;
                        c.tri_tests++;
                        if (IntersectsTriangleBackfaceCulling(ray, *triangles[i], currentU, currentV, distance) && distance < minDistance) {
                            minDistance = distance;
                            intersectionU = currentU;
                            intersectionV = currentV;
                            intersectedTriangle = triangles[i];
                            leaf = cuboidGroup[k]->dfs;
                            intersectionFound = true;
                        }
                    }
                }
            }
        }
        if (intersectionFound) {
            Vector3 p1 = intersectedTriangle->v2 - intersectedTriangle->v1;
            Vector3 p2 = intersectedTriangle->v3 - intersectedTriangle->v1;
            Vector3 interpolatedPosition = intersectedTriangle->v1 + (p1 * intersectionU) + (p2 * intersectionV);
            result.triangle = intersectedTriangle;
            result.u = intersectionU;
            result.v = intersectionV;
            result.d = minDistance;
            result.objectSpacePosition = interpolatedPosition;
            result.leaf_dfs = leaf;
        }
        return intersectionFound;
    }
};

// ---- MESH:9-40 -----------------------------------------------------------------------------------
struct Mesh {
    int id = -1;
    std::vector<Triangle> Triangles;
    Material MeshMaterial;
    BoundingBox MeshBoundingBox{};
    MeshOctree Octree;
    void Init(int threshold) {   // MESH:27-32
        Octree = MeshOctree();
        Octree.itemTreshold = threshold;
        for (auto &t : Triangles) Octree.objects.push_back(&t);
        Octree.Build();
    }
    bool RayIntersects(const Ray &ray) const {   // MESH:34-39
        float distance;
        return Intersects(MeshBoundingBox, ray, distance);
    }
};

// ---- SO:12-199 (transform + mesh list only) ---------------------------------------------------------
struct SceneObject {
    int id = -1;
    std::vector<Mesh *> Meshes;   // shared by reference between instances (SO:126-127)
    Matrix World, InverseWorld;
    BoundingBox BoundingBox_{}, WorldBoundingBox{};
};

struct IntersectionResult {   // OSM:11-33
    const Mesh *mesh = nullptr;
    const Triangle *triangle = nullptr;
    const SceneObject *object = nullptr;   // not a reference field; for the parity record only
    int leaf_dfs = -1;
    float u = 0, v = 0, d = 0;
    Vector3 worldPosition{0, 0, 0};
};

// ---- OSM:35-486 ------------------------------------------------------------------------------------
struct OctreeSpatialManager {
    struct CubeNode {   // OSM:37-48
        uint32_t id = 0;
        BoundingBox bounds{};
        std::unique_ptr<CubeNode> children[8];
        bool hasChildren = false;
        std::vector<SceneObject *> containingObjects;
        int dfs = -1, depthLevel = 0;
    };
    std::unique_ptr<CubeNode> root;
    int itemTreshold = 20;   // OSM:50
    std::vector<SceneObject *> objects;
    bool overflow = false;
    int nodeCount = 0;

    void Build() {   // OSM:64-99
        BoundingBox box{V3(0, 0, 0), V3(0, 0, 0)};
        root.reset(new CubeNode());
        root->containingObjects = objects;
        for (size_t i = 0; i < objects.size(); i++) {
            BoundingBox transformedBox = objects[i]->BoundingBox_;
            const Matrix &world = objects[i]->World;
            transformedBox.Min = Transform(transformedBox.Min, world);
            transformedBox.Max = Transform(transformedBox.Max, world);
            BoundingBox objectBox{Min(transformedBox.Min, transformedBox.Max), Max(transformedBox.Min, transformedBox.Max)};
            if (i == 0) box = objectBox;
            else box = CreateMerged(box, objectBox);
        }
        root->bounds = box;
        BuildTree(root.get(), 0);
        int counter = 0;
        Number(root.get(), counter, 0);
        nodeCount = counter;
    }
    int builtNodes = 1;
    void BuildTree(CubeNode *parent, int level) {   // OSM:101-113
        if (overflow) return;
        if ((int)parent->containingObjects.size() > itemTreshold) {
            builtNodes += 8;
            if (level >= 24 || builtNodes > (1 << 20)) { overflow = true; return; }   // the C# would recurse forever (stack overflow)
            SplitCuboid(parent);
            for (int i = 0; i < 8; i++) BuildTree(parent->children[i].get(), level + 1);
        }
    }
    void SplitCuboid(CubeNode *parent) {   // OSM:218-248
        Vector3 cubeSize = (parent->bounds.Max - parent->bounds.Min) / 2.0f;
        parent->hasChildren = true;
        uint32_t index = 0;
        for (int i = 0; i < 2; i++)
            for (int j = 0; j < 2; j++)
                for (int k = 0; k < 2; k++) {
                    CubeNode *child = new CubeNode();
                    Vector3 cubePosition = parent->bounds.Min + V3(cubeSize.X * (float)i, cubeSize.Y * (float)j, cubeSize.Z * (float)k);
                    child->bounds = BoundingBox{cubePosition, cubePosition + cubeSize};
                    child->id = parent->id + index + 2;
                    parent->children[index++].reset(child);
                    for (size_t o = 0; o < parent->containingObjects.size(); o++)
                        if (Intersects(parent->containingObjects[o]->WorldBoundingBox, child->bounds))   // OSM:240
                            child->containingObjects.push_back(parent->containingObjects[o]);
                }
    }
    void Number(CubeNode *n, int &counter, int level) {
        n->dfs = counter++;
        n->depthLevel = level;
        if (n->hasChildren) for (int i = 0; i < 8; i++) Number(n->children[i].get(), counter, level + 1);
    }
    // OSM:457-482
    void GetRayCubeNodeIntersections(const Ray &ray, CubeNode *current, SortedBuckets<CubeNode *> &cuboids, Counters &c) const {
        float result;
        c.scene_node_tests++;
        if (Intersects(current->bounds, ray, result)) {
            if (current->hasChildren) {
                for (int i = 0; i < 8; i++) GetRayCubeNodeIntersections(ray, current->children[i].get(), cuboids, c);
            } else {
                cuboids.AddToBucket(result, current);
            }
        }
    }
    // OSM:312-455.  `ignoreObject` is the dead parameter of SURVEY Q8: a Mesh reference compared with
    // an ISpatialBody reference (OSM:343) is never equal, so it is not taken here at all.
    bool GetRayIntersection(const Ray &ray, IntersectionResult &result, const Triangle *ignoreTriangle, Counters &c) const {
        SortedBuckets<CubeNode *> cubeoids;
        GetRayCubeNodeIntersections(ray, root.get(), cubeoids, c);
        if (cubeoids.keys.empty()) return false;
        size_t cubeoidIndex = 0;
        float minDistance = FLT_MAX;
        bool intersectionFound = false;
        const Mesh *intersectedMesh = nullptr;
        const SceneObject *intersectedSceneObject = nullptr;
        MeshOctree::TriangleIntersectionResult intersectedTriangleResult, triangleResult;
        while (!intersectionFound && cubeoidIndex < cubeoids.keys.size()) {
            const std::vector<CubeNode *> &cuboidGroup = cubeoids.values[cubeoidIndex++];
            for (size_t k = 0; k < cuboidGroup.size(); k++) {
                const std::vector<SceneObject *> &objs = cuboidGroup[k]->containingObjects;
                for (size_t i = 0; i < objs.size(); i++) {
                    const SceneObject *sceneObject = objs[i];
                    c.instance_visits++;
                    const Matrix &inverseWorld = sceneObject->InverseWorld;
                    Vector3 rayDirPosition = ray.Position + ray.Direction;            // OSM:358
                    Vector3 v1 = Transform(ray.Position, inverseWorld);              // OSM:360
                    Vector3 v2 = Transform(rayDirPosition, inverseWorld);            // OSM:361
                    rayDirPosition = v2 - v1;                                        // OSM:362
                    Ray transformedRay{v1, Normalize(rayDirPosition)};               // OSM:363-364
                    for (size_t meshIndex = 0; meshIndex < sceneObject->Meshes.size(); meshIndex++) {
                        c.mesh_aabb_tests++;
                        if (sceneObject->Meshes[meshIndex]->RayIntersects(transformedRay)) {
                            if (sceneObject->Meshes[meshIndex]->Octree.GetRayIntersection(transformedRay, triangleResult, ignoreTriangle, c) &&
                                triangleResult.d < minDistance) {
                                minDistance = triangleResult.d;
                                intersectedTriangleResult = triangleResult;
                                intersectedMesh = sceneObject->Meshes[meshIndex];
                                intersectedSceneObject = sceneObject;
                                intersectionFound = true;
                            }
                        }
                    }
                }
            }
        }
        if (intersectionFound) {
            result.mesh = intersectedMesh;
            result.triangle = intersectedTriangleResult.triangle;
            result.object = intersectedSceneObject;
            result.leaf_dfs = intersectedTriangleResult.leaf_dfs;
            result.u = intersectedTriangleResult.u;
            result.v = intersectedTriangleResult.v;
            result.d = minDistance;
            result.worldPosition = Transform(intersectedTriangleResult.objectSpacePosition, intersectedSceneObject->World);   // OSM:441-443
        }
        return intersectionFound;
    }
};

// ---- lights: SPOT:10-63, DIR:10-31 ---------------------------------------------------------------
// Math.Pow(x, 12) (SPOT:55) is defined as this double multiply chain on both CPU and GPU.
static inline double Pow12(double x) { double x2 = x * x; double x4 = x2 * x2; double x8 = x4 * x4; return x8 * x4; }

struct Light {
    int kind;
    Vector3 Position, Direction, Color;
    float Intensity, DecayExponent;
    float angleCosine;     // SPOT:25 (float)Math.Cos(spotAngle * 0.5f)
    double decayDenom;     // Math.Pow((1 - angleCosine), DecayExponent), SPOT:54 — constant per light
    bool IsPositionable() const { return kind == XRT_LIGHT_SPOT; }
    Vector3 GetLightForFragment(Vector3 position, Vector3 normal) const {
        if (kind == XRT_LIGHT_SPOT) {   // SPOT:37-62
            Vector3 dirToLight = Normalize(Position - position);
            float surfaceDot = Dot(dirToLight, normal);
            if (surfaceDot < 0.0f) return V3(0, 0, 0);
            float lightDot = Dot(-dirToLight, Direction);
            if (lightDot > angleCosine) {
                float spotIntensity = Intensity * (float)((double)(lightDot - angleCosine) / decayDenom);
                return Color * spotIntensity * surfaceDot + (V3(1, 1, 1) * (float)Pow12((double)surfaceDot));
            }
            return V3(0, 0, 0);
        }
        // DIR:23-30
        float surfaceDot = Dot(Direction, normal);
        if (surfaceDot < 0.0f) surfaceDot = 0.0f;
        return Color * surfaceDot * Intensity;
    }
};

static Light MakeLight(const xrt_light &l) {
    Light r;
    r.kind = l.kind;
    r.Position = V3(l.position[0], l.position[1], l.position[2]);
    r.Direction = V3(l.direction[0], l.direction[1], l.direction[2]);
    r.Color = V3(l.color[0], l.color[1], l.color[2]);
    r.Intensity = l.intensity;
    r.DecayExponent = l.decay_exponent;
    r.angleCosine = (float)std::cos((double)(l.spot_angle * 0.5f));
    r.decayDenom = std::pow((double)(1 - r.angleCosine), (double)r.DecayExponent);
    if (l.kind == XRT_LIGHT_DIRECTIONAL) r.Position = V3(0, 0, 0);   // DIR:14
    return r;
}

static Matrix ToMatrix(const float *m) {
    return Matrix{m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10], m[11], m[12], m[13], m[14], m[15]};
}

}  // namespace

// ---- the scene handle -------------------------------------------------------------------------------
struct orc_scene {
    std::vector<std::unique_ptr<Mesh>> meshes;
    std::vector<std::unique_ptr<SceneObject>> objects;
    OctreeSpatialManager manager;
    bool built = false;
    std::string error;
};

namespace {

// ---- RT:13-751 -----------------------------------------------------------------------------------------
struct RayTracer {
    const orc_scene *scene;
    std::vector<Light> lights;
    int MaxReflections = 0, AddressMode = XRT_ADDRESS_WRAP, TextureFiltering = XRT_FILTER_POINT;
    int MultisampleQuality = 0;
    Matrix view, proj;
    Viewport viewport;
    bool bad_lookup = false;

    // RT:465-502
    float IsLightPathObstructed(const IntersectionResult &result, const Light &light, Counters &c) const {
        Vector3 dirToLight;
        float distanceToLight;
        if (light.IsPositionable()) {
            dirToLight = light.Position - result.worldPosition;
            distanceToLight = Length(dirToLight);
            dirToLight = Normalize(dirToLight);
        } else {
            dirToLight = -light.Direction;
            distanceToLight = FLT_MAX;
        }
        Ray shadowRay{result.worldPosition, dirToLight};
        IntersectionResult intersectionResult;
        c.rays_shadow++;
        if (scene->manager.GetRayIntersection(shadowRay, intersectionResult, result.triangle, c)) {
            c.hits_shadow++;
            if (intersectionResult.d < distanceToLight) {
                if (intersectionResult.mesh->MeshMaterial.Transparent) return intersectionResult.triangle->color.W;
                return 1;
            }
        }
        return 0;
    }

    Vector3 SurfaceColor(const IntersectionResult &result, const Material &material) {   // RT:568-581 / 711-724
        Vector3 surfaceColor;
        if (material.UseTexture) {
            Vector2 uv1 = result.triangle->uv2 - result.triangle->uv1;
            Vector2 uv2 = result.triangle->uv3 - result.triangle->uv1;
            Vector2 interpolatedUV = result.triangle->uv1 + (uv1 * result.u) + (uv2 * result.v);
            if (!material.LookupUV(interpolatedUV, AddressMode, TextureFiltering, surfaceColor)) { bad_lookup = true; surfaceColor = V3(0, 0, 0); }
        } else {
            surfaceColor = V3(result.triangle->color.X, result.triangle->color.Y, result.triangle->color.Z);
        }
        return surfaceColor;
    }

    // RT:506-737.  colorVectorOut receives the value handed to `new Color(...)` (RT:705/726/732).
    // The debug `addRayPoints` list (RT:543,701,740-747) does not influence the image and is dropped.
    void CastRay(Ray &ray, uint32_t &resultColor, int iteration, const Triangle *origin, float currentRefIndex, Counters &c, Vector3 *colorVectorOut) {
        IntersectionResult result;
        c.rays_closest++;
        if (scene->manager.GetRayIntersection(ray, result, origin, c)) {
            c.hits_closest++;
            c.shaded_hits++;
            const Material &material = result.mesh->MeshMaterial;
            Vector3 fragmentNormal;
            if (material.InterpolateNormals) {   // RT:520-527
                Vector3 n1 = result.triangle->n2 - result.triangle->n1;
                Vector3 n2 = result.triangle->n3 - result.triangle->n1;
                fragmentNormal = result.triangle->n1 + (n1 * result.u) + (n2 * result.v);
                fragmentNormal = Normalize(fragmentNormal);
            } else {
                fragmentNormal = result.triangle->surfaceNormal;
            }
            Vector3 lightResult = V3(0, 0, 0);   // RT:534-542
            for (size_t i = 0; i < lights.size(); i++) {
                float lightAmount = IsLightPathObstructed(result, lights[i], c);
                if (lightAmount != 1.0f)
                    lightResult = lightResult + lights[i].GetLightForFragment(result.worldPosition, fragmentNormal) * (1.0f - lightAmount);
            }
            if (iteration < MaxReflections) {   // RT:545-707
                Ray r;
                r.Position = result.worldPosition;
                r.Direction = Reflect(ray.Direction, fragmentNormal);
                r.Direction = Normalize(r.Direction);
                uint32_t reflectionColor;
                // convexGeometry is never set (TRI:22), so RT:557 is unreachable; both branches pass
                // result.triangle as origin.
                CastRay(r, reflectionColor, iteration + 1, result.triangle, currentRefIndex, c, nullptr);
                Vector3 surfaceColor = SurfaceColor(result, material);
                Vector3 colorVector = Lerp(ColorToVector3(reflectionColor), surfaceColor, 1.0f - material.Reflectiveness) * lightResult;   // RT:584
                if (material.Transparent) {   // RT:586-702
                    float n1, n2;
                    if (currentRefIndex == material.RefractionIndex) { n1 = 1.0f; n2 = currentRefIndex; }
                    else { n1 = material.RefractionIndex; n2 = 1.0f; }
                    float cos1 = Dot(fragmentNormal, -ray.Direction);
                    // Math.Pow(x, 2.0) is x*x in double (exact for float-valued x); SURVEY Q14.
                    double ratio = (double)(n1 / n2);
                    double c1 = (double)cos1;
                    float cos2 = (float)std::sqrt(1 - (ratio * ratio) * (1 - (c1 * c1)));
                    Vector3 refract;
                    if (cos1 >= 0) refract = (n1 / n2) * ray.Direction + ((n1 / n2) * cos1 - cos2) * fragmentNormal;
                    else refract = (n1 / n2) * ray.Direction - ((n1 / n2) * cos1 - cos2) * fragmentNormal;
                    ray.Position = result.worldPosition;   // RT:692-694 mutates the caller's ray (Q15)
                    ray.Direction = Normalize(refract);
                    uint32_t refractColor;
                    CastRay(ray, refractColor, iteration + 1, result.triangle, n2, c, nullptr);
                    colorVector = Lerp(ColorToVector3(refractColor), colorVector, result.triangle->color.W);
                }
                if (colorVectorOut) *colorVectorOut = colorVector;
                resultColor = ColorFromVector3(colorVector);   // RT:705
            } else {   // RT:708-727
                Vector3 surfaceColor = SurfaceColor(result, material);
                Vector3 colorVector = lightResult * surfaceColor;
                if (colorVectorOut) *colorVectorOut = colorVector;
                resultColor = ColorFromVector3(colorVector);
            }
        } else {
            if (colorVectorOut) *colorVectorOut = V3(0, 0, 0);
            resultColor = ColorFromVector3(V3(0, 0, 0));   // RT:732
        }
    }

    Ray PrimaryRay(float sx, float sy) const {   // RT:412-421 / 224-232
        Ray ray;
        Vector3 screenSpaceCoord = V3(sx, sy, 0);
        ray.Position = Unproject(viewport, screenSpaceCoord, proj, view, Identity());
        screenSpaceCoord.Z = 1;
        Vector3 vector2 = Unproject(viewport, screenSpaceCoord, proj, view, Identity());
        ray.Direction = Normalize(vector2 - ray.Position);
        return ray;
    }

    // RT:215-311, including the RT:305 bug (the lower-right recursion writes urColor).
    static constexpr float TRESHOLD = 0.5f;   // RT:340
    void GetColorForQuadrant(float centerX, float centerY, float size, int iteration, uint32_t &result, Counters &c) {
        float quarterSize = size * 0.25f;
        uint32_t ulColor, urColor, llColor, lrColor;
        Ray ray;
        ray = PrimaryRay(centerX - quarterSize, centerY - quarterSize); CastRay(ray, ulColor, 0, nullptr, 1.0f, c, nullptr);
        ray = PrimaryRay(centerX + quarterSize, centerY - quarterSize); CastRay(ray, urColor, 0, nullptr, 1.0f, c, nullptr);
        ray = PrimaryRay(centerX - quarterSize, centerY + quarterSize); CastRay(ray, llColor, 0, nullptr, 1.0f, c, nullptr);
        ray = PrimaryRay(centerX + quarterSize, centerY + quarterSize); CastRay(ray, lrColor, 0, nullptr, 1.0f, c, nullptr);
        if (iteration < MultisampleQuality) {
            Vector3 ul = ColorToVector3(ulColor), ur = ColorToVector3(urColor), ll = ColorToVector3(llColor), lr = ColorToVector3(lrColor);
            Vector3 average = (ul + ur + ll + lr) / 4.0f;
            float average_length = Length(average);
            if (std::fabs(average_length - Length(ul)) > TRESHOLD) GetColorForQuadrant(centerX - quarterSize, centerY - quarterSize, size / 2.0f, iteration + 1, ulColor, c);
            if (std::fabs(average_length - Length(ur)) > TRESHOLD) GetColorForQuadrant(centerX + quarterSize, centerY - quarterSize, size / 2.0f, iteration + 1, urColor, c);
            if (std::fabs(average_length - Length(ll)) > TRESHOLD) GetColorForQuadrant(centerX - quarterSize, centerY + quarterSize, size / 2.0f, iteration + 1, llColor, c);
            if (std::fabs(average_length - Length(lr)) > TRESHOLD) GetColorForQuadrant(centerX + quarterSize, centerY + quarterSize, size / 2.0f, iteration + 1, urColor, c);   // RT:305
        }
        result = ColorFromVector3((ColorToVector3(ulColor) + ColorToVector3(urColor) + ColorToVector3(llColor) + ColorToVector3(lrColor)) / 4.0f);
    }
    // XRT_MS_FIXED16: the 16 positions the level-1 subdivision samples, each corner
    // = mean of its 4 sub-rays quantised (RT:309), pixel = mean of the 4 corner colours quantised.
    void GetColorFixed16(float centerX, float centerY, uint32_t &result, Counters &c) {
        uint32_t corner[4];
        const float sx[4] = {-1, 1, -1, 1}, sy[4] = {-1, -1, 1, 1};
        for (int q = 0; q < 4; q++) {
            float cx = centerX + sx[q] * 0.25f, cy = centerY + sy[q] * 0.25f;
            uint32_t sub[4];
            for (int s = 0; s < 4; s++) {
                Ray ray = PrimaryRay(cx + sx[s] * 0.125f, cy + sy[s] * 0.125f);
                CastRay(ray, sub[s], 0, nullptr, 1.0f, c, nullptr);
            }
            corner[q] = ColorFromVector3((ColorToVector3(sub[0]) + ColorToVector3(sub[1]) + ColorToVector3(sub[2]) + ColorToVector3(sub[3])) / 4.0f);
        }
        result = ColorFromVector3((ColorToVector3(corner[0]) + ColorToVector3(corner[1]) + ColorToVector3(corner[2]) + ColorToVector3(corner[3])) / 4.0f);
    }
};

static void FillHit(xrt_hit &h, bool found, const IntersectionResult &r) {
    std::memset(&h, 0, sizeof(h));
    h.hit = found ? 1 : 0;
    h.object = h.mesh = h.tri = h.leaf = -1;
    if (!found) return;
    h.object = r.object ? r.object->id : -1;
    h.mesh = r.mesh ? r.mesh->id : -1;
    h.tri = r.triangle->id;
    h.leaf = r.leaf_dfs;
    h.u = r.u; h.v = r.v; h.d = r.d;
    h.wx = r.worldPosition.X; h.wy = r.worldPosition.Y; h.wz = r.worldPosition.Z;
}

static void FillStats(xrt_stats *s, const Counters &c, uint64_t pixels, double ms) {
    if (!s) return;
    std::memset(s, 0, sizeof(*s));
    s->rays_closest = c.rays_closest; s->rays_shadow = c.rays_shadow;
    s->hits_closest = c.hits_closest; s->hits_shadow = c.hits_shadow;
    s->scene_node_tests = c.scene_node_tests; s->instance_visits = c.instance_visits;
    s->mesh_aabb_tests = c.mesh_aabb_tests; s->mesh_queries = c.mesh_queries;
    s->node_tests = c.node_tests; s->leaf_refs = c.leaf_refs; s->tri_tests = c.tri_tests;
    s->shaded_hits = c.shaded_hits; s->pixels = pixels;
    // B_ray = 32 + 32 N_node + 4 N_ref + 48 N_tri + 48 (+ two-level terms) ; shading 76+4 per
    // shaded hit ; 4 B pixel write.
    uint64_t rays = c.rays_closest + c.rays_shadow, hits = c.hits_closest + c.hits_shadow;
    s->algorithmic_bytes = rays * (32 + 48) + 32 * c.node_tests + 4 * c.leaf_refs + 48 * c.tri_tests +
                           32 * c.scene_node_tests + 64 * c.instance_visits + 24 * c.mesh_aabb_tests + 64 * hits +
                           80 * c.shaded_hits + 4 * pixels;
    s->ms_total = ms;
}

}  // namespace

// ---- C API (ctypes) ------------------------------------------------------------------------------------
extern "C" {

orc_scene *orc_scene_create() { return new orc_scene(); }
void orc_scene_destroy(orc_scene *s) { delete s; }
const char *orc_last_error(orc_scene *s) { return s->error.c_str(); }

int orc_scene_add_mesh(orc_scene *s, const float *v, const float *n, const float *uv, const float *surf_n, const float *color,
                       int32_t ntri, const xrt_material *m, const float bbox[6]) {
    auto mesh = std::make_unique<Mesh>();
    mesh->id = (int)s->meshes.size();
    mesh->Triangles.resize(ntri);
    for (int i = 0; i < ntri; i++) {
        Triangle &t = mesh->Triangles[i];
        t.id = i;
        t.v1 = V3(v[i * 9 + 0], v[i * 9 + 1], v[i * 9 + 2]);
        t.v2 = V3(v[i * 9 + 3], v[i * 9 + 4], v[i * 9 + 5]);
        t.v3 = V3(v[i * 9 + 6], v[i * 9 + 7], v[i * 9 + 8]);
        if (n) {
            t.n1 = V3(n[i * 9 + 0], n[i * 9 + 1], n[i * 9 + 2]);
            t.n2 = V3(n[i * 9 + 3], n[i * 9 + 4], n[i * 9 + 5]);
            t.n3 = V3(n[i * 9 + 6], n[i * 9 + 7], n[i * 9 + 8]);
        } else t.n1 = t.n2 = t.n3 = V3(0, 0, 0);
        if (uv) { t.uv1 = Vector2{uv[i * 6 + 0], uv[i * 6 + 1]}; t.uv2 = Vector2{uv[i * 6 + 2], uv[i * 6 + 3]}; t.uv3 = Vector2{uv[i * 6 + 4], uv[i * 6 + 5]}; }
        else t.uv1 = t.uv2 = t.uv3 = Vector2{0, 0};
        t.surfaceNormal = V3(surf_n[i * 3 + 0], surf_n[i * 3 + 1], surf_n[i * 3 + 2]);
        if (color) t.color = Vector4{color[i * 4 + 0], color[i * 4 + 1], color[i * 4 + 2], color[i * 4 + 3]};
        else t.color = Vector4{1, 1, 1, 1};
    }
    Material &mm = mesh->MeshMaterial;
    mm.Reflectiveness = m->reflectiveness;
    mm.Transparent = m->transparent != 0;
    mm.RefractionIndex = m->refraction_index;
    mm.InterpolateNormals = m->interpolate_normals != 0;
    mm.UseTexture = m->use_texture != 0;
    if (mm.UseTexture) {
        if (!m->tex_argb || m->tex_width <= 0 || m->tex_height <= 0) { s->error = "use_texture without texels"; return -1; }
        mm.Width = m->tex_width; mm.Height = m->tex_height;
        mm.argb.assign(m->tex_argb, m->tex_argb + (size_t)m->tex_width * m->tex_height);
        const uint32_t *pa = m->tex_pargb ? m->tex_pargb : m->tex_argb;
        mm.colorData.assign(pa, pa + (size_t)m->tex_width * m->tex_height);
    }
    mesh->MeshBoundingBox = BoundingBox{V3(bbox[0], bbox[1], bbox[2]), V3(bbox[3], bbox[4], bbox[5])};
    s->meshes.push_back(std::move(mesh));
    return (int)s->meshes.size() - 1;
}

int orc_scene_add_object(orc_scene *s, const int32_t *mesh_ids, int32_t n, const float world[16], const float inv_world[16],
                         const float bbox[6], const float world_bbox[6]) {
    auto o = std::make_unique<SceneObject>();
    o->id = (int)s->objects.size();
    for (int i = 0; i < n; i++) {
        if (mesh_ids[i] < 0 || mesh_ids[i] >= (int)s->meshes.size()) { s->error = "bad mesh id"; return -1; }
        o->Meshes.push_back(s->meshes[mesh_ids[i]].get());
    }
    o->World = ToMatrix(world);
    o->InverseWorld = ToMatrix(inv_world);
    o->BoundingBox_ = BoundingBox{V3(bbox[0], bbox[1], bbox[2]), V3(bbox[3], bbox[4], bbox[5])};
    o->WorldBoundingBox = BoundingBox{V3(world_bbox[0], world_bbox[1], world_bbox[2]), V3(world_bbox[3], world_bbox[4], world_bbox[5])};
    s->objects.push_back(std::move(o));
    return (int)s->objects.size() - 1;
}

int orc_scene_build(orc_scene *s, int32_t mesh_threshold, int32_t scene_threshold) {
    if (mesh_threshold <= 0) mesh_threshold = 50;
    if (scene_threshold <= 0) scene_threshold = 20;
    for (auto &m : s->meshes) {
        m->Init(mesh_threshold);
        if (m->Octree.overflow) { s->error = "MeshOctree recursion would not terminate (more than threshold triangles share a vertex, SURVEY Q5)"; return -2; }
    }
    s->manager = OctreeSpatialManager();
    s->manager.itemTreshold = scene_threshold;
    for (auto &o : s->objects) s->manager.objects.push_back(o.get());
    s->manager.Build();
    if (s->manager.overflow) { s->error = "scene octree recursion would not terminate"; return -2; }
    s->built = true;
    return 0;
}

// Tree inspection in DFS pre-order; mesh_id == -1: scene octree (refs = object ids).
static void DumpMeshNode(const MeshOctree::CubeNode *n, xrt_node_info *nodes, int64_t &ni, int32_t *refs, int64_t &ri) {
    if (nodes) {
        xrt_node_info &o = nodes[ni];
        o.bmin[0] = n->bounds.Min.X; o.bmin[1] = n->bounds.Min.Y; o.bmin[2] = n->bounds.Min.Z;
        o.bmax[0] = n->bounds.Max.X; o.bmax[1] = n->bounds.Max.Y; o.bmax[2] = n->bounds.Max.Z;
        o.is_leaf = n->hasChildren ? 0 : 1; o.count = (int)n->containingObjects.size(); o.dfs_index = n->dfs; o.depth = n->depthLevel;
        o.first_ref = n->hasChildren ? -1 : (int)ri; o.reserved = 0;
    }
    ni++;
    if (!n->hasChildren) {
        for (auto *t : n->containingObjects) { if (refs) refs[ri] = t->id; ri++; }
    } else for (int i = 0; i < 8; i++) DumpMeshNode(n->children[i].get(), nodes, ni, refs, ri);
}
static void DumpSceneNode(const OctreeSpatialManager::CubeNode *n, xrt_node_info *nodes, int64_t &ni, int32_t *refs, int64_t &ri) {
    if (nodes) {
        xrt_node_info &o = nodes[ni];
        o.bmin[0] = n->bounds.Min.X; o.bmin[1] = n->bounds.Min.Y; o.bmin[2] = n->bounds.Min.Z;
        o.bmax[0] = n->bounds.Max.X; o.bmax[1] = n->bounds.Max.Y; o.bmax[2] = n->bounds.Max.Z;
        o.is_leaf = n->hasChildren ? 0 : 1; o.count = (int)n->containingObjects.size(); o.dfs_index = n->dfs; o.depth = n->depthLevel;
        o.first_ref = n->hasChildren ? -1 : (int)ri; o.reserved = 0;
    }
    ni++;
    if (!n->hasChildren) {
        for (auto *t : n->containingObjects) { if (refs) refs[ri] = t->id; ri++; }
    } else for (int i = 0; i < 8; i++) DumpSceneNode(n->children[i].get(), nodes, ni, refs, ri);
}
int orc_scene_get_tree(const orc_scene *s, int32_t mesh_id, xrt_node_info *nodes, int64_t *n_nodes, int32_t *refs, int64_t *n_refs) {
    if (!s->built) return -1;
    int64_t ni = 0, ri = 0;
    if (mesh_id < 0) DumpSceneNode(s->manager.root.get(), nodes, ni, refs, ri);
    else {
        if (mesh_id >= (int)s->meshes.size()) return -1;
        DumpMeshNode(s->meshes[mesh_id]->Octree.root.get(), nodes, ni, refs, ri);
    }
    *n_nodes = ni; *n_refs = ri;
    return 0;
}

static const Triangle *ResolveIgnore(const orc_scene *s, const xrt_ray &r) {
    if (r.ignore_tri < 0 || r.ignore_mesh < 0 || r.ignore_mesh >= (int)s->meshes.size()) return nullptr;
    const Mesh *m = s->meshes[r.ignore_mesh].get();
    if (r.ignore_tri >= (int)m->Triangles.size()) return nullptr;
    return &m->Triangles[r.ignore_tri];
}

// ISpatialManager.GetRayIntersection over a batch (ISM:15)
int orc_scene_intersect(const orc_scene *s, const xrt_ray *rays, int64_t n, xrt_hit *hits, xrt_stats *stats) {
    if (!s->built) return -1;
    Counters c;
    for (int64_t i = 0; i < n; i++) {
        Ray ray{V3(rays[i].o[0], rays[i].o[1], rays[i].o[2]), V3(rays[i].d[0], rays[i].d[1], rays[i].d[2])};
        IntersectionResult r;
        c.rays_closest++;
        bool found = s->manager.GetRayIntersection(ray, r, ResolveIgnore(s, rays[i]), c);
        if (found) c.hits_closest++;
        FillHit(hits[i], found, r);
    }
    FillStats(stats, c, 0, 0);
    return 0;
}

// MeshOctree.GetRayIntersection over a batch (MO:259)
int orc_mesh_intersect(const orc_scene *s, int32_t mesh_id, const xrt_ray *rays, int64_t n, xrt_hit *hits, xrt_stats *stats) {
    if (!s->built || mesh_id < 0 || mesh_id >= (int)s->meshes.size()) return -1;
    const Mesh *m = s->meshes[mesh_id].get();
    Counters c;
    for (int64_t i = 0; i < n; i++) {
        Ray ray{V3(rays[i].o[0], rays[i].o[1], rays[i].o[2]), V3(rays[i].d[0], rays[i].d[1], rays[i].d[2])};
        MeshOctree::TriangleIntersectionResult tr;
        const Triangle *ign = (rays[i].ignore_tri >= 0 && rays[i].ignore_mesh == mesh_id && rays[i].ignore_tri < (int)m->Triangles.size()) ? &m->Triangles[rays[i].ignore_tri] : nullptr;
        c.rays_closest++;
        bool found = m->Octree.GetRayIntersection(ray, tr, ign, c);
        IntersectionResult r;
        if (found) { c.hits_closest++; r.mesh = m; r.triangle = tr.triangle; r.leaf_dfs = tr.leaf_dfs; r.u = tr.u; r.v = tr.v; r.d = tr.d; r.worldPosition = tr.objectSpacePosition; }
        FillHit(hits[i], found, r);
    }
    FillStats(stats, c, 0, 0);
    return 0;
}

// RT:410-421 for a whole frame
int orc_generate_primary_rays(const xrt_camera *cam, xrt_ray *rays) {
    RayTracer rt{};
    rt.view = ToMatrix(cam->view); rt.proj = ToMatrix(cam->proj);
    rt.viewport = Viewport{cam->vp_x, cam->vp_y, cam->vp_width, cam->vp_height, cam->vp_min_depth, cam->vp_max_depth};
    for (int y = 0; y < cam->vp_height; y++)
        for (int x = 0; x < cam->vp_width; x++) {
            Ray r = rt.PrimaryRay((float)x, (float)y);
            xrt_ray &o = rays[(size_t)y * cam->vp_width + x];
            o.o[0] = r.Position.X; o.o[1] = r.Position.Y; o.o[2] = r.Position.Z;
            o.d[0] = r.Direction.X; o.d[1] = r.Direction.Y; o.d[2] = r.Direction.Z;
            o.ignore_mesh = -1; o.ignore_tri = -1;
        }
    return 0;
}

// RayTracer.RenderInternal / RenderInternalWithMultisampling (RT:103-168) over rows [row_begin,row_end)
// with `nthreads` render threads stealing rows through an atomic counter (RT:49-52, 105-120; the
// shipped code hard-wires 1 thread, RT:99).  rgba_out / rgb_f32_out are full-frame buffers; only the
// requested rows are written.
int orc_render(const orc_scene *s, const xrt_camera *cam, const xrt_light *lights, int32_t n_lights, const xrt_render_opts *opts,
               uint32_t *rgba_out, float *rgb_f32_out, xrt_stats *stats, int32_t nthreads, int32_t row_begin, int32_t row_end) {
    if (!s->built) return -1;
    const int W = cam->vp_width, H = cam->vp_height;
    if (row_begin < 0) row_begin = 0;
    if (row_end > H || row_end < 0) row_end = H;
    if (nthreads < 1) nthreads = 1;
    std::atomic<int> scanline{row_begin - 1};   // RT:48
    std::vector<Counters> counters(nthreads);
    std::atomic<bool> bad{false};
    auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](int tid) {
        RayTracer rt{};
        rt.scene = s;
        for (int i = 0; i < n_lights; i++) rt.lights.push_back(MakeLight(lights[i]));
        rt.MaxReflections = opts->max_reflections;
        rt.AddressMode = opts->address_mode;
        rt.TextureFiltering = opts->filtering;
        rt.MultisampleQuality = opts->multisample_quality;
        rt.view = ToMatrix(cam->view); rt.proj = ToMatrix(cam->proj);
        rt.viewport = Viewport{cam->vp_x, cam->vp_y, cam->vp_width, cam->vp_height, cam->vp_min_depth, cam->vp_max_depth};
        Counters &c = counters[tid];
        for (;;) {
            int y = ++scanline;   // Interlocked.Increment, RT:51
            if (y >= row_end) break;
            for (int x = 0; x < W; x++) {
                uint32_t color;
                Vector3 cv = V3(0, 0, 0);
                if (opts->use_multisampling == XRT_MS_ADAPTIVE) {
                    rt.GetColorForQuadrant((float)x, (float)y, 1.0f, 0, color, c);   // RT:195
                    cv = ColorToVector3(color);
                } else if (opts->use_multisampling == XRT_MS_FIXED16) {
                    rt.GetColorFixed16((float)x, (float)y, color, c);
                    cv = ColorToVector3(color);
                } else {
                    Ray ray = rt.PrimaryRay((float)x, (float)y);
                    // RT:418 recomputes (and discards) the inverse view-projection once more per pixel.
                    volatile float sink = Invert(Multiply(Multiply(Identity(), rt.view), rt.proj)).M11; (void)sink;
                    rt.CastRay(ray, color, 0, nullptr, 1.0f, c, &cv);
                }
                rgba_out[(size_t)y * W + x] = color;   // RT:425
                if (rgb_f32_out) { float *p = rgb_f32_out + ((size_t)y * W + x) * 3; p[0] = cv.X; p[1] = cv.Y; p[2] = cv.Z; }
            }
        }
        if (rt.bad_lookup) bad = true;
    };
    std::vector<std::thread> threads;
    for (int i = 1; i < nthreads; i++) threads.emplace_back(worker, i);
    worker(0);
    for (auto &t : threads) t.join();
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    Counters total;
    for (auto &c : counters) total.add(c);
    FillStats(stats, total, (uint64_t)W * (row_end - row_begin), ms);
    if (bad) return -3;   // ArgumentException of MAT:85/97 (or an unsupported filter)
    return 0;
}

// ---- unit-level entry points for the known-answer tests ------------------------------
int orc_kat_triangle(const float o[3], const float d[3], const float v[9], const float sn[3], float out_uvd[3]) {
    Triangle t{};
    t.v1 = V3(v[0], v[1], v[2]); t.v2 = V3(v[3], v[4], v[5]); t.v3 = V3(v[6], v[7], v[8]);
    t.surfaceNormal = V3(sn[0], sn[1], sn[2]);
    Ray r{V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2])};
    float u, vv, dist;
    bool hit = IntersectsTriangleBackfaceCulling(r, t, u, vv, dist);
    out_uvd[0] = u; out_uvd[1] = vv; out_uvd[2] = dist;
    return hit ? 1 : 0;
}
int orc_kat_box(const float o[3], const float d[3], const float box[6], float *key) {
    BoundingBox b{V3(box[0], box[1], box[2]), V3(box[3], box[4], box[5])};
    Ray r{V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2])};
    float k = -1;
    bool hit = Intersects(b, r, k);
    *key = k;
    return hit ? 1 : 0;
}
uint32_t orc_kat_pack_color(const float rgb[3]) { return ColorFromVector3(V3(rgb[0], rgb[1], rgb[2])); }
void orc_kat_unpack_color(uint32_t c, float rgb[3]) { Vector3 v = ColorToVector3(c); rgb[0] = v.X; rgb[1] = v.Y; rgb[2] = v.Z; }
void orc_kat_look_at(const float pos[3], const float target[3], const float up[3], float out[16]) {
    Matrix m = CreateLookAt(V3(pos[0], pos[1], pos[2]), V3(target[0], target[1], target[2]), V3(up[0], up[1], up[2]));
    std::memcpy(out, &m, 64);
}
void orc_kat_perspective(float fov, float aspect, float n, float f, float out[16]) {
    Matrix m = CreatePerspectiveFieldOfView(fov, aspect, n, f);
    std::memcpy(out, &m, 64);
}
void orc_kat_invert(const float in[16], float out[16]) { Matrix m = Invert(ToMatrix(in)); std::memcpy(out, &m, 64); }
void orc_kat_multiply(const float a[16], const float b[16], float out[16]) { Matrix m = Multiply(ToMatrix(a), ToMatrix(b)); std::memcpy(out, &m, 64); }
// SceneObject.BuildWorld (SO:183-199): World, InverseWorld and the un-normalised WorldBoundingBox.
void orc_kat_build_world(const float scale[3], const float rot[3], const float pos[3], const float bbox[6], float world[16], float inv_world[16], float world_bbox[6]) {
    Matrix scaleMatrix = CreateScale(V3(scale[0], scale[1], scale[2]));
    Matrix rotationMatrix = Multiply(Multiply(CreateRotationX(rot[0]), CreateRotationY(rot[1])), CreateRotationZ(rot[2]));
    Matrix translationMatrix = CreateTranslation(V3(pos[0], pos[1], pos[2]));
    Matrix w = Multiply(Multiply(scaleMatrix, rotationMatrix), translationMatrix);
    Vector3 mx = Transform(V3(bbox[3], bbox[4], bbox[5]), w);
    Vector3 mn = Transform(V3(bbox[0], bbox[1], bbox[2]), w);
    Matrix iw = Invert(w);
    std::memcpy(world, &w, 64); std::memcpy(inv_world, &iw, 64);
    world_bbox[0] = mn.X; world_bbox[1] = mn.Y; world_bbox[2] = mn.Z; world_bbox[3] = mx.X; world_bbox[4] = mx.Y; world_bbox[5] = mx.Z;
}
void orc_kat_spot_light(const xrt_light *l, const float pos[3], const float normal[3], float out[3]) {
    Light L = MakeLight(*l);
    Vector3 r = L.GetLightForFragment(V3(pos[0], pos[1], pos[2]), V3(normal[0], normal[1], normal[2]));
    out[0] = r.X; out[1] = r.Y; out[2] = r.Z;
}
int orc_kat_lookup_uv(const xrt_material *m, const float uv[2], int address, int filtering, float out[3]) {
    Material mm; mm.UseTexture = true; mm.Width = m->tex_width; mm.Height = m->tex_height;
    mm.argb.assign(m->tex_argb, m->tex_argb + (size_t)m->tex_width * m->tex_height);
    { const uint32_t *pa = m->tex_pargb ? m->tex_pargb : m->tex_argb; mm.colorData.assign(pa, pa + (size_t)m->tex_width * m->tex_height); }
    Vector3 c;
    bool ok = mm.LookupUV(Vector2{uv[0], uv[1]}, address, filtering, c);
    if (!ok) return -1;
    out[0] = c.X; out[1] = c.Y; out[2] = c.Z;
    return 0;
}

}  // extern "C"
