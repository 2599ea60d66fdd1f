// TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may build, link or call anything under oracle/.
//
// xna_math.h — CPU restatement of the Microsoft.Xna.Framework 4.0.0.0 math the hot path calls
// (SURVEY §8a rows A17-A19, §8c).  The XNA assembly is closed source and NOT vendored in the
// reference (RayTracerTypeLibrary/RayTracerTypeLibrary.csproj:48), and the reference has no tests,
// golden vectors or fixtures that pin it:
//
//                      *** PARITY UNPINNED ***
//
// These definitions are the published / well-known XNA 4.0 formulas, one IEEE-754 binary32 rounding
// per operation, no FMA contraction (build with -ffp-contract=off), `double` exactly where the C#
// calls System.Math.  Each function cites the reference call site(s) that use it.
#pragma once
#include <cmath>
#include <cstdint>
#include <cfloat>

namespace xna {

struct Vector2 { float X, Y; };
struct Vector3 { float X, Y, Z; };
struct Vector4 { float X, Y, Z, W; };
struct Matrix {
    float M11, M12, M13, M14, M21, M22, M23, M24, M31, M32, M33, M34, M41, M42, M43, M44;
};
struct Ray { Vector3 Position, Direction; };
struct BoundingBox { Vector3 Min, Max; };

// ---- Vector3 operators (RT:285,471,540,584,685; MO:207,217,310-312; SPOT:39,55) ----------------
static inline Vector3 V3(float x, float y, float z) { return Vector3{x, y, z}; }
static inline Vector3 operator+(Vector3 a, Vector3 b) { return V3(a.X + b.X, a.Y + b.Y, a.Z + b.Z); }
static inline Vector3 operator-(Vector3 a, Vector3 b) { return V3(a.X - b.X, a.Y - b.Y, a.Z - b.Z); }
static inline Vector3 operator-(Vector3 a) { return V3(-a.X, -a.Y, -a.Z); }
static inline Vector3 operator*(Vector3 a, float s) { return V3(a.X * s, a.Y * s, a.Z * s); }
static inline Vector3 operator*(float s, Vector3 a) { return V3(a.X * s, a.Y * s, a.Z * s); }
static inline Vector3 operator*(Vector3 a, Vector3 b) { return V3(a.X * b.X, a.Y * b.Y, a.Z * b.Z); }
// Vector3 / float multiplies by the reciprocal (MO:207 `/ 2f`, RT:285,309 `/ 4.0f`, Unproject `/ a`).
static inline Vector3 operator/(Vector3 a, float d) { float num = 1.0f / d; return V3(a.X * num, a.Y * num, a.Z * num); }

static inline Vector2 operator+(Vector2 a, Vector2 b) { return Vector2{a.X + b.X, a.Y + b.Y}; }
static inline Vector2 operator-(Vector2 a, Vector2 b) { return Vector2{a.X - b.X, a.Y - b.Y}; }
static inline Vector2 operator*(Vector2 a, float s) { return Vector2{a.X * s, a.Y * s}; }

// Vector3.Dot (RE:49,61-66; SPOT:44,50; DIR:25; RT:675)
static inline float Dot(Vector3 a, Vector3 b) { return (a.X * b.X + a.Y * b.Y) + a.Z * b.Z; }
// Vector3.Cross (RE:58-59; TMP:202)
static inline Vector3 Cross(Vector3 a, Vector3 b) {
    return V3(a.Y * b.Z - a.Z * b.Y, a.Z * b.X - a.X * b.Z, a.X * b.Y - a.Y * b.X);
}
// Vector3.Length (RT:286,288,472): (float)Math.Sqrt(x*x + y*y + z*z) with the sum in float.
static inline float Length(Vector3 a) {
    float num = (a.X * a.X + a.Y * a.Y) + a.Z * a.Z;
    return (float)std::sqrt((double)num);
}
// Vector3.Normalize (RT:421,473,525,550,694; OSM:364; SPOT:41; TMP:203)
static inline Vector3 Normalize(Vector3 a) {
    float num = (a.X * a.X + a.Y * a.Y) + a.Z * a.Z;
    float num2 = 1.0f / (float)std::sqrt((double)num);
    return V3(a.X * num2, a.Y * num2, a.Z * num2);
}
// Vector3.Min / Max (MO:71-77; OSM:84-85)
static inline Vector3 Min(Vector3 a, Vector3 b) {
    return V3(a.X < b.X ? a.X : b.X, a.Y < b.Y ? a.Y : b.Y, a.Z < b.Z ? a.Z : b.Z);
}
static inline Vector3 Max(Vector3 a, Vector3 b) {
    return V3(a.X > b.X ? a.X : b.X, a.Y > b.Y ? a.Y : b.Y, a.Z > b.Z ? a.Z : b.Z);
}
// Vector3.Transform(position, matrix) (OSM:80-81,360-361,443; SO:195-196; Unproject)
static inline Vector3 Transform(Vector3 p, const Matrix &m) {
    return V3(((p.X * m.M11 + p.Y * m.M21) + p.Z * m.M31) + m.M41,
              ((p.X * m.M12 + p.Y * m.M22) + p.Z * m.M32) + m.M42,
              ((p.X * m.M13 + p.Y * m.M23) + p.Z * m.M33) + m.M43);
}
// Vector3.Reflect (RT:549)
static inline Vector3 Reflect(Vector3 v, Vector3 n) {
    float num = (v.X * n.X + v.Y * n.Y) + v.Z * n.Z;
    return V3(v.X - (2.0f * num) * n.X, v.Y - (2.0f * num) * n.Y, v.Z - (2.0f * num) * n.Z);
}
// Vector3.Lerp (RT:584,699)
static inline Vector3 Lerp(Vector3 a, Vector3 b, float t) {
    return V3(a.X + (b.X - a.X) * t, a.Y + (b.Y - a.Y) * t, a.Z + (b.Z - a.Z) * t);
}

// System.Math.Max / Min (float) as used by MathHelper.Max/Min inside BoundingBox.Intersects(Ray).
static inline float MathMax(float a, float b) { return a > b ? a : (std::isnan(a) ? a : b); }
static inline float MathMin(float a, float b) { return a < b ? a : (std::isnan(a) ? a : b); }

// BoundingBox.Intersects(ref Ray, out float?) (MO:331; OSM:460; MESH:37; SO:256).  Returns true and
// the entry distance (>= 0) or false for null.
static inline bool Intersects(const BoundingBox &b, const Ray &ray, float &result) {
    float num = 0.0f;
    float num2 = FLT_MAX;
    if (std::fabs(ray.Direction.X) < 1e-06f) {
        if (ray.Position.X < b.Min.X || ray.Position.X > b.Max.X) return false;
    } else {
        float num3 = 1.0f / ray.Direction.X;
        float num4 = (b.Min.X - ray.Position.X) * num3;
        float num5 = (b.Max.X - ray.Position.X) * num3;
        if (num4 > num5) { float t = num4; num4 = num5; num5 = t; }
        num = MathMax(num4, num);
        num2 = MathMin(num5, num2);
        if (num > num2) return false;
    }
    if (std::fabs(ray.Direction.Y) < 1e-06f) {
        if (ray.Position.Y < b.Min.Y || ray.Position.Y > b.Max.Y) return false;
    } else {
        float num3 = 1.0f / ray.Direction.Y;
        float num4 = (b.Min.Y - ray.Position.Y) * num3;
        float num5 = (b.Max.Y - ray.Position.Y) * num3;
        if (num4 > num5) { float t = num4; num4 = num5; num5 = t; }
        num = MathMax(num4, num);
        num2 = MathMin(num5, num2);
        if (num > num2) return false;
    }
    if (std::fabs(ray.Direction.Z) < 1e-06f) {
        if (ray.Position.Z < b.Min.Z || ray.Position.Z > b.Max.Z) return false;
    } else {
        float num3 = 1.0f / ray.Direction.Z;
        float num4 = (b.Min.Z - ray.Position.Z) * num3;
        float num5 = (b.Max.Z - ray.Position.Z) * num3;
        if (num4 > num5) { float t = num4; num4 = num5; num5 = t; }
        num = MathMax(num4, num);
        num2 = MathMin(num5, num2);
        if (num > num2) return false;
    }
    result = num;
    return true;
}
// BoundingBox.Contains(Vector3) != Disjoint (MO:101-103,226-228): inclusive on all faces.
static inline bool ContainsPoint(const BoundingBox &b, Vector3 p) {
    return b.Min.X <= p.X && p.X <= b.Max.X && b.Min.Y <= p.Y && p.Y <= b.Max.Y && b.Min.Z <= p.Z && p.Z <= b.Max.Z;
}
// BoundingBox.Intersects(BoundingBox) (OSM:117,240)
static inline bool Intersects(const BoundingBox &a, const BoundingBox &b) {
    if (a.Max.X < b.Min.X || a.Min.X > b.Max.X) return false;
    if (a.Max.Y < b.Min.Y || a.Min.Y > b.Max.Y) return false;
    return a.Max.Z >= b.Min.Z && a.Min.Z <= b.Max.Z;
}
// BoundingBox.CreateMerged (OSM:93; SO:131)
static inline BoundingBox CreateMerged(const BoundingBox &a, const BoundingBox &b) {
    return BoundingBox{Min(a.Min, b.Min), Max(a.Max, b.Max)};
}

// ---- Matrix (RT:418; SO:187-198; CAM:42,51; Viewport.Unproject) --------------------------------
static inline Matrix Identity() { return Matrix{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; }
static inline Matrix Multiply(const Matrix &a, const Matrix &b) {
    Matrix r;
    r.M11 = ((a.M11 * b.M11 + a.M12 * b.M21) + a.M13 * b.M31) + a.M14 * b.M41;
    r.M12 = ((a.M11 * b.M12 + a.M12 * b.M22) + a.M13 * b.M32) + a.M14 * b.M42;
    r.M13 = ((a.M11 * b.M13 + a.M12 * b.M23) + a.M13 * b.M33) + a.M14 * b.M43;
    r.M14 = ((a.M11 * b.M14 + a.M12 * b.M24) + a.M13 * b.M34) + a.M14 * b.M44;
    r.M21 = ((a.M21 * b.M11 + a.M22 * b.M21) + a.M23 * b.M31) + a.M24 * b.M41;
    r.M22 = ((a.M21 * b.M12 + a.M22 * b.M22) + a.M23 * b.M32) + a.M24 * b.M42;
    r.M23 = ((a.M21 * b.M13 + a.M22 * b.M23) + a.M23 * b.M33) + a.M24 * b.M43;
    r.M24 = ((a.M21 * b.M14 + a.M22 * b.M24) + a.M23 * b.M34) + a.M24 * b.M44;
    r.M31 = ((a.M31 * b.M11 + a.M32 * b.M21) + a.M33 * b.M31) + a.M34 * b.M41;
    r.M32 = ((a.M31 * b.M12 + a.M32 * b.M22) + a.M33 * b.M32) + a.M34 * b.M42;
    r.M33 = ((a.M31 * b.M13 + a.M32 * b.M23) + a.M33 * b.M33) + a.M34 * b.M43;
    r.M34 = ((a.M31 * b.M14 + a.M32 * b.M24) + a.M33 * b.M34) + a.M34 * b.M44;
    r.M41 = ((a.M41 * b.M11 + a.M42 * b.M21) + a.M43 * b.M31) + a.M44 * b.M41;
    r.M42 = ((a.M41 * b.M12 + a.M42 * b.M22) + a.M43 * b.M32) + a.M44 * b.M42;
    r.M43 = ((a.M41 * b.M13 + a.M42 * b.M23) + a.M43 * b.M33) + a.M44 * b.M43;
    r.M44 = ((a.M41 * b.M14 + a.M42 * b.M24) + a.M43 * b.M34) + a.M44 * b.M44;
    return r;
}
// Matrix.Invert: cofactor expansion, the six 2x2 minors of rows 3-4 first, invDet = 1f / det.
static inline Matrix Invert(const Matrix &m) {
    float n5 = m.M11, n4 = m.M12, n3 = m.M13, n2 = m.M14;
    float n9 = m.M21, n8 = m.M22, n7 = m.M23, n6 = m.M24;
    float n17 = m.M31, n16 = m.M32, n15 = m.M33, n14 = m.M34;
    float n13 = m.M41, n12 = m.M42, n11 = m.M43, n10 = m.M44;
    float n23 = n15 * n10 - n14 * n11;
    float n22 = n16 * n10 - n14 * n12;
    float n21 = n16 * n11 - n15 * n12;
    float n20 = n17 * n10 - n14 * n13;
    float n19 = n17 * n11 - n15 * n13;
    float n18 = n17 * n12 - n16 * n13;
    float n39 = (n8 * n23 - n7 * n22) + n6 * n21;
    float n38 = -((n9 * n23 - n7 * n20) + n6 * n19);
    float n37 = (n9 * n22 - n8 * n20) + n6 * n18;
    float n36 = -((n9 * n21 - n8 * n19) + n7 * n18);
    float num = 1.0f / (((n5 * n39 + n4 * n38) + n3 * n37) + n2 * n36);
    Matrix r;
    r.M11 = n39 * num;
    r.M21 = n38 * num;
    r.M31 = n37 * num;
    r.M41 = n36 * num;
    r.M12 = -((n4 * n23 - n3 * n22) + n2 * n21) * num;
    r.M22 = ((n5 * n23 - n3 * n20) + n2 * n19) * num;
    r.M32 = -((n5 * n22 - n4 * n20) + n2 * n18) * num;
    r.M42 = ((n5 * n21 - n4 * n19) + n3 * n18) * num;
    float n35 = n7 * n10 - n6 * n11;
    float n34 = n8 * n10 - n6 * n12;
    float n33 = n8 * n11 - n7 * n12;
    float n32 = n9 * n10 - n6 * n13;
    float n31 = n9 * n11 - n7 * n13;
    float n30 = n9 * n12 - n8 * n13;
    r.M13 = ((n4 * n35 - n3 * n34) + n2 * n33) * num;
    r.M23 = -((n5 * n35 - n3 * n32) + n2 * n31) * num;
    r.M33 = ((n5 * n34 - n4 * n32) + n2 * n30) * num;
    r.M43 = -((n5 * n33 - n4 * n31) + n3 * n30) * num;
    float n29 = n7 * n14 - n6 * n15;
    float n28 = n8 * n14 - n6 * n16;
    float n27 = n8 * n15 - n7 * n16;
    float n26 = n9 * n14 - n6 * n17;
    float n25 = n9 * n15 - n7 * n17;
    float n24 = n9 * n16 - n8 * n17;
    r.M14 = -((n4 * n29 - n3 * n28) + n2 * n27) * num;
    r.M24 = ((n5 * n29 - n3 * n26) + n2 * n25) * num;
    r.M34 = -((n5 * n28 - n4 * n26) + n2 * n24) * num;
    r.M44 = ((n5 * n27 - n4 * n25) + n3 * n24) * num;
    return r;
}
// Matrix.CreateLookAt (CAM:42)
static inline Matrix CreateLookAt(Vector3 pos, Vector3 target, Vector3 up) {
    Vector3 z = Normalize(pos - target);
    Vector3 x = Normalize(Cross(up, z));
    Vector3 y = Cross(z, x);
    Matrix m;
    m.M11 = x.X; m.M12 = y.X; m.M13 = z.X; m.M14 = 0.0f;
    m.M21 = x.Y; m.M22 = y.Y; m.M23 = z.Y; m.M24 = 0.0f;
    m.M31 = x.Z; m.M32 = y.Z; m.M33 = z.Z; m.M34 = 0.0f;
    m.M41 = -Dot(x, pos); m.M42 = -Dot(y, pos); m.M43 = -Dot(z, pos); m.M44 = 1.0f;
    return m;
}
// Matrix.CreatePerspectiveFieldOfView (CAM:51)
static inline Matrix CreatePerspectiveFieldOfView(float fov, float aspect, float nearP, float farP) {
    float num = 1.0f / (float)std::tan((double)(fov * 0.5f));
    float num9 = num / aspect;
    Matrix m{};
    m.M11 = num9;
    m.M22 = num;
    m.M33 = farP / (nearP - farP);
    m.M34 = -1.0f;
    m.M43 = (nearP * farP) / (nearP - farP);
    return m;
}
// Matrix.CreateScale / CreateRotationX,Y,Z / CreateTranslation (SO:187-191)
static inline Matrix CreateScale(Vector3 s) { Matrix m = Identity(); m.M11 = s.X; m.M22 = s.Y; m.M33 = s.Z; return m; }
static inline Matrix CreateRotationX(float r) {
    float c = (float)std::cos((double)r), s = (float)std::sin((double)r);
    Matrix m = Identity(); m.M22 = c; m.M23 = s; m.M32 = -s; m.M33 = c; return m;
}
static inline Matrix CreateRotationY(float r) {
    float c = (float)std::cos((double)r), s = (float)std::sin((double)r);
    Matrix m = Identity(); m.M11 = c; m.M13 = -s; m.M31 = s; m.M33 = c; return m;
}
static inline Matrix CreateRotationZ(float r) {
    float c = (float)std::cos((double)r), s = (float)std::sin((double)r);
    Matrix m = Identity(); m.M11 = c; m.M12 = s; m.M21 = -s; m.M22 = c; return m;
}
static inline Matrix CreateTranslation(Vector3 p) { Matrix m = Identity(); m.M41 = p.X; m.M42 = p.Y; m.M43 = p.Z; return m; }

// Viewport (RT:397) and Viewport.Unproject (RT:415,419,227-273)
struct Viewport { int X, Y, Width, Height; float MinDepth, MaxDepth; };
static inline bool WithinEpsilon(float a, float b) {
    float num = a - b;
    return (-1.401298E-45f <= num) && (num <= 1.401298E-45f);
}
static inline Vector3 Unproject(const Viewport &vp, Vector3 source, const Matrix &projection, const Matrix &view, const Matrix &world) {
    Matrix matrix = Invert(Multiply(Multiply(world, view), projection));
    source.X = (((source.X - (float)vp.X) / ((float)vp.Width)) * 2.0f) - 1.0f;
    source.Y = -((((source.Y - (float)vp.Y) / ((float)vp.Height)) * 2.0f) - 1.0f);
    source.Z = (source.Z - vp.MinDepth) / (vp.MaxDepth - vp.MinDepth);
    Vector3 vector = Transform(source, matrix);
    float a = (((source.X * matrix.M14) + (source.Y * matrix.M24)) + (source.Z * matrix.M34)) + matrix.M44;
    if (!WithinEpsilon(a, 1.0f)) vector = vector / a;
    return vector;
}

// Color (RT:508,584,699,705,726,732): packed RGBA8, R in the low byte.
static inline double ClampAndRound(float value, float min, float max) {
    if (std::isnan(value)) return 0.0;
    if (std::isinf(value)) return value < 0 ? (double)min : (double)max;
    if (value < min) return (double)min;
    if (value > max) return (double)max;
    return std::nearbyint((double)value);   // Math.Round(double): half to even
}
static inline uint32_t PackUNorm(float bitmask, float value) {
    value *= bitmask;
    return (uint32_t)ClampAndRound(value, 0.0f, bitmask);
}
static inline uint32_t ColorFromVector3(Vector3 v) {   // new Color(Vector3): alpha = 1
    uint32_t r = PackUNorm(255.0f, v.X);
    uint32_t g = PackUNorm(255.0f, v.Y) << 8;
    uint32_t b = PackUNorm(255.0f, v.Z) << 16;
    uint32_t a = PackUNorm(255.0f, 1.0f) << 24;
    return r | g | b | a;
}
static inline Vector3 ColorToVector3(uint32_t packed) {
    return V3((float)(packed & 0xff) / 255.0f, (float)((packed >> 8) & 0xff) / 255.0f, (float)((packed >> 16) & 0xff) / 255.0f);
}

}  // namespace xna
