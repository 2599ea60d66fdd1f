"""Host-side mirror of the Microsoft.Xna.Framework 4.0 math the C# host evaluates BEFORE it calls
the hot path (Camera.cs:42,51 view/projection; SceneObject.cs:183-199 World / InverseWorld /
WorldBoundingBox).  In the real drop-in these values come from XNA itself and cross the C-ABI as
plain float[16]; this module lets the Python host mirror produce the same inputs.

Strict binary32: every operation is between numpy float32 scalars (one rounding per operation),
`double` only where XNA calls System.Math (tan, sin, cos, sqrt).  Matrices are 16-element lists
M11..M44, row-vector convention (v' = v * M).
"""
import math
import numpy as np

f32 = np.float32
_1 = f32(1.0)
_0 = f32(0.0)


def vec3(x, y, z):
    return [f32(x), f32(y), f32(z)]


def dot(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def sub(a, b):
    return [a[0] - b[0], a[1] - b[1], a[2] - b[2]]


def normalize(a):
    num = (a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]
    num2 = _1 / f32(math.sqrt(float(num)))
    return [a[0] * num2, a[1] * num2, a[2] * num2]


def identity():
    m = [_0] * 16
    m[0] = m[5] = m[10] = m[15] = _1
    return m


def multiply(a, b):
    r = [None] * 16
    for i in range(4):
        for j in range(4):
            r[4 * i + j] = ((a[4 * i] * b[j] + a[4 * i + 1] * b[4 + j]) + a[4 * i + 2] * b[8 + j]) + a[4 * i + 3] * b[12 + j]
    return r


def invert(m):
    n5, n4, n3, n2 = m[0], m[1], m[2], m[3]
    n9, n8, n7, n6 = m[4], m[5], m[6], m[7]
    n17, n16, n15, n14 = m[8], m[9], m[10], m[11]
    n13, n12, n11, n10 = m[12], m[13], m[14], m[15]
    n23 = n15 * n10 - n14 * n11
    n22 = n16 * n10 - n14 * n12
    n21 = n16 * n11 - n15 * n12
    n20 = n17 * n10 - n14 * n13
    n19 = n17 * n11 - n15 * n13
    n18 = n17 * n12 - n16 * n13
    n39 = (n8 * n23 - n7 * n22) + n6 * n21
    n38 = -((n9 * n23 - n7 * n20) + n6 * n19)
    n37 = (n9 * n22 - n8 * n20) + n6 * n18
    n36 = -((n9 * n21 - n8 * n19) + n7 * n18)
    num = _1 / (((n5 * n39 + n4 * n38) + n3 * n37) + n2 * n36)
    r = [None] * 16
    r[0] = n39 * num
    r[4] = n38 * num
    r[8] = n37 * num
    r[12] = n36 * num
    r[1] = -((n4 * n23 - n3 * n22) + n2 * n21) * num
    r[5] = ((n5 * n23 - n3 * n20) + n2 * n19) * num
    r[9] = -((n5 * n22 - n4 * n20) + n2 * n18) * num
    r[13] = ((n5 * n21 - n4 * n19) + n3 * n18) * num
    n35 = n7 * n10 - n6 * n11
    n34 = n8 * n10 - n6 * n12
    n33 = n8 * n11 - n7 * n12
    n32 = n9 * n10 - n6 * n13
    n31 = n9 * n11 - n7 * n13
    n30 = n9 * n12 - n8 * n13
    r[2] = ((n4 * n35 - n3 * n34) + n2 * n33) * num
    r[6] = -((n5 * n35 - n3 * n32) + n2 * n31) * num
    r[10] = ((n5 * n34 - n4 * n32) + n2 * n30) * num
    r[14] = -((n5 * n33 - n4 * n31) + n3 * n30) * num
    n29 = n7 * n14 - n6 * n15
    n28 = n8 * n14 - n6 * n16
    n27 = n8 * n15 - n7 * n16
    n26 = n9 * n14 - n6 * n17
    n25 = n9 * n15 - n7 * n17
    n24 = n9 * n16 - n8 * n17
    r[3] = -((n4 * n29 - n3 * n28) + n2 * n27) * num
    r[7] = ((n5 * n29 - n3 * n26) + n2 * n25) * num
    r[11] = -((n5 * n28 - n4 * n26) + n2 * n24) * num
    r[15] = ((n5 * n27 - n4 * n25) + n3 * n24) * num
    return r


def transform(p, m):
    return [((p[0] * m[0] + p[1] * m[4]) + p[2] * m[8]) + m[12],
            ((p[0] * m[1] + p[1] * m[5]) + p[2] * m[9]) + m[13],
            ((p[0] * m[2] + p[1] * m[6]) + p[2] * m[10]) + m[14]]


def create_look_at(pos, target, up):
    """Matrix.CreateLookAt (Camera.cs:42)."""
    pos, target, up = vec3(*pos), vec3(*target), vec3(*up)
    z = normalize(sub(pos, target))
    x = normalize(cross(up, z))
    y = cross(z, x)
    return [x[0], y[0], z[0], _0, x[1], y[1], z[1], _0, x[2], y[2], z[2], _0,
            -dot(x, pos), -dot(y, pos), -dot(z, pos), _1]


def create_perspective_fov(fov, aspect, near, far):
    """Matrix.CreatePerspectiveFieldOfView (Camera.cs:51)."""
    fov, aspect, near, far = f32(fov), f32(aspect), f32(near), f32(far)
    num = _1 / f32(math.tan(float(fov * f32(0.5))))
    num9 = num / aspect
    m = [_0] * 16
    m[0] = num9
    m[5] = num
    m[10] = far / (near - far)
    m[11] = f32(-1.0)
    m[14] = (near * far) / (near - far)
    return m


def create_scale(s):
    m = identity()
    m[0], m[5], m[10] = f32(s[0]), f32(s[1]), f32(s[2])
    return m


def create_rotation_x(r):
    c, s = f32(math.cos(float(f32(r)))), f32(math.sin(float(f32(r))))
    m = identity()
    m[5], m[6], m[9], m[10] = c, s, -s, c
    return m


def create_rotation_y(r):
    c, s = f32(math.cos(float(f32(r)))), f32(math.sin(float(f32(r))))
    m = identity()
    m[0], m[2], m[8], m[10] = c, -s, s, c
    return m


def create_rotation_z(r):
    c, s = f32(math.cos(float(f32(r)))), f32(math.sin(float(f32(r))))
    m = identity()
    m[0], m[1], m[4], m[5] = c, s, -s, c
    return m


def create_translation(p):
    m = identity()
    m[12], m[13], m[14] = f32(p[0]), f32(p[1]), f32(p[2])
    return m


def build_world(scale, rotation, position, bbox):
    """SceneObject.BuildWorld (SceneObject.cs:183-199): returns (World, InverseWorld, WorldBoundingBox[6])
    where the world box is the un-normalised {Min*W, Max*W} pair of SO:195-196."""
    rot = multiply(multiply(create_rotation_x(rotation[0]), create_rotation_y(rotation[1])), create_rotation_z(rotation[2]))
    world = multiply(multiply(create_scale(scale), rot), create_translation(position))
    mx = transform(vec3(*bbox[3:6]), world)
    mn = transform(vec3(*bbox[0:3]), world)
    return world, invert(world), mn + mx


def aspect_ratio(width, height):
    """Viewport.AspectRatio."""
    return f32(width) / f32(height)


def as_array(m):
    return np.array([float(x) for x in m], dtype=np.float32)
