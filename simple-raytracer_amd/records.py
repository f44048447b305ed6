"""numpy views of the records that cross the C ABI (include/srt_types.h).

Byte layouts are those of the reference's device structs
(/root/reference/src/render.cl:5-105) and host mirrors (include/shape.hpp:15-111,
include/material.hpp:10-38, include/tracer.hpp:48-80): float3 = 16 bytes.
Pure host code: no GPU, no oracle.
"""
import numpy as np

F3 = (np.float32, (3,))

MATERIAL = np.dtype(
    {
        "names": ["smoothness", "metallic", "specular", "emission_strength", "transmittance", "refraction_index", "color", "emission"],
        "formats": [np.float32] * 6 + [F3, F3],
        "offsets": [0, 4, 8, 12, 16, 20, 32, 48],
        "itemsize": 64,
    }
)

VERTEX = np.dtype({"names": ["normal", "pos"], "formats": [F3, F3], "offsets": [0, 16], "itemsize": 32})

TRIANGLE = np.dtype({"names": ["v"], "formats": [(VERTEX, (3,))], "offsets": [0], "itemsize": 96})

# union members overlap on purpose (render.cl:69-77)
SHAPE = np.dtype(
    {
        "names": [
            "type", "material",
            "sphere_position", "sphere_radius",
            "plane_position", "plane_normal",
            "triangle_index", "num_triangles", "bounding_min", "bounding_max", "transform",
        ],
        "formats": [
            np.int32, np.int32,
            F3, np.float32,
            F3, F3,
            np.uint32, np.uint32, F3, F3, (np.float32, (4, 4)),
        ],
        "offsets": [0, 4, 16, 32, 16, 32, 16, 20, 32, 48, 64],
        "itemsize": 128,
    }
)

RENDER_DATA = np.dtype(
    {
        "names": ["width", "height", "num_samples", "num_bounces", "aspect_ratio", "fov_scale", "show_normals", "camera_to_world", "time", "tick"],
        "formats": [np.int32, np.int32, np.int32, np.int32, np.float32, np.float32, np.uint8, (np.float32, (4, 4)), np.uint32, np.uint32],
        "offsets": [0, 4, 8, 12, 16, 20, 24, 32, 96, 100],
        "itemsize": 112,
    }
)

SCENE_DATA = np.dtype(
    {
        "names": ["num_shapes", "sun_focus", "sun_intensity", "horizon_color", "zenith_color", "ground_color", "sun_color", "sun_direction"],
        "formats": [np.int32, np.float32, np.float32, F3, F3, F3, F3, F3],
        "offsets": [0, 4, 8, 16, 32, 48, 64, 80],
        "itemsize": 96,
    }
)

SHAPE_SPHERE, SHAPE_PLANE, SHAPE_MODEL = 0, 1, 2

assert MATERIAL.itemsize == 64 and TRIANGLE.itemsize == 96 and SHAPE.itemsize == 128
assert RENDER_DATA.itemsize == 112 and SCENE_DATA.itemsize == 96


def f32(x):
    return np.float32(x)


def concat(dtype, *arrays):
    """Concatenate record arrays WITHOUT losing the padded layout (np.concatenate
    silently repacks padded structured dtypes)."""
    n = sum(len(a) for a in arrays)
    out = np.zeros(n, dtype)
    i = 0
    for a in arrays:
        out[i:i + len(a)] = as_records(a, dtype)
        i += len(a)
    return out


def as_records(a, dtype):
    """Return `a` as a C-contiguous array of exactly `dtype` (the ABI layout)."""
    a = np.asarray(a)
    if a.dtype == dtype:
        return a if a.flags.c_contiguous else np.ascontiguousarray(a)
    if a.dtype.names is None and a.size == 0:
        return np.zeros(0, dtype)
    out = np.zeros(a.shape, dtype)

    def copy_fields(dst, src):
        for name in dst.dtype.names:
            if dst.dtype[name].names is not None:
                copy_fields(dst[name], src[name])
            else:
                dst[name] = src[name]

    copy_fields(out, a)
    return out


def material(color=(1, 1, 1), smoothness=0.0, metallic=0.0, specular=0.0, transmittance=0.0,
             refraction_index=1.0, emission=(0, 0, 0), emission_strength=0.0):
    """Same argument order and defaults as Material(...) in include/material.hpp:23-37."""
    m = np.zeros((), MATERIAL)
    m["color"] = color
    m["smoothness"] = smoothness
    m["metallic"] = metallic
    m["specular"] = specular
    m["transmittance"] = transmittance
    m["refraction_index"] = refraction_index
    m["emission"] = emission
    m["emission_strength"] = emission_strength
    return m


def sphere(material_index, position, radius):
    s = np.zeros((), SHAPE)
    s["type"] = SHAPE_SPHERE
    s["material"] = material_index
    s["sphere_position"] = position
    s["sphere_radius"] = radius
    return s


def plane(material_index, position, normal):
    s = np.zeros((), SHAPE)
    s["type"] = SHAPE_PLANE
    s["material"] = material_index
    s["plane_position"] = position
    s["plane_normal"] = normal
    return s


def identity4():
    return np.eye(4, dtype=np.float32)


def translate(t):
    """glm::translate(t): column-major, m[col][row]; translation in column 3."""
    m = identity4()
    m[3, :3] = np.asarray(t, np.float32)
    return m


def mat_mul(a, b):
    """glm a*b for column-major m[col][row] arrays, float32, glm's operation order:
    result[c] = ((a[0]*b[c][0] + a[1]*b[c][1]) + a[2]*b[c][2]) + a[3]*b[c][3]."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    r = np.zeros((4, 4), np.float32)
    for c in range(4):
        r[c] = ((a[0] * b[c, 0] + a[1] * b[c, 1]) + a[2] * b[c, 2]) + a[3] * b[c, 3]
    return r


def scale_matrix(s):
    m = identity4()
    m[0, 0], m[1, 1], m[2, 2] = np.float32(s[0]), np.float32(s[1]), np.float32(s[2])
    return m


def euler_yxz(yaw, pitch, roll=0.0):
    """glm::eulerAngleYXZ(yaw, pitch, roll) (include/helper.hpp:21-26), float32."""
    ch, sh = np.float32(np.cos(np.float32(yaw))), np.float32(np.sin(np.float32(yaw)))
    cp, sp = np.float32(np.cos(np.float32(pitch))), np.float32(np.sin(np.float32(pitch)))
    cb, sb = np.float32(np.cos(np.float32(roll))), np.float32(np.sin(np.float32(roll)))
    m = identity4()
    m[0, 0] = ch * cb + sh * sp * sb
    m[0, 1] = sb * cp
    m[0, 2] = -sh * cb + ch * sp * sb
    m[1, 0] = -ch * sb + sh * sp * cb
    m[1, 1] = cb * cp
    m[1, 2] = sb * sh + ch * sp * cb
    m[2, 0] = sh * cp
    m[2, 1] = -sp
    m[2, 2] = ch * cp
    return m


def camera_matrix(position, yaw=0.0, pitch=0.0):
    """Camera::camera_matrix (include/helper.hpp:21-26): translate(pos) * eulerAngleYXZ."""
    return mat_mul(translate(position), euler_yxz(yaw, pitch, 0.0))


def transform_points(m, pts, w=1.0):
    """glm mat4*vec4 order: (m0*x + m1*y) + (m2*z + m3*w); pts (N,3) float32 -> (N,3).
    This is the HOST order used by Model::compute_bounding_box (src/shape.cpp:45-58 via
    helper.hpp:72-74); the kernel's own order (render.cl:114-120) is left-associated."""
    m = np.asarray(m, np.float32)
    p = np.asarray(pts, np.float32).reshape(-1, 3)
    wv = np.float32(w)
    out = (m[0][None, :3] * p[:, 0:1] + m[1][None, :3] * p[:, 1:2]) + (m[2][None, :3] * p[:, 2:3] + m[3][None, :3] * wv)
    return out.astype(np.float32)


def model(material_index, triangles, triangle_index, num_triangles, transform=None):
    """Model ctor + compute_bounding_box (src/shape.cpp:37-58): world AABB of the
    transformed vertices; triangles stay in model space."""
    s = np.zeros((), SHAPE)
    s["type"] = SHAPE_MODEL
    s["material"] = material_index
    s["triangle_index"] = triangle_index
    s["num_triangles"] = num_triangles
    t = identity4() if transform is None else np.asarray(transform, np.float32)
    s["transform"] = t
    if num_triangles > 0:
        pos = triangles["v"]["pos"][triangle_index:triangle_index + num_triangles].reshape(-1, 3)
        world = transform_points(t, pos, 1.0)
        s["bounding_min"] = world.min(axis=0)
        s["bounding_max"] = world.max(axis=0)
    else:
        s["bounding_min"] = np.inf
        s["bounding_max"] = -np.inf
    return s


def flat_triangle(normal, p0, p1, p2):
    """Triangle(normal, v0, v1, v2) (src/shape.cpp:20-27)."""
    t = np.zeros((), TRIANGLE)
    for j, p in enumerate((p0, p1, p2)):
        t["v"][j]["normal"] = normal
        t["v"][j]["pos"] = p
    return t


def box_triangles():
    """Box::create_triangle (src/shape.cpp:91-119): 12 flat triangles of the +-1 cube,
    normals flipped to point away from the centre and normalised."""
    vertices = np.array(
        [[-1, -1, 1], [-1, 1, 1], [-1, -1, -1], [-1, 1, -1], [1, -1, 1], [1, 1, 1], [1, -1, -1], [1, 1, -1]], np.float32
    )
    table = [(1, 2, 0), (3, 6, 2), (7, 4, 6), (5, 0, 4), (6, 0, 2), (3, 5, 7),
             (1, 3, 2), (3, 7, 6), (7, 5, 4), (5, 1, 0), (6, 4, 0), (3, 1, 5)]
    tris = np.zeros(12, TRIANGLE)
    for i, (a, b, c) in enumerate(table):
        v1, v2, v3 = vertices[a], vertices[b], vertices[c]
        A, B = v2 - v1, v3 - v1
        n = np.array([A[1] * B[2] - A[2] * B[1], A[2] * B[0] - A[0] * B[2], A[0] * B[1] - A[1] * B[0]], np.float32)
        d = np.float32((v1[0] * n[0] + v1[1] * n[1]) + v1[2] * n[2])
        n = n * (np.float32(1.0) if d > 0 else np.float32(-1.0))
        ln = np.float32(np.sqrt(np.float32((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2])))
        n = (n / ln).astype(np.float32)
        tris[i] = flat_triangle(n, v1, v2, v3)
    return tris


def box_model(material_index, box_triangle_index, position, size=(2, 2, 2)):
    """Box::model (src/shape.cpp:76-89): shares the 12 box triangles, transform =
    translate(position), bounds = position -+ size/2 (only consistent for size 2)."""
    s = np.zeros((), SHAPE)
    s["type"] = SHAPE_MODEL
    s["material"] = material_index
    s["triangle_index"] = box_triangle_index
    s["num_triangles"] = 12
    p = np.asarray(position, np.float32)
    sz = np.asarray(size, np.float32)
    s["bounding_min"] = p - sz * np.float32(0.5)
    s["bounding_max"] = p + sz * np.float32(0.5)
    s["transform"] = translate(p)
    return s


def render_data(width, height, num_samples=4, num_bounces=10, fov_scale=1.0, camera_to_world=None,
                time=12345, tick=0, show_normals=False, aspect_ratio=None):
    """Tracer::RenderData (include/tracer.hpp:48-67) as main.cpp:283-288 fills it."""
    r = np.zeros((), RENDER_DATA)
    r["width"], r["height"] = width, height
    r["num_samples"], r["num_bounces"] = num_samples, num_bounces
    r["aspect_ratio"] = np.float32(width) / np.float32(height) if aspect_ratio is None else aspect_ratio
    r["fov_scale"] = fov_scale
    r["show_normals"] = 1 if show_normals else 0
    r["camera_to_world"] = identity4() if camera_to_world is None else camera_to_world
    r["time"] = np.uint32(time & 0xFFFFFFFF)
    r["tick"] = tick
    return r


def color_from_hex(v):
    """color::from_hex (include/color.hpp:12-14)."""
    return np.array([((v >> 16) & 255), ((v >> 8) & 255), (v & 255)], np.float32) / np.float32(255.0)


def scene_data(num_shapes, sun_focus=25.0, sun_intensity=1.0, sun_color=None, sun_direction=None):
    """Tracer::SceneData with the defaults main.cpp:119-126 sets."""
    s = np.zeros((), SCENE_DATA)
    s["num_shapes"] = num_shapes
    s["sun_focus"] = sun_focus
    s["sun_intensity"] = sun_intensity
    s["horizon_color"] = color_from_hex(0x374F62)
    s["zenith_color"] = color_from_hex(0x11334A)
    s["ground_color"] = color_from_hex(0x777777)
    s["sun_color"] = color_from_hex(0xFFFFD3) if sun_color is None else sun_color
    if sun_direction is None:
        d = np.array([1.0, -1.0, 0.0], np.float32)
        d = d / np.float32(np.sqrt(np.float32(2.0)))
        sun_direction = d
    s["sun_direction"] = sun_direction
    return s
