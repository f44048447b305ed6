"""Scene builders for the BASELINE.json configs and the parity tests.

The reference ships no scene files and no meshes: scenes are
built interactively in its GUI. These builders fix the inputs once; they only produce
the arrays `Tracer::update_scene` takes (src/tracer.cpp:70-96). Pure host code.
"""
import numpy as np

from . import records as R

SKY_W, SKY_H = 2048, 1024  # same size as the reference's assets/skybox.png


def synthetic_sky(w=SKY_W, h=SKY_H):
    """RGBA32F sky defined by integer arithmetic only (no libm), so every machine
    regenerates the same bits. Row 0 is the BOTTOM of the picture (v = 0 <=> d.y = -1),
    matching stbi_set_flip_vertically_on_load(1) in src/tracer.cpp:44-45. Texel =
    (byte/255)^2, a gamma-2.0 stand-in for stb's pow(byte/255, 2.2) (lib/stb_image.h:
    1857-1878) that needs only one IEEE division."""
    y, x = np.mgrid[0:h, 0:w].astype(np.int64)
    # ground (dark) below the horizon, brightening sky above it, plus integer "clouds"
    horizon = h // 2
    up = np.clip(y - horizon, 0, None)
    cloud = (((x * 5 + y * 3) >> 5) ^ ((x * 3 - y * 7) >> 6)) & 31
    base = np.where(y < horizon, 70 + (y * 40) // max(horizon, 1), 230 - (up * 120) // max(h - horizon, 1))
    r = np.clip(base - 40 + cloud, 0, 255)
    g = np.clip(base - 10 + cloud, 0, 255)
    b = np.clip(base + 20 + cloud // 2, 0, 255)
    b = np.where(y < horizon, np.clip(base - 20 + cloud, 0, 255), b)
    rgb = np.stack([r, g, b], axis=-1)
    lin = (rgb * rgb).astype(np.float32) / np.float32(65025.0)
    out = np.ones((h, w, 4), np.float32)
    out[..., :3] = lin
    return np.ascontiguousarray(out)


def skybox_from_rgb8(img):
    """Texel preparation of the reference's skybox upload for an 8-bit image (h, w, c) uint8 with c in 1..4, first row = top
    of the picture: what stbi_loadf(..., 4) with stbi_set_flip_vertically_on_load(1) returns (src/tracer.cpp:42-46;
    lib/stb_image.h:1857-1878): four channels, rows flipped, colour = (float)pow(byte / 255.0f, 2.2f) with the quotient in
    float and pow in double, alpha = byte / 255.0f. Python mirror of host/skybox.hpp (srt_skybox_from_rgb8)."""
    img = np.asarray(img, np.uint8)
    if img.ndim == 2:
        img = img[..., None]
    h, w, c = img.shape
    q = np.arange(256, dtype=np.float32) / np.float32(255.0)
    lut = np.power(q.astype(np.float64), np.float64(np.float32(2.2))).astype(np.float32)
    out = np.empty((h, w, 4), np.float32)
    src = img[::-1]
    grey = c < 3
    out[..., 0] = lut[src[..., 0]]
    out[..., 1] = lut[src[..., 0 if grey else 1]]
    out[..., 2] = lut[src[..., 0 if grey else 2]]
    out[..., 3] = q[src[..., c - 1]] if c in (2, 4) else np.float32(1.0)
    return np.ascontiguousarray(out)


def _stack(items, dtype):
    a = np.zeros(len(items), dtype)
    for i, it in enumerate(items):
        a[i] = it
    return a


def sphere_scene_materials():
    """the 7 materials of the Cornell-style sphere scene."""
    return _stack(
        [
            R.material((0.8, 0.8, 0.9)),
            R.material((0.9, 0.3, 0.3)),
            R.material((0.3, 0.9, 0.4)),
            R.material((0.9, 0.95, 1.0)),
            R.material((1, 1, 1), smoothness=1.0, transmittance=1.0, refraction_index=1.5),
            R.material((0.2, 0.3, 0.9), smoothness=1.0, metallic=1.0),
            R.material((1, 1, 1), emission=(1.0, 0.2, 0.2), emission_strength=5.0),
        ],
        R.MATERIAL,
    )


def sphere_scene():
    """configs[0], [1], [3] of BASELINE.json: 3 planes + 4 spheres (readme/spheres.png
    approximated; coordinates frozen here). Returns
    (shapes, triangles, materials). The 12 box triangles are always present because
    main.cpp:102 creates them before any shape exists."""
    shapes = _stack(
        [
            R.plane(0, (0, -1, 0), (0, 1, 0)),
            R.plane(1, (-4, 0, 0), (1, 0, 0)),
            R.plane(2, (0, 0, -6), (0, 0, 1)),
            R.sphere(3, (-2, 0, -1), 1.5),
            R.sphere(4, (0.5, 0.8, -3), 1.0),
            R.sphere(5, (2.5, 0.5, -1.5), 1.0),
            R.sphere(6, (0.6, -1, 0), 0.6),
        ],
        R.SHAPE,
    )
    return shapes, R.box_triangles(), sphere_scene_materials()


def default_camera():
    return R.camera_matrix((0.0, 0.5, 5.0), 0.0, 0.0)


def blob_mesh(lon, bands, seed=1, radius=1.0, bump=0.18, smooth=True):
    """Seeded procedural closed mesh: a lat/long sphere with a few low-frequency radial
    bumps. 2*lon*(bands-1) triangles (lon=22, bands=23 -> 968 like a triangulated
    Suzanne; lon=224, bands=224 -> 99,904). smooth=True gives per-vertex normals (OBJ
    style), smooth=False gives the face normal on all three vertices (STL style)."""
    rng = np.random.RandomState(seed)
    k = rng.randint(1, 5, size=(4, 2))
    ph = rng.uniform(0, 2 * np.pi, size=(4, 2))
    amp = rng.uniform(0.3, 1.0, size=4)

    def radial(theta, phi):
        r = np.ones_like(theta)
        for i in range(4):
            r = r + bump * amp[i] * np.sin(k[i, 0] * theta + ph[i, 0]) * np.sin(k[i, 1] * phi + ph[i, 1]) / 4.0
        return radius * r

    # rings 1..bands-1 (interior), poles at ring 0 and ring `bands`
    th = np.pi * np.arange(1, bands) / bands
    ph_ = 2 * np.pi * np.arange(lon) / lon
    T, P = np.meshgrid(th, ph_, indexing="ij")
    rr = radial(T, P)
    ring = np.stack([rr * np.sin(T) * np.cos(P), rr * np.cos(T), rr * np.sin(T) * np.sin(P)], axis=-1)
    north = np.array([0.0, radial(np.zeros(1), np.zeros(1))[0], 0.0])
    south = np.array([0.0, -radial(np.full(1, np.pi), np.zeros(1))[0], 0.0])
    verts = np.concatenate([north[None], ring.reshape(-1, 3), south[None]], axis=0).astype(np.float32)
    nv = len(verts)
    S = nv - 1

    def vid(i, j):
        return 1 + i * lon + (j % lon)

    faces = []
    for j in range(lon):
        faces.append((0, vid(0, j + 1), vid(0, j)))
    for i in range(bands - 2):
        for j in range(lon):
            a, b, c, d = vid(i, j), vid(i, j + 1), vid(i + 1, j), vid(i + 1, j + 1)
            faces.append((a, b, c))
            faces.append((b, d, c))
    for j in range(lon):
        faces.append((S, vid(bands - 2, j), vid(bands - 2, j + 1)))
    faces = np.asarray(faces, np.int64)

    p0, p1, p2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    fn = np.cross((p1 - p0).astype(np.float64), (p2 - p0).astype(np.float64))
    # orient outward
    cen = (p0 + p1 + p2).astype(np.float64) / 3.0
    flip = np.sum(fn * cen, axis=1) < 0
    fn[flip] *= -1
    fl = np.linalg.norm(fn, axis=1, keepdims=True)
    fl[fl == 0] = 1.0
    fn_unit = fn / fl
    if smooth:
        vn = np.zeros((nv, 3), np.float64)
        for c in range(3):
            np.add.at(vn, faces[:, c], fn)
        vl = np.linalg.norm(vn, axis=1, keepdims=True)
        vl[vl == 0] = 1.0
        vn = (vn / vl).astype(np.float32)
        normals = [vn[faces[:, 0]], vn[faces[:, 1]], vn[faces[:, 2]]]
    else:
        f32n = fn_unit.astype(np.float32)
        normals = [f32n, f32n, f32n]
    tris = np.zeros(len(faces), R.TRIANGLE)
    for c, p in enumerate((p0, p1, p2)):
        tris["v"][:, c]["pos"] = p
        tris["v"][:, c]["normal"] = normals[c]
    return tris


def mesh_scene(n_instances=2, lon=22, bands=23, smooth=True):
    """configs[2] (two ~1k-triangle meshes + plane) / configs[4] (one 100k mesh + plane)."""
    mats = _stack(
        [
            R.material((0.8, 0.8, 0.8)),
            R.material((0.4, 0.9, 0.5), smoothness=1.0, transmittance=1.0, refraction_index=1.3),
            R.material((0.9, 0.7, 0.3), smoothness=0.7, metallic=0.6),
        ],
        R.MATERIAL,
    )
    box = R.box_triangles()
    mesh = blob_mesh(lon, bands, seed=7, smooth=smooth)
    tris = R.concat(R.TRIANGLE, box, mesh)
    shapes = [R.plane(0, (0, -1.2, 0), (0, 1, 0))]
    if n_instances >= 1:
        t1 = R.mat_mul(R.translate((-1.3, 0.1, -1.0)), R.mat_mul(R.euler_yxz(0.6, 0.2, 0.0), R.scale_matrix((1.1, 1.1, 1.1))))
        if n_instances == 1:
            t1 = R.mat_mul(R.translate((0.0, 0.2, -1.0)), R.mat_mul(R.euler_yxz(0.6, 0.2, 0.0), R.scale_matrix((1.6, 1.6, 1.6))))
        shapes.append(R.model(1, tris, 12, len(mesh), t1))
    if n_instances >= 2:
        t2 = R.mat_mul(R.translate((1.4, 0.0, -1.6)), R.mat_mul(R.euler_yxz(-0.9, -0.1, 0.0), R.scale_matrix((1.0, 1.25, 0.9))))
        shapes.append(R.model(2, tris, 12, len(mesh), t2))
    return _stack(shapes, R.SHAPE), tris, mats


def mixed_test_scene():
    """Everything at once for parity tests: planes, spheres (incl. camera inside none),
    box instances with rotated / non-uniformly scaled transforms, a small smooth mesh, a
    small flat mesh, glass, mirror, specular coat, emitter."""
    mats = _stack(
        [
            R.material((0.8, 0.8, 0.9)),
            R.material((0.9, 0.3, 0.3), specular=0.3, smoothness=0.9),
            R.material((1, 1, 1), smoothness=1.0, transmittance=0.9, refraction_index=1.5),
            R.material((0.2, 0.3, 0.9), smoothness=0.8, metallic=1.0),
            R.material((1, 1, 1), emission=(0.3, 1.0, 0.4), emission_strength=4.0),
            R.material((0.9, 0.8, 0.2), smoothness=0.5, transmittance=0.5, refraction_index=1.2, specular=0.2),
        ],
        R.MATERIAL,
    )
    box = R.box_triangles()
    smooth = blob_mesh(10, 11, seed=3, smooth=True)   # 200 triangles
    flat = blob_mesh(8, 7, seed=5, smooth=False)       # 96 triangles
    tris = R.concat(R.TRIANGLE, box, smooth, flat)
    i_s, i_f = 12, 12 + len(smooth)
    shapes = [
        R.plane(0, (0, -1, 0), (0, 1, 0)),
        R.sphere(2, (-1.6, 0.0, -0.5), 1.0),
        R.box_model(1, 0, (1.8, 0.0, -2.5)),
        R.model(3, tris, 0, 12, R.mat_mul(R.translate((-0.2, -0.4, 1.2)), R.mat_mul(R.euler_yxz(0.7, 0.3, 0.0), R.scale_matrix((0.5, 0.3, 0.7))))),
        R.model(5, tris, i_s, len(smooth), R.mat_mul(R.translate((0.6, 0.3, -0.8)), R.scale_matrix((0.9, 1.2, 0.8)))),
        R.sphere(4, (2.2, 1.9, -0.5), 0.4),
        R.model(1, tris, i_f, len(flat), R.mat_mul(R.translate((-2.6, 1.2, -2.2)), R.euler_yxz(1.1, -0.4, 0.0))),
        R.plane(0, (0, 0, -7), (0, 0, 2.0)),  # un-normalised normal on purpose (render.cl:206-221)
    ]
    return _stack(shapes, R.SHAPE), tris, mats


# ---- model files in the syntax the reference's loaders read (src/parser.cpp) ---------
def write_stl(path, tris):
    """Binary STL: 80-byte header, u32 count, 50-byte records (normal, v1, v2, v3, u16)."""
    tris = R.as_records(tris, R.TRIANGLE)
    rec = np.zeros(len(tris), np.dtype([("f", "<f4", (12,)), ("attr", "<u2")]))
    rec["f"][:, 0:3] = tris["v"][:, 0]["normal"]
    for c in range(3):
        rec["f"][:, 3 + 3 * c:6 + 3 * c] = tris["v"][:, c]["pos"]
    with open(path, "wb") as f:
        f.write(b"srt synthetic mesh".ljust(80, b" "))
        f.write(np.uint32(len(tris)).tobytes())
        f.write(rec.tobytes())


def write_obj(path, tris):
    """Wavefront OBJ with `v`, `vn` and `f a//a b//b c//c` (one v/vn pair per corner, so
    the loader's index handling is exercised without any vertex welding). %.9g
    round-trips binary32 exactly."""
    tris = R.as_records(tris, R.TRIANGLE)
    with open(path, "w") as f:
        f.write("# srt synthetic mesh\ns off\n")
        for t in tris:
            for c in range(3):
                p = t["v"][c]["pos"]
                f.write("v %.9g %.9g %.9g\n" % (p[0], p[1], p[2]))
        for t in tris:
            for c in range(3):
                n = t["v"][c]["normal"]
                f.write("vn %.9g %.9g %.9g\n" % (n[0], n[1], n[2]))
        for i in range(len(tris)):
            a = 3 * i + 1
            f.write("f %d//%d %d//%d %d//%d\n" % (a, a, a + 1, a + 1, a + 2, a + 2))
