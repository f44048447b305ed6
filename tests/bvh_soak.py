"""Long BVH-against-array-scan run (GPU): python tests/bvh_soak.py N [seed]. Each iteration: a seeded procedural mesh of
100 .. 6,000 triangles (smooth or flat), instanced once or twice under random scales (10^-3 .. 10^3, anisotropic), rotations and
translations (up to 10^4 model sizes from the world's origin), a plane, glass / metal / diffuse materials, a camera 1.5 .. 300
model sizes away that looks at the mesh; rendered 48x32 at 2 spp and 6 bounces through the array-order scan (the reference's
result, checked against the oracle elsewhere) and through the BVH: canvases must agree bit for bit, rays / sky / paths counters
too. What this adds to tests/fuzz_soak.py (whose meshes have a handful of triangles): hierarchies several levels deep, far
from the origin, seen from far away -- where the boxes' padding is what keeps an accepted hit inside them."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import srt_pkg  # noqa: E402

srt_pkg.load()
from simple_raytracer_amd import records as R, scenes as S, tracer as T  # noqa: E402


def bits_equal(a, b):
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb])


def scene(rng):
    lon, bands = int(rng.randint(6, 56)), int(rng.randint(5, 56))
    mesh = S.blob_mesh(lon, bands, seed=int(rng.randint(1 << 30)), smooth=bool(rng.randint(2)))
    tris = R.concat(R.TRIANGLE, R.box_triangles(), mesh)
    size = 10.0 ** rng.uniform(-3, 3)
    centre = rng.normal(size=3) * size * (10.0 ** rng.uniform(0, 4) if rng.randint(3) == 0 else 2.0)
    mats = np.zeros(3, R.MATERIAL)
    mats[0] = R.material((0.8, 0.8, 0.8))
    mats[1] = R.material((0.4, 0.9, 0.5), smoothness=1.0, transmittance=float(rng.randint(2)), refraction_index=1.3)
    mats[2] = R.material((0.9, 0.7, 0.3), smoothness=0.7, metallic=0.6)
    n_inst = int(rng.randint(1, 3))
    shapes = np.zeros(1 + n_inst, R.SHAPE)
    shapes[0] = R.plane(0, tuple(centre - np.array([0, 1.3 * size, 0])), (0, 1, 0))
    for i in range(n_inst):
        sc = size * rng.uniform(0.3, 1.5, size=3) if rng.randint(2) else (size, size, size)
        xf = R.mat_mul(R.translate(tuple(centre + rng.normal(size=3) * size * 0.8 * i)), R.mat_mul(R.euler_yxz(*rng.uniform(-3, 3, size=3)), R.scale_matrix(tuple(sc))))
        shapes[1 + i] = R.model(1 + i, tris, 12, len(mesh), xf)
    dist = size * 10.0 ** rng.uniform(0.2, 2.5)
    yaw, pitch = rng.uniform(-np.pi, np.pi), rng.uniform(-0.6, 0.6)
    fwd = np.array([-np.sin(yaw) * np.cos(pitch), np.sin(pitch), -np.cos(yaw) * np.cos(pitch)])
    cam = R.camera_matrix(tuple(centre - fwd * dist), float(yaw), float(pitch))
    return shapes, tris, mats, cam


def main():
    n = int(sys.argv[1])
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.RandomState(seed)
    sky = S.synthetic_sky()
    w, h = 48, 32
    ts = []
    for accel in (0, 1):
        t = T.Tracer(w, h)
        t.set_skybox(sky)
        t.set_acceleration(accel)
        t.count_triangles(True)
        ts.append(t)
    bad = hits = walked = 0
    t0 = time.time()
    for it in range(n):
        shapes, tris, mats, cam = scene(rng)
        rd = R.render_data(w, h, 2, 6, camera_to_world=cam, time=int(rng.randint(1 << 30)))
        out = []
        for t in ts:
            t.options, t.scene_data = rd, R.scene_data(len(shapes))
            t.update_scene(shapes, tris, mats)
            t.clear_canvas()
            t.reset_counters()
            t.trace()
            out.append((t.read_canvas(), t.counters()))
        (c0, k0), (c1, k1) = out
        same = bits_equal(c0, c1) and all(k0[k] == k1[k] for k in ("rays", "sky", "paths", "nan_pixels")) and k1["watchdog"] == 0
        hits += int(k0["rays"] > k0["paths"])  # some path bounced: the mesh or the plane was hit
        walked += int(k1["tri_tests"] > 0)  # some ray walked a hierarchy down to a leaf
        if not same:
            bad += 1
            print(f"MISMATCH iteration {it} seed {seed}: rays {k0['rays']} / {k1['rays']}, differing pixels {int((c0.view(np.uint32) != c1.view(np.uint32)).any(axis=2).sum())}", flush=True)
        if (it + 1) % 5000 == 0:
            print(f"{it + 1}/{n} scenes, {bad} mismatches, {hits} with bounces, {walked} with walks that reached triangles, {time.time() - t0:.0f} s", flush=True)
    print(f"DONE {n} scenes: {bad} mismatches ({hits} scenes with bounced paths, {walked} with walks that reached triangles)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
