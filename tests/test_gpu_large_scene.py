"""GPU: scenes that leave the small-scene fast paths — more shapes than fit the LDS
winner/material staging (global-memory fallback), sphere runs longer than one 4-sphere
scalar block, many runs of alternating types, many box instances."""
import numpy as np
import pytest

from conftest import bits_equal
from simple_raytracer_amd import records as R, scenes as S

pytestmark = pytest.mark.gpu


def _many_shapes(n_spheres, n_boxes, seed):
    rng = np.random.RandomState(seed)
    mats = np.zeros(40, R.MATERIAL)
    for i in range(40):
        kind = i % 5
        col = rng.uniform(0.2, 1.0, 3)
        if kind == 0:
            mats[i] = R.material(col)
        elif kind == 1:
            mats[i] = R.material(col, smoothness=rng.uniform(0.3, 1.0), metallic=rng.uniform(0.3, 1.0))
        elif kind == 2:
            mats[i] = R.material(col, smoothness=1.0, transmittance=rng.uniform(0.5, 1.0), refraction_index=rng.uniform(1.1, 1.8))
        elif kind == 3:
            mats[i] = R.material(col, specular=rng.uniform(0.1, 0.6), smoothness=0.8)
        else:
            mats[i] = R.material(col, emission=col, emission_strength=rng.uniform(0.5, 3.0))
    tris = R.box_triangles()
    shapes = [R.plane(0, (0, -1, 0), (0, 1, 0))]
    for i in range(n_spheres):
        p = rng.uniform([-5, -0.8, -8], [5, 3, 1])
        shapes.append(R.sphere(int(rng.randint(40)), p, rng.uniform(0.08, 0.35)))
        if i % 37 == 5:  # break the sphere runs with planes and boxes
            shapes.append(R.plane(int(rng.randint(40)), (0, 0, -9 - i * 0.01), (0, 0, 1)))
        if i % 53 == 7 and n_boxes:
            shapes.append(R.model(int(rng.randint(40)), tris, 0, 12,
                                  R.mat_mul(R.translate(rng.uniform([-4, -0.5, -6], [4, 2, 0])), R.scale_matrix(rng.uniform(0.1, 0.4, 3)))))
    for i in range(n_boxes):
        shapes.append(R.model(int(rng.randint(40)), tris, 0, 12,
                              R.mat_mul(R.translate(rng.uniform([-4, -0.5, -6], [4, 2, 0])), R.mat_mul(R.euler_yxz(rng.uniform(0, 3), rng.uniform(0, 1), 0), R.scale_matrix(rng.uniform(0.1, 0.4, 3))))))
    a = np.zeros(len(shapes), R.SHAPE)
    for i, s in enumerate(shapes):
        a[i] = s
    return a, tris, mats


@pytest.mark.parametrize("n_spheres,n_boxes", [(300, 0), (260, 60), (5, 300)])
def test_many_shapes_vs_oracle(n_spheres, n_boxes, sky, oracle):
    from simple_raytracer_amd import build, tracer as T
    build.build_hip()
    shapes, tris, mats = _many_shapes(n_spheres, n_boxes, seed=n_spheres + n_boxes)
    assert len(shapes) * 32 + len(mats) * 64 > 8192  # beyond the LDS staging threshold
    w, h = 48, 32
    rd = R.render_data(w, h, 2, 6, camera_to_world=S.default_camera(), time=777)
    sd = R.scene_data(len(shapes))
    t = T.Tracer(w, h)
    t.set_skybox(sky)
    t.options, t.scene_data = rd, sd
    t.count_triangles(True)
    t.update_scene(shapes, tris, mats)
    t.clear_canvas()
    t.trace()
    got = t.read_canvas()
    c = t.counters()
    want, oc = oracle.render(rd, sd, shapes, tris, mats, sky, counters=True)
    assert bits_equal(got, want)
    for k in ("paths", "rays", "sky", "tri_tests", "tri_pass_u"):
        assert c[k] == oc[k], k
    t.close()


def _big_models_between_shapes(seed):
    """Big models (>= 128 triangles: the array scan suspends the rays that enter their boxes, csrc/kernels.hip) with
    spheres, planes, boxes and a second instance of the same mesh BEHIND them in the shape array, overlapping boxes,
    glass and mirror materials: a suspended ray has to come back with its closest hit so far and see every later shape."""
    rng = np.random.RandomState(seed)
    mats = np.zeros(6, R.MATERIAL)
    mats[0] = R.material((0.8, 0.8, 0.8))
    mats[1] = R.material((0.5, 0.9, 0.6), smoothness=1.0, transmittance=0.9, refraction_index=1.4)
    mats[2] = R.material((0.9, 0.7, 0.3), smoothness=0.8, metallic=0.7)
    mats[3] = R.material((0.3, 0.4, 0.9), specular=0.3, smoothness=0.9)
    mats[4] = R.material((1, 1, 1), emission=(1.0, 0.8, 0.5), emission_strength=2.0)
    mats[5] = R.material((0.9, 0.2, 0.2))
    box = R.box_triangles()
    mesh_a = S.blob_mesh(14, 9, seed=seed, smooth=True)        # 2 * 14 * 8 = 224+ triangles: big
    mesh_b = S.blob_mesh(10, 9, seed=seed + 1, smooth=False)   # big as well, flat shaded
    tris = R.concat(R.TRIANGLE, box, mesh_a, mesh_b)
    ia, ib = 12, 12 + len(mesh_a)
    assert len(mesh_a) >= 128 and len(mesh_b) >= 128
    xf = lambda p, yaw, s: R.mat_mul(R.translate(p), R.mat_mul(R.euler_yxz(yaw, 0.2, 0.0), R.scale_matrix(s)))
    shapes = [
        R.sphere(2, (-2.4, 0.2, -1.5), 0.7),
        R.model(1, tris, ia, len(mesh_a), xf((-0.6, 0.1, -1.0), 0.5, (1.0, 1.0, 1.0))),   # big, glass
        R.plane(0, (0, -1.1, 0), (0, 1, 0)),                                               # after a big model
        R.sphere(4, (0.3, 2.2, -1.0), 0.5),
        R.model(2, tris, ib, len(mesh_b), xf((1.2, 0.0, -1.8), -0.8, (0.9, 1.2, 0.9))),   # big, overlaps the first one's box
        R.model(3, tris, 0, 12, xf((0.2, -0.6, 0.4), 0.3, (0.3, 0.3, 0.3))),               # small (box) after the big ones
        R.model(5, tris, ia, len(mesh_a), xf((0.4, 0.3, -2.6), 2.0, (0.8, 0.8, 0.8))),    # second instance of mesh a
        R.sphere(3, (2.3, -0.3, -0.4), 0.6),
    ]
    a = np.zeros(len(shapes), R.SHAPE)
    for i, s in enumerate(shapes):
        a[i] = s
    return a, tris, mats


@pytest.mark.parametrize("seed,spp", [(11, 3), (12, 24)])
def test_suspended_scans_resume_where_they_left(seed, spp, sky, oracle):
    from simple_raytracer_amd import build, tracer as T
    build.build_hip()
    shapes, tris, mats = _big_models_between_shapes(seed)
    w, h = 64, 40
    rd = R.render_data(w, h, spp, 8, camera_to_world=S.default_camera(), time=4242 + seed)
    sd = R.scene_data(len(shapes))
    t = T.Tracer(w, h)
    t.set_skybox(sky)
    t.options, t.scene_data = rd, sd
    for count in (True, False):  # the instrumented and the plain kernel
        t.count_triangles(count)
        t.update_scene(shapes, tris, mats)
        t.clear_canvas()
        t.reset_counters()
        t.trace()
        got = t.read_canvas()
        c = t.counters()
        want, oc = oracle.render(rd, sd, shapes, tris, mats, sky, counters=True)
        assert bits_equal(got, want)
        for k in ("paths", "rays", "sky") + (("tri_tests", "tri_pass_u") if count else ()):
            assert c[k] == oc[k], k
        assert c["watchdog"] == 0
    d = t.debug_counters()
    assert d["scans"] > 0 and d["scan_lanes"] >= d["scans"]  # the scan queue was really in play
    t.close()
