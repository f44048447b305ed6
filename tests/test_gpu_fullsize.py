"""GPU (MI355X): every BASELINE config at the sample count it is QUOTED on, against the CPU oracle.

The other parity tests run 1-16 spp; the bench times 1024 spp in one dispatch (2.1e9 work-items,
6.4e9 radiance floats: the only place where item indices pass 2^32 and a 25 GB batch exists). Here:
  configs[1]  1920x1080, 1024 spp, ONE dispatch of the whole frame; 4 bands of 8 rows vs the oracle
  configs[3]  3840x2160, 4096 spp, one 8-row block of the frame, forced through 4 sample batches
  configs[2]  two 968-triangle meshes + plane, 512 spp, on a 128x72 canvas (array scan and BVH)
  configs[4]  99,904-triangle mesh + plane, 256 spp, on a 32x18 canvas (array scan and BVH)
Bar: canvas bit-identical (NaN == NaN), work counters equal."""
import os

import numpy as np
import pytest

from conftest import bits_equal
from simple_raytracer_amd import records as R, scenes as S

pytestmark = pytest.mark.gpu
THREADS = max(1, min(len(os.sched_getaffinity(0)), 64))


@pytest.fixture(scope="module")
def T():
    from simple_raytracer_amd import build, tracer
    build.build_hip()
    return tracer


def tracer_for(T, sky, shapes, tris, mats, rd, accel=None):
    t = T.Tracer(int(rd["width"]), int(rd["height"]))
    t.set_skybox(sky)
    t.options = rd.copy()
    t.scene_data = R.scene_data(len(shapes))
    if accel is not None:
        t.set_acceleration(accel)
    t.update_scene(shapes, tris, mats)
    t.clear_canvas()
    return t


def test_config1_full_frame_1024spp_one_dispatch(T, sky, oracle):
    """The frame bench.py times, as it times it: 1920x1080x1024 spp in one srt_trace."""
    shapes, tris, mats = S.sphere_scene()
    rd = R.render_data(1920, 1080, 1024, 10, camera_to_world=S.default_camera(), time=12345)
    t = tracer_for(T, sky, shapes, tris, mats, rd)
    t.reset_counters()
    t.trace()
    got = t.read_canvas()
    c = t.counters()
    assert c["paths"] == 1920 * 1080 * 1024 and c["watchdog"] == 0
    sd = R.scene_data(len(shapes))
    rays = 0
    for y0 in (0, 360, 712, 1072):  # sky rows, sphere rows, floor rows, the last block of the frame
        want, oc = oracle.render(rd, sd, shapes, tris, mats, sky, rows=(y0, y0 + 8), nthreads=THREADS, counters=True)
        assert bits_equal(got[y0:y0 + 8], want[y0:y0 + 8]), f"rows {y0}..{y0 + 8}"
        rays += oc["rays"]
    assert 0 < rays < c["rays"]
    # the same frame as 32 progressive frames x 32 spp is a different sum (other seeds), but one frame of it doubles exactly
    t.close()


def test_config3_block_4096spp_through_sample_batches(T, sky, oracle):
    """configs[3] (3840x2160x4096 spp) is 408 GB of radiance: it runs as sample batches with the ordered
    sum carried between them. One 8-row block of that frame, its budget cut so that it takes 4 batches."""
    shapes, tris, mats = S.sphere_scene()
    rd = R.render_data(3840, 2160, 4096, 10, camera_to_world=S.default_camera(), time=12345)
    t = tracer_for(T, sky, shapes, tris, mats, rd)
    t.set_partition(100, 270, 8)
    assert t.owned_rows == 8
    t.set_radiance_budget(8 * 3840 * 12 * 1024)
    t.reset_counters()
    t.trace()
    rows = t.read_canvas()
    c = t.counters()
    want, oc = oracle.render(rd, R.scene_data(len(shapes)), shapes, tris, mats, sky, rows=(800, 808), nthreads=THREADS, counters=True)
    assert bits_equal(rows, want[800:808])
    for k in ("paths", "rays", "sky", "nan_pixels"):
        assert c[k] == oc[k], (k, c, oc)
    assert c["watchdog"] == 0
    t.close()


@pytest.mark.parametrize("accel", ["scan", "bvh"])
def test_config2_meshes_512spp_crop(T, sky, oracle, accel):
    shapes, tris, mats = S.mesh_scene(2)
    rd = R.render_data(128, 72, 512, 10, camera_to_world=S.default_camera(), time=12345)
    t = tracer_for(T, sky, shapes, tris, mats, rd, T.ACCEL_BVH if accel == "bvh" else None)
    t.count_triangles(accel == "scan")
    t.reset_counters()
    t.trace()
    got = t.read_canvas()
    c = t.counters()
    want, oc = oracle.render(rd, R.scene_data(len(shapes)), shapes, tris, mats, sky, nthreads=THREADS, counters=True)
    assert bits_equal(got, want)
    keys = ("paths", "rays", "sky", "nan_pixels") + (("tri_tests", "tri_pass_u") if accel == "scan" else ())
    for k in keys:
        assert c[k] == oc[k], (k, c, oc)
    t.close()


@pytest.mark.parametrize("accel", ["scan", "bvh"])
def test_config4_mesh100k_256spp_crop(T, sky, oracle, accel):
    shapes, tris, mats = S.mesh_scene(1, 224, 224, smooth=False)
    assert len(tris) == 99904 + 12
    rd = R.render_data(32, 18, 256, 10, camera_to_world=S.default_camera(), time=12345)
    t = tracer_for(T, sky, shapes, tris, mats, rd, T.ACCEL_BVH if accel == "bvh" else None)
    t.count_triangles(accel == "scan")
    t.reset_counters()
    t.trace()
    got = t.read_canvas()
    c = t.counters()
    want, oc = oracle.render(rd, R.scene_data(len(shapes)), shapes, tris, mats, sky, nthreads=THREADS, counters=True)
    assert bits_equal(got, want)
    keys = ("paths", "rays", "sky", "nan_pixels") + (("tri_tests", "tri_pass_u") if accel == "scan" else ())
    for k in keys:
        assert c[k] == oc[k], (k, c, oc)
    t.close()
