"""GPU (MI355X): the HIP path, called through the C ABI, against (1) the golden vectors
generated from the reference kernel and (2) the CPU oracle on the same seeded inputs.
Bar: the accumulation canvas is BIT-IDENTICAL (NaN == NaN) — stricter than the 1e-4 of
BASELINE.json's north_star — and the ARGB bytes are identical."""
import numpy as np
import pytest

import cases as C
import golden_io
from conftest import bits_equal
from simple_raytracer_amd import records as R, scenes as S

pytestmark = pytest.mark.gpu
CASES = golden_io.load_cases()


@pytest.fixture(scope="module")
def T():
    from simple_raytracer_amd import build, tracer
    build.build_hip()
    return tracer


def make_tracer(T, g, sky, rd=None, lib=None):
    rd = g["rd"] if rd is None else rd
    t = T.Tracer(int(rd["width"]), int(rd["height"]), lib=lib)
    t.set_skybox(sky)
    t.options = rd.copy()
    t.scene_data = g["sd"].copy()
    t.update_scene(g["shapes"], g["tris"], g["mats"])
    t.clear_canvas()
    return t


@pytest.mark.parametrize("name", sorted(CASES))
def test_golden_canvas_and_bytes(name, T, sky):
    g = CASES[name]
    t = make_tracer(T, g, sky)
    out = None
    for i, tm in enumerate(g["frames"]):
        t.options["time"] = np.uint32(tm)
        out = t.render(i + 1)  # ticks_stopped = frames so far, as main.cpp:290,337
    canvas = t.read_canvas()
    assert bits_equal(canvas, g["canvas"]), f"{name}: max |diff| = {np.nanmax(np.abs(canvas - g['canvas']))}"
    assert np.array_equal(out.reshape(g["argb"].shape), g["argb"])
    t.close()


@pytest.mark.parametrize("name", ["mixed", "glass", "mesh_smooth"])
def test_counters_equal_oracle(name, T, sky, oracle):
    g = CASES[name]
    t = make_tracer(T, g, sky)
    t.count_triangles(True)
    t.reset_counters()
    t.trace()
    c = t.counters()
    _, oc = oracle.render(g["rd"], g["sd"], g["shapes"], g["tris"], g["mats"], sky, counters=True)
    for k in ("paths", "rays", "sky", "tri_tests", "tri_pass_u", "nan_pixels"):
        assert c[k] == oc[k], (k, c, oc)
    t.close()


def test_config0_sphere_scene_256x256x16_vs_oracle(T, sky, oracle):
    """BASELINE.json configs[0] in full against the oracle."""
    shapes, tris, mats = S.sphere_scene()
    rd = R.render_data(256, 256, 16, 10, camera_to_world=S.default_camera(), time=12345)
    g = dict(shapes=shapes, tris=tris, mats=mats, rd=rd, sd=R.scene_data(len(shapes)))
    t = make_tracer(T, g, sky)
    t.trace()
    got = t.read_canvas()
    want = oracle.render(rd, g["sd"], shapes, tris, mats, sky)
    assert bits_equal(got, want)
    t.close()


def test_mesh_scene_968_triangles_vs_oracle(T, sky, oracle):
    """configs[2] geometry (two ~1k-triangle instances + plane) on a crop-sized canvas."""
    shapes, tris, mats = S.mesh_scene(2)
    rd = R.render_data(160, 90, 4, 10, camera_to_world=S.default_camera(), time=424242)
    g = dict(shapes=shapes, tris=tris, mats=mats, rd=rd, sd=R.scene_data(len(shapes)))
    t = make_tracer(T, g, sky)
    t.trace()
    assert bits_equal(t.read_canvas(), oracle.render(rd, g["sd"], shapes, tris, mats, sky))
    t.close()


def test_scan_pool_sizes_are_invisible(T, sky, oracle, monkeypatch):
    """Array scan: at the end of a launch the waves pool the rays left in their scan stacks (kernels.hip). Whatever the pool
    holds -- nothing at all, one block of 64 rays (it overflows at once: the waves keep what does not fit), seven blocks, the
    full size -- and however many launches (sample batches) there are, the canvas is the oracle's bit for bit."""
    shapes, tris, mats = S.mesh_scene(2)
    rd = R.render_data(160, 90, 6, 10, camera_to_world=S.default_camera(), time=777)
    g = dict(shapes=shapes, tris=tris, mats=mats, rd=rd, sd=R.scene_data(len(shapes)))
    want = oracle.render(rd, g["sd"], shapes, tris, mats, sky)
    dev = T.load_dev_library()  # the knobs below exist in the -DSRT_DEV_KNOBS build only
    for env, val in ((None, None), ("SRT_POOL_BLOCKS", "1"), ("SRT_POOL_BLOCKS", "7"), ("SRT_NO_SCAN_POOL", "1")):
        if env:
            monkeypatch.setenv(env, val)
        for budget in (0, 160 * 90 * 12 * 2):
            t = make_tracer(T, g, sky, lib=dev)
            if budget:
                t.set_radiance_budget(budget)  # three launches of two samples
            t.trace()
            assert bits_equal(t.read_canvas(), want), f"{env}={val}, budget {budget}"
            assert t.counters()["watchdog"] == 0
            t.close()
        if env:
            monkeypatch.delenv(env)


@pytest.mark.parametrize("accel", [0, 1])
def test_chunk_sizes_are_invisible(T, sky, oracle, monkeypatch, accel):
    """How many sub-jobs the work cursor hands out at a time (srt_trace's rule, or SRT_JOB_CAP_SUBS) decides which wave traces
    which path and nothing else: array scan and BVH, chunks of 1 / 3 / 16 sub-jobs and the rule's own choice give the oracle's canvas."""
    shapes, tris, mats = S.mesh_scene(2)
    rd = R.render_data(96, 54, 40, 10, camera_to_world=S.default_camera(), time=31337)
    g = dict(shapes=shapes, tris=tris, mats=mats, rd=rd, sd=R.scene_data(len(shapes)))
    want = oracle.render(rd, g["sd"], shapes, tris, mats, sky)
    for cap in (None, "1", "3", "16"):
        if cap:
            monkeypatch.setenv("SRT_JOB_CAP_SUBS", cap)
        t = T.Tracer(96, 54, lib=T.load_dev_library())
        t.set_skybox(sky)
        if accel:
            t.set_acceleration(T.ACCEL_BVH)
        t.options, t.scene_data = rd.copy(), g["sd"].copy()
        t.update_scene(shapes, tris, mats)
        t.clear_canvas()
        t.trace()
        assert bits_equal(t.read_canvas(), want), f"accel {accel}, chunks of {cap} sub-jobs"
        assert t.counters()["watchdog"] == 0
        t.close()
        if cap:
            monkeypatch.delenv("SRT_JOB_CAP_SUBS")


def test_100k_triangle_mesh_vs_oracle(T, sky, oracle):
    """configs[4] geometry (one 99,904-triangle flat mesh + plane), tiny canvas so the
    brute-force oracle finishes in seconds."""
    shapes, tris, mats = S.mesh_scene(1, 224, 224, smooth=False)
    rd = R.render_data(40, 24, 1, 10, camera_to_world=S.default_camera(), time=99)
    g = dict(shapes=shapes, tris=tris, mats=mats, rd=rd, sd=R.scene_data(len(shapes)))
    t = make_tracer(T, g, sky)
    t.trace()
    assert bits_equal(t.read_canvas(), oracle.render(rd, g["sd"], shapes, tris, mats, sky))
    t.close()


@pytest.mark.parametrize("world,rpb", [(2, 8), (8, 8), (3, 5)])
def test_virtual_tiles_reproduce_untiled_canvas(T, sky, world, rpb):
    """Tile invariance: every rank's packed rows, unpermuted, equal the
    single-device canvas bit for bit — all 'ranks' run on this one GPU."""
    g = CASES["mixed"]
    want = g["canvas"]
    h = int(g["rd"]["height"])
    padded = T.padded_rows(h, world, rpb)
    gathered = np.zeros((world * padded,) + want.shape[1:], np.float32)
    for r in range(world):
        t = make_tracer(T, g, sky)
        t.set_partition(r, world, rpb)
        t.trace()
        rows = t.read_canvas()
        gathered[r * padded:r * padded + rows.shape[0]] = rows
        t.close()
    assert bits_equal(T.unpermute(gathered, h, world, rpb), want)


def test_update_scene_semantics(T, sky, oracle):
    """SceneData takes effect only at update_scene (src/tracer.cpp:95); RenderData at
    every render (:105); empty update keeps rendering sky; buffers only grow."""
    g = CASES["spheres"]
    t = make_tracer(T, g, sky)
    t.scene_data["sun_intensity"] = 7.0          # edited but NOT pushed
    t.trace()
    assert bits_equal(t.read_canvas(), g["canvas"])
    t.clear_canvas()
    t.update_scene(g["shapes"][:0], g["tris"], g["mats"])  # now empty scene, new sun
    t.trace()
    sd = g["sd"].copy()
    sd["sun_intensity"] = 7.0
    sd["num_shapes"] = 0
    want = oracle.render(g["rd"], sd, g["shapes"][:0], g["tris"], g["mats"], sky)
    assert bits_equal(t.read_canvas(), want)
    t.close()


def test_errors_are_loud(T, sky):
    g = CASES["spheres"]
    t = T.Tracer(32, 32)
    with pytest.raises(T.SrtError):      # no skybox yet
        t.trace()
    t.set_skybox(sky)
    bad = g["shapes"].copy()
    bad["material"][0] = 99
    with pytest.raises(T.SrtError):
        t.update_scene(bad, g["tris"], g["mats"])
    t.options = R.render_data(64, 32)
    with pytest.raises(T.SrtError):      # no resize
        t.trace()
    t.close()


def test_full_size_properties_1080p(T, sky):
    """At BASELINE size (1920x1080) the oracle is too slow; check size-independent
    properties: (a) a 64x36 window of the full frame equals the same pixels rendered by
    a second handle restricted by partition to those rows... (b) determinism: two runs
    are bit-identical; (c) linearity of accumulation: two frames with the same seed give
    exactly 2x one frame; (d) counters: rays = paths + bounces >= paths, sky <= paths."""
    shapes, tris, mats = S.sphere_scene()
    rd = R.render_data(1920, 1080, 4, 10, camera_to_world=S.default_camera(), time=2024)
    g = dict(shapes=shapes, tris=tris, mats=mats, rd=rd, sd=R.scene_data(len(shapes)))
    t = make_tracer(T, g, sky)
    t.reset_counters()
    t.trace()
    a = t.read_canvas()
    c = t.counters()
    assert c["paths"] == 1920 * 1080 * 4 and c["rays"] >= c["paths"] and c["sky"] <= c["paths"]
    t.trace()
    b = t.read_canvas()
    assert bits_equal(b, a + a)          # same seed twice: x + x is exact in binary fp
    t.clear_canvas()
    t.trace()
    assert bits_equal(t.read_canvas(), a)
    # (a) rows 536..544 via a partition of 135 ranks x 8-row blocks: rank 67 owns block 67
    t.set_partition(67, 135, 8)
    t.trace()
    rows = t.read_canvas()
    assert rows.shape[0] == 8 and bits_equal(rows, a[536:544])
    t.close()


def test_oracle_agrees_on_rows_of_the_1080p_frame(T, sky, oracle):
    """...and those 8 full-width rows of the 1080p frame against the oracle."""
    shapes, tris, mats = S.sphere_scene()
    rd = R.render_data(1920, 1080, 4, 10, camera_to_world=S.default_camera(), time=2024)
    g = dict(shapes=shapes, tris=tris, mats=mats, rd=rd, sd=R.scene_data(len(shapes)))
    t = make_tracer(T, g, sky)
    t.set_partition(67, 135, 8)
    t.trace()
    rows = t.read_canvas()
    want = oracle.render(rd, g["sd"], shapes, tris, mats, sky, rows=(536, 544))
    assert bits_equal(rows, want[536:544])
    t.close()


def test_device_math_equals_host_math(T, oracle):
    """Bit-for-bit determinism of csrc/detmath.h across x86-64 and gfx950, and of the
    kernel-local sqrt/log/cos specialisations against their generic definitions.
    Mismatch counts: EVERY one of the 2^32 RNG outputs (stride 1). Checksums vs the host
    build: every 61st (the CPU side takes a few seconds)."""
    t = T.Tracer(8, 8)
    full = t.selftest_math(1)
    assert full[0] == 0, f"sqrt_ieee differs from IEEE sqrt on {full[0]} of 2^32 bit patterns"
    assert full[1] == 0, f"log_unit differs from dm_logf on {full[1]} RNG outputs"
    assert full[2] == 0, f"cos_2pi differs from dm_cosf on {full[2]} RNG outputs"
    assert full[8] == 0, f"div3 differs from IEEE division on {full[8]} of 2^32 random operand sets"
    assert full[10] == 0, f"unguarded Box-Muller sqrt differs from IEEE sqrt on {full[10]} RNG outputs"
    assert full[11] == 0, f"folding the 2^-32 RNG scaling changes log or theta on {full[11]} RNG outputs"
    assert full[12] == 0, f"sqrt_rsq differs from IEEE sqrt on {full[12]} floats of [2^-96, inf)"
    assert full[14] == 0, f"sign_fast differs from dm_sign on {full[14]} bit patterns"
    assert full[15] == 0, f"powi_uniform differs from dm_powi on {full[15]} (value, exponent) pairs"
    assert full[13] == 0, f"division by the image size through the host's reciprocal differs from IEEE division on {full[13]} of 2^32 quotients"
    dev = t.selftest_math(61)
    host = oracle.math_checksums(61)
    assert dev[:3] == [0, 0, 0] and dev[8] == 0 and dev[10:15] == [0, 0, 0, 0, 0]
    assert dev[3:8] == host[3:8], (dev, host)
    assert dev[9] == host[0], "the built-in normalize (division-free rsqrt) differs between gfx950 and x86-64"
    t.close()


@pytest.mark.parametrize("name", ["spheres", "mixed", "spheres_accum"])
def test_sample_batches_are_invisible(name, T, sky):
    """A radiance budget too small for the dispatch forces several sample batches (ordered
    sum carried across them): the canvas must not change by a single bit."""
    g = CASES[name]
    t = make_tracer(T, g, sky)
    w, h, ns = int(g["rd"]["width"]), int(g["rd"]["height"]), int(g["rd"]["num_samples"])
    t.set_radiance_budget(w * h * 12 * max(1, ns // 3))  # ~3 batches, one of them ragged
    for i, tm in enumerate(g["frames"]):
        t.options["time"] = np.uint32(tm)
        t.trace()
    assert bits_equal(t.read_canvas(), g["canvas"])
    t.set_radiance_budget(1)  # degenerate: one sample per batch
    t.clear_canvas()
    for i, tm in enumerate(g["frames"]):
        t.options["time"] = np.uint32(tm)
        t.trace()
    assert bits_equal(t.read_canvas(), g["canvas"])
    t.close()


def test_odd_frame_in_tiny_batches(T, sky, oracle):
    """A frame whose pixel count is not a multiple of 4 (37 x 41) traced in batches of one and three samples: the two radiance
    buffers of overlapping batches must both start on 16-byte boundaries, and the magic-number pixel / row
    arithmetic of the camera rays must agree with the oracle's divisions for an odd width."""
    g = CASES["spheres"]
    rd = g["rd"].copy()
    rd["width"], rd["height"], rd["num_samples"] = 37, 41, 7
    rd["aspect_ratio"] = np.float32(37 / 41)
    want = oracle.render(rd, g["sd"], g["shapes"], g["tris"], g["mats"], sky)
    for budget in (1, 37 * 41 * 12 * 3):
        t = make_tracer(T, g, sky, rd)
        t.set_radiance_budget(budget)
        t.trace()
        assert bits_equal(t.read_canvas(), want), f"budget {budget}"
        t.close()


def test_zero_and_negative_sample_counts(T, sky, oracle):
    """num_samples = 0 -> colour 0/0 = NaN added to every pixel; negative -> -0 (render.cl:520)."""
    g = CASES["spheres"]
    for ns in (0, -3):
        rd = g["rd"].copy()
        rd["num_samples"] = ns
        t = make_tracer(T, g, sky, rd)
        t.trace()
        want = oracle.render(rd, g["sd"], g["shapes"], g["tris"], g["mats"], sky)
        assert bits_equal(t.read_canvas(), want)
        t.close()


def test_nan_poisoning_is_reproduced_and_counted(T, sky, oracle):
    """The reference lets a NaN radiance poison a pixel for good (render.cl:522, SURVEY
    H4). Force it with a NaN material colour: same pixels NaN on GPU and oracle, the
    nan_pixels counter agrees, and the resolve maps NaN to byte 0 instead of crashing."""
    g = CASES["spheres"]
    mats = g["mats"].copy()
    mats["color"][3] = (np.nan, 0.5, 0.5)  # the big diffuse sphere
    t = make_tracer(T, dict(g, mats=mats), sky)
    t.reset_counters()
    out = t.render(1)
    got = t.read_canvas()
    want, oc = oracle.render(g["rd"], g["sd"], g["shapes"], g["tris"], mats, sky, counters=True)
    assert np.isnan(want[..., 0]).sum() > 100
    assert bits_equal(got, want)
    assert t.counters()["nan_pixels"] == oc["nan_pixels"] > 0
    assert np.array_equal(out.reshape(want.shape[:2] + (4,)), oracle.average(1, want))
    assert t.counters()["watchdog"] == 0
    t.close()


def test_rank_without_rows_and_4k_frame_slice(T, sky, oracle):
    """(a) More ranks than row blocks: a rank that owns nothing must trace nothing and not
    fail. (b) BASELINE configs[3] geometry: 8 rows of the 3840x2160 frame (one of 270
    blocks) against the oracle — exercises 64-bit item / radiance offsets at 4K width."""
    g = CASES["spheres"]
    t = T.Tracer(64, 5)
    t.set_skybox(sky)
    t.options = R.render_data(64, 5, 2, 4, camera_to_world=S.default_camera())
    t.scene_data = g["sd"].copy()
    t.update_scene(g["shapes"], g["tris"], g["mats"])
    t.set_partition(7, 8, 8)
    assert t.owned_rows == 0
    t.reset_counters()
    t.trace()
    assert t.read_canvas().shape == (0, 64, 4) and t.counters()["paths"] == 0
    t.close()

    shapes, tris, mats = S.sphere_scene()
    rd = R.render_data(3840, 2160, 16, 10, camera_to_world=S.default_camera(), time=4096 + 1)
    sd = R.scene_data(len(shapes))
    t = T.Tracer(3840, 2160)
    t.set_skybox(sky)
    t.options, t.scene_data = rd, sd
    t.update_scene(shapes, tris, mats)
    t.set_partition(200, 270, 8)
    t.trace()
    rows = t.read_canvas()
    want = oracle.render(rd, sd, shapes, tris, mats, sky, rows=(1600, 1608))
    assert rows.shape[0] == 8 and bits_equal(rows, want[1600:1608])
    assert t.counters()["watchdog"] == 0
    t.close()


def test_async_progressive_frames_equal_blocking_frames(T, sky):
    """The front-end loop (src/main.cpp:277-337) with srt_render_async: enqueue several
    progressive frames back to back, wait once; last image and canvas equal the blocking
    sequence of the golden case."""
    g = CASES["spheres_accum"]
    t = make_tracer(T, g, sky)
    h, w = g["argb"].shape[:2]
    outs = [np.zeros(w * h * 4, np.uint8) for _ in g["frames"]]
    for i, tm in enumerate(g["frames"]):
        t.options["time"] = np.uint32(tm)
        t.render_async(i + 1, outs[i])
    t.synchronize()
    assert bits_equal(t.read_canvas(), g["canvas"])
    assert np.array_equal(outs[-1].reshape(g["argb"].shape), g["argb"])
    assert not np.array_equal(outs[0], outs[-1])  # earlier frames were really delivered separately
    t.close()


def test_caller_owned_canvas_stream_and_external_resolve(T, sky):
    """The zero-copy hand-off bench.py relies on: canvas in a torch tensor (srt_bind_canvas),
    launches on a torch stream (srt_bind_stream), resolve of a caller-owned buffer
    (srt_resolve_external), device pointers reported by srt_device_buffers."""
    import torch
    g = CASES["mixed"]
    w, h = int(g["rd"]["width"]), int(g["rd"]["height"])
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        t = make_tracer(T, g, sky)
        t.bind_stream(stream.cuda_stream)
        canvas = torch.zeros((h, w, 4), dtype=torch.float32, device=dev)
        t.bind_canvas(canvas.data_ptr(), canvas.numel() * 4)
        cp, cb, ap, ab = t.device_buffers()
        assert cp == canvas.data_ptr() and cb == canvas.numel() * 4 and ap and ab >= w * h * 4
        t.clear_canvas()
        t.trace()
        argb = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
        t.resolve_external(canvas.data_ptr(), w * h, 1, argb.data_ptr())
        stream.synchronize()
        assert bits_equal(canvas.cpu().numpy(), g["canvas"])
        assert np.array_equal(argb.cpu().numpy(), g["argb"])
        with pytest.raises(T.SrtError):
            t.bind_canvas(canvas.data_ptr(), 16)  # too small for the owned rows
        t.bind_canvas(0, 0)                        # back to the handle's own canvas
        t.bind_stream(0)
        t.clear_canvas()
        t.trace()
        assert bits_equal(t.read_canvas(), g["canvas"])
        t.close()


@pytest.mark.parametrize("case", ["plain", "minus_zero_colour", "nan_colour", "tiny_specular", "one_specular", "no_glass"])
def test_draws_the_scene_decides(T, sky, oracle, case):
    """srt_update_scene tells the kernel what holds for EVERY material (device_types.h SRT_MF_*): with no specular material
    the specular draw is not made (the generator steps over it) and mix(colour, 1, 0) is the colour itself -- valid for finite
    colours that are not -0 only. Each edge of that rule against the oracle, bit for bit."""
    shapes, tris, mats = S.sphere_scene()
    mats = mats.copy()
    if case == "minus_zero_colour":
        mats["color"][1] = (-0.0, 0.5, 1.0)  # fma(1 - (-0), 0, -0) = +0: the shortcut would keep -0
    if case == "nan_colour":
        mats["color"][2] = (np.nan, 0.5, np.inf)
    if case == "tiny_specular":
        mats["specular"][0] = 1e-12  # true for exactly one output of the generator (u = 0): the draw must be made
    if case == "one_specular":
        mats["specular"][3] = 0.35
    if case == "no_glass":
        mats["transmittance"][:] = 0.0
    rd = R.render_data(64, 40, 24, 10, camera_to_world=S.default_camera(), time=4711)
    g = dict(shapes=shapes, tris=tris, mats=mats, rd=rd, sd=R.scene_data(len(shapes)))
    t = make_tracer(T, g, sky)
    t.trace()
    assert bits_equal(t.read_canvas(), oracle.render(rd, g["sd"], shapes, tris, mats, sky)), case
    t.close()


def test_render_fuses_the_resolve_and_times_only_on_request(T, sky):
    """srt_render = trace + ordered reduction + resolve with the last two in ONE launch (the reduction resolves each pixel it has
    just accumulated) and no timer events -- unless srt_set_kernel_timers asks for them. Bytes and canvas equal the golden's
    either way, and equal trace() + resolve() as two calls."""
    g = CASES["spheres_accum"]
    for timers in (False, True):
        t = make_tracer(T, g, sky)
        t.set_kernel_timers(timers)
        out = None
        for i, tm in enumerate(g["frames"]):
            t.options["time"] = np.uint32(tm)
            out = t.render(i + 1)
        trace_ms, resolve_ms = t.last_kernel_ms()
        assert (trace_ms > 0.0) == timers and resolve_ms == 0.0
        assert bits_equal(t.read_canvas(), g["canvas"]) and np.array_equal(out.reshape(g["argb"].shape), g["argb"])
        t.close()
    t = make_tracer(T, g, sky)
    for i, tm in enumerate(g["frames"]):
        t.options["time"] = np.uint32(tm)
        t.trace()
        t.resolve(i + 1)
    assert t.last_kernel_ms()[0] > 0.0 and np.array_equal(t.read_argb().reshape(g["argb"].shape), g["argb"])
    t.close()
    # a dispatch that runs as several sample batches: only the LAST reduction resolves
    t = make_tracer(T, g, sky)
    h, w = g["argb"].shape[:2]
    t.set_radiance_budget(w * h * 12)  # one sample per batch
    out = None
    for i, tm in enumerate(g["frames"]):
        t.options["time"] = np.uint32(tm)
        out = t.render(i + 1)
    assert t.last_trace_launches()[0] == int(g["rd"]["num_samples"])
    assert bits_equal(t.read_canvas(), g["canvas"]) and np.array_equal(out.reshape(g["argb"].shape), g["argb"])
    t.close()
