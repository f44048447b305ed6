"""ctypes binding of the C ABI (include/srt_abi.h) and a `Tracer` with the reference's
class shape (/root/reference/include/tracer.hpp:26-88): ctor(width, height), public
`options` / `scene_data`, update_scene(shapes, triangles, materials), clear_canvas(),
render(ticks_stopped, output).

There is NO CPU fallback here: if lib/libsrt_hip.so is missing or no GPU is present the
calls raise. The oracle is never imported from this package.
"""
import ctypes as C
from pathlib import Path

import numpy as np

from . import build as B
from . import records as R

PKG = Path(__file__).resolve().parent
import os  # noqa: E402

# SRT_LIB selects an alternative build of the SAME library (A/B experiments); there is
# still no non-HIP path.
LIB_PATH = Path(os.environ["SRT_LIB"]) if os.environ.get("SRT_LIB") else PKG / "lib" / "libsrt_hip.so"

# every symbol include/srt_abi.h declares
ABI_SYMBOLS = [
    "srt_create", "srt_destroy", "srt_last_error", "srt_set_skybox", "srt_update_scene", "srt_clear_canvas",
    "srt_render", "srt_render_async", "srt_trace", "srt_set_radiance_budget", "srt_resolve", "srt_resolve_external", "srt_synchronize", "srt_read_canvas", "srt_read_argb",
    "srt_get_counters", "srt_set_count_triangles", "srt_reset_counters", "srt_last_kernel_ms", "srt_set_kernel_timers", "srt_last_trace_kernel_ms", "srt_last_trace_launches",
    "srt_device_buffers", "srt_bind_canvas", "srt_bind_stream", "srt_set_partition",
    "srt_partition_owned_rows", "srt_partition_padded_rows", "srt_partition_global_row",
    "srt_partition_unpermute", "srt_selftest_math", "srt_version", "srt_set_acceleration", "srt_acceleration_info", "srt_bvh_build_host", "srt_bvh_wide_host", "srt_debug_counters", "srt_debug_region_counters",
    "srt_comm_unique_id", "srt_comm_init", "srt_gather", "srt_resolve_gathered", "srt_gathered_buffers", "srt_read_gathered",
    "srt_group_create", "srt_group_destroy", "srt_group_last_error", "srt_group_size", "srt_group_tracer", "srt_group_set_skybox",
    "srt_group_set_acceleration", "srt_group_update_scene", "srt_group_clear_canvas", "srt_group_trace_and_gather", "srt_group_render",
    "srt_group_read_canvas", "srt_group_get_counters", "srt_render_pipelined", "srt_pipeline_flush", "srt_unpermute_device",
]

ACCEL_NONE, ACCEL_BVH = 0, 1


BVH_NODE = np.dtype({"names": ["lo", "skip", "hi", "leaf"], "formats": [(np.float32, (3,)), np.uint32, (np.float32, (3,)), np.uint32],
                     "offsets": [0, 12, 16, 28], "itemsize": 32})
BVH_END = 0xFFFFFFFF


def bvh_build_host(model_shape, triangles):
    """(nodes, order) of the BVH the library builds for one model shape record. Host only: no GPU needed."""
    lib = load_library()
    shape = np.zeros(1, R.SHAPE)
    shape[0] = model_shape
    tris = R.as_records(triangles, R.TRIANGLE)
    n = C.c_size_t(0)
    rc = lib.srt_bvh_build_host(_ptr(shape), _ptr(tris), len(tris), None, 0, None, 0, C.byref(n))
    if rc:
        raise SrtError(f"srt_bvh_build_host failed ({rc})")
    nodes = np.zeros(n.value, BVH_NODE)
    order = np.zeros(int(shape[0]["num_triangles"]), np.uint32)
    rc = lib.srt_bvh_build_host(_ptr(shape), _ptr(tris), len(tris), _ptr(nodes), len(nodes), _ptr(order), len(order), C.byref(n))
    if rc:
        raise SrtError(f"srt_bvh_build_host failed ({rc})")
    return nodes, order


BVH_NONE = 0xFFFFFFFF
BVH_LEAF_BIT = 0x80000000
BVH_INDEX_MASK = 0x0FFFFFFF
BVH_STACK_CAP = 64


def bvh_wide_host(model_shape, triangles, force_balanced=False):
    """The four-wide hierarchy the kernel walks for one model shape record (csrc/device_types.h): dict with
    blocks (n x 32 uint32; an inner block: origin, grid exponents + count, boxes as bytes, tags, first), dest (per record:
    leaf block << 2 | slot), root, stack_need, balanced. Host only: no GPU needed; leaf blocks come back empty (the device
    writes the triangles)."""
    lib = load_library()
    lib.srt_bvh_wide_host.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                      C.POINTER(C.c_size_t), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
    shape = np.zeros(1, R.SHAPE)
    shape[0] = model_shape
    tris = R.as_records(triangles, R.TRIANGLE)
    n, root, need, bal = C.c_size_t(0), C.c_uint32(0), C.c_uint32(0), C.c_int(0)
    flags = int(bool(force_balanced))
    rc = lib.srt_bvh_wide_host(_ptr(shape), _ptr(tris), len(tris), flags, None, 0, None, 0, C.byref(n), None, None, None)
    if rc:
        raise SrtError(f"srt_bvh_wide_host failed ({rc})")
    blocks = np.zeros((n.value, 32), np.uint32)
    dest = np.zeros(int(shape[0]["num_triangles"]), np.uint32)
    rc = lib.srt_bvh_wide_host(_ptr(shape), _ptr(tris), len(tris), flags, _ptr(blocks), len(blocks), _ptr(dest), len(dest),
                               C.byref(n), C.byref(root), C.byref(need), C.byref(bal))
    if rc:
        raise SrtError(f"srt_bvh_wide_host failed ({rc})")
    return {"blocks": blocks, "dest": dest, "root": root.value, "stack_need": need.value, "balanced": bool(bal.value)}


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("paths", "rays", "sky", "tri_tests", "tri_pass_u", "nan_pixels", "watchdog")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class SrtError(RuntimeError):
    """Raised where the reference would throw a boost::compute exception."""


_lib = None


def load_library():
    """dlopen lib/libsrt_hip.so. Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise SrtError(f"{LIB_PATH} not built: run `python __graft_entry__.py` / simple-raytracer_amd/build.py (needs hipcc)")
    _lib = _bind(C.CDLL(str(LIB_PATH)))
    return _lib


_variants = {}


def load_dev_library():
    """The -DSRT_DEV_KNOBS build (lib/variants/dev/, build.py build_dev()): the same library, which additionally reads the
    development knobs SRT_WAVES_PER_CU / SRT_SCAN_PAIRS / SRT_JOB_CAP_SUBS / SRT_POOL_BLOCKS / SRT_NO_SCAN_POOL from the
    environment. Tests that force rare paths pass it to Tracer(lib=...); the product library ignores those variables."""
    path = B.DEV_LIB
    if str(path) not in _variants:
        if not path.exists():
            raise SrtError(f"{path} not built: simple_raytracer_amd.build.build_dev()")
        lib = _bind(C.CDLL(str(path)))
        assert b"dev knobs" in lib.srt_version(), lib.srt_version()
        _variants[str(path)] = lib
    return _variants[str(path)]


def _bind(lib):
    vp, i, sz = C.c_void_p, C.c_int, C.c_size_t
    lib.srt_create.argtypes = [i, i, i, C.POINTER(vp)]
    lib.srt_destroy.argtypes = [vp]
    lib.srt_destroy.restype = None
    lib.srt_last_error.argtypes = [vp]
    lib.srt_last_error.restype = C.c_char_p
    lib.srt_set_skybox.argtypes = [vp, vp, i, i]
    lib.srt_update_scene.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp]
    lib.srt_clear_canvas.argtypes = [vp]
    lib.srt_render.argtypes = [vp, vp, C.c_uint32, vp]
    lib.srt_trace.argtypes = [vp, vp]
    if hasattr(lib, "srt_render_async"):
        lib.srt_render_async.argtypes = [vp, vp, C.c_uint32, vp]
    lib.srt_resolve.argtypes = [vp, C.c_uint32]
    if hasattr(lib, "srt_set_radiance_budget"):
        lib.srt_set_radiance_budget.argtypes = [vp, sz]
    lib.srt_synchronize.argtypes = [vp]
    lib.srt_resolve_external.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp]
    lib.srt_read_canvas.argtypes = [vp, vp]
    lib.srt_read_argb.argtypes = [vp, vp]
    lib.srt_get_counters.argtypes = [vp, C.POINTER(Counters)]
    lib.srt_set_count_triangles.argtypes = [vp, i]
    lib.srt_reset_counters.argtypes = [vp]
    lib.srt_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    if hasattr(lib, "srt_set_kernel_timers"):
        lib.srt_set_kernel_timers.argtypes = [vp, i]
    if hasattr(lib, "srt_last_trace_kernel_ms"):
        lib.srt_last_trace_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.srt_device_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz)]
    lib.srt_bind_canvas.argtypes = [vp, vp, sz]
    lib.srt_bind_stream.argtypes = [vp, vp]
    lib.srt_set_partition.argtypes = [vp, i, i, i]
    lib.srt_partition_owned_rows.argtypes = [i, i, i, i]
    lib.srt_partition_padded_rows.argtypes = [i, i, i]
    lib.srt_partition_global_row.argtypes = [i, i, i, i, i]
    lib.srt_partition_unpermute.argtypes = [vp, vp, i, i, i, sz]
    if hasattr(lib, "srt_selftest_math"):  # absent only in older A/B builds selected through SRT_LIB
        lib.srt_selftest_math.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint64)]
    lib.srt_version.restype = C.c_char_p
    if hasattr(lib, "srt_set_acceleration"):
        lib.srt_set_acceleration.argtypes = [vp, i]
        lib.srt_acceleration_info.argtypes = [vp, C.POINTER(C.c_uint64)]
        lib.srt_bvh_build_host.argtypes = [vp, vp, sz, vp, sz, vp, sz, C.POINTER(sz)]
    if hasattr(lib, "srt_gather"):
        lib.srt_comm_unique_id.argtypes = [vp]
        lib.srt_comm_init.argtypes = [vp, vp, i, i]
        lib.srt_gather.argtypes = [vp, i]
        lib.srt_resolve_gathered.argtypes = [vp, C.c_uint32]
        lib.srt_gathered_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
        lib.srt_read_gathered.argtypes = [vp, vp, vp]
        lib.srt_group_create.argtypes = [i, i, i, vp, i, C.POINTER(vp)]
        lib.srt_group_destroy.argtypes = [vp]
        lib.srt_group_destroy.restype = None
        lib.srt_group_last_error.argtypes = [vp]
        lib.srt_group_last_error.restype = C.c_char_p
        lib.srt_group_size.argtypes = [vp]
        lib.srt_group_tracer.argtypes = [vp, i]
        lib.srt_group_tracer.restype = vp
        lib.srt_group_set_skybox.argtypes = [vp, vp, i, i]
        lib.srt_group_set_acceleration.argtypes = [vp, i]
        lib.srt_group_update_scene.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp]
        lib.srt_group_clear_canvas.argtypes = [vp]
        lib.srt_group_trace_and_gather.argtypes = [vp, vp]
        lib.srt_group_render.argtypes = [vp, vp, C.c_uint32, vp]
        lib.srt_group_read_canvas.argtypes = [vp, vp]
        lib.srt_group_get_counters.argtypes = [vp, C.POINTER(Counters)]
        lib.srt_render_pipelined.argtypes = [vp, vp, C.c_uint32, vp, C.POINTER(C.c_longlong)]
        lib.srt_pipeline_flush.argtypes = [vp, vp, C.POINTER(C.c_longlong)]
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Tracer:
    """Mirror of the reference's `class Tracer`. Field and method names are the
    reference's; records are numpy scalars of records.RENDER_DATA / SCENE_DATA."""

    def __init__(self, width, height, device=0, lib=None):
        self.lib = lib if lib is not None else load_library()
        self._h = C.c_void_p()
        rc = self.lib.srt_create(width, height, device, C.byref(self._h))
        if rc:
            raise SrtError(self.lib.srt_last_error(None).decode())
        self.width, self.height = width, height
        # RenderData(width, height): num_samples = 4, num_bounces = 10 (tracer.hpp:61-66)
        self.options = R.render_data(width, height, num_samples=4, num_bounces=10)
        self.scene_data = R.scene_data(0)
        self.owned_rows = height

    # -- plumbing --
    def _check(self, rc):
        if rc:
            raise SrtError(self.lib.srt_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.srt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- the reference's interface --
    def set_skybox(self, rgba):
        rgba = np.ascontiguousarray(rgba, np.float32)
        assert rgba.ndim == 3 and rgba.shape[2] == 4
        self._check(self.lib.srt_set_skybox(self._h, _ptr(rgba), rgba.shape[1], rgba.shape[0]))

    def update_scene(self, shapes, triangles, materials):
        shapes = R.as_records(shapes, R.SHAPE)
        triangles = R.as_records(triangles, R.TRIANGLE)
        materials = R.as_records(materials, R.MATERIAL)
        sd = R.as_records(self.scene_data, R.SCENE_DATA)
        self._check(self.lib.srt_update_scene(self._h, _ptr(shapes), len(shapes), _ptr(triangles), len(triangles),
                                              _ptr(materials), len(materials), _ptr(sd)))
        self.scene_data["num_shapes"] = len(shapes)  # src/tracer.cpp:94

    def clear_canvas(self):
        self._check(self.lib.srt_clear_canvas(self._h))

    def render(self, ticks_stopped, output=None):
        """Tracer::render: trace + resolve + blocking read-back into `output`
        (uint8, owned_rows*width*4, bytes A,R,G,B)."""
        if output is None:
            output = np.zeros(self.owned_rows * self.width * 4, np.uint8)
        assert output.dtype == np.uint8 and output.size >= self.owned_rows * self.width * 4
        rd = R.as_records(self.options, R.RENDER_DATA)
        self._check(self.lib.srt_render(self._h, _ptr(rd), ticks_stopped, _ptr(output)))
        return output

    # -- extras --
    def render_async(self, ticks_stopped, output):
        """Enqueue trace + resolve + copy into `output`; valid after synchronize()."""
        assert output.dtype == np.uint8 and output.size >= self.owned_rows * self.width * 4
        rd = R.as_records(self.options, R.RENDER_DATA)
        self._check(self.lib.srt_render_async(self._h, _ptr(rd), ticks_stopped, _ptr(output)))

    def trace(self):
        rd = R.as_records(self.options, R.RENDER_DATA)
        self._check(self.lib.srt_trace(self._h, _ptr(rd)))

    def set_radiance_budget(self, nbytes):
        self._check(self.lib.srt_set_radiance_budget(self._h, nbytes))

    def resolve(self, ticks_stopped):
        self._check(self.lib.srt_resolve(self._h, ticks_stopped))

    def resolve_external(self, canvas_ptr, num_pixels, ticks_stopped, argb_ptr):
        self._check(self.lib.srt_resolve_external(self._h, C.c_void_p(canvas_ptr), num_pixels, ticks_stopped, C.c_void_p(argb_ptr)))

    def synchronize(self):
        self._check(self.lib.srt_synchronize(self._h))

    def read_canvas(self):
        out = np.zeros((self.owned_rows, self.width, 4), np.float32)
        self._check(self.lib.srt_read_canvas(self._h, _ptr(out)))
        return out

    def read_argb(self):
        out = np.zeros((self.owned_rows, self.width, 4), np.uint8)
        self._check(self.lib.srt_read_argb(self._h, _ptr(out)))
        return out

    def counters(self):
        c = Counters()
        self._check(self.lib.srt_get_counters(self._h, C.byref(c)))
        return c.as_dict()

    def debug_counters(self):
        out = (C.c_uint64 * 18)()
        self.lib.srt_debug_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        self._check(self.lib.srt_debug_counters(self._h, out))
        v = [int(x) for x in out]
        return {"rays": v[0], "sky": v[1], "paths": v[2], "orphans": v[5] & ((1 << 40) - 1), "evictions": v[5] >> 40,
                "iterations": v[6], "shade_phases": v[7], "waves_per_cu": v[8], "grid": v[9],
                "phase_cycles": dict(zip(("extend", "ring", "shade", "park", "deliver", "refill", "head", "kernel"), v[10:18])),
                # array-scan kernels without -DSRT_PHASE_CLOCK reuse the first two clock slots: big-model scans and the lanes in them
                "scans": v[10], "scan_lanes": v[11],
                # ... and the next four: the launch-end ray pool (blocks taken, records handed in, sum of squares of the blocks a wave took, blocks the last wave took)
                "pool_taken": v[12], "pool_given": v[13], "pool_taken_sq": v[14], "pool_last_taken": v[15]}

    def debug_region_counters(self):
        """[(waves, lanes)] per region of SRT_REGION_LIST for a -DSRT_REGION_COUNT build; [] for the product build."""
        out = (C.c_uint64 * 128)()
        n = C.c_int(0)
        self.lib.srt_debug_region_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_int)]
        self._check(self.lib.srt_debug_region_counters(self._h, out, 128, C.byref(n)))
        return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n.value // 2)]

    def reset_counters(self):
        self._check(self.lib.srt_reset_counters(self._h))

    def count_triangles(self, enable=True):
        self._check(self.lib.srt_set_count_triangles(self._h, 1 if enable else 0))

    def set_acceleration(self, mode):
        """ACCEL_NONE (array-order triangle scan, the parity mode) or ACCEL_BVH; applies at the next update_scene."""
        self._check(self.lib.srt_set_acceleration(self._h, int(mode)))

    def acceleration_info(self):
        out = (C.c_uint64 * 7)()
        self._check(self.lib.srt_acceleration_info(self._h, out))
        return dict(zip(("nodes", "leaves", "depth", "build_us", "models_built", "models_reused", "models_refitted"), (int(v) for v in out)))

    def selftest_math(self, stride=1):
        out = (C.c_uint64 * 16)()
        self._check(self.lib.srt_selftest_math(self._h, stride, out))
        return [int(v) for v in out]

    def set_kernel_timers(self, enable=True):
        """the render calls record the kernel timers' events too (default: only trace() does)"""
        self._check(self.lib.srt_set_kernel_timers(self._h, int(bool(enable))))

    def last_kernel_ms(self):
        a, b = C.c_float(), C.c_float()
        self._check(self.lib.srt_last_kernel_ms(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def last_trace_launches(self):
        """(srt_trace_kernel launches of the last trace = its sample batches, whether they overlapped on two streams)"""
        n, o = C.c_int(0), C.c_int(0)
        self.lib.srt_last_trace_launches.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        self._check(self.lib.srt_last_trace_launches(self._h, C.byref(n), C.byref(o)))
        return n.value, bool(o.value)

    def last_trace_kernel_ms(self):
        a = C.c_float()
        self._check(self.lib.srt_last_trace_kernel_ms(self._h, C.byref(a)))
        return a.value

    def device_buffers(self):
        cp, cb, ap, ab = C.c_void_p(), C.c_size_t(), C.c_void_p(), C.c_size_t()
        self._check(self.lib.srt_device_buffers(self._h, C.byref(cp), C.byref(cb), C.byref(ap), C.byref(ab)))
        return cp.value, cb.value, ap.value, ab.value

    def bind_canvas(self, device_ptr, nbytes):
        self._check(self.lib.srt_bind_canvas(self._h, C.c_void_p(device_ptr), nbytes))

    def bind_stream(self, hip_stream):
        self._check(self.lib.srt_bind_stream(self._h, C.c_void_p(hip_stream)))

    # -- collecting a partitioned frame over RCCL (one process per GPU) --
    @staticmethod
    def comm_unique_id():
        """128 bytes for srt_comm_init; made by ONE rank and shipped to all others by the caller."""
        buf = C.create_string_buffer(128)
        if load_library().srt_comm_unique_id(buf):
            raise SrtError(load_library().srt_last_error(None).decode())
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        self._check(self.lib.srt_comm_init(self._h, C.c_char_p(unique_id), rank, world))

    def gather(self, root=0):
        """ONE ncclGather of the packed canvases to `root` + unpermute there (asynchronous, collective)."""
        self._check(self.lib.srt_gather(self._h, root))

    def resolve_gathered(self, ticks_stopped):
        self._check(self.lib.srt_resolve_gathered(self._h, ticks_stopped))

    def gathered_buffers(self):
        cp, ap = C.c_void_p(), C.c_void_p()
        self._check(self.lib.srt_gathered_buffers(self._h, C.byref(cp), C.byref(ap)))
        return cp.value, ap.value

    def read_gathered(self, canvas=True, argb=False):
        c = np.zeros((self.height, self.width, 4), np.float32) if canvas else None
        a = np.zeros((self.height, self.width, 4), np.uint8) if argb else None
        self._check(self.lib.srt_read_gathered(self._h, _ptr(c) if canvas else None, _ptr(a) if argb else None))
        return c, a

    # -- frame pipeline --
    def render_pipelined(self, ticks_stopped, output):
        """Enqueue this frame, receive the previous one in `output`; returns its index or -1."""
        assert output.dtype == np.uint8 and output.size >= self.owned_rows * self.width * 4
        rd = R.as_records(self.options, R.RENDER_DATA)
        n = C.c_longlong(-1)
        self._check(self.lib.srt_render_pipelined(self._h, _ptr(rd), ticks_stopped, _ptr(output), C.byref(n)))
        return n.value

    def pipeline_flush(self, output):
        n = C.c_longlong(-1)
        self._check(self.lib.srt_pipeline_flush(self._h, _ptr(output), C.byref(n)))
        return n.value

    def set_partition(self, rank, world, rows_per_block=8):
        self._check(self.lib.srt_set_partition(self._h, rank, world, rows_per_block))
        self.owned_rows = self.lib.srt_partition_owned_rows(self.height, rank, world, rows_per_block)


class TracerGroup:
    """One process driving several GPUs (srt_group_*): the reference's Tracer interface over a row partition,
    collected with one ncclGather per frame."""

    def __init__(self, width, height, n_devices=1, devices=None, rows_per_block=8):
        self.lib = load_library()
        self._g = C.c_void_p()
        devs = (C.c_int * n_devices)(*devices) if devices is not None else None
        rc = self.lib.srt_group_create(width, height, n_devices, devs, rows_per_block, C.byref(self._g))
        if rc:
            raise SrtError(self.lib.srt_last_error(None).decode())
        self.width, self.height, self.n_devices = width, height, n_devices
        self.options = R.render_data(width, height, num_samples=4, num_bounces=10)
        self.scene_data = R.scene_data(0)

    def _check(self, rc):
        if rc:
            raise SrtError(self.lib.srt_group_last_error(self._g).decode())

    def close(self):
        if getattr(self, "_g", None) is not None and self._g.value:
            self.lib.srt_group_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_skybox(self, rgba):
        rgba = np.ascontiguousarray(rgba, np.float32)
        self._check(self.lib.srt_group_set_skybox(self._g, _ptr(rgba), rgba.shape[1], rgba.shape[0]))

    def set_acceleration(self, mode):
        self._check(self.lib.srt_group_set_acceleration(self._g, int(mode)))

    def update_scene(self, shapes, triangles, materials):
        shapes = R.as_records(shapes, R.SHAPE)
        triangles = R.as_records(triangles, R.TRIANGLE)
        materials = R.as_records(materials, R.MATERIAL)
        sd = R.as_records(self.scene_data, R.SCENE_DATA)
        self._check(self.lib.srt_group_update_scene(self._g, _ptr(shapes), len(shapes), _ptr(triangles), len(triangles), _ptr(materials),
                                                    len(materials), _ptr(sd)))
        self.scene_data["num_shapes"] = len(shapes)

    def clear_canvas(self):
        self._check(self.lib.srt_group_clear_canvas(self._g))

    def render(self, ticks_stopped, output=None):
        if output is None:
            output = np.zeros(self.height * self.width * 4, np.uint8)
        rd = R.as_records(self.options, R.RENDER_DATA)
        self._check(self.lib.srt_group_render(self._g, _ptr(rd), ticks_stopped, _ptr(output)))
        return output

    def read_canvas(self):
        out = np.zeros((self.height, self.width, 4), np.float32)
        self._check(self.lib.srt_group_read_canvas(self._g, _ptr(out)))
        return out

    def counters(self):
        c = Counters()
        self._check(self.lib.srt_group_get_counters(self._g, C.byref(c)))
        return c.as_dict()


# ---- pure-host partition helpers (no GPU) ------------------------------------------
def owned_rows(height, rank, world, rows_per_block):
    return load_library().srt_partition_owned_rows(height, rank, world, rows_per_block)


def padded_rows(height, world, rows_per_block):
    return load_library().srt_partition_padded_rows(height, world, rows_per_block)


def global_row(height, rank, world, rows_per_block, local_row):
    return load_library().srt_partition_global_row(height, rank, world, rows_per_block, local_row)


def unpermute(gathered, height, world, rows_per_block):
    """gathered: (world*padded_rows, ...) array, rank-major -> (height, ...) image."""
    g = np.ascontiguousarray(gathered)
    row_bytes = g.strides[0]
    out = np.zeros((height,) + g.shape[1:], g.dtype)
    rc = load_library().srt_partition_unpermute(_ptr(g), _ptr(out), height, world, rows_per_block, row_bytes)
    if rc:
        raise SrtError("srt_partition_unpermute failed")
    return out
