"""ctypes loaders for the CPU checkers under oracle/.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py. The product package never imports this module.

`Oracle("oracle")` -> oracle/libsrt_oracle.so  (our C restatement, prefix orc_)
`Oracle("ref")`    -> oracle/_ref/libsrt_ref.so (the reference's render.cl compiled for
                      x86-64, prefix ref_; exists only where it was built)
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
COUNTER_NAMES = ["paths", "rays", "sphere_tests", "plane_tests", "aabb_tests", "tri_tests", "tri_pass_u", "bounces", "sky", "nan_pixels"]


def build(ref=False):
    """Compile the checker(s); `ref` additionally needs /root/reference."""
    targets = ["all"] + (["ref"] if ref else [])
    subprocess.run(["make", "-C", str(HERE)] + targets, check=True, capture_output=True)


def ref_available():
    return (HERE / "_ref" / "libsrt_ref.so").exists()


def ref_libm_available():
    return (HERE / "_ref" / "libsrt_ref_libm.so").exists()


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _rec():
    import sys
    sys.path.insert(0, str(HERE.parent))
    import srt_pkg
    srt_pkg.load()
    from simple_raytracer_amd import records
    return records


def _scene_arrays(shapes, tris, mats, rd, sd):
    R = _rec()
    return (R.as_records(shapes, R.SHAPE), R.as_records(tris, R.TRIANGLE), R.as_records(mats, R.MATERIAL),
            R.as_records(rd, R.RENDER_DATA), R.as_records(sd, R.SCENE_DATA))


class Oracle:
    def __init__(self, kind="oracle"):
        self.kind = kind
        if kind == "oracle":
            path, self.px = HERE / "libsrt_oracle.so", "orc_"
            if not path.exists():
                build()
        elif kind == "ref":
            path, self.px = HERE / "_ref" / "libsrt_ref.so", "ref_"
        elif kind == "ref_libm":  # the reference object with glibc behind cos / log / pow / atan2 (statistics only)
            path, self.px = HERE / "_ref" / "libsrt_ref_libm.so", "ref_"
        else:
            raise ValueError(kind)
        self.lib = C.CDLL(str(path))
        f = self._f
        f("render").restype = None
        f("average").restype = None
        f("trace_paths").restype = None
        f("random_float").restype = C.c_float
        f("shlick_reflectance").restype = C.c_float
        f("shlick_reflectance").argtypes = [C.c_float, C.c_float]
        f("intersection_aabb").argtypes = [C.c_void_p] * 4 + [C.c_float]
        for n in ("intersect_sphere", "intersect_plane", "intersect_triangle", "intersection_aabb"):
            f(n).restype = C.c_int

    def _f(self, name):
        return getattr(self.lib, self.px + name)

    # ---- kernels ---------------------------------------------------------------
    def render(self, rd, sd, shapes, tris, mats, sky, canvas=None, rows=None, nthreads=0, counters=False, row_stride=1):
        """`render` kernel over rows [y0,y1) (default all): canvas += colour. Returns the
        canvas (h, w, 4) float32, and the counter dict when counters=True (oracle only)."""
        shapes, tris, mats, rd, sd = _scene_arrays(shapes, tris, mats, rd, sd)
        w, h = int(rd["width"]), int(rd["height"])
        if canvas is None:
            canvas = np.zeros((h, w, 4), np.float32)
        y0, y1 = (0, h) if rows is None else rows
        sky = np.ascontiguousarray(sky, np.float32)
        args = [_p(rd), _p(sd), _p(canvas), _p(shapes), _p(tris), _p(mats), _p(sky),
                C.c_int(sky.shape[1]), C.c_int(sky.shape[0]), C.c_int(y0), C.c_int(y1)]
        ctr = None
        if self.kind == "oracle":
            ctr = np.zeros(len(COUNTER_NAMES), np.uint64)
            args += [C.c_int(row_stride), C.c_int(nthreads), _p(ctr)]
            self.lib.orc_render_strided.restype = None
            self.lib.orc_render_strided(*args)
        else:
            assert row_stride == 1
            args.append(C.c_int(nthreads))
            self._f("render")(*args)
        if counters:
            return canvas, dict(zip(COUNTER_NAMES, (int(v) for v in ctr))) if ctr is not None else None
        return canvas

    def average(self, num_steps, canvas):
        canvas = np.ascontiguousarray(canvas, np.float32)
        n = canvas.size // 4
        out = np.zeros(n * 4, np.uint8)
        self._f("average")(C.c_uint32(num_steps), _p(canvas), _p(out), C.c_size_t(n))
        return out.reshape(canvas.shape[:-1] + (4,))

    def trace_paths(self, rd, sd, shapes, tris, mats, sky, pixel_ids, samples):
        pixel_ids = np.ascontiguousarray(pixel_ids, np.int32)
        samples = np.ascontiguousarray(samples, np.int32)
        out = np.zeros((len(pixel_ids), 3), np.float32)
        shapes, tris, mats, rd, sd = _scene_arrays(shapes, tris, mats, rd, sd)
        sky = np.ascontiguousarray(sky, np.float32)
        self._f("trace_paths")(_p(rd), _p(sd), _p(shapes), _p(tris), _p(mats), _p(sky), C.c_int(sky.shape[1]),
                               C.c_int(sky.shape[0]), _p(pixel_ids), _p(samples), C.c_int(len(pixel_ids)), _p(out))
        return out

    # ---- function-level known-answer entry points ------------------------------
    def random_floats(self, seed, n):
        s = C.c_uint32(seed)
        out = np.zeros(n, np.float32)
        for i in range(n):
            out[i] = self._f("random_float")(C.byref(s))
        return out, s.value

    def shlick(self, mu, c):
        return np.float32(self._f("shlick_reflectance")(C.c_float(mu), C.c_float(c)))

    def intersect_sphere(self, center, radius, o, d):
        from_records = np.zeros(8, np.float32)
        from_records[:3] = center
        from_records[4] = radius
        o, d = np.asarray(o, np.float32), np.asarray(d, np.float32)
        t = C.c_float(np.nan)
        hit = self._f("intersect_sphere")(_p(from_records), _p(o), _p(d), C.byref(t))
        return hit, np.float32(t.value)

    def intersect_plane(self, pos, normal, o, d):
        rec = np.zeros(8, np.float32)
        rec[:3] = pos
        rec[4:7] = normal
        o, d = np.asarray(o, np.float32), np.asarray(d, np.float32)
        t = C.c_float(np.nan)
        hit = self._f("intersect_plane")(_p(rec), _p(o), _p(d), C.byref(t))
        return hit, np.float32(t.value)

    def intersect_triangle(self, p0, p1, p2, o, d):
        a = [np.ascontiguousarray(v, np.float32) for v in (p0, p1, p2, o, d)]
        t = C.c_float(np.nan)
        hit = self._f("intersect_triangle")(*[_p(v) for v in a], C.byref(t))
        return hit, np.float32(t.value)

    def intersection_aabb(self, bmin, bmax, o, inv_dir, tmax):
        a = [np.ascontiguousarray(v, np.float32) for v in (bmin, bmax, o, inv_dir)]
        return self._f("intersection_aabb")(*[_p(v) for v in a], C.c_float(tmax))

    def matrix_by_vector(self, m, v):
        m = np.ascontiguousarray(m, np.float32)
        v = np.ascontiguousarray(v, np.float32)
        out = np.zeros(4, np.float32)
        self._f("matrix_by_vector")(_p(m), _p(v), _p(out))
        return out

    def barycentric_weights(self, p0, p1, p2, p):
        a = [np.ascontiguousarray(v, np.float32) for v in (p0, p1, p2, p)]
        out = np.zeros(3, np.float32)
        self._f("barycentric_weights")(*[_p(v) for v in a], _p(out))
        return out

    def sky_box(self, sd, sky, direction):
        sky = np.ascontiguousarray(sky, np.float32)
        sd = _rec().as_records(sd, _rec().SCENE_DATA)
        d = np.ascontiguousarray(direction, np.float32)
        out = np.zeros(3, np.float32)
        self._f("sky_box")(_p(sd), _p(sky), C.c_int(sky.shape[1]), C.c_int(sky.shape[0]), _p(d), _p(out))
        return out

    def math_checksums(self, stride):
        out = (C.c_uint64 * 8)()
        self.lib.orc_math_checksums.restype = None
        self.lib.orc_math_checksums(C.c_uint32(stride), out)
        return [int(v) for v in out]

    def aces(self, rgb):
        a = np.ascontiguousarray(rgb, np.float32)
        out = np.zeros(3, np.float32)
        self._f("aces")(_p(a), _p(out))
        return out
