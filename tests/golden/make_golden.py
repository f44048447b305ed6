#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE kernel.

Needs /root/reference (read where it lies): `make -C oracle ref` compiles its
src/render.cl for x86-64 into oracle/_ref/libsrt_ref.so (see oracle/Makefile and
oracle/cl_builtins_shim.cpp). Every expected value below is an OUTPUT OF THAT BUILD;
inputs are stored next to them so the fixtures are self-contained data. The reference
itself never travels: the GPU box and later rounds use only these vectors.

Run:  python tests/golden/make_golden.py
"""
import sys
import zlib
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

import srt_pkg  # noqa: E402

srt_pkg.load()
from simple_raytracer_amd import records as R, scenes as S  # noqa: E402
from oracle import oracle_py  # noqa: E402
import cases as C  # noqa: E402


def raw(a):
    return np.frombuffer(np.ascontiguousarray(a).tobytes(), np.uint8)


def gen_cases(ref, sky):
    out = {}
    for name, case in C.build_cases().items():
        def fn(rd, canvas, case=case):
            return ref.render(rd, case["sd"], case["shapes"], case["tris"], case["mats"], sky, canvas=canvas, nthreads=8)
        canvas = C.render_case(fn, case, sky)
        # An OpenCL float3 occupies 16 bytes and the 4th lane is unspecified (the x86-64 build of render.cl leaves
        # whatever its vector registers held there): only x, y, z are results. Stored as 0, like a cleared canvas.
        canvas[..., 3] = 0.0
        argb = ref.average(len(case["frames"]), canvas)
        rd = case["rd"]
        w, h, ns = int(rd["width"]), int(rd["height"]), int(rd["num_samples"])
        rng = np.random.RandomState(zlib.crc32(name.encode()) % (2**31))
        pid = rng.randint(0, w * h, size=160).astype(np.int32)
        smp = rng.randint(0, ns, size=160).astype(np.int32)
        rad = ref.trace_paths(rd, case["sd"], case["shapes"], case["tris"], case["mats"], sky, pid, smp)
        out[name] = dict(
            shapes=raw(case["shapes"]), tris=raw(case["tris"]), mats=raw(case["mats"]), rd=raw(rd), sd=raw(case["sd"]),
            frames=np.asarray(case["frames"], np.uint32), canvas=canvas, argb=argb, path_pixel=pid, path_sample=smp,
            path_radiance=rad,
        )
        print(f"{name}: {w}x{h}x{ns} frames={case['frames']} mean={np.nanmean(canvas[..., :3]):.5f} nan={int(np.isnan(canvas).sum())}")
    return out


def gen_kats(ref, sky):
    k = {}
    rng = np.random.RandomState(2024)
    # a-1 RNG stream: 16 outputs for 32 seeds (incl. 0 and 0xffffffff)
    seeds = np.concatenate([[0, 1, 0xFFFFFFFF, 0x80000000], rng.randint(0, 2**32, size=28, dtype=np.uint64)]).astype(np.uint32)
    streams, finals = [], []
    for s in seeds:
        v, f = ref.random_floats(int(s), 16)
        streams.append(v)
        finals.append(f)
    k["rng_seed"], k["rng_out"], k["rng_final"] = seeds, np.stack(streams), np.asarray(finals, np.uint32)

    def unit(v):
        v = np.asarray(v, np.float32)
        return (v / np.float32(np.sqrt(np.float32((v * v).sum())))).astype(np.float32)

    # a-5 spheres: random + grazing + origin inside + behind
    sph = []
    for i in range(200):
        c = rng.uniform(-3, 3, 3).astype(np.float32)
        r = np.float32(rng.uniform(0.2, 2.0))
        o = rng.uniform(-4, 4, 3).astype(np.float32)
        d = unit(rng.normal(size=3))
        if i % 5 == 1:
            o = (c + rng.uniform(-0.3, 0.3, 3).astype(np.float32) * r).astype(np.float32)  # inside
        if i % 5 == 2:
            # grazing: aim at the silhouette
            to_c = c - o
            perp = unit(np.cross(to_c, rng.normal(size=3)))
            d = unit(to_c + perp * r * np.float32(0.99999 + 0.00002 * rng.rand()))
        hit, t = ref.intersect_sphere(c, r, o, d)
        sph.append(np.concatenate([c, [r], o, d, [hit, t]]))
    k["sphere"] = np.asarray(sph, np.float32)
    # a-6 planes incl. denom == 0 and un-normalised normals
    pl = []
    for i in range(120):
        p = rng.uniform(-3, 3, 3).astype(np.float32)
        n = rng.normal(size=3).astype(np.float32) * np.float32(rng.uniform(0.1, 3))
        o = rng.uniform(-4, 4, 3).astype(np.float32)
        d = unit(rng.normal(size=3))
        if i % 6 == 0:
            n = np.array([0, 1, 0], np.float32)
            d = unit([rng.normal(), 0.0, rng.normal()])  # parallel: denom == 0
        hit, t = ref.intersect_plane(p, n, o, d)
        pl.append(np.concatenate([p, n, o, d, [hit, t]]))
    k["plane"] = np.asarray(pl, np.float32)
    # a-8 triangles incl. edge / vertex hits and a == 0
    tr = []
    for i in range(300):
        p0, p1, p2 = (rng.uniform(-2, 2, 3).astype(np.float32) for _ in range(3))
        o = rng.uniform(-4, 4, 3).astype(np.float32)
        bary = rng.dirichlet([1, 1, 1]).astype(np.float32)
        if i % 7 == 1:
            bary = np.array([0.5, 0.5, 0.0], np.float32)  # on an edge
        if i % 7 == 2:
            bary = np.array([1.0, 0.0, 0.0], np.float32)  # on a vertex
        target = (p0 * bary[0] + p1 * bary[1] + p2 * bary[2]).astype(np.float32)
        d = unit(target - o)
        if i % 7 == 3:
            d = unit(p1 - p0)  # in-plane direction: a == 0 (or nearly)
        if i % 7 == 4:
            d = unit(rng.normal(size=3))
        hit, t = ref.intersect_triangle(p0, p1, p2, o, d)
        tr.append(np.concatenate([p0, p1, p2, o, d, [hit, t]]))
    k["triangle"] = np.asarray(tr, np.float32)
    # a-7 AABB incl. axis-parallel rays (inv_dir = +-inf) and origin on a slab (0*inf)
    bb = []
    for i in range(200):
        lo = rng.uniform(-2, 0, 3).astype(np.float32)
        hi = (lo + rng.uniform(0.2, 3, 3)).astype(np.float32)
        o = rng.uniform(-4, 4, 3).astype(np.float32)
        d = unit(rng.normal(size=3))
        if i % 4 == 1:
            d = np.array([0, 0, -1], np.float32)
        if i % 4 == 2:
            d = np.array([1, 0, 0], np.float32)
            o[1] = lo[1]  # exactly on a slab plane: (lo - o) * inf = 0 * inf = NaN
        with np.errstate(divide="ignore"):
            inv = (np.float32(1.0) / d).astype(np.float32)
        tmax = np.float32(np.inf if i % 3 else rng.uniform(0.5, 6))
        hit = ref.intersection_aabb(lo, hi, o, inv, tmax)
        bb.append(np.concatenate([lo, hi, o, inv, [tmax, hit]]))
    k["aabb"] = np.asarray(bb, np.float32)
    # a-4 matrix_by_vector, a-9 barycentric weights
    mv, bw = [], []
    for i in range(100):
        m = rng.normal(size=16).astype(np.float32)
        v = rng.normal(size=4).astype(np.float32)
        v[3] = i % 2
        mv.append(np.concatenate([m, v, ref.matrix_by_vector(m, v)]))
        p0, p1, p2 = (rng.uniform(-2, 2, 3).astype(np.float32) for _ in range(3))
        b = rng.dirichlet([1, 1, 1]).astype(np.float32)
        p = (p0 * b[0] + p1 * b[1] + p2 * b[2]).astype(np.float32)
        bw.append(np.concatenate([p0, p1, p2, p, ref.barycentric_weights(p0, p1, p2, p)]))
    k["matvec"], k["bary"] = np.asarray(mv, np.float32), np.asarray(bw, np.float32)
    # a-13 Schlick (fp64 inside)
    sh = []
    for i in range(200):
        mu = np.float32(rng.choice([1 / 1.5, 1.5, 1 / 1.33, 1.33, 1.0, rng.uniform(0.3, 3)]))
        c = np.float32(rng.uniform(-0.1, 1.0))
        sh.append([mu, c, ref.shlick(mu, c)])
    k["schlick"] = np.asarray(sh, np.float32)
    # a-12 sky_box on the synthetic sky: random dirs + poles + seam + sun centre
    sd = R.scene_data(0)
    dirs = [unit(rng.normal(size=3)) for _ in range(200)]
    dirs += [np.array(v, np.float32) for v in ([0, 1, 0], [0, -1, 0], [-1, 0, 0], [-1, 0, 1e-7], [-1, 0, -1e-7], [1, 0, 0], [0, 0, 1])]
    dirs.append(unit([-1, 1, 0]))  # looking into the sun (-sun_direction)
    k["sky_dir"] = np.asarray(dirs, np.float32)
    k["sky_out"] = np.asarray([ref.sky_box(sd, sky, d) for d in dirs], np.float32)
    k["sky_sd"] = raw(sd)
    # a-15 aces
    xs = np.concatenate([rng.uniform(0, 4, size=(150, 3)), [[0, 0, 0], [1e-8, 100.0, 0.5], [-0.1, -1.0, 1e6]]]).astype(np.float32)
    k["aces_in"] = xs
    k["aces_out"] = np.asarray([ref.aces(x) for x in xs], np.float32)
    return k


def main():
    oracle_py.build(ref=True)
    ref = oracle_py.Oracle("ref")
    orc = oracle_py.Oracle("oracle")
    sky = S.synthetic_sky()
    cases = gen_cases(ref, sky)
    flat = {}
    for name, d in cases.items():
        for key, v in d.items():
            flat[f"{name}/{key}"] = v
    np.savez_compressed(HERE / "cases.npz", **flat)
    kats = gen_kats(ref, sky)
    np.savez_compressed(HERE / "kats.npz", **kats)
    # checksum of the synthetic sky the vectors were made with
    np.savez_compressed(HERE / "sky_probe.npz", checksum=np.asarray([int(sky.view(np.uint32).astype(np.uint64).sum())], np.uint64),
                        probe=sky[::97, ::131].copy())
    # sanity: the restatement must reproduce every canvas bit for bit
    for name, case in C.build_cases().items():
        def fn(rd, canvas, case=case):
            return orc.render(rd, case["sd"], case["shapes"], case["tris"], case["mats"], sky, canvas=canvas, nthreads=8)
        c = C.render_case(fn, case, sky)
        g = cases[name]["canvas"]
        ok = np.array_equal(np.isnan(c), np.isnan(g)) and np.array_equal(c.view(np.uint32)[~np.isnan(c)], g.view(np.uint32)[~np.isnan(g)])
        print(f"oracle == ref on {name}: {ok}")
        assert ok, name
    for f in sorted(HERE.glob("*.npz")):
        print(f.name, f.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
