"""Loads tests/golden/*.npz (vectors produced from the reference kernel by
tests/golden/make_golden.py) back into ABI record arrays."""
from pathlib import Path

import numpy as np

from simple_raytracer_amd import records as R

GOLDEN = Path(__file__).resolve().parent / "golden"


def _rec(raw, dtype, scalar=False):
    a = np.frombuffer(bytearray(raw.tobytes()), dtype)  # exact bytes incl. padding, writable
    return a[0] if scalar else a


def load_cases():
    z = np.load(GOLDEN / "cases.npz")
    names = sorted({k.split("/")[0] for k in z.files if "/" in k})  # (the file also carries `detmath_revision`)
    out = {}
    for n in names:
        g = lambda key: z[f"{n}/{key}"]
        out[n] = dict(
            shapes=_rec(g("shapes"), R.SHAPE), tris=_rec(g("tris"), R.TRIANGLE), mats=_rec(g("mats"), R.MATERIAL),
            rd=_rec(g("rd"), R.RENDER_DATA, True), sd=_rec(g("sd"), R.SCENE_DATA, True),
            frames=[int(v) for v in g("frames")], canvas=g("canvas"), argb=g("argb"),
            path_pixel=g("path_pixel"), path_sample=g("path_sample"), path_radiance=g("path_radiance"),
        )
    return out


def load_kats():
    z = np.load(GOLDEN / "kats.npz")
    return {k: z[k] for k in z.files if k != "detmath_revision"}


def load_sky_probe():
    z = np.load(GOLDEN / "sky_probe.npz")
    return int(z["checksum"][0]), z["probe"]


def golden_revisions():
    """detmath revision stamped into each fixture file by make_golden.py"""
    return {n: int(np.load(GOLDEN / f"{n}.npz")["detmath_revision"][0]) for n in ("cases", "kats")}
