"""CPU: the parts of bench.py that do not need a GPU -- the W_ops / W_bytes bookkeeping of
SURVEY.md 8(d) and the cpu_baseline leg (which is allowed to use oracle/)."""
import importlib.util
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("srt_bench", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_algorithmic_work_formula(bench):
    from simple_raytracer_amd import scenes as S
    shapes, _, _ = S.sphere_scene()  # 4 spheres + 3 planes
    c = {"rays": 1000, "paths": 200, "sky": 150, "tri_tests": 0, "tri_pass_u": 0}
    assert bench.w_ops(c, shapes) == 1000 * 4 * 17 + 1000 * 3 * 14 + 1000 * 3 + 800 * 250 + 150 * 40 + 200 * 30
    shapes, _, _ = S.mesh_scene(2)  # 1 plane + 2 models
    c = {"rays": 10, "paths": 4, "sky": 3, "tri_tests": 500, "tri_pass_u": 20}
    assert bench.w_ops(c, shapes) == 10 * 14 + 10 * 2 * 24 + 10 * 3 + 500 * 30 + 20 * 22 + 6 * 250 + 3 * 40 + 4 * 30
    assert bench.w_bytes({"paths": 100, "sky": 2}, 50, 1000, 10**9) == 24 * 100 + 32 * 50 + 1000 + 128


def test_cpu_baseline_leg(bench, sky):
    from oracle import oracle_py
    r = bench.cpu_baseline("spheres_256_16spp", sky, target_seconds=1.5)
    assert r["unit"] == "Mray/s" and r["value"] > 0 and r["cores"] >= 1 and r["mpath_per_s"] > 0
    if oracle_py.ref_available():
        assert r["kind"] == "reference" and r["reference_equals_port"] is True and "bit-identical" in r["sample"]
    else:
        assert r["kind"] == "port"
