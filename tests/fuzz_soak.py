"""Long differential fuzz run (GPU): python tests/fuzz_soak.py N_benign N_hostile [seed].
Same generator and checks as tests/test_gpu_fuzz.py, with progress lines so that a long
run is visibly alive. Prints every failing iteration and a final summary; exit code 1 on
any mismatch. SRT_FUZZ_ACCEL=1 runs the GPU side through the BVH (triangle counters are then
not compared: the walk tests fewer triangles by design)."""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import srt_pkg  # noqa: E402

srt_pkg.load()
from simple_raytracer_amd import build, records as R, scenes as S, tracer as T  # noqa: E402
from oracle import oracle_py  # noqa: E402
from test_gpu_fuzz import random_scene  # noqa: E402


def bits_equal(a, b):
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb])


def main():
    n_benign, n_hostile = int(sys.argv[1]), int(sys.argv[2])
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    verbose = len(sys.argv) > 4
    build.build_hip()
    oracle_py.build()
    orc = oracle_py.Oracle("oracle")
    sky = S.synthetic_sky()
    w, h = 24, 16
    t = T.Tracer(w, h)
    t.set_skybox(sky)
    t.count_triangles(True)
    accel = int(os.environ.get("SRT_FUZZ_ACCEL", "0"))
    t.set_acceleration(accel)
    keys = ("paths", "rays", "sky", "nan_pixels") if accel else ("paths", "rays", "sky", "tri_tests", "tri_pass_u", "nan_pixels")
    fails = 0
    t0 = time.time()
    for hostile, n in ((False, n_benign), (True, n_hostile)):
        rng = np.random.RandomState(seed + int(hostile) if seed >= 10000 else seed * 1000 + int(hostile))  # >= 10000: raw seed, as the pytest case
        for it in range(n):
            shapes, tris, mats, cam = random_scene(rng, hostile)
            rd = R.render_data(w, h, int(rng.randint(1, 5)), int(rng.choice([1, 2, 5, 10])), fov_scale=float(rng.uniform(0.3, 2.0)),
                               camera_to_world=cam, time=int(rng.randint(1, 2**31)), show_normals=bool(rng.rand() < 0.1))
            sd = R.scene_data(len(shapes), sun_focus=float(rng.choice([25.0, 1.0, 32.0, 7.5, 0.0, 100.0])), sun_intensity=float(rng.uniform(0, 3)))
            t.options, t.scene_data = rd, sd
            t.update_scene(shapes, tris, mats)
            t.clear_canvas()
            t.reset_counters()
            t.trace()
            got = t.read_canvas()
            c = t.counters()
            with np.errstate(all="ignore"):
                want, oc = orc.render(rd, sd, shapes, tris, mats, sky, counters=True, nthreads=8)
            ok = bits_equal(got, want) and all(c[k] == oc[k] for k in keys) and c["watchdog"] == 0
            if not ok:
                fails += 1
                print(f"MISMATCH hostile={hostile} it={it}: counters gpu {c} oracle {oc}", flush=True)
            if it % 100 == 99 or verbose:
                print(f"hostile={hostile} {it + 1}/{n} scenes, {fails} mismatches, {time.time() - t0:.0f} s", flush=True)
    t.close()
    print(f"DONE {n_benign} benign + {n_hostile} hostile scenes: {fails} mismatches", flush=True)
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
