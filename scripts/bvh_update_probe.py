"""Host cost of srt_update_scene under SRT_ACCEL_BVH for the 10^5-triangle mesh: a fresh build, an unchanged scene (hierarchy kept),
a moved model (refit). Wall time of the call and the library's own figure (srt_acceleration_info: microseconds on the hierarchies).
With an argument N > 1: the same through srt_group_update_scene on N virtual devices (one GPU), which prepares the scene once
and uploads it N times -- to be within 1.5x of the single handle. usage: bvh_update_probe.py [N]"""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import srt_pkg
srt_pkg.load()
from simple_raytracer_amd import records as R, scenes as S
from simple_raytracer_amd.tracer import Tracer, TracerGroup

shapes, tris, mats = S.mesh_scene(1, 224, 224, smooth=False)
n_dev = int(sys.argv[1]) if len(sys.argv) > 1 else 1
t = Tracer(64, 64) if n_dev == 1 else TracerGroup(64, 64, n_dev)
t.set_skybox(S.synthetic_sky())
t.set_acceleration(1)
t.scene_data = R.scene_data(len(shapes))
out = {"devices": n_dev}
for name in ("build", "unchanged", "moved", "moved_again"):
    if name.startswith("moved"):
        shapes = shapes.copy()
        m = shapes[1]["transform"].copy()
        m[3][0] += np.float32(0.25)
        shapes[1]["transform"] = m
    t0 = time.perf_counter()
    t.update_scene(shapes, tris, mats)
    dt = time.perf_counter() - t0
    if n_dev == 1:
        info = t.acceleration_info()
        out[name] = {"call_ms": round(dt * 1e3, 2), "hierarchy_ms": round(info["build_us"] / 1e3, 2), "built": info["models_built"], "reused": info["models_reused"], "refitted": info["models_refitted"]}
    else:
        out[name] = {"call_ms": round(dt * 1e3, 2)}
print(json.dumps(out))
t.close()
