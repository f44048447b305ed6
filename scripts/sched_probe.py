"""Scheduling diagnostics of the trace kernel (srt_debug_counters) next to its time, for one or more
workloads of scripts/time_one.py's table: iterations of the waves' main loop, SHADE phases, lanes per
phase, paths that outlived their staging buffer. usage: sched_probe.py [spheres64,...] [spp]"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import srt_pkg
srt_pkg.load()
from simple_raytracer_amd import records as R, scenes as S
from simple_raytracer_amd.tracer import Tracer

sky = S.synthetic_sky()
WORK = {
    "spheres": (S.sphere_scene, 1920, 1080, 64, 0),
    "spheres_noglass": (lambda: (lambda sh, tr, ma: (sh, tr, np.concatenate([ma[:4], ma[3:4], ma[5:]])))(*S.sphere_scene()), 1920, 1080, 64, 0),  # what the glass branch costs
    "mesh2": (lambda: S.mesh_scene(2), 1920, 1080, 8, 0),
    "mesh2_bvh": (lambda: S.mesh_scene(2), 1920, 1080, 8, 1),
    "mesh100k_bvh": (lambda: S.mesh_scene(1, 224, 224, smooth=False), 1920, 1080, 16, 1),
    "interactive": (S.sphere_scene, 960, 540, 2, 0),
}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["spheres"]
for name in names:
    builder, w, h, spp, accel = WORK[name]
    if len(sys.argv) > 2:
        spp = int(sys.argv[2])
    shapes, tris, mats = builder()
    t = Tracer(w, h)
    t.set_skybox(sky)
    if accel:
        t.set_acceleration(accel)
    t.options = R.render_data(w, h, spp, 10, camera_to_world=S.default_camera())
    t.scene_data = R.scene_data(len(shapes))
    t.update_scene(shapes, tris, mats)
    ms = []
    for i in range(4):
        t.clear_canvas(); t.reset_counters(); t.trace(); t.synchronize()
        ms.append(t.last_trace_kernel_ms())
    d = t.debug_counters() if hasattr(t.lib, "srt_debug_counters") else {}
    c = t.counters()
    out = {"workload": name, "spp": spp, "kernel_ms": round(min(ms[1:]), 3), "mray_s": round(c["rays"] / min(ms[1:]) / 1e3, 1), **d}
    if d:
        out["rays_per_iteration"] = round(d["rays"] / max(d["iterations"], 1), 2)
        out["orphan_fraction"] = round(d["orphans"] / max(d["paths"], 1), 4)
    print(json.dumps(out))
    t.close()
