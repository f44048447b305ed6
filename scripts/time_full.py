"""Times full-size BASELINE configs for the library selected by SRT_LIB (development aid for A/B of kernel variants).
usage: time_full.py name[:bvh][,name...] [steps]   names of bench.py's WORKLOADS"""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import srt_pkg
srt_pkg.load()
from simple_raytracer_amd import records as R, scenes as S
from simple_raytracer_amd.tracer import Tracer, ACCEL_BVH
W = {
    "spheres": (S.sphere_scene, 1920, 1080, 1024), "meshes": (lambda: S.mesh_scene(2), 1920, 1080, 512),
    "mesh100k": (lambda: S.mesh_scene(1, 224, 224, smooth=False), 1920, 1080, 256),
}
sky = S.synthetic_sky()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
out = {}
for spec in sys.argv[1].split(","):
    name, _, acc = spec.partition(":")
    builder, w, h, spp = W[name]
    shapes, tris, mats = builder()
    t = Tracer(w, h)
    t.set_skybox(sky)
    if acc == "bvh":
        t.set_acceleration(ACCEL_BVH)
    t.options = R.render_data(w, h, spp, 10, camera_to_world=S.default_camera(), time=12345)
    t.scene_data = R.scene_data(len(shapes))
    t.update_scene(shapes, tris, mats)
    ms = []
    for i in range(steps + 1):
        t.clear_canvas(); t.trace(); t.synchronize()
        ms.append(t.last_trace_kernel_ms())
    d = t.debug_counters()
    out[spec] = {"kernel_ms_min": round(min(ms[1:]), 3), "kernel_ms_all": [round(x, 2) for x in ms[1:]], "checksum": float(np.nansum(t.read_canvas().astype(np.float64))),
                 "scans": d.get("scans"), "scan_lanes": d.get("scan_lanes"), "pool": [d.get("pool_taken"), d.get("pool_given"), d.get("pool_taken_sq"), d.get("pool_last_taken")], "iterations": d.get("iterations")}
    t.close()
print(json.dumps(out))
