"""Times the trace kernel of the currently selected library (SRT_LIB) on a few
workloads; prints one JSON line. Development aid for A/B runs."""
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
    "spheres64": (S.sphere_scene, 1920, 1080, 64),
    "mesh2_8": (lambda: S.mesh_scene(2), 1920, 1080, 8),
    "mesh100k": (lambda: S.mesh_scene(1, 224, 224, smooth=False), 480, 270, 1),
    # through the BVH (srt_set_acceleration)
    "mesh2_8_bvh": (lambda: S.mesh_scene(2), 1920, 1080, 8, 1),
    "mesh100k_bvh": (lambda: S.mesh_scene(1, 224, 224, smooth=False), 480, 270, 1, 1),
    "mesh100k_bvh_16": (lambda: S.mesh_scene(1, 224, 224, smooth=False), 1920, 1080, 16, 1),
}
names = sys.argv[1].split(",") if len(sys.argv) > 1 else list(WORK)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
out = {}
for name in names:
    builder, w, h, spp, *accel = WORK[name]
    shapes, tris, mats = builder()
    t = Tracer(w, h)
    t.set_skybox(sky)
    if accel:
        t.set_acceleration(accel[0])
    t.options = R.render_data(w, h, spp, 10, camera_to_world=S.default_camera())
    t.scene_data = R.scene_data(len(shapes))
    t.update_scene(shapes, tris, mats)
    ms = []
    for i in range(reps + 1):
        t.clear_canvas(); t.reset_counters(); t.trace(); t.synchronize()
        ms.append(t.last_kernel_ms()[0])
        kms = t.last_trace_kernel_ms()
    c = t.counters()
    if (shapes["type"] == 2).any():
        t.count_triangles(True); t.clear_canvas(); t.reset_counters(); t.trace(); t.synchronize()
        c2 = t.counters(); c["tri_tests"], c["tri_pass_u"] = c2["tri_tests"], c2["tri_pass_u"]
    info = t.acceleration_info() if accel else {}
    out[name] = {**info, "trace_kernel_only_ms": round(kms, 3), "tri_tests": c["tri_tests"], "tri_pass_u": c["tri_pass_u"], "gtri_per_s": round(c["tri_tests"] / max(np.min(ms[1:]), 1e-9) / 1e6, 2),"ms": round(float(np.min(ms[1:])), 3), "med": round(float(np.median(ms[1:])), 3), "mray_s": round(c["rays"] / np.min(ms[1:]) / 1e3, 1),
                 "checksum": float(np.nansum(t.read_canvas().astype(np.float64)))}
    t.close()
print(json.dumps(out))
