"""Per-phase wave cycles of a full-size BASELINE config for a -DSRT_PHASE_CLOCK build (development aid).
usage: SRT_LIB=<variant .so> phase_probe.py name[:bvh]   names as scripts/time_full.py"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import srt_pkg
srt_pkg.load()
from simple_raytracer_amd import records as R, scenes as S
from simple_raytracer_amd.tracer import Tracer, ACCEL_BVH
W = {
    "spheres": (S.sphere_scene, 1920, 1080, 1024), "meshes": (lambda: S.mesh_scene(2), 1920, 1080, 512),
    "mesh100k": (lambda: S.mesh_scene(1, 224, 224, smooth=False), 1920, 1080, 256),
}
name, _, acc = sys.argv[1].partition(":")
builder, w, h, spp = W[name]
shapes, tris, mats = builder()
t = Tracer(w, h)
t.set_skybox(S.synthetic_sky())
if acc == "bvh":
    t.set_acceleration(ACCEL_BVH)
t.options = R.render_data(w, h, spp, 10, camera_to_world=S.default_camera(), time=12345)
t.scene_data = R.scene_data(len(shapes))
t.update_scene(shapes, tris, mats)
t.clear_canvas(); t.trace(); t.synchronize()
d0 = t.debug_counters()
t.clear_canvas(); t.trace(); t.synchronize()
d = t.debug_counters()
ph = {k: d["phase_cycles"][k] - d0["phase_cycles"][k] for k in d["phase_cycles"]}
tot = ph["kernel"] or 1
print(json.dumps({"config": sys.argv[1], "ms": round(t.last_trace_kernel_ms(), 2), "rays": d["rays"] - d0["rays"], "paths": d["paths"] - d0["paths"],
                  "share": {k: round(v / tot, 4) for k, v in ph.items()}, "grid": d["grid"]}))
