"""Region frequencies of the trace kernel from a -DSRT_REGION_COUNT build (SRT_LIB=lib/variants/regions/libsrt_hip.so):
per region of kernels.hip's SRT_REGION_LIST how often a wave ran it and with how many lanes, next to the launch's exact
work counters. Input of scripts/isa_phase_mix.py merge. usage: region_probe.py <workload of sched_probe.py> [spp] > freq.json"""
import json, re, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import srt_pkg
srt_pkg.load()
from simple_raytracer_amd import records as R, scenes as S
from simple_raytracer_amd.tracer import Tracer

src = (ROOT / "simple-raytracer_amd/csrc/kernels.hip").read_text()
lst = src[src.index("#define SRT_REGION_LIST(X)"):src.index("enum SrtRegion")]
NAMES = re.findall(r"X\((\w+)\)", lst)
WORK = {
    "spheres": (S.sphere_scene, 1920, 1080, 64, 0),
    "mesh2": (lambda: S.mesh_scene(2), 1920, 1080, 8, 0),
    "mesh2_bvh": (lambda: S.mesh_scene(2), 1920, 1080, 8, 1),
    "mesh100k": (lambda: S.mesh_scene(1, 224, 224, smooth=False), 1920, 1080, 1, 0),
    "mesh100k_bvh": (lambda: S.mesh_scene(1, 224, 224, smooth=False), 1920, 1080, 16, 1),
}
name = sys.argv[1] if len(sys.argv) > 1 else "spheres"
builder, w, h, spp, accel = WORK[name]
if len(sys.argv) > 2:
    spp = int(sys.argv[2])
shapes, tris, mats = builder()
t = Tracer(w, h)
t.set_skybox(S.synthetic_sky())
if accel:
    t.set_acceleration(accel)
t.options = R.render_data(w, h, spp, 10, camera_to_world=S.default_camera())
t.scene_data = R.scene_data(len(shapes))
t.update_scene(shapes, tris, mats)
t.clear_canvas(); t.trace(); t.synchronize()  # warm-up
t.clear_canvas(); t.reset_counters(); t.trace(); t.synchronize()
rc = t.debug_region_counters()
if not rc:
    raise SystemExit("not a -DSRT_REGION_COUNT build (set SRT_LIB)")
d = t.debug_counters()
print(json.dumps({"workload": f"{name} {w}x{h}x{spp}spp", "kernel_ms_instrumented": round(t.last_trace_kernel_ms(), 3), "counters": t.counters(),
                  "iterations": d["iterations"], "shade_phases": d["shade_phases"], "orphans": d["orphans"],
                  "regions": {n: list(rc[i]) for i, n in enumerate(NAMES)}}))
t.close()
