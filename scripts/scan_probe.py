"""Array-order triangle scan (no BVH) on BASELINE configs[2] / configs[4] geometry: time, triangle tests per second, VALU
roofline fraction from the W_ops formula, and the canvas checksum. usage: scan_probe.py [mesh2|mesh100k] [spp] [w h]"""
import json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import srt_pkg
srt_pkg.load()
from simple_raytracer_amd import records as R, scenes as S
from simple_raytracer_amd.tracer import Tracer

name = sys.argv[1] if len(sys.argv) > 1 else "mesh100k"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
shapes, tris, mats = S.mesh_scene(2) if name == "mesh2" else S.mesh_scene(1, 224, 224, smooth=False)
t = Tracer(w, h)
t.set_skybox(S.synthetic_sky())
t.options = R.render_data(w, h, spp, 10, camera_to_world=S.default_camera())
t.scene_data = R.scene_data(len(shapes))
t.update_scene(shapes, tris, mats)
ms = []
for i in range(3):
    t.clear_canvas(); t.reset_counters(); t.trace(); t.synchronize()
    ms.append(t.last_trace_kernel_ms())
c = t.counters()
d = t.debug_counters()
t.count_triangles(True); t.clear_canvas(); t.reset_counters(); t.trace(); t.synchronize()
ci = t.counters()
n_pl = int((shapes["type"] == 1).sum()); n_mod = int((shapes["type"] == 2).sum())
ops = c["rays"] * (n_pl * 14 + n_mod * 24 + 3) + ci["tri_tests"] * 30 + ci["tri_pass_u"] * 22 + (c["rays"] - c["paths"]) * 250 + c["sky"] * 40 + c["paths"] * 30
best = min(ms[1:])
print(json.dumps({"workload": name, "size": f"{w}x{h}x{spp}", "kernel_ms": round(best, 3), "tri_tests": ci["tri_tests"], "gtri_per_s": round(ci["tri_tests"] / best / 1e6, 1),
                  "valu_frac": round(ops / (best * 1e-3) / 78.6e12, 4), "mray_s": round(c["rays"] / best / 1e3, 1), "waves_per_cu": d["waves_per_cu"], "scans": d["scans"], "lanes_per_scan": round(d["scan_lanes"] / max(d["scans"], 1), 1),
                  "checksum": float(np.nansum(t.read_canvas().astype(np.float64)))}))
t.close()
