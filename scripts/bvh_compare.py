"""BVH versus the array-order triangle scan at sizes too slow for the unit tests: number of
canvas pixels whose bits differ, ray counts, kernel times. usage: bvh_compare.py [name ...]"""
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
    "mesh2_1080p_16": (lambda: S.mesh_scene(2), 1920, 1080, 16),
    "mesh100k_1080p_1": (lambda: S.mesh_scene(1, 224, 224, smooth=False), 1920, 1080, 1),
    "mesh100k_smooth_1080p_1": (lambda: S.mesh_scene(1, 224, 224, smooth=True), 1920, 1080, 1),
    "mesh100k_960_4": (lambda: S.mesh_scene(1, 224, 224, smooth=False), 960, 540, 4),
}
for name in (sys.argv[1:] or list(WORK)):
    builder, w, h, spp = WORK[name]
    shapes, tris, mats = builder()
    res = []
    for accel in (0, 1):
        t = Tracer(w, h)
        t.set_skybox(sky)
        t.set_acceleration(accel)
        t.options = R.render_data(w, h, spp, 10, camera_to_world=S.default_camera(), time=12345)
        t.scene_data = R.scene_data(len(shapes))
        t.update_scene(shapes, tris, mats)
        t.clear_canvas(); t.reset_counters(); t.trace(); t.synchronize()
        res.append((t.read_canvas(), t.counters(), t.last_trace_kernel_ms()))
        t.close()
    (c0, k0, ms0), (c1, k1, ms1) = res
    same = (c0.view(np.uint32) == c1.view(np.uint32)) | (np.isnan(c0) & np.isnan(c1))
    print(json.dumps({"work": name, "pixels": w * h, "paths": k0["paths"], "differing_pixels": int((~same).any(axis=-1).sum()),
                      "rays": [k0["rays"], k1["rays"]], "sky": [k0["sky"], k1["sky"]], "scan_ms": round(ms0, 2), "bvh_ms": round(ms1, 2)}), flush=True)
