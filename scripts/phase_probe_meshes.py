import json, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import srt_pkg
srt_pkg.load()
from simple_raytracer_amd import records as R, scenes as S
from simple_raytracer_amd.tracer import Tracer
shapes, tris, mats = S.mesh_scene(2)
w, h, spp = 1920, 1080, 512
t = Tracer(w, h)
t.set_skybox(S.synthetic_sky())
t.options = R.render_data(w, h, spp, 10, camera_to_world=S.default_camera(), time=12345)
t.scene_data = R.scene_data(len(shapes))
t.update_scene(shapes, tris, mats)
t.clear_canvas(); t.trace(); t.synchronize()
t.reset_counters() if hasattr(t, "reset_counters") else None
t.clear_canvas(); t.trace(); t.synchronize()
d = t.debug_counters()
print(json.dumps({"ms": t.last_trace_kernel_ms(), "phase": d["phase_cycles"], "grid": d["grid"]}))
