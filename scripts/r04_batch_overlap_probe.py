"""Whole-step time of BASELINE configs[1] (1920x1080x1024 spp) with the samples in 1 / 2 / 4 batches (dev build, SRT_FORCE_BATCH): does the
ordered reduction of one batch hide under the trace kernel of the next? usage: r04_batch_overlap_probe.py   (GPU box)"""
import json, os, sys, time, subprocess
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1:
    sys.path.insert(0, str(ROOT))
    import numpy as np
    import srt_pkg
    srt_pkg.load()
    from simple_raytracer_amd import records as R, scenes as S
    from simple_raytracer_amd.tracer import Tracer, load_dev_library
    shapes, tris, mats = S.sphere_scene()
    t = Tracer(1920, 1080, lib=load_dev_library())
    t.set_skybox(S.synthetic_sky())
    t.options = R.render_data(1920, 1080, 1024, 10, camera_to_world=S.default_camera(), time=12345)
    t.scene_data = R.scene_data(len(shapes))
    t.update_scene(shapes, tris, mats)
    ms = []
    for i in range(6):
        t.clear_canvas(); t.synchronize()
        t0 = time.perf_counter(); t.trace(); t.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"force_batch": os.environ.get("SRT_FORCE_BATCH"), "step_ms_min": round(min(ms[1:]), 3), "step_ms": [round(x, 2) for x in ms[1:]], "launches": t.last_trace_launches(),
                      "checksum": float(np.nansum(t.read_canvas().astype(np.float64)))}))
else:
    for rnd in range(2):
        for b in ("", "512", "256", "128"):
            env = dict(os.environ)
            if b:
                env["SRT_FORCE_BATCH"] = b
            subprocess.run([sys.executable, __file__, "child"], env=env, check=True)
