"""What the reference's front-end actually does: 960x540, 2 spp per frame, render() every frame with a blocking ARGB
read-back (src/main.cpp:42-46,116,283-290) -- against the two-deep frame pipeline (srt_render_pipelined: frame N's
resolve + read-back run under frame N+1's trace; the loop shows each frame one call late). Wall time per call, host
side, and frames per second of both loops; prints one JSON line per size."""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import srt_pkg
srt_pkg.load()
from simple_raytracer_amd import records as R, scenes as S
from simple_raytracer_amd.tracer import Tracer

FRAMES = 400
for (w, h, spp) in ((960, 540, 2), (960, 540, 32), (256, 256, 16)):
    shapes, tris, mats = S.sphere_scene()
    res = {"size": f"{w}x{h}x{spp}spp"}
    for mode in ("blocking", "pipelined"):
        t = Tracer(w, h)
        t.set_skybox(S.synthetic_sky())
        t.options = R.render_data(w, h, spp, 10, camera_to_world=S.default_camera())
        t.scene_data = R.scene_data(len(shapes))
        t.update_scene(shapes, tris, mats)
        out = np.zeros(w * h * 4, np.uint8)
        for rep in range(2):  # first pass warms up
            t.clear_canvas()
            t.synchronize()
            t0 = time.perf_counter()
            for frame in range(FRAMES):
                t.options["time"] = np.uint32(1000 + frame)
                if mode == "blocking":
                    t.render(frame + 1, out)
                else:
                    t.render_pipelined(frame + 1, out)
            if mode == "pipelined":
                t.pipeline_flush(out)
            dt = time.perf_counter() - t0
        res[mode] = {"ms_per_frame": round(dt / FRAMES * 1e3, 4), "fps": round(FRAMES / dt, 1), "last_frame_checksum": int(out.astype(np.uint64).sum())}
        t.close()
    res["speedup"] = round(res["pipelined"]["fps"] / res["blocking"]["fps"], 3)
    res["same_last_frame"] = res["blocking"]["last_frame_checksum"] == res["pipelined"]["last_frame_checksum"]
    print(json.dumps(res))
