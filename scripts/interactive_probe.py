"""What the reference's front-end actually does: 960x540, 2 spp per frame, render() every
frame with a blocking ARGB read-back (src/main.cpp:42-46,116,283-290). Wall time per
srt_render call, host side."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import srt_pkg
srt_pkg.load()
from simple_raytracer_amd import records as R, scenes as S
from simple_raytracer_amd.tracer import Tracer

for (w, h, spp) in ((960, 540, 2), (960, 540, 32), (256, 256, 16)):
    shapes, tris, mats = S.sphere_scene()
    t = Tracer(w, h)
    t.set_skybox(S.synthetic_sky())
    t.options = R.render_data(w, h, spp, 10, camera_to_world=S.default_camera())
    t.scene_data = R.scene_data(len(shapes))
    out = np.zeros(w * h * 4, np.uint8)
    walls = []
    for frame in range(40):
        t0 = time.perf_counter()
        if frame % 10 == 0:
            t.clear_canvas()
            t.update_scene(shapes, tris, mats)
        t.options["time"] = np.uint32(1000 + frame)
        t.render(frame % 10 + 1, out)
        walls.append(time.perf_counter() - t0)
    ms, _ = t.last_kernel_ms()
    walls = np.array(walls[5:]) * 1e3
    print(f"{w}x{h}x{spp}spp: srt_render wall median {np.median(walls):.3f} ms (min {walls.min():.3f}), device trace+reduce {ms:.3f} ms, {1e3/np.median(walls):.0f} FPS")
    t.close()
