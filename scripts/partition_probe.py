"""How well does 1/N of the frame fill one GPU? Renders each rank's row blocks of an
N-way partition on the single local GPU and compares with 1/N of the full-frame time
(GPUs are independent on this path, so this predicts strong-scaling efficiency)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import srt_pkg
srt_pkg.load()
from simple_raytracer_amd import records as R, scenes as S
from simple_raytracer_amd.tracer import Tracer

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rpb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w, h = 1920, 1080
shapes, tris, mats = S.sphere_scene()
t = Tracer(w, h)
t.set_skybox(S.synthetic_sky())
t.options = R.render_data(w, h, spp, 10, camera_to_world=S.default_camera())
t.scene_data = R.scene_data(len(shapes))
t.update_scene(shapes, tris, mats)
def timed():
    best = 1e9
    for _ in range(3):
        t.clear_canvas(); t.trace(); t.synchronize()
        best = min(best, t.last_kernel_ms()[0])
    return best
full = timed()
print(f"full frame: {full:.3f} ms")
for world in (2, 4, 8):
    ms = []
    for r in range(world):
        t.set_partition(r, world, rpb)
        ms.append(timed())
    print(f"world {world} rpb {rpb}: per-rank ms {[round(m, 2) for m in ms]} max {max(ms):.2f} ideal {full / world:.2f} efficiency {full / world / max(ms):.3f}")
t.close()
