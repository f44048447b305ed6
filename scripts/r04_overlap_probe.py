"""Why did configs[4]'s array scan take 5.6 s inside bench.py and 4.3 s alone (round 4)? measure_config of that workload
(a) in a fresh process, (b) behind other handles that came and went, as in bench.py's other_configs loop."""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
from simple_raytracer_amd import scenes as S
sky = S.synthetic_sky()
mode = sys.argv[1] if len(sys.argv) > 1 else "alone"
if mode == "behind":
    for name, accel, k, wu in (("spheres_256_16spp", "none", 20, 2), ("meshes_1080p_512spp", "none", 1, 1), ("meshes_1080p_512spp", "bvh", 1, 1), ("mesh100k_1080p_256spp", "bvh", 1, 1)):
        bench.measure_config(name, sky, accel, k, wu, 0)
o = bench.measure_config("mesh100k_1080p_256spp", sky, "none", 1, 0, 0)
print(json.dumps({"mode": mode, "ms_per_step": o["ms_per_step"], "kernel_ms": o["kernel_ms"], "launches": o["launches_per_step"]}))
