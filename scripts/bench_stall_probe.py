"""What stalled one 0.2 ms step of configs[0] for 74.7 ms in BENCH_r03.json? Host timestamps around every
ABI call of a step (clear_canvas / trace / resolve / synchronize), for 30 steps of configs[0] in three situations:
  hot      right after warm launches, nothing freed
  freed    right after a handle that held the headline's 25.5 GB radiance buffer was destroyed (hipFree)
  idle     after 13 s with the GPU idle (what the CPU baseline leaves behind)
Round 4's run (profiles/r04_bench_stall_probe.jsonl): no situation reproduces a 75 ms step -- the worst is 0.38 ms (the first step after the
idle stretch, at a low clock); bench.py now measures other_configs before it frees anything or idles, warmed up by time.
usage: bench_stall_probe.py   (one JSON line per situation)"""
import json, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import srt_pkg
srt_pkg.load()
from simple_raytracer_amd import records as R, scenes as S
from simple_raytracer_amd.tracer import Tracer

sky = S.synthetic_sky()
shapes, tris, mats = S.sphere_scene()


def handle(w, h, spp):
    t = Tracer(w, h)
    t.set_skybox(sky)
    t.options = R.render_data(w, h, spp, 10, camera_to_world=S.default_camera(), time=12345)
    t.scene_data = R.scene_data(len(shapes))
    t.update_scene(shapes, tris, mats)
    return t


def steps(t, n):
    rows = []
    for _ in range(n):
        a = time.perf_counter(); t.clear_canvas()
        b = time.perf_counter(); t.trace()
        c = time.perf_counter(); t.resolve(1)
        d = time.perf_counter(); t.synchronize()
        e = time.perf_counter()
        rows.append([(b - a) * 1e3, (c - b) * 1e3, (d - c) * 1e3, (e - d) * 1e3, (e - a) * 1e3, t.last_trace_kernel_ms()])
    r = np.array(rows)
    worst = int(np.argmax(r[:, 4]))
    return {"median_step_ms": round(float(np.median(r[:, 4])), 3), "max_step_ms": round(float(r[worst, 4]), 3), "worst_step_index": worst,
            "worst_step_ms": dict(zip(("clear_canvas", "trace", "resolve", "synchronize", "step", "kernel"), [round(float(x), 3) for x in r[worst]])),
            "first_five_steps_ms": [round(float(x), 3) for x in r[:5, 4]]}


if "--torch" in sys.argv:
    # round 3's bench.py, step by step: torch's context and stream, the headline handle bound to them, one 1024-spp dispatch,
    # 13 s of CPU work, the handle destroyed, the canvas tensor dropped, torch.cuda.empty_cache() -- then a NEW handle whose
    # third step onwards is timed (warm-up by count: two steps)
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    big = handle(1920, 1080, 1024)
    big.bind_stream(stream.cuda_stream)
    canvas_t = torch.zeros((1080, 1920, 4), dtype=torch.float32, device=dev)
    big.bind_canvas(canvas_t.data_ptr(), canvas_t.numel() * 4)
    big.clear_canvas(); big.trace(); big.resolve(1); big.synchronize()
    time.sleep(13.0)
    t0 = time.perf_counter()
    big.close(); del canvas_t; torch.cuda.empty_cache()
    free_ms = (time.perf_counter() - t0) * 1e3
    small = handle(256, 256, 16)
    for _ in range(2):
        small.clear_canvas(); small.trace(); small.resolve(1)
    small.synchronize()
    print(json.dumps({"situation": f"round 3's sequence (torch context, headline handle + canvas freed in {free_ms:.1f} ms after 13 s idle, two warm-up steps)", **steps(small, 30)}))
    small.close()
    sys.exit(0)
small = handle(256, 256, 16)
small.clear_canvas(); small.trace(); small.resolve(1); small.synchronize()  # first use: allocations
print(json.dumps({"situation": "cold (first steps of the handle, two launches behind it)", **steps(small, 30)}))
print(json.dumps({"situation": "hot", **steps(small, 30)}))
big = handle(1920, 1080, 1024)
big.clear_canvas(); big.trace(); big.synchronize()
t0 = time.perf_counter(); big.close(); free_ms = (time.perf_counter() - t0) * 1e3
print(json.dumps({"situation": f"freed (25.5 GB radiance buffer destroyed {free_ms:.1f} ms ago)", **steps(small, 30)}))
time.sleep(13.0)
print(json.dumps({"situation": "idle (13 s without a launch)", **steps(small, 30)}))
small.close()
