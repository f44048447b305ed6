#!/usr/bin/env python3
"""bench.py — headline metric of BASELINE.json: Mray/s of the path-tracing hot path.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
  N > 1 either way: `python bench.py --gpus N ...` starts its own N ranks (children of
  `python -m torch.distributed.run`, before this process has touched a GPU), and
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
  --master-port P bench.py --gpus N ...` (the driver's form) is used as it comes.
  The documented 8-GPU line is BASELINE configs[3]: --gpus 8 --workload spheres_4k_4096spp.

Step   = one dispatch of the hot path over the whole frame of the workload: clear the
         accumulation canvas, trace (the `render` kernel), [N > 1: ONE gather of the
         packed per-rank canvases to rank 0 over RCCL + unpermute], resolve (the
         `average` kernel) on rank 0. Inputs and outputs stay in HBM.
Default workload = BASELINE.json configs[1]: the 7-shape sphere scene, 1920x1080,
         1024 spp in one dispatch, 10 bounces, one MI355X. With N GPUs the SAME frame
         is split in interleaved 8-row blocks ("strong" scaling).
value  = rays (closest-hit queries, counted exactly by the kernel) of all ranks per
         second of max-over-ranks wall time, in Mray/s.
roofline: the trace kernel is VALU-issue bound, not HBM bound:
         achieved = algorithmic lane-ops (W_ops formula of SURVEY.md §8d from the
         kernel's deterministic counters) / HIP-event kernel time; the HBM side
         (W_bytes / time against 8 TB/s) is reported next to it.
cpu_baseline: the reference's own render.cl compiled for x86-64 (oracle/_ref, kind
         "reference") when that prebuilt library is present, else the CPU oracle (kind "port"),
         on a bounded row sample of the same frame, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import srt_pkg  # noqa: E402

srt_pkg.load()
from simple_raytracer_amd import build as B, multi, records as R, scenes as S, tracer as T  # noqa: E402

VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9  # 78.6e12 lane-ops/s: 256 CU x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md)
HBM_PEAK = 8.0e12                            # B/s spec (MI355X_MICROARCH.md §HBM)

WORKLOADS = {
    # name: (scene builder, width, height, spp, bounces, description)
    "spheres_1080p_1024spp": (S.sphere_scene, 1920, 1080, 1024, 10, "BASELINE configs[1]: 7-shape sphere scene 1920x1080 1024spp"),
    "spheres_256_16spp": (S.sphere_scene, 256, 256, 16, 10, "BASELINE configs[0]: 7-shape sphere scene 256x256 16spp"),
    "meshes_1080p_512spp": (lambda: S.mesh_scene(2), 1920, 1080, 512, 10, "BASELINE configs[2]: plane + 2 x 968-triangle meshes 1920x1080 512spp"),
    "spheres_4k_4096spp": (S.sphere_scene, 3840, 2160, 4096, 10, "BASELINE configs[3]: sphere scene 3840x2160 4096spp"),
    "mesh100k_1080p_256spp": (lambda: S.mesh_scene(1, 224, 224, smooth=False), 1920, 1080, 256, 10, "BASELINE configs[4]: plane + 99,904-triangle mesh 1920x1080 256spp"),
}


def w_ops(c, shapes):
    """Algorithmic lane-ops of SURVEY.md §8(d) from deterministic counters."""
    n_sph = int((shapes["type"] == R.SHAPE_SPHERE).sum())
    n_pl = int((shapes["type"] == R.SHAPE_PLANE).sum())
    n_mod = int((shapes["type"] == R.SHAPE_MODEL).sum())
    rays, paths, sky = c["rays"], c["paths"], c["sky"]
    bounces = rays - paths  # every path has exactly one non-bouncing final segment
    return (rays * n_sph * 17 + rays * n_pl * 14 + rays * n_mod * 24 + rays * 3 + c["tri_tests"] * 30
            + c["tri_pass_u"] * 22 + bounces * 250 + sky * 40 + paths * 30)


def w_bytes(c, pixels, scene_bytes, sky_bytes):
    """HBM bytes the trace step has to move by design: 12 B of radiance written per path by
    the trace kernel and read once by the ordered reduction, canvas RMW 32 B/pixel, scene,
    touched sky texels (the resolve's 20 B/pixel belong to the resolve kernel)."""
    return 24 * c["paths"] + 32 * pixels + scene_bytes + min(sky_bytes, 64 * c["sky"])


def w_bytes_survey(c, pixels, scene_bytes, sky_bytes):
    """SURVEY.md §8(d) W_bytes: what the ALGORITHM has to move (canvas RMW 32 B + resolve 20 B per pixel, scene, touched
    sky texels) -- without the per-path radiance spill this design adds to keep the sample sum ordered."""
    return 52 * pixels + scene_bytes + min(sky_bytes, 64 * c["sky"])


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def same_bits(a, b):
    """bit compare of float arrays, NaN == NaN"""
    na, nb = np.isnan(a), np.isnan(b)
    return bool(np.array_equal(na, nb) and np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb]))


def self_launch(argv, n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children of
    torch.distributed.run and pass their exit code on. Runs before this process has made any HIP
    call (never re-exec or fork a process that has initialised the GPU)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + argv
    # a wedged collective must end the run, not hang it: the children get a deadline and are killed as a group past it
    limit = float(os.environ.get("SRT_BENCH_TIMEOUT_S", "1500"))
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(proc.pid, signal.SIGKILL)
        proc.wait()
        print(f"bench.py: the {n}-rank run did not finish within {limit:.0f} s and was killed", file=sys.stderr)
        return 124


def cpu_baseline(name, sky, target_seconds=15.0, spp_override=0):
    """CPU rate on the host cores over a sample of the SAME frame (same scene, seeds, spp).
    When oracle/_ref/libsrt_ref.so is present (the reference's own render.cl compiled for
    x86-64, built where /root/reference exists and shipped prebuilt) it is what gets timed
    (kind "reference") on a few bands of rows spread over the frame; the port runs the same
    bands to count their rays, and the two canvases must agree bit for bit. Otherwise the
    port alone is timed on rows y0, y0+stride, ... (kind "port")."""
    from oracle import oracle_py  # cpu_baseline leg only
    oracle_py.build()
    orc = oracle_py.Oracle("oracle")
    builder, w, h, spp, nb, _ = WORKLOADS[name]
    spp = spp_override or spp
    shapes, tris, mats = builder()
    rd = R.render_data(w, h, spp, nb, camera_to_world=S.default_camera(), time=12345)
    sd = R.scene_data(len(shapes))
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(cores, 64))
    # calibrate on a thin sample, then size the real one for ~target_seconds
    probe_rd = rd.copy()
    probe_rd["num_samples"] = max(1, min(spp, 16))
    t0 = time.time()
    _, c0 = orc.render(probe_rd, sd, shapes, tris, mats, sky, rows=(4, h), row_stride=max(1, h // 16), nthreads=threads, counters=True)
    dt0 = max(time.time() - t0, 1e-4)
    paths_per_s = c0["paths"] / dt0
    ref = None
    if oracle_py.ref_available():
        try:
            ref = oracle_py.Oracle("ref")
        except OSError:
            ref = None
    if ref is not None:
        # bands of whole rows (the reference build parallelises over rows of one contiguous range)
        rows_wanted = int(max(8, min(h, 0.5 * target_seconds * paths_per_s / (w * spp))))
        n_bands = 8 if h >= 64 else 1
        band = max(1, rows_wanted // n_bands)
        starts = [int((i + 0.5) * h / n_bands - band / 2) for i in range(n_bands)]
        bands = [(max(0, y), min(h, max(0, y) + band)) for y in starts]
        rays = paths = 0
        dt_port = dt_ref = 0.0
        same = True
        port_rows = {}
        for y0, y1 in bands:
            t0 = time.time()
            cp, c = orc.render(rd, sd, shapes, tris, mats, sky, rows=(y0, y1), nthreads=threads, counters=True)
            dt_port += time.time() - t0
            t0 = time.time()
            cr = ref.render(rd, sd, shapes, tris, mats, sky, rows=(y0, y1), nthreads=threads)
            dt_ref += time.time() - t0
            rays += c["rays"]
            paths += c["paths"]
            port_rows[(y0, y1)] = cp[y0:y1, :, :3].copy()
            a, b = cp[y0:y1, :, :3], cr[y0:y1, :, :3]  # the float3's padding lane is unspecified (the reference build carries NaNs into it)
            same = same and bool(np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a.view(np.uint32)[~np.isnan(a)], b.view(np.uint32)[~np.isnan(b)]))
        n_rows = sum(y1 - y0 for y0, y1 in bands)
        return {
            "value": round(rays / dt_ref / 1e6, 3), "unit": "Mray/s", "cores": threads, "threads": threads, "host_cores": cores, "cpu_model": cpu_model(), "kind": "reference", "_port_rows": port_rows,
            "cores_is": "the threads that ran (OpenMP over pixels, capped at 64); host_cores = the cores this process may use",
            "sample": f"oracle/_ref/libsrt_ref.so (the reference's src/render.cl compiled for x86-64 with ROCm clang; its 18 OpenCL built-ins -- cos, log, pow, atan2pi, normalize, dot, cross, mix, read_imagef ... -- are THIS repo's csrc/detmath.h through oracle/cl_builtins_shim.cpp, not an OpenCL runtime's; OpenMP over pixels), {len(bands)} bands of {band} rows spread over the frame ({n_rows} of {h} rows x {w} px x {spp} spp = {paths} paths, {rays} rays counted by the port on the same rows) in {dt_ref:.1f} s; the port took {dt_port:.1f} s on the same rows and its canvas is {'bit-identical' if same else 'DIFFERENT'}",
            "mpath_per_s": round(paths / dt_ref / 1e6, 3), "port_value": round(rays / dt_port / 1e6, 3), "reference_equals_port": same,
        }
    rows_wanted = int(max(1, min(h, target_seconds * paths_per_s / (w * spp))))
    stride = max(1, h // rows_wanted)
    y0 = stride // 2
    t0 = time.time()
    cp, c = orc.render(rd, sd, shapes, tris, mats, sky, rows=(y0, h), row_stride=stride, nthreads=threads, counters=True)
    dt = time.time() - t0
    n_rows = len(range(y0, h, stride))
    port_rows = {(y, y + 1): cp[y:y + 1, :, :3].copy() for y in range(y0, h, stride)}
    return {
        "value": round(c["rays"] / dt / 1e6, 3), "unit": "Mray/s", "cores": threads, "threads": threads, "host_cores": cores, "cpu_model": cpu_model(), "kind": "port", "_port_rows": port_rows,
        "cores_is": "the threads that ran (OpenMP over pixels, capped at 64); host_cores = the cores this process may use",
        "sample": f"oracle/srt_oracle.c (bit-identical port of render.cl), rows {y0}::{stride} ({n_rows} of {h} rows x {w} px x {spp} spp = {c['paths']} paths, {c['rays']} rays) in {dt:.1f} s",
        "mpath_per_s": round(c["paths"] / dt / 1e6, 3),
    }


def kernels_hash():
    """sha256 of the kernel source a measurement belongs to (profiles/traffic.json carries the one it was taken with)"""
    import hashlib
    return hashlib.sha256((ROOT / "simple-raytracer_amd" / "csrc" / "kernels.hip").read_bytes()).hexdigest()


def measure_config(name, sky, accel, steps, warmup, device, keep_canvas=False):
    """One BASELINE config on one GPU through a handle of its own, outside the headline's timed region: ms per step (wall,
    outputs resident), the trace kernel's own time (HIP events), Mray/s, and the VALU roofline fraction where W_ops is defined
    (array-order scan: triangle counters from the instrumented kernel variant, one extra untimed dispatch)."""
    builder, w, h, spp, nb, desc = WORKLOADS[name]
    shapes, tris, mats = builder()
    t = T.Tracer(w, h, device=device)
    t.set_skybox(sky)
    t.options = R.render_data(w, h, spp, nb, camera_to_world=S.default_camera(), time=12345)
    t.scene_data = R.scene_data(len(shapes))
    if accel == "bvh":
        t.set_acceleration(T.ACCEL_BVH)
    t.update_scene(shapes, tris, mats)
    # Warm up by TIME as well as by count: at least `warmup` steps and at least 100 ms of launches. A 0.2 ms step measured two
    # launches after the handle's first use still has first-use costs in it (round 3's driver run: one step of configs[0] took
    # 74.7 ms with 0.15 ms inside the kernel's events; scripts/bench_stall_probe.py).
    t_w, n_w = time.perf_counter(), 0
    while n_w < warmup or (n_w < 2000 and time.perf_counter() - t_w < 0.1):
        t.clear_canvas(); t.trace(); t.resolve(1)
        t.synchronize()
        n_w += 1
    t.reset_counters()
    kms, walls, calls = [], [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        t.clear_canvas()
        t1 = time.perf_counter()
        t.trace()
        t2 = time.perf_counter()
        t.resolve(1)
        t3 = time.perf_counter()
        t.synchronize()
        t4 = time.perf_counter()
        walls.append(t4 - t0)
        calls.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
        kms.append(t.last_trace_kernel_ms())
    dt = float(np.median(walls))
    worst = int(np.argmax(walls))
    c = t.counters()
    launches, overlapped = t.last_trace_launches()
    per = {"rays": c["rays"] // steps, "paths": c["paths"] // steps, "sky": c["sky"] // steps, "tri_tests": 0, "tri_pass_u": 0}
    has_models = bool((shapes["type"] == R.SHAPE_MODEL).any())
    frac = None
    if not has_models or accel != "bvh":
        if has_models:
            t.count_triangles(True); t.reset_counters(); t.clear_canvas(); t.trace(); t.synchronize()
            ci = t.counters()
            per["tri_tests"], per["tri_pass_u"] = ci["tri_tests"], ci["tri_pass_u"]
            t.count_triangles(False)
        frac = round(w_ops(per, shapes) / (float(np.median(kms)) * 1e-3) / VALU_PEAK_LANE_OPS, 4)
    out = {"workload": desc + (" [BVH, srt_set_acceleration]" if accel == "bvh" else ""), "steps": steps, "ms_per_step": round(dt * 1e3, 3), "ms_per_step_is": "median of the steps' wall times", "ms_per_step_max": round(max(walls) * 1e3, 3),
           "kernel_ms": round(float(np.median(kms)), 3), "kernel_ms_max": round(float(max(kms)), 3), "kernel_ms_is": ("span of overlapping sample-batch launches" if overlapped else "sum of the launches' own durations") + " (median step)",
           "launches_per_step": launches, "mray_s": round(per["rays"] / dt / 1e6, 1), "rays_per_step": per["rays"], "frac": frac,
           "warmup_steps": n_w, "watchdog": c["watchdog"],
           "slowest_step_host_ms": dict(zip(("clear_canvas", "trace", "resolve", "synchronize"), [round(x * 1e3, 3) for x in calls[worst]]))}
    if frac is not None:
        out["frac_is"] = "ALGORITHMIC W_ops (SURVEY 8d: every ray x shape test at its full cost) / kernel time / VALU peak -- not executed VALU: the kernels skip work the formula charges (division-free triangle rejects, pre-transformed triangles), so it can approach or pass 1 without the pipe being full"
    if accel == "bvh":
        out["frac_note"] = "W_ops (SURVEY 8d) has no term for hierarchy steps: no roofline fraction for the BVH walk"
    if keep_canvas:
        out["_canvas"] = t.read_canvas()
    t.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="spheres_1080p_1024spp", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel (development only; makes the line non-comparable)")
    ap.add_argument("--rows-per-block", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other_configs section (the other BASELINE configs on one GPU, measured after the headline)")
    ap.add_argument("--accel", default="none", choices=["none", "bvh"],
                    help="bvh = opt-in acceleration structure for model shapes (srt_set_acceleration); the headline line uses none")
    ap.add_argument("--gather", default="lib", choices=["lib", "torch"],
                    help="N > 1: lib = the library's own ncclGather + unpermute kernel (srt_gather, C ABI); torch = torch.distributed.gather + index_select")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = REHEARSAL ONLY on a box with fewer GPUs than ranks: ranks share cuda:0 and the gather is staged through host memory")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            raise SystemExit(self_launch(sys.argv[1:], args.gpus))  # nothing above has touched a GPU
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # stdout carries the ONE JSON line and nothing else: RCCL / gloo print banners to fd 1 when a communicator comes up
    # ("RCCL version : ...", "[Gloo] Rank 0 is connected to ..."), so fd 1 points at stderr until the line is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    B.build_hip()
    rehearsal = args.backend == "gloo"
    if rehearsal:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SRT_BENCH_COLLECT=1 under a one-rank torchrun walks the N > 1 path (process group, RCCL id broadcast, srt_gather,
    # gathered == single check) with world = 1: the rehearsal a one-GPU box allows for the RCCL leg
    collect = world > 1 or (bool(os.environ.get("SRT_BENCH_COLLECT")) and "WORLD_SIZE" in os.environ)
    if collect:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    builder, w, h, spp, nb, desc = WORKLOADS[args.workload]
    if args.spp:
        spp = args.spp
    shapes, tris, mats = builder()
    sky = S.synthetic_sky()

    t = T.Tracer(w, h, device=local_rank)
    # One non-default torch stream carries BOTH the library's launches and the RCCL
    # gather, so trace -> gather -> resolve are ordered without host syncs. (torch's
    # default stream is the NULL stream, which a non-blocking stream does not order with.)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    t.bind_stream(stream.cuda_stream)
    t.set_skybox(sky)
    t.options = R.render_data(w, h, spp, nb, camera_to_world=S.default_camera(), time=12345)
    t.scene_data = R.scene_data(len(shapes))
    if args.accel == "bvh":
        t.set_acceleration(T.ACCEL_BVH)
    t.update_scene(shapes, tris, mats)
    part = multi.RowPartition(h, rank, world, args.rows_per_block)
    canvas_t = torch.zeros((part.padded, w, 4), dtype=torch.float32, device=dev)
    t.set_partition(rank, world, args.rows_per_block)
    t.bind_canvas(canvas_t.data_ptr(), canvas_t.numel() * 4)
    argb_t = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev) if rank == 0 else None
    unperm = torch.as_tensor(multi.unpermute_index(h, world, args.rows_per_block, part.padded), device=dev) if (rank == 0 and collect) else None
    gather_bufs = [torch.empty_like(canvas_t) for _ in range(world)] if (rank == 0 and collect) else None

    # the library's own collective (srt_comm_init / srt_gather): rank 0 makes the RCCL id, torch.distributed ships it
    gather_mode = "none" if not collect else ("host-staged gloo (rehearsal)" if rehearsal else "torch.distributed.gather")
    if collect and not rehearsal and args.gather == "lib":
        # Every rank takes part in every collective below whatever fails locally: rank 0 ALWAYS broadcasts (the id, or None when it
        # could not make one), every rank then reports whether it can go on, and the communicator (itself a collective) is only
        # initialised when all of them can. A rank that fails alone must not leave the others waiting in a different collective.
        ident, why = None, ""
        if rank == 0:
            try:
                if os.environ.get("SRT_BENCH_FAIL_COMM_ID"):  # test hook: exercise the fallback leg
                    raise RuntimeError("SRT_BENCH_FAIL_COMM_ID is set")
                ident = T.Tracer.comm_unique_id()
            except Exception as e:
                why = f"srt_comm_unique_id failed: {e}"
        ids = [ident]
        dist.broadcast_object_list(ids, src=0)
        can = ids[0] is not None and hasattr(t.lib, "srt_comm_init")
        dist.all_gather_object(allc := [None] * world, bool(can))
        if all(allc):
            ok_here = True
            try:
                t.comm_init(ids[0], rank, world)
            except Exception as e:
                ok_here, why = False, f"srt_comm_init failed: {e}"
            dist.all_gather_object(allf := [None] * world, ok_here)
            gather_mode = "srt_gather (ncclGather inside libsrt_hip.so)" if all(allf) else f"torch.distributed.gather ({why or 'srt_comm_init failed on another rank'})"
        else:
            gather_mode = f"torch.distributed.gather ({why or 'no RCCL id from rank 0'})"
    use_lib_gather = gather_mode.startswith("srt_gather")

    trace_ms, resolve_ms, kernel_only_ms, gather_ms = [], [], [], []
    g_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if collect else None

    def step(record):
        t.clear_canvas()
        t.trace()
        if collect:
            g_ev[0].record(stream)
            if rehearsal:
                host = canvas_t.cpu()
                bufs = [torch.empty_like(host) for _ in range(world)] if rank == 0 else None
                dist.gather(host, bufs, dst=0)
                if rank == 0:
                    for dst_buf, src in zip(gather_bufs, bufs):
                        dst_buf.copy_(src)
            elif use_lib_gather:
                t.gather(0)  # the ONE collective of the path: ncclGather over xGMI + unpermute kernel on rank 0, inside the library
                if rank == 0:
                    t.resolve_gathered(1)
            else:
                dist.gather(canvas_t, gather_bufs, dst=0)  # the same collective through torch
            if rank == 0 and not use_lib_gather:
                full = torch.cat(gather_bufs, dim=0).index_select(0, unperm)
                t.resolve_external(full.data_ptr(), w * h, 1, argb_t.data_ptr())
                step.full = full
            g_ev[1].record(stream)  # gather + unpermute + (rank 0) resolve of the gathered frame
        elif rank == 0:
            t.resolve_external(canvas_t.data_ptr(), w * h, 1, argb_t.data_ptr())
        if record:
            a, b = t.last_kernel_ms()  # HIP events on the launch stream (synchronises it)
            trace_ms.append(a)
            resolve_ms.append(b)
            kernel_only_ms.append(t.last_trace_kernel_ms())
            step.launches = t.last_trace_launches()
            if collect:
                g_ev[1].synchronize()
                gather_ms.append(g_ev[0].elapsed_time(g_ev[1]))

    def fence():
        torch.cuda.synchronize(dev)
        if collect:
            dist.barrier()
        torch.cuda.synchronize(dev)

    step.full = None
    step.launches = (1, False)

    for _ in range(args.warmup):
        step(False)
    fence()
    t.reset_counters()
    t.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0

    c = t.counters()
    red_dev = torch.device("cpu") if rehearsal else dev
    k_mean = float(np.mean(kernel_only_ms)) if kernel_only_ms else 0.0
    g_mean = float(np.mean(gather_ms)) if gather_ms else 0.0
    stats = torch.tensor([elapsed, k_mean, -k_mean, g_mean], dtype=torch.float64, device=red_dev)
    cnt = torch.tensor([c["rays"], c["paths"], c["sky"], c["nan_pixels"]], dtype=torch.int64, device=red_dev)
    if collect:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    elapsed_max, trace_ms_max, trace_ms_min, gather_ms_max = float(stats[0]), float(stats[1]), -float(stats[2]), float(stats[3])
    rays, paths, nsky, nan_px = (int(v) for v in cnt.tolist())

    check = None
    if rank == 0 and collect and os.environ.get("SRT_BENCH_CHECK", "1") != "0":
        # outside the timed region: the gathered, unpermuted canvas must equal a single-handle render of the whole frame bit for bit
        ref_t = T.Tracer(w, h, device=local_rank)
        ref_t.set_skybox(sky)
        ref_t.options, ref_t.scene_data = t.options.copy(), t.scene_data.copy()
        ref_t.update_scene(shapes, tris, mats)  # array-order scan even when the ranks use the BVH
        ref_t.clear_canvas()
        ref_t.trace()
        want = ref_t.read_canvas()
        got = t.read_gathered(canvas=True)[0] if use_lib_gather else step.full.cpu().numpy()
        check = same_bits(want, got)
        ref_t.close()
    if rank == 0:
        steps = max(args.steps, 1)
        per = {"rays": rays // steps, "paths": paths // steps, "sky": nsky // steps, "tri_tests": 0, "tri_pass_u": 0}
        if (shapes["type"] == R.SHAPE_MODEL).any() and world == 1:
            # triangle counters come from the instrumented kernel variant, outside the timed region
            t.count_triangles(True)
            t.reset_counters()
            t.clear_canvas()
            t.trace()
            ci = t.counters()
            per["tri_tests"], per["tri_pass_u"] = ci["tri_tests"], ci["tri_pass_u"]
            t.count_triangles(False)
        ops = w_ops(per, shapes)
        scene_bytes = shapes.nbytes + tris.nbytes + mats.nbytes
        nbytes = w_bytes(per, w * h, scene_bytes, sky.nbytes)
        nbytes_alg = w_bytes_survey(per, w * h, scene_bytes, sky.nbytes)
        kt = trace_ms_max * 1e-3 if trace_ms_max > 0 else elapsed_max / steps
        achieved = ops / kt / 1e12 / world  # per-GPU rate: each GPU ran 1/world of the ops in kt
        # HBM bytes per launch from the PMC passes of scripts/pmc_traffic.sh -- only when that file was measured on THIS kernel source
        # (it carries the sha256 of csrc/kernels.hip); a stale measurement reads as null
        traffic = None
        tf = ROOT / "profiles" / "traffic.json"
        if tf.exists() and world == 1:  # measured for the whole frame on one GPU
            try:
                tj = json.loads(tf.read_text())
                if tj.get("kernels_hip_sha256") == kernels_hash():
                    traffic = tj.get(args.workload if not args.spp else "", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "Mray/s at 1920x1080x1024spp" if args.workload == "spheres_1080p_1024spp" and not args.spp else f"Mray/s ({args.workload})",
            "value": round(rays / elapsed_max / 1e6, 2),
            "unit": "Mray/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": desc + (f" [spp overridden to {spp}]" if args.spp else ""), "width": w, "height": h, "spp": spp,
                       "bounces": nb, "shapes": int(len(shapes)), "triangles": int(len(tris)), "partition": f"{world} x interleaved {args.rows_per_block}-row blocks" if world > 1 else "single GPU", "gather": gather_mode,
                       "mode": "parity (fp-contract off, IEEE div/sqrt; canvas bit-identical to the CPU oracle)",
                       **({"accel": {"kind": "bvh", **t.acceleration_info(), "note": "triangle counters are the BVH walk's leaf tests; node box tests are not in W_ops"}} if args.accel == "bvh" else {})},
            "mpath_per_s": round(paths / elapsed_max / 1e6, 2),
            "rays_per_step": per["rays"], "paths_per_step": per["paths"], "nan_pixels": nan_px,
            "kernel_ms": {("trace_launch_span" if step.launches[1] else "srt_trace_kernel"): round(trace_ms_max, 3),
                          "trace_plus_ordered_reduce": round(float(np.mean(trace_ms)) if trace_ms else 0.0, 3),
                          "resolve": round(float(np.mean(resolve_ms)) if resolve_ms else 0.0, 4)},
            "roofline": {
                "bound": "valu", "achieved": round(achieved, 3), "peak": round(VALU_PEAK_LANE_OPS / 1e12, 1), "unit": "Tlane-op/s",
                "frac": round(achieved * 1e12 / VALU_PEAK_LANE_OPS, 4), "traffic": traffic,
                "kernel": "srt_trace_kernel", "algorithmic_ops_per_launch": int(ops // world // max(step.launches[0], 1)),
                "launches_per_step": step.launches[0],
                **({"launch_time": "the step's launches overlap on two streams (srt_trace): `achieved` = ops of the step / span from the first launch's start to the last one's end, "
                                   "not ops per launch / a launch's own duration (a launch shares the GPU with its neighbour for part of its life)"} if step.launches[1] else {}),
                "note": "VALU-issue roofline (SURVEY.md §8d/H8): 256 CU x 4 SIMD x 32 lanes x 2.4 GHz; ops = W_ops formula over exact kernel counters",
                "hbm": {"achieved": round(nbytes / world / kt / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                        "frac": round(nbytes / world / kt / HBM_PEAK, 6), "design_bytes_per_launch": int(nbytes // world // max(step.launches[0], 1)),
                        "algorithmic_bytes_per_launch": int(nbytes_alg // world // max(step.launches[0], 1)),
                        "note": "design bytes = algorithmic bytes (SURVEY.md 8d W_bytes) + 24 B per path of radiance written by the trace kernel and read by the ordered reduction"},
            },
        }
        if collect:  # (also the one-rank rehearsal of the N > 1 path, SRT_BENCH_COLLECT=1)
            line["per_rank"] = {"trace_kernel_ms_min": round(trace_ms_min, 3), "trace_kernel_ms_max": round(trace_ms_max, 3),
                                "gather_ms": round(gather_ms_max, 3),
                                "gather_ms_is": "max over ranks of: collective + unpermute + (rank 0) resolve of the gathered frame, on the launch stream, behind the rank's trace"}
        ok = True
        if check is not None:
            line["gathered_equals_single_gpu"] = check
            ok = ok and check
        if world == 1 and args.workload == "spheres_1080p_1024spp" and not args.spp and not args.no_other_configs:
            # The other BASELINE configs on this GPU (handles of their own; headline fields unchanged): right behind the headline's
            # timed region, while the device is busy and BEFORE anything is freed or the CPU baseline leaves it idle for 15 s --
            # round 3 measured them after both and one 0.2 ms step took 74.7 ms.
            others, canv = [], {}
            for name, accel, k, wu in (("spheres_256_16spp", "none", 20, 2), ("meshes_1080p_512spp", "none", 3, 1), ("meshes_1080p_512spp", "bvh", 3, 1),
                                       ("mesh100k_1080p_256spp", "bvh", 3, 1), ("mesh100k_1080p_256spp", "none", 1, 0), ("spheres_4k_4096spp", "none", 1, 1)):  # (configs[3]: one warm-up step, which also allocates its 2 x 48 GB of radiance)
                try:
                    o = measure_config(name, sky, accel, k, wu, local_rank, keep_canvas=name.startswith("mesh"))
                    if "_canvas" in o:
                        canv[(name, accel)] = o.pop("_canvas")
                    others.append(o)
                except Exception as e:  # the headline line is still written
                    others.append({"workload": name, "accel": accel, "error": str(e)})
            # The full-size array scans run ~5,000 waves through the launch-end ray pool (kernels.hip REFILL_POOL), more than any
            # test does: their canvases must equal the BVH's of the same frame bit for bit (same triangle test, same tie rule),
            # and no wave may have hit a watchdog.
            for o in others:
                for name in ("meshes_1080p_512spp", "mesh100k_1080p_256spp"):
                    if o.get("workload", "").startswith(WORKLOADS[name][5]) and "BVH" not in o["workload"] and (name, "none") in canv and (name, "bvh") in canv:
                        o["canvas_equals_bvh_canvas"] = same_bits(canv[(name, "none")][..., :3], canv[(name, "bvh")][..., :3])
                        ok = ok and o["canvas_equals_bvh_canvas"]
                if o.get("watchdog"):
                    ok = False
            del canv
            line["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.workload, sky, spp_override=args.spp)
            # the frame that was TIMED, checked where the CPU has already rendered it: the canvas of the last timed step
            # (one dispatch at the workload's full spp) against the port's rows, bit for bit
            port_rows = cb.pop("_port_rows")
            gpu = canvas_t.cpu().numpy()
            cb["gpu_equals_port"] = all(same_bits(gpu[y0:y1, :, :3], rows) for (y0, y1), rows in port_rows.items())
            cb["gpu_rows_checked"] = int(sum(y1 - y0 for y0, y1 in port_rows))
            line["cpu_baseline"] = cb
            ok = ok and cb["gpu_equals_port"] and cb.get("reference_equals_port", True)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
        if not ok:
            if t is not None:
                t.close()
            raise SystemExit("bench.py: a canvas differs from its checker, or a wave hit a watchdog (see gpu_equals_port / gathered_equals_single_gpu / other_configs[*].canvas_equals_bvh_canvas, watchdog in the line above)")
    if t is not None:
        t.close()
    if collect:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
