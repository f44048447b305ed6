"""hipcc recipes for csrc/ -> lib/libsrt_hip.so (in-tree, so it travels with gpurun).

-ffp-contract=off is load-bearing: parity with the CPU oracle needs unfused IEEE
arithmetic (DESIGN.md "Numerics"). hipcc's defaults -fhip-fp32-correctly-rounded-
divide-sqrt and f32 denormal support stay ON (never pass -ffast-math,
-fgpu-flush-denormals-to-zero or -fno-hip-fp32-correctly-rounded-divide-sqrt here).

-fno-slp-vectorize: the SLP vectoriser pairs scalar float ops into v_pk_mul/add_f32. On
gfx950 a packed fp32 op issues at half the rate of a plain one, so nothing is gained, and
assembling the register pairs costs v_mov's: measured -11 % (spheres) to -19 % (meshes)
trace time with it off (profiles/README.md). It reorders nothing, so results are unchanged.
"""
import contextlib
import fcntl
import json
import os
import re
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
LIB = LIBDIR / "libsrt_hip.so"
SOURCES = ["kernels.hip", "srt_abi.hip", "srt_collect.hip"]
HEADERS = ["detmath.h", "device_types.h", "srt_internal.h", "issue_probe.h", "../../include/srt_abi.h", "../../include/srt_types.h"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-fno-gpu-rdc", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         "-pthread", "-Rpass-analysis=kernel-resource-usage"]  # -pthread: the host BVH build runs subtrees on std::async threads
RESOURCES = LIBDIR / "kernel_resources.json"  # registers / spills / scratch per kernel instantiation of the last build


def parse_resource_remarks(stderr):
    """{demangled-ish kernel name: {sgprs, vgprs, scratch, occupancy, sgpr_spill, vgpr_spill, lds}} from the
    -Rpass-analysis=kernel-resource-usage remarks of one hipcc run."""
    out, cur = {}, None
    keys = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes_per_lane",
            "Occupancy [waves/SIMD]": "occupancy_waves_per_simd", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
            "LDS Size [bytes/block]": "lds_bytes"}
    for line in stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            try:
                name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            except OSError:
                pass
            cur = out.setdefault(name.split("(")[0].replace("void ", ""), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+) \[-Rpass", line)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    return out


def print_resources(res=None):
    res = res if res is not None else (json.loads(RESOURCES.read_text()) if RESOURCES.exists() else {})
    for name, r in sorted(res.items()):
        print(f"  {name}: {r.get('vgprs')} VGPR, {r.get('sgprs')} SGPR, spills {r.get('sgpr_spill')} SGPR / {r.get('vgpr_spill')} VGPR, "
              f"scratch {r.get('scratch_bytes_per_lane')} B/lane, occupancy {r.get('occupancy_waves_per_simd')}")


def hipcc():
    return os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def stale():
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = [CSRC / s for s in SOURCES] + [(CSRC / h).resolve() for h in HEADERS] + [Path(__file__)]
    return any(d.stat().st_mtime > t for d in deps)


@contextlib.contextmanager
def _build_lock():
    """Serialises builds across processes (torchrun starts one rank per GPU, all of which
    call build_hip(); only one may run hipcc and write the .so)."""
    LIBDIR.mkdir(exist_ok=True)
    with open(LIBDIR / ".build.lock", "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


DEV_LIB = LIBDIR / "variants" / "dev" / "libsrt_hip.so"


def build_dev(force=False):
    """The -DSRT_DEV_KNOBS build of the same sources -> lib/variants/dev/libsrt_hip.so: reads the development knobs
    (SRT_WAVES_PER_CU, SRT_SCAN_PAIRS, SRT_JOB_CAP_SUBS, SRT_ITEMS_PER_WAVE, SRT_FORCE_BATCH, SRT_POOL_BLOCKS, SRT_NO_SCAN_POOL) from the environment, which the
    product library does not. For the tests that force rare paths and for scripts/."""
    deps = [CSRC / s for s in SOURCES] + [(CSRC / h).resolve() for h in HEADERS] + [Path(__file__)]
    if not force and DEV_LIB.exists() and all(d.stat().st_mtime <= DEV_LIB.stat().st_mtime for d in deps):
        return DEV_LIB
    with _build_lock():
        return build_variant("dev", ["-DSRT_DEV_KNOBS"])


def build_variant(name, extra_flags):
    """A/B experiments: the same sources with extra -D flags -> lib/variants/<name>/libsrt_hip.so."""
    out = LIBDIR / "variants" / name
    out.mkdir(parents=True, exist_ok=True)
    cmd = [hipcc(), *FLAGS, *extra_flags, "-shared", "-o", str(out / "libsrt_hip.so")] + [str(CSRC / s) for s in SOURCES] + ["-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
    return out / "libsrt_hip.so"


def build_hip(force=False, verbose=False, extra_flags=()):
    """Compile the HIP library if sources are newer than the .so. Returns its path."""
    if os.environ.get("SRT_LIB"):
        return Path(os.environ["SRT_LIB"])
    if not force and not stale():
        return LIB
    with _build_lock():
        if not force and not stale():  # another process built it while we waited
            return LIB
        tmp = LIBDIR / f".libsrt_hip.{os.getpid()}.so"
        cmd = [hipcc(), *FLAGS, *extra_flags, "-shared", "-o", str(tmp)] + [str(CSRC / s) for s in SOURCES] + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        res = parse_resource_remarks(r.stderr)
        RESOURCES.write_text(json.dumps(res, indent=1, sort_keys=True))
        if verbose:
            print("\n".join(l for l in r.stderr.splitlines() if "warning" in l or "error" in l))
            print_resources(res)
        os.replace(tmp, LIB)  # atomic: a concurrent dlopen sees the old or the new file, never half of one
    return LIB


HEADLESS = LIBDIR / "srt_headless"


def build_headless(force=False):
    """g++ build of tools/srt_headless.cpp (the C++ host mirror's driver) against the .so."""
    src = PKG / "tools" / "srt_headless.cpp"
    deps = [src] + sorted((PKG / "host").glob("*.hpp"))
    build_hip()
    if not force and HEADLESS.exists() and all(d.stat().st_mtime <= HEADLESS.stat().st_mtime for d in deps + [LIB]):
        return HEADLESS
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-ffp-contract=off", str(src), "-o", str(HEADLESS), f"-L{LIBDIR}", "-lsrt_hip",
           "-Wl,-rpath,$ORIGIN"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("g++ failed:\n" + r.stdout + r.stderr)
    return HEADLESS


if __name__ == "__main__":
    import sys
    print(build_hip(force="--force" in sys.argv, verbose=True))
