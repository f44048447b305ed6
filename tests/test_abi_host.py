"""CPU: the C-ABI library loads without a GPU, exports every symbol include/srt_abi.h
declares, fails loudly (never falls back) when asked to compute without a device, and
its pure-host partition helpers are right."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    from simple_raytracer_amd import build, tracer
    build.build_hip()
    return tracer.load_library()


def declared_symbols():
    text = (ROOT / "include/srt_abi.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(srt_[a-z_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from simple_raytracer_amd import tracer
    names = declared_symbols()
    assert len(names) >= 24
    assert sorted(tracer.ABI_SYMBOLS) == names
    for n in names:
        assert hasattr(lib, n), n


def test_header_compiles_as_c_and_cxx(tmp_path):
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "srt_abi.h"\nint main(void){return sizeof(srt_shape)==128?0:1;}\n')
    for cc, std in (("gcc", "-std=c11"), ("g++", "-std=c++17")):
        exe = tmp_path / ("t_" + cc)
        subprocess.run([cc, std, "-x", "c" if cc == "gcc" else "c++", f"-I{ROOT}/include", str(src), "-o", str(exe)], check=True)
        assert subprocess.run([str(exe)]).returncode == 0


def test_no_silent_cpu_fallback(lib):
    """Without a GPU srt_create must FAIL with a message; with one it must succeed."""
    import torch
    h = C.c_void_p()
    rc = lib.srt_create(16, 16, 0, C.byref(h))
    if torch.cuda.is_available():
        assert rc == 0
        lib.srt_destroy(h)
    else:
        assert rc != 0 and not h.value
        assert b"HIP" in lib.srt_last_error(None) or b"device" in lib.srt_last_error(None)


def test_create_rejects_bad_sizes(lib):
    h = C.c_void_p()
    assert lib.srt_create(0, 16, 0, C.byref(h)) != 0
    assert lib.srt_create(16, -1, 0, C.byref(h)) != 0


@pytest.mark.parametrize("height,world,rpb", [(1080, 8, 8), (1080, 1, 8), (29, 2, 8), (2160, 8, 16), (7, 4, 2), (64, 3, 8), (5, 8, 8)])
def test_partition_helpers(lib, height, world, rpb):
    from simple_raytracer_amd import tracer
    seen = np.zeros(height, np.int32)
    padded = tracer.padded_rows(height, world, rpb)
    total = 0
    for r in range(world):
        own = tracer.owned_rows(height, r, world, rpb)
        assert 0 <= own <= padded
        total += own
        ys = [tracer.global_row(height, r, world, rpb, lr) for lr in range(padded)]
        real = [y for y in ys if y >= 0]
        assert len(real) == own and real == sorted(real)
        assert ys[:own] == real  # padding only at the end
        for y in real:
            seen[y] += 1
    assert total == height and (seen == 1).all()
    # unpermute: rank-major packed rows -> image
    gathered = np.full((world * padded, 3), -1, np.int32)
    for r in range(world):
        for lr in range(padded):
            y = tracer.global_row(height, r, world, rpb, lr)
            if y >= 0:
                gathered[r * padded + lr] = y
    img = tracer.unpermute(gathered, height, world, rpb)
    assert (img == np.arange(height)[:, None]).all()


def test_product_library_reads_no_development_knobs():
    """The scheduling knobs (SRT_WAVES_PER_CU, SRT_SCAN_PAIRS, SRT_JOB_CAP_SUBS, SRT_POOL_BLOCKS, SRT_NO_SCAN_POOL) exist in the
    -DSRT_DEV_KNOBS build only (lib/variants/dev): their names are not even in the product library, whose one environment
    variable is SRT_RADIANCE_BUDGET_MB; the dev build says what it is in srt_version()."""
    from simple_raytracer_amd import build, tracer
    product = Path(build.build_hip()).read_bytes()
    dev = Path(build.build_dev()).read_bytes()
    knobs = [b"SRT_WAVES_PER_CU", b"SRT_SCAN_PAIRS", b"SRT_JOB_CAP_SUBS", b"SRT_POOL_BLOCKS", b"SRT_NO_SCAN_POOL"]
    assert not [k for k in knobs if k in product]
    assert all(k in dev for k in knobs)
    assert b"SRT_RADIANCE_BUDGET_MB" in product
    assert b"dev knobs" not in tracer.load_library().srt_version() and b"dev knobs" in tracer.load_dev_library().srt_version()
