import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import srt_pkg  # noqa: E402

srt_pkg.load()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _hip_device_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a HIP device stays green: gpu-marked tests are skipped
    there (the product itself never falls back: srt_create fails loudly, tests/test_abi_host.py)."""
    if _hip_device_present():
        return
    skip = pytest.mark.skip(reason="no HIP device")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def sky():
    from simple_raytracer_amd import scenes
    return scenes.synthetic_sky()


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (oracle/srt_oracle.c), compiled on demand with gcc."""
    from oracle import oracle_py
    oracle_py.build()
    return oracle_py.Oracle("oracle")


@pytest.fixture(scope="session")
def ref():
    """The reference's render.cl compiled for x86-64; only where it has been built."""
    from oracle import oracle_py
    if not oracle_py.ref_available():
        if Path("/root/reference/src/render.cl").exists():
            oracle_py.build(ref=True)
        else:
            pytest.skip("oracle/_ref not built and /root/reference absent")
    return oracle_py.Oracle("ref")


def bits_equal(a, b):
    """Bit-exact comparison that treats any NaN as equal to any NaN (payload/sign of a
    NaN differs between x86 SSE and gfx950)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return False
    return np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb])
