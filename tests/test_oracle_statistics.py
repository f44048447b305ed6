"""CPU: are detmath's built-ins (cos / log / pow / atan2pi as FMA polynomials, dot / cross / mix fused, normalize
through a Newton rsqrt) UNBIASED, not just inside the OpenCL ULP bounds? A path tracer is chaotic, so renders made with
different realisations of the built-ins cannot be compared bit for bit -- but they must agree as statistics.

Both sides are the reference's own render.cl compiled for x86-64 (oracle/Makefile `ref`): once with detmath behind the
built-ins (oracle/_ref/libsrt_ref.so, what the goldens come from), once with glibc's libm and textbook unfused vector
helpers (libsrt_ref_libm.so). For every golden scene, at a raised sample count:
  * the image means agree within Monte-Carlo error,
  * the RMS difference between the two images is no larger than the RMS difference between two renders of ONE
    implementation with different seeds (the noise floor): there is no systematic per-pixel offset either.
Needs the prebuilt oracle/_ref (built where /root/reference exists; travels with gpurun)."""
import numpy as np
import pytest

import cases as C
from oracle import oracle_py

pytestmark = pytest.mark.skipif(not (oracle_py.ref_available() and oracle_py.ref_libm_available()), reason="oracle/_ref not built")
SPP = 256


@pytest.fixture(scope="module")
def refs():
    return oracle_py.Oracle("ref"), oracle_py.Oracle("ref_libm")


def _render(o, case, sky, time_seed):
    rd = case["rd"].copy()
    rd["num_samples"] = SPP
    rd["time"] = np.uint32(time_seed)
    img = o.render(rd, case["sd"], case["shapes"], case["tris"], case["mats"], sky, nthreads=8)[..., :3]
    return np.nan_to_num(img.astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0)  # NaN-poisoned pixels carry no statistics


@pytest.mark.parametrize("name", sorted(C.build_cases()))
def test_detmath_built_ins_agree_with_libm_statistically(name, refs, sky):
    det, libm = refs
    case = C.build_cases()[name]
    if bool(case["rd"]["show_normals"]):
        pytest.skip("show_normals images carry no Monte-Carlo noise: covered bit for bit by the goldens")
    a = _render(det, case, sky, 12345)
    b = _render(libm, case, sky, 12345)
    a2 = _render(det, case, sky, 777)  # same implementation, other seeds: the noise floor
    clip = lambda x: np.minimum(x, 4.0)  # fireflies (a few paths into the sun lobe) would dominate every RMS otherwise
    a, b, a2 = clip(a), clip(b), clip(a2)
    mean_a, mean_b = a.mean(), b.mean()
    noise = np.sqrt(((a - a2) ** 2).mean())
    diff = np.sqrt(((a - b) ** 2).mean())
    n = a.size
    # means: difference below 5 standard errors of the difference of two noisy means (plus 0.1 % for exactly-empty scenes)
    assert abs(mean_a - mean_b) <= 5.0 * noise / np.sqrt(n) + 1e-3 * max(mean_a, 1e-6), (name, mean_a, mean_b, noise)
    # per-pixel: same implementation / other seeds differ as much as the two implementations do
    assert diff <= 1.25 * noise + 1e-6, (name, diff, noise)
