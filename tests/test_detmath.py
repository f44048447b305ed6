"""CPU: detmath (the pinned realisation of the OpenCL built-ins) stays inside the
OpenCL ULP bounds against glibc's double libm: cos <= 4, log <= 3, pow <= 16,
atan2pi <= 6 ulp. tests/csrc/ulp_check.c with stride 1 is exhaustive over the kernel's
domains (cos on every float in [0, 2pi], log on every float in (0, 1]); measured maxima
with stride 1 / 7 (revision 2, FMA forms): cos 1.55, log 0.81, pow 0.50 (integer exponents 1..32: 0.50),
atan2pi 2.86, normalize 2.74 per component (bound 2 + 3), its division-free rsqrt 1.00 ulp."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BOUNDS = {"cos": 4.0, "cos_wide": 4.0, "log01": 3.0, "log_big": 3.0, "pow": 16.0, "powi": 16.0, "atan2pi": 6.0,
          "normalize": 5.0, "rsqrt": 2.0}  # normalize: 2 + n ulp for n = 3 components (OpenCL 3.0 table of geometric functions)


@pytest.fixture(scope="module")
def ulp_report(tmp_path_factory):
    exe = tmp_path_factory.mktemp("ulp") / "ulp_check"
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", f"-I{ROOT}/simple-raytracer_amd/csrc",
                    str(ROOT / "tests/csrc/ulp_check.c"), "-o", str(exe), "-lm"], check=True)
    out = subprocess.run([str(exe), "61"], check=True, capture_output=True, text=True).stdout
    rep = {}
    for line in out.strip().splitlines():
        name, worst, n = line.split()
        rep[name] = (float(worst), int(n))
    return rep


def test_ulp_bounds(ulp_report):
    for name, bound in BOUNDS.items():
        worst, n = ulp_report[name]
        assert n > 1000000, name
        assert worst <= bound, f"{name}: {worst} ulp > {bound}"
    # we do far better than the spec requires; keep it that way
    assert ulp_report["cos"][0] < 2.0 and ulp_report["log01"][0] < 1.0 and ulp_report["powi"][0] < 1.0 and ulp_report["atan2pi"][0] < 4.0
    assert ulp_report["normalize"][0] < 3.5 and ulp_report["rsqrt"][0] < 1.5


def test_special_values(ulp_report):
    assert ulp_report["specials"][0] == 0.0
