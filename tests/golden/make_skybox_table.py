"""Generates tests/golden/skybox_l2h_table.json: the 256 colour texels and 256 alpha texels stb_image's float loader makes of
an 8-bit channel ((float)pow(byte / 255.0f, 2.2f), byte / 255.0f; /root/reference/lib/stb_image.h:1857-1878, :1572), as float
bit patterns. Values come from tests/csrc/skybox_table.cpp (= simple-raytracer_amd/host/skybox.hpp, glibc pow) in the build
container and are cross-checked here against 60-digit decimal arithmetic: each colour texel must be the float nearest to
RN_double(q^g) for q = float(byte / 255), g = double(2.2f) -- i.e. what a correctly rounded pow gives.
usage: python tests/golden/make_skybox_table.py"""
import json, struct, subprocess, sys
from decimal import Decimal, getcontext
from pathlib import Path
import numpy as np

ROOT = Path(__file__).resolve().parents[2]
exe = "/tmp/skybox_table_gen"
subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", str(ROOT / "tests/csrc/skybox_table.cpp"), "-o", exe], check=True)
out = subprocess.run([exe], input="", capture_output=True, text=True, check=True).stdout.split()
colour = [int(out[2 * i], 16) for i in range(256)]
alpha = [int(out[2 * i + 1], 16) for i in range(256)]

getcontext().prec = 60
g = Decimal(float(np.float32(2.2)))
for b in range(256):
    q = np.float32(b) / np.float32(255.0)
    assert struct.unpack("<I", struct.pack("<f", q))[0] == alpha[b], b
    exact = Decimal(0) if b == 0 else (Decimal(float(q)).ln() * g).exp()
    as_double = float(exact)  # correctly rounded to double (decimal -> float conversion rounds to nearest)
    want = struct.unpack("<I", struct.pack("<f", np.float32(as_double)))[0]
    assert want == colour[b], (b, hex(want), hex(colour[b]))
(ROOT / "tests/golden/skybox_l2h_table.json").write_text(json.dumps({
    "what": "stb_image float loader, 8-bit channel -> linear float: colour = (float)pow(byte / 255.0f, 2.2f), alpha = byte / 255.0f (float bit patterns, hex)",
    "reference": "/root/reference/lib/stb_image.h:1857-1878 (stbi__ldr_to_hdr), :1572 (gamma 2.2f, scale 1.0f); /root/reference/src/tracer.cpp:42-46",
    "generator": "tests/golden/make_skybox_table.py (host/skybox.hpp in the build container, cross-checked against 60-digit decimal arithmetic)",
    "colour": [f"{v:08x}" for v in colour], "alpha": [f"{v:08x}" for v in alpha]}, indent=0) + "\n")
print("wrote tests/golden/skybox_l2h_table.json; colour[1], [128], [255] =", [hex(colour[i]) for i in (1, 128, 255)])
