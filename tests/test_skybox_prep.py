"""The skybox texel preparation of SURVEY.md 8 a-12 that the C ABI leaves to the caller: 8-bit RGB(A) -> the RGBA32F image
stbi_loadf(..., 4) + flip gives the reference (src/tracer.cpp:42-46, lib/stb_image.h:1857-1878). CPU only: the C++ header
(host/skybox.hpp), its Python mirror (scenes.skybox_from_rgb8) and the committed 256-entry table must agree bit for bit."""
import json
import subprocess
from pathlib import Path

import numpy as np
import pytest

from simple_raytracer_amd import scenes as S

ROOT = Path(__file__).resolve().parent.parent
GOLD = json.loads((ROOT / "tests/golden/skybox_l2h_table.json").read_text())


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = tmp_path_factory.mktemp("sky") / "skybox_table"
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", str(ROOT / "tests/csrc/skybox_table.cpp"), "-o", str(out)], check=True)
    return str(out)


def run(exe, img=None):
    text = ""
    if img is not None:
        h, w, c = img.shape
        text = f"{w} {h} {c}\n" + " ".join(str(int(v)) for v in img.reshape(-1))
    words = subprocess.run([exe], input=text, capture_output=True, text=True, check=True).stdout.split()
    table = np.array([int(x, 16) for x in words[:512]], np.uint32).reshape(256, 2)
    image = np.array([int(x, 16) for x in words[513:]], np.uint32) if img is not None else None
    return table, image


def test_table_equals_golden(exe):
    table, _ = run(exe)
    assert [f"{v:08x}" for v in table[:, 0]] == GOLD["colour"]
    assert [f"{v:08x}" for v in table[:, 1]] == GOLD["alpha"]
    # the ends of the map: 0 -> 0, 255 -> 1, and it is monotone
    col = table[:, 0].view(np.float32)
    assert col[0] == 0.0 and col[255] == 1.0 and np.all(np.diff(col) > 0)


def test_python_mirror_equals_golden():
    ramp = np.arange(256, dtype=np.uint8).reshape(1, 256, 1)
    rgba = S.skybox_from_rgb8(np.concatenate([ramp, ramp], axis=2))  # grey + alpha
    assert [f"{v:08x}" for v in rgba[0, :, 0].view(np.uint32)] == GOLD["colour"]
    assert [f"{v:08x}" for v in rgba[0, :, 3].view(np.uint32)] == GOLD["alpha"]


@pytest.mark.parametrize("channels", [1, 2, 3, 4])
def test_image_flip_and_channels(exe, channels):
    rng = np.random.RandomState(channels)
    img = rng.randint(0, 256, size=(5, 7, channels)).astype(np.uint8)
    _, image = run(exe, img)
    want = S.skybox_from_rgb8(img)
    assert np.array_equal(image.reshape(5, 7, 4), want.view(np.uint32))
    # row 0 of the result is the LAST row of the file (stbi_set_flip_vertically_on_load(1)), alpha 1 where the file has none
    col = np.array([int(x, 16) for x in GOLD["colour"]], np.uint32).view(np.float32)
    assert want[0, 0, 0] == col[img[-1, 0, 0]]
    if channels in (1, 3):
        assert np.all(want[..., 3] == 1.0)
    if channels < 3:
        assert np.array_equal(want[..., 0], want[..., 1]) and np.array_equal(want[..., 0], want[..., 2])
