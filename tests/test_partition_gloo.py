"""CPU, world_size 2 over gloo: the multi-GPU data flow of bench.py / the north star —
each rank renders ONLY its interleaved row blocks, one gather of the packed per-rank
canvases to rank 0, unpermute — must reproduce the single-device canvas bit for bit.
The per-rank render here is the oracle restricted to the rank's rows (no GPU in this
container); the partition, packing, gather and unpermute code is the product's
(simple_raytracer_amd.tracer / multi.py), the same code the GPU path uses."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, out_path, rpb):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    import srt_pkg
    srt_pkg.load()
    from simple_raytracer_amd import multi, scenes as S, tracer
    from oracle.oracle_py import Oracle
    import golden_io

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = golden_io.load_cases()["mixed"]
        sky = S.synthetic_sky()
        orc = Oracle("oracle")
        w, h = int(g["rd"]["width"]), int(g["rd"]["height"])
        part = multi.RowPartition(h, rank, world, rpb)
        # render only the owned rows, then pack them the way the device canvas is packed
        full = np.zeros((h, w, 4), np.float32)
        for y0, y1 in part.owned_row_ranges():
            orc.render(g["rd"], g["sd"], g["shapes"], g["tris"], g["mats"], sky, canvas=full, rows=(y0, y1), nthreads=2)
        packed = part.pack(full)  # (padded_rows, w, 4)
        image = multi.gather_rows(torch.from_numpy(packed), part, dst=0)
        if rank == 0:
            np.save(out_path, image.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("rpb", [8, 5])
def test_two_rank_gather_reproduces_single_device_canvas(tmp_path, rpb):
    import golden_io
    out = tmp_path / "img.npy"
    port = 29500 + (os.getpid() % 2000) + rpb
    mp.spawn(_worker, args=(2, port, str(out), rpb), nprocs=2, join=True)
    got = np.load(out)
    want = golden_io.load_cases()["mixed"]["canvas"]
    from conftest import bits_equal
    assert bits_equal(got, want)
