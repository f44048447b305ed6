"""Multi-GPU tile split: one process per GPU, rows dealt in interleaved blocks, ONE
gather of the packed per-rank buffers to rank 0 (RCCL over xGMI when the backend is
"nccl"; gloo on CPU for tests), then an unpermute. New work: the reference is
single-device (src/tracer.cpp:13). Pixels are independent and seeded by their GLOBAL
index (render.cl:488,496), so no other exchange exists on this path.

Why interleaved blocks: sky rows are cheap, floor/glass rows expensive; dealing blocks
of `rows_per_block` scanlines round-robin gives every GPU the same mix. Why a gather
and not a ring: each peer has its own xGMI link to the root, so 7 independent
transfers of 1/8 canvas arrive in parallel."""
import numpy as np
import torch
import torch.distributed as dist

from . import tracer as T


class RowPartition:
    def __init__(self, height, rank, world, rows_per_block=8):
        self.height, self.rank, self.world, self.rpb = height, rank, world, rows_per_block
        self.owned = T.owned_rows(height, rank, world, rows_per_block)
        self.padded = T.padded_rows(height, world, rows_per_block)

    def local_to_global(self):
        return [T.global_row(self.height, self.rank, self.world, self.rpb, lr) for lr in range(self.padded)]

    def owned_row_ranges(self):
        """[(y0, y1), ...] contiguous global row ranges owned by this rank."""
        ys = [y for y in self.local_to_global() if y >= 0]
        ranges = []
        for y in ys:
            if ranges and ranges[-1][1] == y:
                ranges[-1][1] = y + 1
            else:
                ranges.append([y, y + 1])
        return [tuple(r) for r in ranges]

    def pack(self, image):
        """Full image (height, ...) -> this rank's packed rows (padded_rows, ...); padding rows zero."""
        out = np.zeros((self.padded,) + image.shape[1:], image.dtype)
        for lr, y in enumerate(self.local_to_global()):
            if y >= 0:
                out[lr] = image[y]
        return out


def gather_rows(local, part, dst=0, group=None):
    """ONE gather of equal-size packed buffers (padded_rows, width, C) to `dst`, then the
    unpermute to (height, width, C). Returns the image on dst, None elsewhere. Works on
    CUDA tensors (RCCL) and CPU tensors (gloo)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert local.shape[0] == part.padded
    local = local.contiguous()
    if world == 1:
        gathered = local
    else:
        bufs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
        dist.gather(local, bufs, dst=dst, group=group)
        if rank != dst:
            return None
        gathered = torch.cat(bufs, dim=0)
    idx = unpermute_index(part.height, world, part.rpb, part.padded)
    return gathered.index_select(0, torch.as_tensor(idx, device=gathered.device))


def unpermute_index(height, world, rpb, padded):
    """index[y] = row of the rank-major gathered buffer that holds global row y."""
    idx = np.zeros(height, np.int64)
    for r in range(world):
        for lr in range(padded):
            y = T.global_row(height, r, world, rpb, lr)
            if y >= 0:
                idx[y] = r * padded + lr
    return idx
