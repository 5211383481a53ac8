"""Tile partition of one frame over the ranks of a node, and the gather of the per-pixel accumulators.

The reference's only parallelism is shared-memory work sharing over 32x32 image tiles
(crates/rt-cli/src/executor.rs:307-330, crates/rt-cli/src/tile.rs:56-97).  Here each rank (one process per GPU) owns
every world_size-th tile -- interleaved, because cost per tile varies with what it sees -- holds a full copy of the
scene, and renders all samples of its tiles.  Samples depend only on (seed, x, y, sample index), so the assembled
frame is bit-identical to a single-GPU render.  The one exchange step is a gather of the accumulator planes to
rank 0 (torch.distributed: RCCL over xGMI on GPUs, gloo on CPU); nothing is reduced.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from ._lib import ACCUM_PLANES


def partition_tiles(n_tiles: int, world_size: int, rank: int) -> range:
    """Tile indices owned by `rank`: rank, rank + world_size, ..."""
    return range(rank, n_tiles, world_size)


def tile_pixels(tiler, tile_indices):
    """(x, y) of every pixel of the listed tiles, tile after tile, row-major inside a tile."""
    parts = [tiler.tile(i).pixels() for i in tile_indices]
    return np.concatenate(parts, 0) if parts else np.zeros((0, 2), np.uint32)


def gather_frame(accum_local: torch.Tensor, xy_local: np.ndarray, width: int, height: int, offset=(0, 0), group=None):
    """Gathers every rank's [ACCUM_PLANES, n_local] accumulators to rank 0 and scatters them into one
    [ACCUM_PLANES, height * width] frame (row-major pixels).  Returns the frame on rank 0, None elsewhere.
    Without an initialised process group this is the single-rank case."""
    device = accum_local.device
    lin_local = torch.from_numpy((xy_local[:, 1].astype(np.int64) - offset[1]) * width +
                                 (xy_local[:, 0].astype(np.int64) - offset[0])).to(device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        frame = torch.zeros((ACCUM_PLANES, width * height), dtype=torch.float32, device=device)
        frame.index_copy_(1, lin_local, accum_local)
        return frame
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_local = torch.tensor([accum_local.shape[1]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    # payload: accumulator planes plus the pixel index (as f32 bit patterns of two 16-bit halves would lose nothing,
    # but a separate int64 gather is simpler and tiny)
    pad_acc = torch.zeros((ACCUM_PLANES, n_max), dtype=torch.float32, device=device)
    pad_acc[:, :accum_local.shape[1]] = accum_local
    pad_lin = torch.zeros((n_max,), dtype=torch.int64, device=device)
    pad_lin[:lin_local.shape[0]] = lin_local
    if rank == 0:
        acc_list = [torch.empty_like(pad_acc) for _ in range(world)]
        lin_list = [torch.empty_like(pad_lin) for _ in range(world)]
    else:
        acc_list = lin_list = None
    dist.gather(pad_acc, acc_list, dst=0, group=group)
    dist.gather(pad_lin, lin_list, dst=0, group=group)
    if rank != 0:
        return None
    frame = torch.zeros((ACCUM_PLANES, width * height), dtype=torch.float32, device=device)
    for r in range(world):
        frame.index_copy_(1, lin_list[r][:counts[r]], acc_list[r][:, :counts[r]])
    return frame
