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


def tile_pixels(tiler, tile_indices, block=None):
    """(x, y) of every pixel of the listed tiles, tile after tile, row-major inside a tile.  With block = (w, h) the
    pixels of a tile are listed block after block of w x h pixels instead (row-major inside a block): the renderer
    traces 64 consecutive pixels of the list per wave, and a compact 8 x 8 block of camera rays stays together in the
    tree longer than a 32 x 2 strip.  The order of the list never changes a pixel's result."""
    parts = []
    for i in tile_indices:
        px = tiler.tile(i).pixels()
        if block is not None and px.shape[0]:
            bw, bh = block
            x0, y0 = int(px[:, 0].min()), int(px[:, 1].min())
            bx, by = (px[:, 0].astype(np.int64) - x0) // bw, (px[:, 1].astype(np.int64) - y0) // bh
            nbx = int(bx.max()) + 1
            key = ((by * nbx + bx) << 32) | ((px[:, 1].astype(np.int64) - y0) << 16) | (px[:, 0].astype(np.int64) - x0)
            px = px[np.argsort(key, kind="stable")]
        parts.append(px)
    return np.concatenate(parts, 0) if parts else np.zeros((0, 2), np.uint32)


class FrameGatherer:
    """The gather of one rank's accumulators into rank 0's frame, with everything that does not change from frame to
    frame done once: the pixel -> frame index of every rank (gathered at construction), the peers' pixel counts and
    the padded exchange buffers.  Per frame it costs one gather of [ACCUM_PLANES, n_max] floats and, on rank 0, one
    index_copy per rank.  Without an initialised process group this is the single-rank case."""

    def __init__(self, xy_local: np.ndarray, width: int, height: int, device, offset=(0, 0), group=None,
                 always_collective=False):
        self.width, self.height, self.group = width, height, group
        self.device = torch.device(device)
        lin = (xy_local[:, 1].astype(np.int64) - offset[1]) * width + (xy_local[:, 0].astype(np.int64) - offset[0])
        self.lin_local = torch.from_numpy(lin).to(self.device)
        self.n_local = int(xy_local.shape[0])
        # always_collective: go through the collectives even in a group of one (exercises the backend in tests)
        self.distributed = dist.is_available() and dist.is_initialized() and \
            (dist.get_world_size(group) > 1 or always_collective)
        self.rank = dist.get_rank(group) if self.distributed else 0
        self.world = dist.get_world_size(group) if self.distributed else 1
        if not self.distributed:
            return
        n_local = torch.tensor([self.n_local], dtype=torch.int64, device=self.device)
        counts = [torch.zeros_like(n_local) for _ in range(self.world)]
        dist.all_gather(counts, n_local, group=group)
        self.counts = [int(c.item()) for c in counts]
        self.n_max = max(self.counts)
        pad_lin = torch.zeros((self.n_max,), dtype=torch.int64, device=self.device)
        pad_lin[:self.n_local] = self.lin_local
        lin_list = [torch.empty_like(pad_lin) for _ in range(self.world)] if self.rank == 0 else None
        dist.gather(pad_lin, lin_list, dst=0, group=group)
        self.lin = [lin_list[r][:self.counts[r]] for r in range(self.world)] if self.rank == 0 else None
        self.pad_acc = torch.zeros((ACCUM_PLANES, self.n_max), dtype=torch.float32, device=self.device)
        self.acc_list = ([torch.empty_like(self.pad_acc) for _ in range(self.world)] if self.rank == 0 else None)

    def gather(self, accum_local: torch.Tensor, frame: torch.Tensor = None):
        """Returns the [ACCUM_PLANES, height * width] frame (row-major pixels) on rank 0, None elsewhere.  `frame`, if
        given, is overwritten where pixels were rendered and returned."""
        if self.rank == 0 and frame is None:
            frame = torch.zeros((ACCUM_PLANES, self.width * self.height), dtype=torch.float32, device=self.device)
        if not self.distributed:
            frame.index_copy_(1, self.lin_local, accum_local)
            return frame
        self.pad_acc[:, :self.n_local] = accum_local
        dist.gather(self.pad_acc, self.acc_list, dst=0, group=self.group)
        if self.rank != 0:
            return None
        for r in range(self.world):
            frame.index_copy_(1, self.lin[r], self.acc_list[r][:, :self.counts[r]])
        return frame


def gather_frame(accum_local: torch.Tensor, xy_local: np.ndarray, width: int, height: int, offset=(0, 0), group=None):
    """One-off form of FrameGatherer: gathers every rank's [ACCUM_PLANES, n_local] accumulators to rank 0 and scatters
    them into one frame.  Returns the frame on rank 0, None elsewhere."""
    return FrameGatherer(xy_local, width, height, accum_local.device, offset, group).gather(accum_local)
