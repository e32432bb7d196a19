"""Multi-GPU frame sharding: one process per GPU, image tiles interleaved across
ranks, ONE gather of RGBA8 tiles to rank 0 per frame (RCCL over xGMI through
torch.distributed's "nccl" backend), then a device-side un-swizzle on rank 0.

The scene is replicated (each rank uploads the same grid); pixels are
independent, so there is no other exchange step.

Tile ownership (must match the kernel and vrt_assemble_frame): tiles are
16x16 pixels, numbered row-major; tile t belongs to rank t % R and is the
(t // R)-th tile of that rank's packed shard; every shard is padded to
tiles_per_rank = ceil(T / R) tiles so the gather has equal counts.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

TILE = 16


def shard_geometry(width: int, height: int, rank: int, world: int) -> dict:
    tiles_x = (width + TILE - 1) // TILE
    tiles_y = (height + TILE - 1) // TILE
    total = tiles_x * tiles_y
    owned = (total - rank + world - 1) // world if total > rank else 0
    return {"tiles_x": tiles_x, "tiles_y": tiles_y, "total_tiles": total, "owned_tiles": owned,
            "tiles_per_rank": (total + world - 1) // world}


def owned_tile_ids(width: int, height: int, rank: int, world: int) -> np.ndarray:
    g = shard_geometry(width, height, rank, world)
    return np.arange(rank, g["total_tiles"], world, dtype=np.int64)


def assemble_reference(gathered: np.ndarray, width: int, height: int, world: int) -> np.ndarray:
    """Host restatement of vrt_assemble_frame for checking (numpy).  gathered: [world, tiles_per_rank, 16, 16, C]."""
    g = shard_geometry(width, height, 0, world)
    c = gathered.shape[-1]
    gathered = gathered.reshape(world, g["tiles_per_rank"], TILE, TILE, c)
    frame = np.zeros((g["tiles_y"] * TILE, g["tiles_x"] * TILE, c), dtype=gathered.dtype)
    for t in range(g["total_tiles"]):
        ty, tx = divmod(t, g["tiles_x"])
        frame[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE] = gathered[t % world, t // world]
    return frame[:height, :width]


def broadcast_grid_delta(grid, upload, rank: int, root: int = 0, group=None) -> list:
    """Replica update of the torch path (the native pipeline has vrt_dist_broadcast): incremental edits made on rank `root`'s
    host BrickGrid reach every rank's device buffers.  Root packs its dirty ranges (Grid.zig:129-194) as (buffer id, byte
    offset, bytes), one broadcast_object_list carries them, and EVERY rank — root too — applies them with `upload(buffer_id,
    byte_offset, uint8 array)` (VoxelRT.upload); root's deltas are reset as VoxelRT.updateGridDelta does (VoxelRT.zig:107-172).
    Collective; any backend.  Returns the ranges applied as (buffer id, byte offset, bytes).
    Only the DEVICE replicas are brought up to date: the host BrickGrid objects of the other ranks are not edited (a host that wants
    them in step applies the same edits to them, or keeps no grid at all on those ranks — the renderer needs none after the upload)."""
    import torch.distributed as dist
    from . import _lib as L
    # the five buffers BrickGrid tracks deltas for, with their element sizes
    tracked = ((L.BUF_BRICK_STATUS, 4), (L.BUF_BRICK_INDEX, 4), (L.BUF_BRICK_OCCUPANCY, 1), (L.BUF_BRICK_START_INDEX, 4), (L.BUF_MATERIAL_INDEX, 1))
    box = [None]
    if rank == root:
        pieces = []
        for buf_id, es in tracked:
            active, a, b = grid.delta(buf_id)
            if active and b > a:
                # (a view of the host array: only the dirty range is copied, not the whole buffer — gigabytes on the large grids)
                view = grid.array_view(buf_id).view(np.uint8).reshape(-1)
                pieces.append((buf_id, a * es, view[a * es:b * es].tobytes()))
        box[0] = pieces
    dist.broadcast_object_list(box, src=root, group=group)
    for buf_id, off, payload in box[0]:
        upload(buf_id, off, np.frombuffer(payload, dtype=np.uint8))
    if rank == root:
        for buf_id, _ in tracked:
            grid.reset_delta(buf_id)
    return [(b, o, len(p)) for b, o, p in box[0]]


class FrameGather:
    """Per-frame collective of the sharded renderer, pipelined over `depth` frames.

    Each rank owns `depth` packed tile buffers; frame f renders into buffer f % depth and its gather
    is started asynchronously, so the traversal kernel of frame f+1 overlaps the gather of frame f
    (RCCL runs on its own stream; the only ordering points are: a buffer is not rewritten before the
    gather that reads it has finished, and rank 0 un-swizzles a frame after its gather).  Works with
    any torch.distributed backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" on CPU tensors in the
    world_size-2 tests.  One gather per frame; no other data-path collective."""

    def __init__(self, width: int, height: int, rank: int, world: int, device, bytes_per_pixel: int = 4, depth: int = 2):
        import torch
        self.torch = torch
        self.width, self.height, self.rank, self.world, self.depth = width, height, rank, world, depth
        self.geom = shard_geometry(width, height, rank, world)
        self.bpp = bytes_per_pixel
        self.shard_bytes = self.geom["tiles_per_rank"] * TILE * TILE * bytes_per_pixel
        self.device = device
        self.shards = [torch.zeros(self.shard_bytes, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.work = [None] * depth
        self.gathered = None
        self.frame = None
        if rank == 0:
            self.gathered = [torch.zeros(world * self.shard_bytes, dtype=torch.uint8, device=device) for _ in range(depth)]
            self.frame = torch.zeros(height * width * bytes_per_pixel, dtype=torch.uint8, device=device)
            self._recv_views = [list(g.view(world, self.shard_bytes).unbind(0)) for g in self.gathered]

    @property
    def shard(self):  # buffer of frame 0 (single-frame use)
        return self.shards[0]

    def shard_for(self, f: int):
        """Buffer frame f renders into.  Call begin_frame(f) first."""
        return self.shards[f % self.depth]

    def begin_frame(self, f: int) -> None:
        """Order the rewrite of buffer f % depth after the gather of frame f - depth that read it."""
        slot = f % self.depth
        w = self.work[slot]
        if w is not None:
            w.wait()  # nccl: the current stream waits (no host block); gloo: host waits
            self.work[slot] = None

    def gather_async(self, f: int) -> None:
        """Start the one collective of frame f: every rank's packed shard -> rank 0, rank-major."""
        import torch.distributed as dist
        slot = f % self.depth
        if self.world == 1:
            self.gathered[slot].copy_(self.shards[slot])
            return
        self.work[slot] = dist.gather(self.shards[slot], gather_list=self._recv_views[slot] if self.rank == 0 else None, dst=0,
                                      async_op=True)

    def complete(self, f: int, renderer=None) -> None:
        """Rank 0: wait for frame f's gather and un-swizzle it into the row-major frame (device kernel
        when a renderer context is given; numpy restatement otherwise — CPU tests only)."""
        slot = f % self.depth
        w = self.work[slot]
        if w is not None:
            w.wait()
            self.work[slot] = None
        if self.rank != 0:
            return
        if renderer is not None:
            renderer.assemble_frame(self.gathered[slot].data_ptr(), self.frame.data_ptr(), self.bpp)
        else:
            g = self.gathered[slot].cpu().numpy().reshape(self.world, -1)
            px = g.reshape(self.world, self.geom["tiles_per_rank"], TILE, TILE, self.bpp)
            out = assemble_reference(px, self.width, self.height, self.world)
            self.frame.copy_(self.torch.from_numpy(np.ascontiguousarray(out).reshape(-1)))

    # single-frame convenience (tests)
    def gather(self) -> None:
        self.gather_async(0)

    def assemble(self, renderer=None) -> None:
        self.complete(0, renderer)

    def frame_numpy(self) -> np.ndarray:
        assert self.rank == 0
        return self.frame.cpu().numpy().reshape(self.height, self.width, self.bpp)
