"""N>1 path on CPU: world_size-2 `gloo` run of the frame sharding + per-frame gather + un-swizzle
(zig_vulkan_amd/dist.py).  No GPU here, so each rank fills its packed tile shard with the oracle;
the collective, the ownership rule and the reassembly are the code the GPU ranks use."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_path: str):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from tests.helpers import oracle_scene_from_grid, push_for
    from zig_vulkan_amd import workloads as W
    from zig_vulkan_amd.dist import TILE, FrameGather, owned_tile_ids, shard_geometry

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = W.Workload("t", 200, 120, 64, 4, 1, 0, True, 0.0)  # ragged: 200x120 is not a multiple of 16
        grid = W.build_grid(w)  # replicated scene: every rank builds the same deterministic grid
        scene = oracle_scene_from_grid(grid)
        fg = FrameGather(w.width, w.height, rank, world, torch.device("cpu"))
        geom = shard_geometry(w.width, w.height, rank, world)

        def render_shard(view):
            pc = push_for(W.camera_for(w, view), W.sun_for(w))
            shard = np.zeros((geom["tiles_per_rank"], TILE, TILE, 4), dtype=np.uint8)
            for i, t in enumerate(owned_tile_ids(w.width, w.height, rank, world)):
                ty, tx = divmod(int(t), geom["tiles_x"])
                ys, xs = np.mgrid[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE]
                inside = (xs < w.width) & (ys < w.height)
                xy = np.stack([xs[inside], ys[inside]], axis=-1)
                _, u8, _ = O.render_pixels(scene, pc, xy)
                tile = np.zeros((TILE, TILE, 4), dtype=np.uint8)
                tile[inside] = u8
                shard[i] = tile
            return pc, shard

        # the pipelined loop of bench.py: 5 frames over 3 views through 2 buffers
        views = ["V2", "V1", "V0", "V2", "V1"]
        ok = 1
        pcs = []
        for f, view in enumerate(views):
            pc, shard = render_shard(view)
            pcs.append(pc)
            fg.begin_frame(f)
            fg.shard_for(f).copy_(torch.from_numpy(shard.reshape(-1)))
            fg.gather_async(f)   # the one collective per frame
            if f >= 1:
                fg.complete(f - 1, None)  # numpy restatement of vrt_assemble_frame
                if rank == 0:
                    _, full, _ = O.render(scene, pcs[f - 1])
                    ok &= int(np.array_equal(fg.frame_numpy(), full))
        fg.complete(len(views) - 1, None)
        if rank == 0:
            _, full, _ = O.render(scene, pcs[-1])
            ok &= int(np.array_equal(fg.frame_numpy(), full))
            np.save(out_path, np.array([ok, geom["total_tiles"]]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_frame_gather(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    ok, total = np.load(out).tolist()
    assert ok == 1 and total == 13 * 8


def test_shard_geometry_partitions_every_tile_once():
    from zig_vulkan_amd.dist import owned_tile_ids, shard_geometry
    for (wd, ht, world) in [(1920, 1080, 8), (3840, 2160, 4), (250, 131, 3), (16, 16, 2)]:
        seen = np.concatenate([owned_tile_ids(wd, ht, r, world) for r in range(world)])
        g = shard_geometry(wd, ht, 0, world)
        assert sorted(seen.tolist()) == list(range(g["total_tiles"]))
        for r in range(world):
            gr = shard_geometry(wd, ht, r, world)
            assert gr["owned_tiles"] == len(owned_tile_ids(wd, ht, r, world)) <= gr["tiles_per_rank"]


def test_assemble_reference_inverts_the_tile_swizzle():
    from zig_vulkan_amd.dist import TILE, assemble_reference, shard_geometry
    wd, ht, world = 100, 70, 3
    g = shard_geometry(wd, ht, 0, world)
    frame = np.arange(ht * wd * 4, dtype=np.uint32).reshape(ht, wd, 4).astype(np.uint8)
    padded = np.zeros((g["tiles_y"] * TILE, g["tiles_x"] * TILE, 4), dtype=np.uint8)
    padded[:ht, :wd] = frame
    gathered = np.zeros((world, g["tiles_per_rank"], TILE, TILE, 4), dtype=np.uint8)
    for t in range(g["total_tiles"]):
        ty, tx = divmod(t, g["tiles_x"])
        gathered[t % world, t // world] = padded[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE]
    assert np.array_equal(assemble_reference(gathered, wd, ht, world), frame)


def _edit_worker(rank: int, world: int, port: int, out_path: str):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from zig_vulkan_amd import _lib as L
    from zig_vulkan_amd import workloads as W
    from zig_vulkan_amd.dist import broadcast_grid_delta

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w = W.Workload("t", 64, 64, 64, 4, 1, 0, True, 0.0)
        grid = W.build_grid(w)           # replicated scene
        bufs = [L.BUF_BRICK_STATUS, L.BUF_BRICK_INDEX, L.BUF_BRICK_OCCUPANCY, L.BUF_BRICK_START_INDEX, L.BUF_MATERIAL_INDEX]
        # stands in for the rank's device buffers (no GPU here): what VoxelRT.upload would write
        device = {b: np.ascontiguousarray(grid.array(b)).view(np.uint8).reshape(-1).copy() for b in bufs}

        def upload(buf_id, off, data):
            device[buf_id][off:off + data.size] = data

        if rank == 0:                    # only this host edits
            for y in range(20, 60):
                for d in range(3):
                    grid.insert(30 + d, y, 30, 7)
                    grid.insert(5, y, 50 + d, 5)
        applied = broadcast_grid_delta(grid, upload, rank, root=0)
        if rank == 1:
            np.save(out_path, np.concatenate([device[b] for b in bufs]))
            np.save(out_path + ".n.npy", np.array([len(applied), sum(n for _, _, n in applied)]))
        else:
            assert not any(grid.delta(b)[0] for b in bufs)   # root's deltas were reset
            np.save(out_path + ".root.npy", np.concatenate([np.ascontiguousarray(grid.array(b)).view(np.uint8).reshape(-1) for b in bufs]))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_replica_update(tmp_path):
    """Edits made on rank 0's host reach rank 1's buffers through zig_vulkan_amd.dist.broadcast_grid_delta (the torch path's
    counterpart of vrt_dist_broadcast): after the collective rank 1 holds byte for byte what rank 0's BrickGrid holds, and
    only the dirty ranges travelled."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "rank1.npy")
    mp.spawn(_edit_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got, want = np.load(out), np.load(out + ".root.npy")
    n_ranges, n_bytes = np.load(out + ".n.npy").tolist()
    assert np.array_equal(got, want)
    assert 1 <= n_ranges <= 5 and 0 < n_bytes < want.size // 2
