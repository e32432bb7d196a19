#!/usr/bin/env python3
"""bench.py — Mrays/s of the brickmap traversal path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload NAME]

A step is one full frame of the workload (all pixels, every ray the shader
casts for them) at one of the three fixed camera views: the first third of the
steps renders V0, the second third V1, the last third V2.  The
scene is resident in HBM before the timed region.  For N>1 the driver launches
one process per GPU (torch.distributed, backend nccl = RCCL): the frame is
sharded by interleaved 16x16 tiles and gathered to rank 0 once per frame.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VIEW_ORDER = ["V0", "V1", "V2"]


def cpu_baseline(w, grid, min_wall_s: float):
    """Oracle (CPU restatement of the reference shader, oracle/vrt_oracle.c) on the host cores of this
    box.  Bounded sample: whole frames of the workload, views cycled V0,V1,V2, until `min_wall_s` of
    wall time has passed (at least one frame per view); rows are handed to the threads one at a time.
    kind = "port": the reference itself cannot be built here (no zig / GLSL compiler / Vulkan ICD)."""
    import ctypes as C
    import threading

    import numpy as np
    from oracle import oracle as O
    from tests.helpers import oracle_scene_from_grid
    from zig_vulkan_amd import workloads as W
    cores = os.cpu_count() or 1
    scene = oracle_scene_from_grid(grid)
    L = O.lib()
    pcs = [O.push_constants(W.camera_for(w, v).blob(), W.sun_for(w).blob()) for v in VIEW_ORDER]
    f32 = np.zeros((w.height, w.width, 4), dtype=np.float32)
    lock = threading.Lock()
    state = {"next": 0, "frames": 0, "stop": False}
    rays_total = [0] * cores
    t0 = time.perf_counter()

    def worker(tid: int) -> None:
        c = O.Counters()
        while True:
            with lock:
                if state["stop"]:
                    break
                i = state["next"]
                state["next"] += 1
                frame, row = divmod(i, w.height)
                if row == 0 and frame >= len(VIEW_ORDER) and time.perf_counter() - t0 >= min_wall_s:
                    state["stop"] = True
                    state["frames"] = frame
                    break
            pc = pcs[frame % len(VIEW_ORDER)]
            L.oracle_render_rows(C.byref(scene.c), pc.ctypes.data, row, row + 1, f32.ctypes.data, None, C.byref(c))
        rays_total[tid] = c.rays

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(cores)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    rays = sum(rays_total)
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"{state['frames']} whole frames of {w.name} (views {'/'.join(VIEW_ORDER)} cycled), {rays} rays in {dt:.2f} s wall "
                      f"= {dt * cores:.0f} core-seconds; oracle/vrt_oracle.c gcc -O2, {cores} threads, rows handed out one at a time"}


def hbm_traffic_from_profile(brick_dimension: int):
    """HBM bytes per launch of the traversal kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/*_pmc.json, written by tools/summarize_prof.py; bench.py cannot run rocprofv3 on
    itself).  FETCH_SIZE and WRITE_SIZE are in KiB, collected in separate passes.  WRITE_SIZE equals the
    RGBA8 frame exactly (8100 KiB at 1080p); FETCH_SIZE on gfx950 reports half of the bytes fetched
    (MI355X_MICROARCH.md) — for scattered dword loads as well: tools/ubench/fetch_calib.hip reads a 2 GiB buffer
    with one dword per 128-byte line and gets 1.00 GiB, memory being fetched in whole lines and tallied at 64 bytes
    per line — so `traffic` uses the doubled figure (the raw one is quoted beside it)."""
    import glob
    # the shipped kernel's passes are r<round>_final_pmc.json (latest round last); other *_pmc.json files are earlier kernels
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_final_pmc.json")))
    if not files:
        return None, "no profiles/r*_final_pmc.json"
    with open(files[-1]) as fh:
        data = json.load(fh)
    for name, c in data.items():
        if "vrt_trace_kernel<%d, false" % brick_dimension in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            fetch, write = c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
            return 2.0 * fetch + write, (f"{os.path.basename(files[-1])}: FETCH_SIZE {fetch / 1e6:.2f} MB raw (x2 = {2 * fetch / 1e6:.2f} MB), "
                                         f"WRITE_SIZE {write / 1e6:.2f} MB per launch; the touched scene data lives in L2/MALL")
    return None, "traversal kernel not found in " + os.path.basename(files[-1])


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default=None)
    ap.add_argument("--variant", type=lambda x: int(x, 0), default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=2.0, help="minimum wall time of the CPU baseline sample")
    ap.add_argument("--frames-in-flight", type=int, default=2, help="1: frames strictly one after another; 2: two frames in flight")
    ap.add_argument("--dist", choices=["native", "torch"], default="native",
                    help="N>1 frame gather: native = RCCL send/recv inside libvrt_hip.so (pipelined, 4 frames in flight); "
                         "torch = torch.distributed.gather from Python (fallback)")
    ap.add_argument("--dist-frames", type=int, default=4, help="launches in flight per rank of the native multi-GPU pipeline")
    ap.add_argument("--dist-batch", type=int, default=4,
                    help="frames traced by one launch and gathered by one collective when world > 1 (a rank owns 1/world of the tiles)")
    ap.add_argument("--root-share", type=int, default=-1,
                    help="native multi-GPU pipeline: rank 0's share of the tiles in percent of an equal share (it also takes in the other "
                         "ranks' shards and un-swizzles every frame); -1: 100 - 40 (world - 1) / 7, i.e. 60 at 8 GPUs, 83 at 4, 94 at 2")
    ap.add_argument("--force-gather", action="store_true", help="run the shard/gather/assemble path even at world size 1")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio (flushed at exit, i.e.
    # after anything Python printed), so everything else written to fd 1 by any library is sent to stderr and the
    # JSON line goes to a private copy of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    from zig_vulkan_amd import workloads as W
    from zig_vulkan_amd.dist import FrameGather

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the traversal path has no CPU implementation")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    w = W.WORKLOADS[args.workload or W.HEADLINE]
    grid = W.build_grid(w)
    sharded = world > 1 or args.force_gather
    stream = torch.cuda.current_stream().cuda_stream

    # ---- rays and algorithmic bytes per view: counted by a counters build of the kernel (untimed) ----
    per_view = {}
    if rank == 0:
        rtc = W.make_renderer(w, grid, enable_counters=True, device_id=local_rank, kernel_variant=args.variant)
        for v in VIEW_ORDER:
            W.set_view(rtc, v)
            rtc.draw()
            c = rtc.counters()
            per_view[v] = {"rays": c["rays"],
                           "bytes": 4 * c["status_loads"] + 4 * c["bricks_entered"] + c["voxel_steps"] + 25 * c["hits"]
                           + 4 * w.width * w.height,
                           "counters": c}
        rtc.deinit()
    if dist is not None and world > 1:
        obj = [per_view]
        dist.broadcast_object_list(obj, src=0)
        per_view = obj[0]

    # ---- the timed renderer ----
    fg = None
    native = False
    if sharded and args.dist == "native":
        # RCCL inside libvrt_hip.so: kernel -> grouped send/recv to rank 0 -> un-swizzle, several frames in flight
        from zig_vulkan_amd import VoxelRT
        ok = 1
        rt = None
        try:
            uid = [VoxelRT.dist_unique_id() if rank == 0 else None]
            if dist is not None and world > 1:
                dist.broadcast_object_list(uid, src=0)
            root_share = args.root_share if args.root_share >= 0 else max(30, 100 - (40 * (world - 1) + 3) // 7)
            rt = W.make_renderer(w, grid, device_id=local_rank, shard_rank=rank, shard_count=world, kernel_variant=args.variant,
                                 shard_root_weight=(root_share if 2 <= world <= 8 else 0))
            rt.dist_init(uid[0], rank, world, args.dist_frames, frames_per_launch=(args.dist_batch if world > 1 else 1))
            if world == 1:
                rt.dist_selftest()
        except Exception as e:  # noqa: BLE001 - any failure means "use the torch path"
            print(f"[bench rank {rank}] native RCCL pipeline unavailable: {e}", file=sys.stderr)
            ok = 0
        if dist is not None and world > 1:  # every rank must take the same path
            t = torch.tensor([ok], dtype=torch.int32, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = int(t.item())
        native = bool(ok)
        if not native and rt is not None:
            rt.deinit()
    if sharded and not native:
        fg = FrameGather(w.width, w.height, rank, world, torch.device("cuda", local_rank))
        rt = W.make_renderer(w, grid, device_id=local_rank, shard_rank=rank, shard_count=world, stream=stream,
                             external_target_rgba8=fg.shard.data_ptr(), kernel_variant=args.variant)
    elif not sharded:
        rt = W.make_renderer(w, grid, device_id=local_rank, kernel_variant=args.variant, frames_in_flight=args.frames_in_flight)
    if native:
        rt.dist_wait()
    else:
        rt.wait()

    cams = {}
    for v in VIEW_ORDER:
        W.set_view(rt, v)
        cams[v] = bytes(rt.camera.d_camera)

    import ctypes as C
    from zig_vulkan_amd import _lib as L

    def view_of(i: int, n: int) -> str:
        # frames of one view are consecutive (a camera moves smoothly; the tile schedule feeds on the
        # previous frame): first third V0, second third V1, last third V2
        return VIEW_ORDER[min(len(VIEW_ORDER) - 1, (i * len(VIEW_ORDER)) // max(n, 1))]

    frame_no = [0]  # frames submitted so far (warm-up included): the gather pipeline's slot counter

    def step(i: int, n: int) -> None:
        v = view_of(i, n)
        C.memmove(C.byref(rt.camera.d_camera), cams[v], 96)
        if not sharded:
            rt.draw()
            return
        if native:
            rt.dist_frame()                     # kernel -> one RCCL gather -> un-swizzle, on this frame's stream
            return
        f = frame_no[0]
        frame_no[0] += 1
        fg.begin_frame(f)                       # buffer f%2 is free once frame f-2's gather is done
        rt.set_target(fg.shard_for(f).data_ptr())
        rt.draw()                               # this rank's tiles of frame f
        fg.gather_async(f)                      # ONE collective per frame, overlapped with frame f+1's kernel
        if f >= 1:
            fg.complete(f - 1, rt)              # rank 0: un-swizzle the previous frame

    def drain() -> None:
        if native:
            rt.dist_wait()
        elif sharded and frame_no[0] >= 1:
            fg.complete(frame_no[0] - 1, rt)

    def barrier() -> None:
        if dist is not None and world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, args.warmup)
    drain()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, args.steps)
    drain()  # the last frame's gather + un-swizzle belong to the timed region
    if not native:
        rt.wait()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None and world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- dominant-kernel duration with HIP events on the kernel's own stream (N=1 only) ----
    roofline = None
    kernel_ms_view = {}
    if rank == 0 and not sharded:
        reps = max(5, args.steps // len(VIEW_ORDER))
        for v in VIEW_ORDER:
            C.memmove(C.byref(rt.camera.d_camera), cams[v], 96)
            # untimed: the camera has just jumped to this view, and the launch order follows the measured tile costs with
            # a lag (re-sorted every 32 frames from a running mean): let it settle as it would under a moving camera
            rt.draw(frames=128)
            rt.draw(frames=reps)
            kernel_ms_view[v] = rt.last_kernel_ms()
        avg_ms = sum(kernel_ms_view.values()) / len(kernel_ms_view)
        avg_bytes = sum(per_view[v]["bytes"] for v in VIEW_ORDER) / len(VIEW_ORDER)
        achieved = avg_bytes / (avg_ms * 1e-3) / 1e9
        traffic, traffic_note = hbm_traffic_from_profile(w.brick_dimension)
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                    "traffic": traffic, "traffic_note": traffic_note, "kernel": rt.kernel_name(), "kernel_ms_avg": avg_ms,
                    "kernel_ms_per_view": kernel_ms_view, "algorithmic_bytes_per_launch": avg_bytes}

    if rank == 0:
        total_rays = sum(per_view[view_of(i, args.steps)]["rays"] for i in range(args.steps))
        out = {
            "metric": "Mrays/s at 1920x1080 on 512^3 brickmap; achieved % of HBM roofline",
            "value": total_rays / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": w.name, "frame": f"{w.width}x{w.height}", "grid": f"{w.voxels}^3 voxels, {w.brick_dimension}^3 bricks",
                       "rays": "primary + shadow" if w.sun_enabled else "primary", "spp": w.spp, "max_bounce": w.max_bounce,
                       "views": VIEW_ORDER, "rays_per_frame": {v: per_view[v]["rays"] for v in VIEW_ORDER},
                       "counters_per_frame": {v: per_view[v]["counters"] for v in VIEW_ORDER},
                       "parallelism": (f"image tiles 16x16 interleaved over {world} GPU(s), 1 RCCL gather per launch to rank 0, "
                                       + (f"native pipeline, {args.dist_batch if world > 1 else 1} frame(s) per launch, {args.dist_frames} launches in flight" if native else "torch.distributed gather, frame f overlaps kernel of f+1"))
                       if sharded else f"1 GPU, whole frame, {args.frames_in_flight} frame(s) in flight"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, grid, args.cpu_seconds)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    rt.deinit()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
