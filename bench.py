#!/usr/bin/env python3
"""bench.py — Mrays/s of the brickmap traversal path on MI355X.

  python bench.py --gpus N --steps K --warmup W [--workload NAME]

A step is one full frame of the workload (all pixels, every ray the shader casts for them) at one of the fixed
camera views: the steps are split evenly over V0, V1, V2 (consecutive frames share a view).  The scene is resident
in HBM before the timed region.  N > 1: one process per GPU (torch.distributed, backend nccl = RCCL); the frame is
sharded by interleaved 16x16 tiles and each frame's shards are gathered to rank 0 once.  Launched
without rank environment (`python bench.py --gpus 8`), the script re-executes itself under
`python -m torch.distributed.run --standalone --nproc-per-node N`; it never reports fewer ranks than asked.
Rank 0 prints ONE JSON line (fields: DESIGN.md §6).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VIEW_ORDER = ["V0", "V1", "V2"]          # cycled by the timed region
EXTRA_VIEWS = ["V1x", "VG"]              # reported per view only: outside-the-box view of round 1, all-ground view
SETTLE_FRAMES = 128                      # untimed frames after a camera jump (the tile schedule follows with a lag)


def metric_name(w) -> str:
    rays = "primary + shadow" if w.sun_enabled else "primary"
    if w.max_bounce > 0:
        rays += f" + {w.max_bounce} bounce(s)"
    return (f"Mrays/s at {w.width}x{w.height} on {w.voxels}^3 brickmap ({w.brick_dimension}^3 bricks, {rays}, {w.spp} spp); "
            "achieved % of HBM roofline")


# ------------------------------------------------------------------------------------------------ CPU baselines
def cpu_baseline_port(w, grid, min_wall_s: float, per_view):
    """Oracle (CPU restatement of the reference shader, oracle/vrt_oracle.c) on all host cores of this box.
    Bounded sample: whole frames, views cycled, until `min_wall_s` of wall time (at least one frame per view)."""
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
                now = time.perf_counter() - t0
                # whole frames, one per view at least — unless the workload's frames are so heavy that three of them
                # exceed 3 x the asked time: then the sample ends mid-frame (the rate is rays traced / time either way)
                if (row == 0 and frame >= len(VIEW_ORDER) and now >= min_wall_s) or now >= 3.0 * min_wall_s:
                    state["stop"] = True
                    state["frames"] = frame + row / w.height
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
            "sample": f"{state['frames']:.2f} frames of {w.name} (views {'/'.join(VIEW_ORDER)} cycled, rows in order), {rays} rays in {dt:.2f} s wall "
                      f"= {dt * cores:.0f} core-seconds; oracle/vrt_oracle.c gcc -O2, {cores} threads, rows handed out one at a time"}


def cpu_baseline_reference(w, grid, min_wall_s: float, per_view):
    """The reference's own shader (brick_raytracer.comp) compiled by Mesa and run by llvmpipe on this box's host cores
    (oracle/_ref; north_star's "reference under lavapipe": same gallivm back end, OpenGL instead of Vulkan front end).
    Scene buffers are uploaded once; each frame pushes the 128 constant bytes and dispatches ceil(W/32) x ceil(H/32)
    workgroups, timed to glFinish.  Returns (dict, None) or (None, reason)."""
    try:
        from oracle import oracle as O
        from oracle import ref_gl
        from tests.helpers import oracle_scene_from_grid
        from zig_vulkan_amd import workloads as W
        reason = ref_gl.available()
        if reason is not None:
            return None, reason
        ref = ref_gl.ReferenceShader(w.brick_dimension, want_float=False)
        scene = oracle_scene_from_grid(grid)
        pcs = [O.push_constants(W.camera_for(w, v).blob(), W.sun_for(w).blob()) for v in VIEW_ORDER]
        try:
            ref.bind(scene, pcs[0])
        except ref_gl.GlRefUnavailable as e:   # a storage block above GL_MAX_SHADER_STORAGE_BLOCK_SIZE (128 MiB)
            return None, str(e)
        ref.frame(pcs[0])  # untimed: JIT of the compute variant, page faults
        t0 = time.perf_counter()
        frames = 0
        rays = 0
        while frames < len(VIEW_ORDER) or time.perf_counter() - t0 < min_wall_s:
            v = VIEW_ORDER[frames % len(VIEW_ORDER)]
            ref.frame(pcs[frames % len(VIEW_ORDER)])
            rays += per_view[v]["rays"]
            frames += 1
        dt = time.perf_counter() - t0
        threads = ref.gl.worker_threads()
        info = ref.gl.info()
        ref.unbind()
        return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "reference",
                "sample": f"{frames} whole frames of {w.name} (views {'/'.join(VIEW_ORDER)} cycled), {rays} rays in {dt:.2f} s wall; "
                          f"assets/shaders/brick_raytracer.comp of the reference as Mesa program binary (oracle/_ref), {info}, "
                          f"{threads} llvmpipe worker threads (Mesa's cap) on a {os.cpu_count()}-core host"}, None
    except Exception as e:  # noqa: BLE001 - a baseline must never take the bench down
        return None, f"{type(e).__name__}: {e}"


# ------------------------------------------------------------------------------------------------ PMC (rocprofv3)
PMC_PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"],
              ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_WAVES",
               "SQ_BUSY_CYCLES"]]


def pmc_child(args) -> None:
    """Run under `rocprofv3 --pmc`: the product kernel alone, single stream, a few settled frames per view.
    No torch import (start-up time), nothing printed."""
    from zig_vulkan_amd import workloads as W
    w = W.WORKLOADS[args.workload or W.HEADLINE]
    grid = W.build_grid(w)
    rt = W.make_renderer(w, grid, kernel_variant=args.variant, frames_in_flight=1)
    for v in VIEW_ORDER:
        W.set_view(rt, v)
        rt.draw(frames=args.pmc_frames or 24)
        rt.wait()
    rt.deinit()


def _pmc_read(dirs, kernel_substr: str):
    """kernel_substr: the brick dimension (first template argument of the product kernels)."""
    import glob
    import sqlite3
    out = {}
    for d in dirs:
        for f in sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True)):
            db = sqlite3.connect(f)
            for c, v, n in db.execute("select counter_name, avg(value), count(*) from counters_collection where kernel_name like ? "
                                      "or kernel_name like ? group by counter_name", (f"%vrt_trace_kernel<{kernel_substr}, false%", f"%vrt_path_kernel<{kernel_substr},%")):
                out[c] = v
                out["_dispatches"] = n
    return out


def pmc_live(args, w):
    """HBM bytes and instruction counts per launch of the traversal kernel, measured now: one `rocprofv3 --kernel-trace
    --pmc <group>` pass per counter group over `bench.py --pmc-child` (counters are never combined with other trace
    domains; FETCH_SIZE and WRITE_SIZE do not fit one pass — MI355X_MICROARCH.md).  Returns (counters, note)."""
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="vrt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    dirs = []
    t0 = time.perf_counter()
    try:
        for i, group in enumerate(PMC_PASSES):
            d = os.path.join(tmp, f"pmc{i}")
            cmd = [exe, "--kernel-trace", "--pmc", *group, "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--workload", w.name, "--variant", str(args.variant), "--pmc-frames", str(args.pmc_frames)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=args.pmc_timeout)
            if r.returncode != 0:
                return None, f"rocprofv3 pass {group} exited {r.returncode}: {r.stdout.decode(errors='replace')[-300:]}"
            dirs.append(d)
        c = _pmc_read(dirs, str(w.brick_dimension))   # the product kernel: vrt_trace_kernel<B, false, ...> or vrt_path_kernel<B, ...>
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            return None, "rocprofv3 ran but the traversal kernel's counters were not found"
        return c, (f"measured in this run: {len(PMC_PASSES)} rocprofv3 --pmc passes over {int(c.get('_dispatches', 0))} launches of the product "
                   f"kernel ({args.pmc_frames} frames per view, single stream), {time.perf_counter() - t0:.0f} s")
    except Exception as e:  # noqa: BLE001
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_from_profile(brick_dimension: int):
    """Fallback: the committed PMC passes of the same command (profiles/r*_final_pmc.json, tools/summarize_prof.py)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_final_pmc.json")))
    if not files:
        return None, "no profiles/r*_final_pmc.json"
    with open(files[-1]) as fh:
        data = json.load(fh)
    for name, c in data.items():
        if "vrt_trace_kernel<%d, false" % brick_dimension in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            return c, f"NOT measured in this run: read from the committed {os.path.basename(files[-1])} (an earlier run of this command)"
    return None, "traversal kernel not found in " + os.path.basename(files[-1])


# ------------------------------------------------------------------------------------------------ launch plumbing
def ensure_ranks(args, argv) -> None:
    """`--gpus N` with N > 1 and no rank environment: re-execute under torch.distributed.run with N ranks on this node.
    Exits non-zero instead of running on fewer GPUs than asked."""
    if args.gpus <= 1 or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return
    if not args.stub:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {have} GPU(s): refusing to run on fewer")
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    print(f"[bench] no rank environment: re-executing as {' '.join(cmd)}", file=sys.stderr)
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")))


class _StubRT:
    """TEST ONLY (`--stub`, tests/test_bench_plumbing.py): stands in for the renderer so that the rank plumbing —
    environment, process group, broadcasts, the native/torch agreement, max-over-ranks timing, the one JSON line —
    runs at world size 2 over gloo on a box without GPUs.  Renders nothing."""

    def __init__(self, fail_native: bool):
        import types
        self.fail_native = fail_native
        self.camera = types.SimpleNamespace(d_camera=(bytearray(96)))
        self.frames = 0

    def draw(self, frames: int = 1):
        self.frames += frames
        time.sleep(0.0005 * frames)

    def dist_init(self, *a, **k):
        if self.fail_native:
            raise RuntimeError("stub: native pipeline unavailable on this rank")

    def dist_frame(self):
        self.draw()

    def dist_info(self):
        return {"rank": int(os.environ.get("RANK", "0")), "world": int(os.environ.get("WORLD_SIZE", "1")), "frames_per_launch": 1,
                "launches_in_flight": 1}

    def set_target(self, *a):
        pass

    def assemble_frame(self, *a):
        pass

    def kernel_name(self):
        return "stub"

    def dist_wait(self):
        pass

    wait = deinit = dist_selftest = dist_wait


def percentiles(ms):
    import numpy as np
    a = np.sort(np.asarray(ms, dtype=np.float64))
    return {"median": float(np.percentile(a, 50)), "p10": float(np.percentile(a, 10)), "p90": float(np.percentile(a, 90)),
            "mean": float(a.mean()), "n": int(a.size)}


def main(argv=None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed frames; default: as many as fit about 2.5 s, at most 600, at least 6")
    ap.add_argument("--warmup", type=int, default=None, help="untimed frames before them; default: min(30, steps / 2)")
    ap.add_argument("--workload", default=None)
    ap.add_argument("--variant", type=lambda x: int(x, 0), default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="minimum wall time of each CPU baseline sample")
    ap.add_argument("--frames-in-flight", type=int, default=2, help="1: frames strictly one after another; 2: two frames in flight")
    ap.add_argument("--pmc", choices=["auto", "live", "profile", "off"], default="auto",
                    help="HBM traffic / instruction counters of the roofline object: live = rocprofv3 passes over a child run now; "
                         "profile = the committed profiles/*_pmc.json; auto = live, falling back to profile")
    ap.add_argument("--pmc-frames", type=int, default=None, help="frames per view of a PMC child run; default: up to 24, about 0.6 s")
    ap.add_argument("--pmc-timeout", type=float, default=240.0)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dist", choices=["native", "torch"], default="native",
                    help="N>1 frame gather: native = RCCL send/recv inside libvrt_hip.so (pipelined); torch = torch.distributed.gather (fallback)")
    ap.add_argument("--dist-frames", type=int, default=4, help="launches in flight per rank of the native multi-GPU pipeline")
    ap.add_argument("--dist-batch", type=int, default=8,
                    help="frames traced by one launch and carried by one collective when world > 1 (every frame is gathered once; "
                         "1 = one collective per frame)")
    ap.add_argument("--root-share", type=int, default=-1,
                    help="native multi-GPU pipeline: rank 0's share of the tiles in percent of an equal share; -1: 100 - 70 (world - 2) / 6 "
                         "(100 % at 2 ranks ... 30 % at 8: one-GPU emulation of root and peers, tools/root_share_sweep.sh)")
    ap.add_argument("--force-gather", action="store_true", help="run the shard/gather/assemble path even at world size 1")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)            # tests/test_bench_plumbing.py
    ap.add_argument("--stub-fail-native-on", type=int, default=-1, help=argparse.SUPPRESS)
    args = ap.parse_args(argv)

    if args.pmc_child:
        pmc_child(args)
        return
    ensure_ranks(args, argv)

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio (flushed at exit, i.e.
    # after anything Python printed), so everything else written to fd 1 by any library is sent to stderr and the
    # JSON line goes to a private copy of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    stub = args.stub
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the traversal path has no CPU implementation")
    if not stub:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cpu") if stub else torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stub:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    def sync() -> None:
        if not stub:
            torch.cuda.synchronize()

    # every rank announces itself: rank 0 reports who took part
    ranks_seen = [rank]
    if use_dist and world > 1:
        t = torch.zeros(world, dtype=torch.int32, device=dev)
        t[rank] = 1
        dist.all_reduce(t)
        ranks_seen = [i for i in range(world) if int(t[i].item()) == 1]
        if len(ranks_seen) != world:
            raise SystemExit(f"bench.py: only ranks {ranks_seen} of {world} answered")

    from zig_vulkan_amd import workloads as W
    w = W.WORKLOADS[args.workload or W.HEADLINE]
    grid = None if stub else W.build_grid(w)
    sharded = world > 1 or args.force_gather
    stream = 0 if stub else torch.cuda.current_stream().cuda_stream
    all_views = VIEW_ORDER + EXTRA_VIEWS

    # ---- rays and bytes per view, counted by counting builds of the kernel (untimed) ----
    per_view = {}
    if rank == 0 and not stub:
        pixels = w.width * w.height
        for mode, key in ((1, "counters"), (2, "issued")):
            rtc = W.make_renderer(w, grid, enable_counters=mode, device_id=local_rank, kernel_variant=args.variant)
            for v in all_views:
                W.set_view(rtc, v)
                rtc.draw()
                c = rtc.counters()
                pv = per_view.setdefault(v, {})
                pv[key] = c
                if mode == 1:
                    # the loads of the REFERENCE algorithm (per-lane word cache, walk to the grid's face)
                    pv["rays"] = c["rays"]
                    pv["bytes"] = 4 * c["status_loads"] + 4 * c["bricks_entered"] + c["voxel_steps"] + 25 * c["hits"] + 4 * pixels
                    primaries = pixels * w.spp
                    if w.max_bounce == 0:
                        ph = (c["rays"] - primaries) if w.sun_enabled else c["hits"]
                        pv["primary_hit_fraction"] = ph / primaries
                    else:
                        pv["primary_hit_fraction"] = None
                else:
                    # what the PRODUCT kernel requests: a status dword per brick-level trip of a walk that ends at
                    # the occupied-cell box; per brick entered the index, the start index and the first occupancy dword; an
                    # occupancy dword per voxel trip; per hit the material id and the 20-byte material; the pixel store
                    pv["issued_bytes"] = (4 * c["grid_steps"] + 12 * c["bricks_entered"] + 4 * c["voxel_steps"] + 21 * c["hits"] + 4 * pixels)
            rtc.deinit()
    elif stub:
        per_view = {v: {"rays": 1000, "bytes": 4000, "issued_bytes": 2000, "counters": {}, "issued": {}, "primary_hit_fraction": 0.5}
                    for v in all_views}
    if use_dist and world > 1:
        obj = [per_view]
        dist.broadcast_object_list(obj, src=0)
        per_view = obj[0]

    # ---- the timed renderer ----
    fg = None
    native = False
    rt = None
    dist_info = None
    if sharded and args.dist == "native":
        # RCCL inside libvrt_hip.so: kernel -> grouped send/recv to rank 0 -> un-swizzle, several launches in flight
        ok = 1
        try:
            root_share = args.root_share if args.root_share >= 0 else max(30, 100 - (70 * max(0, world - 2) + 3) // 6)
            if stub:
                uid = [b"stub"]
                rt = _StubRT(fail_native=(rank == args.stub_fail_native_on))
            else:
                from zig_vulkan_amd import VoxelRT
                uid = [VoxelRT.dist_unique_id() if rank == 0 else None]
            if use_dist and world > 1:
                dist.broadcast_object_list(uid, src=0)
            if not stub:
                rt = W.make_renderer(w, grid, device_id=local_rank, shard_rank=rank, shard_count=world, kernel_variant=args.variant,
                                     shard_root_weight=(root_share if 2 <= world <= 8 else 0))
            rt.dist_init(uid[0], rank, world, args.dist_frames, frames_per_launch=(args.dist_batch if world > 1 else 1))
            if world == 1:
                rt.dist_selftest()
            dist_info = rt.dist_info()
        except Exception as e:  # noqa: BLE001 - any failure means "use the torch path"
            print(f"[bench rank {rank}] native RCCL pipeline unavailable: {e}", file=sys.stderr)
            ok = 0
        if use_dist and world > 1:  # every rank must take the same path
            t = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok = int(t.item())
        native = bool(ok)
        if not native and rt is not None:
            rt.deinit()
            rt = None
    if sharded and not native:
        from zig_vulkan_amd.dist import FrameGather
        fg = FrameGather(w.width, w.height, rank, world, dev)
        if stub:
            rt = _StubRT(False)
        else:
            rt = W.make_renderer(w, grid, device_id=local_rank, shard_rank=rank, shard_count=world, stream=stream,
                                 external_target_rgba8=fg.shard.data_ptr(), kernel_variant=args.variant)
    elif not sharded:
        rt = _StubRT(False) if stub else W.make_renderer(w, grid, device_id=local_rank, kernel_variant=args.variant,
                                                         frames_in_flight=args.frames_in_flight)
    # the communicator's own idea of its size, from every rank
    rccl_world = None
    if native and dist_info is not None:
        rccl_world = dist_info["world"]
        if use_dist and world > 1:
            t = torch.tensor([dist_info["world"]], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            rccl_world = int(t.item())
        if rccl_world != world:
            raise SystemExit(f"bench.py: the RCCL communicator spans {rccl_world} ranks, {world} were launched")
    if native:
        rt.dist_wait()
    else:
        rt.wait()

    import ctypes as C
    cams = {}
    if not stub:
        for v in all_views:
            W.set_view(rt, v)
            cams[v] = bytes(rt.camera.d_camera)

    def set_cam(v: str) -> None:
        if not stub:
            C.memmove(C.byref(rt.camera.d_camera), cams[v], 96)

    def view_of(i: int, n: int) -> str:
        # frames of one view are consecutive (a camera moves smoothly; the tile schedule feeds on the previous frames)
        return VIEW_ORDER[min(len(VIEW_ORDER) - 1, (i * len(VIEW_ORDER)) // max(n, 1))]

    frame_no = [0]  # frames submitted so far (warm-up included): the gather pipeline's slot counter

    def step(i: int, n: int) -> None:
        set_cam(view_of(i, n))
        if not sharded:
            rt.draw()
            return
        if native:
            rt.dist_frame()                     # kernel -> RCCL gather -> un-swizzle, on this launch's stream
            return
        f = frame_no[0]
        frame_no[0] += 1
        fg.begin_frame(f)                       # buffer f%2 is free once frame f-2's gather is done
        rt.set_target(fg.shard_for(f).data_ptr())
        rt.draw()                               # this rank's tiles of frame f
        fg.gather_async(f)                      # ONE collective per frame, overlapped with frame f+1's kernel
        if f >= 1:
            fg.complete(f - 1, None if stub else rt)   # rank 0: un-swizzle the previous frame

    def drain() -> None:
        if native:
            rt.dist_wait()
        elif sharded and frame_no[0] >= 1:
            fg.complete(frame_no[0] - 1, None if stub else rt)

    def barrier() -> None:
        if use_dist and world > 1:
            dist.barrier()
        sync()

    def timed_region(n: int) -> float:
        barrier()
        t0 = time.perf_counter()
        for i in range(n):
            step(i, n)
        drain()  # the last frame's gather + un-swizzle belong to the timed region
        if not native:
            rt.wait()
        barrier()
        dt = time.perf_counter() - t0
        if use_dist and world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # a first look at the frame time (two untimed frames): sizes the run when --steps / --warmup were left to the script
    # and the settling / PMC legs for workloads whose frames take milliseconds instead of microseconds
    barrier()
    t0 = time.perf_counter()
    for i in range(2):
        step(2, 3)
    drain()
    if not native:
        rt.wait()
    sync()
    frame_ms_est = max((time.perf_counter() - t0) / 2 * 1e3, 1e-3)
    if use_dist and world > 1:
        t = torch.tensor([frame_ms_est], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)   # every rank must choose the same step count
        frame_ms_est = float(t.item())
    if args.steps is None:
        args.steps = int(min(600, max(6, 2500.0 / frame_ms_est)))
    if args.warmup is None:
        args.warmup = min(30, args.steps // 2)
    if args.pmc_frames is None:
        args.pmc_frames = int(min(24, max(2, 600.0 / frame_ms_est)))
    settle_frames = int(min(SETTLE_FRAMES, max(3, 1500.0 / frame_ms_est)))

    for i in range(args.warmup):
        step(i, args.warmup)
    drain()
    dt = timed_region(args.steps)

    # ---- N = 1: the same steps strictly one frame after another, and the dominant kernel by HIP events ----
    roofline = None
    single = None
    if rank == 0 and not sharded and not stub:
        rt1 = rt
        if args.frames_in_flight != 1:
            rt.deinit()
            rt1 = rt = W.make_renderer(w, grid, device_id=local_rank, kernel_variant=args.variant, frames_in_flight=1)
        for i in range(args.warmup):
            step(i, args.warmup)
        single = timed_region(args.steps) / args.steps * 1e3
        reps = max(8, args.steps // len(VIEW_ORDER))
        kernel_ms_view, frame_stats = {}, {}
        for v in all_views:
            set_cam(v)
            # untimed: the camera has just jumped to this view, and the launch order follows the measured tile costs with
            # a lag (re-sorted every 32 frames from a running mean): let it settle as it would under a moving camera
            rt1.draw(frames=settle_frames)
            rt1.draw(frames=reps)                           # back to back, one event pair around all of them
            kernel_ms_view[v] = rt1.last_kernel_ms()
            frame_stats[v] = percentiles(rt1.draw_timed(min(reps, 512)))   # an event pair around every frame
        avg_ms = sum(kernel_ms_view[v] for v in VIEW_ORDER) / len(VIEW_ORDER)
        avg_bytes = sum(per_view[v]["bytes"] for v in VIEW_ORDER) / len(VIEW_ORDER)
        avg_issued = sum(per_view[v]["issued_bytes"] for v in VIEW_ORDER) / len(VIEW_ORDER)
        achieved = avg_bytes / (avg_ms * 1e-3) / 1e9
        pmc, pmc_note = (None, "--pmc off")
        if args.pmc in ("auto", "live"):
            rt1.wait()
            pmc, pmc_note = pmc_live(args, w)
        if pmc is None and args.pmc in ("auto", "profile"):
            why = pmc_note
            pmc, pmc_note = pmc_from_profile(w.brick_dimension)
            if args.pmc == "auto":
                pmc_note = f"{pmc_note} (live measurement failed: {why})"
        traffic = issue_ipc = None
        insts = None
        if pmc is not None:
            # FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE on gfx950 tallies 64 B per 128-byte line fetched
            # (MI355X_MICROARCH.md; tools/ubench/fetch_calib.hip confirms it for scattered dword reads): doubled
            traffic = 2.0 * pmc["FETCH_SIZE"] * 1024.0 + pmc["WRITE_SIZE"] * 1024.0
            keys = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_LDS")
            if all(k in pmc for k in keys[:3]):
                insts = {k: pmc.get(k, 0.0) for k in keys}
                from zig_vulkan_amd import _lib as VL
                di = (C.c_int64 * 4)()
                VL.check(VL.lib.vrt_device_info(local_rank, di))
                clock_hz = di[0] * 1e3
                simds = int(di[1]) * 4
                issue_ipc = sum(insts.values()) / (simds * avg_ms * 1e-3 * clock_hz)
        roofline = {
            "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_note": pmc_note,
            "definition": "achieved = bytes the REFERENCE algorithm loads for these frames (4 S + 4 K + V + 25 H per ray + 4 B per pixel, "
                          "SURVEY.md 8(d), counted by the counting build that walks to the grid's face like the shader) / kernel time: a rate "
                          "of useful work, not of bytes moved — the product kernel never asks for the status words of cells it knows to "
                          "be empty (it jumps to the near face of the occupied-cell box and stops at its far face), so frac can exceed 1 on "
                          "views from outside the box.  issued_bytes = what the product kernel's lanes request (4 B per brick-level trip "
                          "taken, 12 per brick entered, 4 per voxel trip, 21 per hit, 4 per pixel). "
                          "traffic = HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE): the touched scene data lives in L2 / MALL.",
            "kernel": rt1.kernel_name(), "settle_frames": settle_frames, "kernel_ms_avg": avg_ms, "kernel_ms_per_view": kernel_ms_view,
            "frame_ms_percentiles_per_view": frame_stats,
            "algorithmic_bytes_per_launch": avg_bytes,
            "issued_bytes": avg_issued, "issued_GBps": avg_issued / (avg_ms * 1e-3) / 1e9,
            "issue_ipc": issue_ipc,
            "issue_ipc_note": "wave-instructions per launch (SQ_INSTS_VALU + SALU + VMEM + SMEM + LDS) / (1024 SIMDs x kernel_ms_avg x "
                              "the device's engine clock): the kernel is bound by instruction issue, not by HBM",
            "insts_per_launch": insts, "clock_hz": clock_hz if insts else None,
        }

    if rank == 0:
        total_rays = sum(per_view[view_of(i, args.steps)]["rays"] for i in range(args.steps))
        par = (f"1 GPU, whole frame, {args.frames_in_flight} frame(s) in flight" if not sharded else
               f"image tiles 16x16 interleaved over {world} GPU(s), every frame gathered once to rank 0, "
               + (f"native RCCL pipeline: {args.dist_batch if world > 1 else 1} frame(s) per launch and per collective, {args.dist_frames} launches in flight"
                  if native else "torch.distributed gather per frame, frame f overlaps the kernel of f+1"))
        out = {
            "metric": metric_name(w),
            "value": total_rays / dt / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_single_stream": single,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "ranks_seen": ranks_seen,
            "rccl_world": rccl_world,
            "dist_path": ("native" if native else "torch") if sharded else None,
            "config": {"workload": w.name, "frame": f"{w.width}x{w.height}", "grid": f"{w.voxels}^3 voxels, {w.brick_dimension}^3 bricks",
                       "rays": "primary + shadow" if w.sun_enabled else "primary", "spp": w.spp, "max_bounce": w.max_bounce,
                       "views": VIEW_ORDER, "views_reported_only": EXTRA_VIEWS,
                       "rays_per_frame": {v: per_view[v]["rays"] for v in all_views},
                       "primary_hit_fraction": {v: per_view[v]["primary_hit_fraction"] for v in all_views},
                       "counters_per_frame": {v: per_view[v]["counters"] for v in all_views},
                       "issued_counters_per_frame": {v: per_view[v]["issued"] for v in all_views},
                       "parallelism": par},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline and not stub:
            port = cpu_baseline_port(w, grid, args.cpu_seconds, per_view)
            ref, why = cpu_baseline_reference(w, grid, args.cpu_seconds, per_view)
            if ref is not None:
                out["cpu_baseline"] = ref
                out["cpu_baseline_port"] = port
            else:
                port["reference_unavailable"] = why
                out["cpu_baseline"] = port
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if rt is not None:
        rt.deinit()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
