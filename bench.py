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
PRECONDITION_MS = 150.0                  # `value_sustained` only: untimed GPU work that brings the clocks to their sustained state
WALL_BUDGET_S = 900.0                    # N > 1: the whole run's wall-clock budget; the secondary leg is skipped when it would not fit


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


def cpu_baseline_reference(w, grid, min_wall_s: float, per_view, hip_frames=None):
    """The reference's own shader (brick_raytracer.comp) compiled by Mesa and run by llvmpipe on this box's host cores
    (oracle/_ref; north_star's "reference under lavapipe": same gallivm back end, OpenGL instead of Vulkan front end).
    Scene buffers are uploaded once; each frame pushes the 128 constant bytes and dispatches ceil(W/32) x ceil(H/32)
    workgroups, timed to glFinish.  Returns (dict, parity, None) or (None, None, reason).
    hip_frames: {"product": {view: (f32, u8)}, "fused": {...}} — frames libvrt_hip rendered for the same 128 push-constant
    bytes; one reference frame per view is then read back (untimed) and compared pixel by pixel: the on-box parity
    certificate (`parity_vs_reference`)."""
    try:
        from oracle import oracle as O
        from oracle import ref_gl
        from tests.helpers import oracle_scene_from_grid
        from zig_vulkan_amd import workloads as W
        reason = ref_gl.available()
        if reason is not None:
            return None, reason
        ref = ref_gl.ReferenceShader(w.brick_dimension, want_float=bool(hip_frames))
        scene = oracle_scene_from_grid(grid)
        pcs = [O.push_constants(W.camera_for(w, v).blob(), W.sun_for(w).blob()) for v in VIEW_ORDER]
        try:
            ref.bind(scene, pcs[0])
        except ref_gl.GlRefUnavailable as e:   # a storage block above GL_MAX_SHADER_STORAGE_BLOCK_SIZE (128 MiB)
            return None, None, str(e)
        ref.frame(pcs[0])  # untimed: JIT of the compute variant, page faults
        parity = None
        if hip_frames:
            import numpy as np
            parity = {"against": "frames of assets/shaders/brick_raytracer.comp (reference) run under Mesa llvmpipe on this box, same 128 push-constant "
                                 "bytes, same scene buffers; float colour (rgba32f build of the shader) and its Rgba8 image",
                      "views": list(VIEW_ORDER), "pixels_per_view": w.width * w.height, "tolerance": 1e-4}
            for build in hip_frames:
                parity[build] = {"pixels_over_1e-4": {}, "max_abs_err": {}, "pixels_not_bit_equal": {}, "rgba8_pixels_differing": {}}
            for i, v in enumerate(VIEW_ORDER):
                rf = ref.frame(pcs[i], read=True, want_float=True)     # untimed
                ru = ref.frame(pcs[i], read=True)
                for build, frames in hip_frames.items():
                    hf, hu = frames[v]
                    d = np.abs(hf[:, :, :3] - rf[:, :, :3]).max(axis=2)
                    parity[build]["pixels_over_1e-4"][v] = int((d > 1e-4).sum())
                    parity[build]["max_abs_err"][v] = float(d.max())
                    parity[build]["pixels_not_bit_equal"][v] = int((hf.view(np.uint32) != rf.view(np.uint32)).any(axis=2).sum())
                    parity[build]["rgba8_pixels_differing"][v] = int((hu != ru).any(axis=2).sum())
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
                          f"{threads} llvmpipe worker threads (Mesa's cap) on a {os.cpu_count()}-core host"}, parity, None
    except Exception as e:  # noqa: BLE001 - a baseline must never take the bench down
        return None, None, f"{type(e).__name__}: {e}"


def hbm_achievable(dev):
    """What this box's HBM delivers to plain streaming kernels ("report that achievable figure too"): a
    device-to-device copy (read + write) and a read-only reduction over 2 GiB buffers, far beyond L2 + MALL, by events."""
    import torch
    n = 2 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev).fill_(1)
    b = torch.empty_like(a)

    def timed(fn, reps=6):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3

    copy = 2 * n / timed(lambda: b.copy_(a)) / 1e9
    af = a.view(torch.float32)
    read = n / timed(lambda: af.sum()) / 1e9
    del a, b, af
    torch.cuda.empty_cache()
    return {"copy_read_plus_write_GBps": copy, "read_only_GBps": read, "method": "torch copy_ / sum over 2 GiB, 6 repetitions, torch.cuda.Event"}


# ------------------------------------------------------------------------------------------------ PMC (rocprofv3)
PMC_PASSES = [["FETCH_SIZE"], ["WRITE_SIZE"],
              ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH",
               "SQ_INSTS"],
              # lane utilisation of the vector instructions: thread-cycles / (64 x instruction-cycles)
              ["SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"]]
PMC_OPTIONAL_PASSES = 1   # (the last pass may fail without taking the HBM / instruction counters down)


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
                                      "or kernel_name like ? or kernel_name like ? group by counter_name",
                                      (f"%vrt_trace_kernel<{kernel_substr}, false%", f"%vrt_path_kernel<{kernel_substr},%", f"%vrt_pool_kernel<{kernel_substr},%")):
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
                if i >= len(PMC_PASSES) - PMC_OPTIONAL_PASSES:
                    continue
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


# ------------------------------------------------------------------------------------------------ N > 1: probe of the native pipeline
PROBE_FRAMES = ["V0", "V1", "V2", "V1", "V2", "V0", "V0", "V2", "V1", "V1", "V0", "V2", "V2", "V0", "V1", "V2"]


def dist_probe_child(args) -> None:
    """`bench.py --dist-probe`: one rank of a short run of the native RCCL pipeline, in a process of its own: a 332 x 210 frame with
    batches of 1 and of 8 frames per collective, then (round 5) a small path trace whose bounce frames the persistent kernels trace
    inside the pipeline — each on its own communicator.  Rank 0 compares every assembled frame with the frame one context renders
    alone.  Exit code 0 = this rank got through.  No torch import; nothing on stdout."""
    import numpy as np
    from zig_vulkan_amd import VoxelRT
    from zig_vulkan_amd import workloads as W
    uids = bytes.fromhex(args.probe_uid)
    rank, world, device = args.probe_rank, args.probe_world, args.probe_device
    plain_w = W.Workload("probe", 332, 210, 64, 4, 1, 0, True, 0.0)
    # (workload, kernel_variant, frames per collective, launch slots, which of the three ids, keep the context alive for the next case)
    # Round 6: the cases go through the communicator pool the way the timed legs do — one id for several contexts
    # (vrt_dist_keep_communicators), a second context made while the first still holds its communicators (the root-share tune keeps three
    # alive), a third that takes everything from the pool — so that a hang in that machinery is this child's timeout, not the run's.
    # (few launch slots per case: every communicator is a collective to make, and the child runs under the parent's time limit)
    cases = [(plain_w, 0, 1, 4, 0, True), (plain_w, 0, 8, 4, 0, False), (plain_w, 0, 1, 4, 0, False),
             (W.Workload("probe_bounce", 330, 210, 256, 8, 2, 2, True, 5.0, "sparse", 0.08, 30000), 1 << 23, 1, 2, 1, False)]
    VoxelRT.dist_keep_communicators(True)
    grids, held = {}, []
    for i, (w, variant, batch, slots, which, hold) in enumerate(cases):
        grid = grids.setdefault(w.name, W.build_grid(w))
        ref = None
        if rank == 0:
            plain = W.make_renderer(w, grid, device_id=device, kernel_variant=variant)
            W.set_view(plain, PROBE_FRAMES[-1])
            plain.draw()
            ref = plain.read_rgba8().copy()
            plain.deinit()
        rt = W.make_renderer(w, grid, device_id=device, shard_rank=rank, shard_count=world, kernel_variant=variant)
        rt.dist_init(uids[128 * which:128 * (which + 1)], rank, world, slots, frames_per_launch=batch, communicators=args.dist_comms)
        for v in PROBE_FRAMES:
            W.set_view(rt, v)
            rt.dist_frame()
        rt.dist_wait()
        if rank == 0 and not np.array_equal(rt.dist_read_frame(), ref):
            print(f"[bench probe] case {i} ({w.name}, batch {batch}): the assembled frame differs from the single-context frame", file=sys.stderr)
            sys.exit(3)
        if hold:
            held.append(rt)
        else:
            rt.deinit()
            for h in held:
                h.deinit()
            held = []
    VoxelRT.dist_release_communicators()
    sys.exit(0)


def native_probe(env, timeout: float):
    """Before any timed leg at N > 1: every rank runs dist_probe_child in a child process (the children form their own
    communicators on the same GPUs).  A hang inside a collective cannot be recovered from in-process; in a child it is a
    timeout, the child is killed, and every rank takes the torch.distributed path instead.  Returns (ok on every rank, report)."""
    uids = None
    if env.rank == 0:
        try:
            from zig_vulkan_amd import VoxelRT
            uids = VoxelRT.dist_unique_id() + VoxelRT.dist_unique_id() + VoxelRT.dist_unique_id()
        except Exception as e:  # noqa: BLE001 - the other ranks wait in the broadcast below: it must happen either way
            print(f"[bench rank 0] no RCCL unique id for the probe: {type(e).__name__}: {e}", file=sys.stderr)
    uids = env.bcast(uids)
    if uids is None:
        return False, {"ok": False, "seconds": 0.0, "this_rank": "rank 0 could not make a RCCL unique id"}
    cmd = [sys.executable, os.path.abspath(__file__), "--dist-probe", "--probe-uid", uids.hex(), "--probe-rank", str(env.rank),
           "--probe-world", str(env.world), "--probe-device", str(env.local_rank), "--dist-comms", str(getattr(getattr(env, "args", None), "dist_comms", 0))]
    child_env = {k: v for k, v in os.environ.items()
                 if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT")
                 and not k.startswith("TORCHELASTIC_")}
    t0 = time.perf_counter()
    why = ""
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=child_env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout)
        ok = int(r.returncode == 0)
        if not ok:
            why = f"exit code {r.returncode}: {r.stderr.decode(errors='replace')[-300:]}"
    except subprocess.TimeoutExpired:   # (subprocess.run has killed the child)
        ok, why = 0, f"no answer within {timeout:.0f} s"
    except Exception as e:  # noqa: BLE001
        ok, why = 0, f"{type(e).__name__}: {e}"
    if not ok:
        print(f"[bench rank {env.rank}] probe of the native RCCL pipeline failed: {why}", file=sys.stderr)
    all_ok = bool(env.all_min_int(ok))
    return all_ok, {"ok": all_ok, "seconds": time.perf_counter() - t0, "this_rank": "ok" if ok else why}


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
    environment, process group, broadcasts, the native/torch agreement, the root-share auto-tune and its agreement across
    ranks, the two timed legs, max-over-ranks timing, the one JSON line — runs at world size 2 over gloo on a box without
    GPUs.  Renders nothing; a frame "takes" a time that depends on the root share so that the tuner has something to choose."""

    def __init__(self, fail_native: bool, root_share: int = 100, batch: int = 1):
        import types
        self.fail_native = fail_native
        self.camera = types.SimpleNamespace(d_camera=(bytearray(96)))
        self.frames = 0
        self.batch = batch
        self.frame_s = 0.0004 * (1.0 + 4.0 * abs(root_share - 60) / 100.0) / (1.0 + 0.25 * (batch > 1))   # 2.6 x between the candidates: clear of a loaded host's sleep jitter

    def draw(self, frames: int = 1):
        self.frames += frames
        time.sleep(self.frame_s * frames)

    def dist_init(self, *a, **k):
        if self.fail_native:
            raise RuntimeError("stub: native pipeline unavailable on this rank")

    def dist_frame(self):
        self.draw()

    def dist_frames(self, cameras, sun=None):
        self.draw(frames=len(cameras))

    def reserve_samples(self, *a):
        pass

    def dist_info(self):
        return {"rank": int(os.environ.get("RANK", "0")), "world": int(os.environ.get("WORLD_SIZE", "1")), "frames_per_launch": self.batch,
                "launches_in_flight": 1}

    def dist_comm_info(self):
        return {"communicators": 1, "made_by_this_init": 1, "library_splits": False, "agreed_by_all_reduce": False}

    def dist_stats(self):
        return {"launches_sampled": 4, "frames_sampled": 4 * self.batch, "kernel_ms_per_launch": self.frame_s * 1e3 * self.batch,
                "collective_ms_per_launch": 0.01, "unswizzle_ms_per_launch": 0.005, "owned_tiles": 10, "shard_bytes_per_frame": 7680,
                "frames_per_launch": self.batch}

    def set_target(self, *a):
        pass

    def assemble_frame(self, *a):
        pass

    def dist_profile(self, *a):
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


def default_root_share(world: int) -> int:
    """Rank 0's share of the tiles in percent of an equal share, before tuning: 100 at 2 ranks ... 30 at 8 (the root also takes
    in every other rank's shards and un-swizzles every frame; one-GPU emulation of both sides, tools/experiments/root_share_sweep.sh)."""
    return max(30, 100 - (70 * max(0, world - 2) + 3) // 6)


def root_share_candidates(world: int):
    """The three shares the warm-up auto-tune times (two at 2 ranks): the emulated default, 0.6 x it and 1.5 x it."""
    t = default_root_share(world)
    return sorted({t, max(10, int(round(t * 0.6))), min(100, int(round(t * 1.5)))})


def count_rays(W, w, grid, views, variant, device_id):
    """Rays and bytes per view by the counting builds of the kernel (untimed): mode 1 the reference algorithm's loads, mode 2
    what the product kernel requests."""
    per_view = {}
    pixels = w.width * w.height
    for mode, key in ((1, "counters"), (2, "issued")):
        rtc = W.make_renderer(w, grid, enable_counters=mode, device_id=device_id, kernel_variant=variant)
        for v in views:
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
    return per_view


def view_of(i: int, n: int) -> str:
    # frames of one view are consecutive (a camera moves smoothly; the tile schedule feeds on the previous frames)
    return VIEW_ORDER[min(len(VIEW_ORDER) - 1, (i * len(VIEW_ORDER)) // max(n, 1))]


class Env:
    """What every leg needs to know about this process."""

    def __init__(self, args, torch, dist, world, rank, local_rank, dev, stub, use_dist):
        self.args, self.torch, self.dist = args, torch, dist
        self.world, self.rank, self.local_rank, self.dev, self.stub, self.use_dist = world, rank, local_rank, dev, stub, use_dist
        self.multi = use_dist and world > 1
        self._uid = None       # the ONE RCCL unique id of this run: every context's communicators come from the library's pool under it

    def shared_uid(self):
        """Rank 0 makes the id once, every rank gets it by one broadcast; later contexts pass the same id, and with
        vrt_dist_keep_communicators(1) the library hands them the communicators earlier contexts made (round 6: a run makes a dozen
        contexts — one per root-share candidate and leg — and a communicator set costs a collective of the order of a second)."""
        if self._uid is None:
            uid = None
            if self.rank == 0:
                try:
                    if self.stub:
                        uid = b"stub"
                    else:
                        from zig_vulkan_amd import VoxelRT
                        VoxelRT.dist_keep_communicators(True)
                        uid = VoxelRT.dist_unique_id()
                except Exception as e:  # noqa: BLE001
                    print(f"[bench rank 0] no RCCL unique id: {type(e).__name__}: {e}", file=sys.stderr)
            elif not self.stub:
                try:
                    from zig_vulkan_amd import VoxelRT
                    VoxelRT.dist_keep_communicators(True)
                except Exception as e:  # noqa: BLE001
                    print(f"[bench rank {self.rank}] vrt_dist_keep_communicators: {type(e).__name__}: {e}", file=sys.stderr)
            self._uid = (self.bcast(uid),)    # (a tuple: None — rank 0 failed — is an answer too, and is not asked for again)
        return self._uid[0]

    def sync(self) -> None:
        if not self.stub:
            self.torch.cuda.synchronize()

    def barrier(self) -> None:
        if self.multi:
            self.dist.barrier()
        self.sync()

    def all_max(self, x: float) -> float:
        if not self.multi:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def all_min_int(self, x: int) -> int:
        if not self.multi:
            return x
        t = self.torch.tensor([x], dtype=self.torch.int32, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
        return int(t.item())

    def bcast(self, obj):
        if not self.multi:
            return obj
        box = [obj]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def gather_objects(self, obj):
        if not self.multi:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out


class Leg:
    """One renderer configuration that can be stepped and timed: the single-GPU context, or one (frames per collective, root
    share) setting of the multi-GPU pipeline (native RCCL inside libvrt_hip.so, or the torch.distributed fallback)."""

    def __init__(self, env: Env, W, w, grid, *, sharded: bool, want_native: bool, batch: int = 1, root_share: int = 100, frames_in_flight: int = 2):
        import ctypes as C
        self.env, self.W, self.w, self.sharded, self.batch, self.root_share = env, W, w, sharded, batch, root_share
        self.native, self.fg, self.rt, self.dist_info, self.rccl_world = False, None, None, None, None
        self.comm_info = None
        self.launches_in_flight = None
        self.frame_no = 0
        args, stub, rank, world = env.args, env.stub, env.rank, env.world
        if sharded and want_native:
            # RCCL inside libvrt_hip.so: kernel -> grouped send/recv to rank 0 -> un-swizzle, several launches in flight
            ok = 1
            # (rank 0 makes the id inside its own guard and every rank executes the broadcast whatever happened: a rank that skipped
            # it would sit in all_min_int's all-reduce while the others sit in the broadcast — ADVICE r03)
            uid = env.shared_uid()
            try:
                if uid is None:
                    raise RuntimeError("rank 0 could not make a RCCL unique id")
                if stub:
                    self.rt = _StubRT(fail_native=(rank == args.stub_fail_native_on), root_share=root_share, batch=batch)
                if not stub:
                    self.rt = W.make_renderer(w, grid, device_id=env.local_rank, shard_rank=rank, shard_count=world, kernel_variant=args.variant,
                                              shard_root_weight=(root_share if (2 <= world <= 8 and root_share < 100) else 0))
                # (one frame per launch: a rank's 1/R of the tiles is a small kernel, and the more of them are in flight the better they
                # overlap — rank 1 of 8 over the RCCL stand-in: 40.6 us per frame with 4 launches in flight, 30.9 with 8, 26.0 with 16 on the runtime's default 4
                # hardware queues; 21.2 / 22.8 / 16.2 on 24 queues, tools/experiments/literal_leg_probe.py)
                self.launches_in_flight = args.dist_frames if (batch > 1 or world == 1) else max(args.dist_frames, min(16, args.literal_launches))
                self.rt.dist_init(uid, rank, world, self.launches_in_flight, frames_per_launch=(batch if world > 1 else 1), communicators=args.dist_comms)
                if world == 1:
                    self.rt.dist_selftest()
                self.dist_info = self.rt.dist_info()
                self.comm_info = self.rt.dist_comm_info()
            except Exception as e:  # noqa: BLE001 - any failure means "use the torch path"
                print(f"[bench rank {rank}] native RCCL pipeline unavailable: {e}", file=sys.stderr)
                ok = 0
            ok = env.all_min_int(ok)          # every rank must take the same path
            self.native = bool(ok)
            if not self.native and self.rt is not None:
                self.rt.deinit()
                self.rt = None
        if sharded and not self.native:
            from zig_vulkan_amd.dist import FrameGather
            self.batch, self.root_share = 1, 100
            self.fg = FrameGather(w.width, w.height, rank, world, env.dev)
            if stub:
                self.rt = _StubRT(False)
            else:
                stream = env.torch.cuda.current_stream().cuda_stream
                self.rt = W.make_renderer(w, grid, device_id=env.local_rank, shard_rank=rank, shard_count=world, stream=stream,
                                          external_target_rgba8=self.fg.shard.data_ptr(), kernel_variant=args.variant)
        elif not sharded:
            self.rt = _StubRT(False) if stub else W.make_renderer(w, grid, device_id=env.local_rank, kernel_variant=args.variant,
                                                                  frames_in_flight=frames_in_flight)
        # the communicator's own idea of its size, from every rank
        if self.native and self.dist_info is not None:
            self.rccl_world = env.all_min_int(self.dist_info["world"])
            if self.rccl_world != world:
                raise SystemExit(f"bench.py: the RCCL communicator spans {self.rccl_world} ranks, {world} were launched")
        if self.native:
            self.rt.dist_wait()
        else:
            self.rt.wait()
        self.cams = {}
        if not stub:
            for v in VIEW_ORDER + EXTRA_VIEWS:
                W.set_view(self.rt, v)
                self.cams[v] = bytes(self.rt.camera.d_camera)
        self._C = C
        self._cam_arrays = {}

    def set_cam(self, v: str) -> None:
        if not self.env.stub:
            self._C.memmove(self._C.byref(self.rt.camera.d_camera), self.cams[v], 96)

    def cam_array(self, v: str, count: int):
        """`count` copies of view v's Camera.Device: the argument of one vrt_dist_frames call."""
        key = (v, count)
        if key not in self._cam_arrays:
            if self.env.stub:
                self._cam_arrays[key] = [None] * count
            else:
                from zig_vulkan_amd import _lib as VL
                arr = (VL.CameraDevice * count)()
                for i in range(count):
                    self._C.memmove(self._C.byref(arr[i]), self.cams[v], 96)
                self._cam_arrays[key] = arr
        return self._cam_arrays[key]

    def steps(self, n: int) -> None:
        """Frames 0 .. n-1 of a run of n.  The native pipeline takes the consecutive frames of one view by ONE call across the ABI
        (vrt_dist_frames, --dist-submit call: what a compiled host's frame loop costs — submitted frame by frame from Python the host
        took 14.6 us of a 30 us frame, tools/experiments/dist_host_probe.py); everything else steps frame by frame."""
        if self.sharded and self.native and self.env.args.dist_submit == "call":
            i = 0
            while i < n:
                v, j = view_of(i, n), i
                while j < n and view_of(j, n) == v:
                    j += 1
                self.set_cam(v)
                self.rt.dist_frames(self.cam_array(v, j - i))
                i = j
            return
        for i in range(n):
            self.step(i, n)

    def step(self, i: int, n: int) -> None:
        self.set_cam(view_of(i, n))
        if not self.sharded:
            self.rt.draw()
            return
        if self.native:
            self.rt.dist_frame()                     # kernel -> RCCL gather -> un-swizzle, on this launch's stream
            return
        f = self.frame_no
        self.frame_no += 1
        self.fg.begin_frame(f)                       # buffer f%2 is free once frame f-2's gather is done
        self.rt.set_target(self.fg.shard_for(f).data_ptr())
        self.rt.draw()                               # this rank's tiles of frame f
        self.fg.gather_async(f)                      # ONE collective per frame, overlapped with frame f+1's kernel
        if f >= 1:
            self.fg.complete(f - 1, None if self.env.stub else self.rt)   # rank 0: un-swizzle the previous frame

    def drain(self) -> None:
        if self.native:
            self.rt.dist_wait()
        elif self.sharded and self.frame_no >= 1:
            self.fg.complete(self.frame_no - 1, None if self.env.stub else self.rt)

    def run(self, n: int) -> None:
        """n untimed frames."""
        self.steps(n)
        self.drain()

    def timed(self, n: int) -> float:
        """EXACTLY n steps between barrier + synchronize on both sides; the maximum over ranks."""
        env = self.env
        events = (not self.sharded) and (not env.stub) and hasattr(self.rt, "region_begin")
        self.device_ms = None
        env.barrier()
        t0 = time.perf_counter()
        if events:
            self.rt.region_begin()   # (HIP events around the same steps, on both streams: `value_device_events`)
        self.steps(n)
        self.drain()  # the last frame's gather + un-swizzle belong to the timed region
        if not self.native:
            self.rt.wait()
        env.barrier()
        dt = env.all_max(time.perf_counter() - t0)
        if events:
            self.device_ms = self.rt.region_end()
        return dt

    def estimate_frame_ms(self) -> float:
        env = self.env
        env.barrier()
        t0 = time.perf_counter()
        for _ in range(2):
            self.step(2, 3)
        self.drain()
        if not self.native:
            self.rt.wait()
        env.sync()
        return env.all_max(max((time.perf_counter() - t0) / 2 * 1e3, 1e-3))   # every rank must choose the same step count

    def breakdown(self, launches: int):
        """Per-rank stage times of the native pipeline (events on each launch's stream, vrt_dist_profile), after the timed region:
        kernel / this rank's part of the collective / root un-swizzle, in us per frame; every rank's row and the maxima."""
        if not (self.native and self.sharded):
            return None
        self.rt.dist_profile(True)
        n = max(1, launches) * max(1, self.batch if self.env.world > 1 else 1)
        for i in range(n):
            self.step(i, n)
        self.rt.dist_wait()
        st = self.rt.dist_stats()
        self.rt.dist_profile(False)
        fpl = max(1, st["frames_per_launch"] if self.env.world > 1 else 1)
        row = {"rank": self.env.rank, "owned_tiles": st["owned_tiles"], "launches_sampled": st["launches_sampled"],
               "kernel_us_per_frame": st["kernel_ms_per_launch"] * 1e3 / fpl,
               "collective_us_per_frame": st["collective_ms_per_launch"] * 1e3 / fpl,
               "unswizzle_us_per_frame": st["unswizzle_ms_per_launch"] * 1e3 / fpl,
               "shard_bytes_per_frame": st["shard_bytes_per_frame"]}
        rows = self.env.gather_objects(row)
        peers = [r for r in rows if r["rank"] != 0] or rows
        return {"per_rank": rows,
                "max_over_ranks": {k: max(r[k] for r in rows) for k in ("kernel_us_per_frame", "collective_us_per_frame", "unswizzle_us_per_frame")},
                "root": {k: rows[0][k] for k in ("kernel_us_per_frame", "collective_us_per_frame", "unswizzle_us_per_frame", "owned_tiles")},
                "peers_max_kernel_us_per_frame": max(r["kernel_us_per_frame"] for r in peers),
                "note": "events on each launch's own stream after the timed region (vrt_dist_profile); collective = the grouped receives on rank 0, "
                        "the send elsewhere, waiting for the other side included"}

    def close(self) -> None:
        if self.rt is not None:
            self.rt.deinit()
            self.rt = None


def tuned_leg(env: Env, W, w, grid, batch: int, want_native: bool, fixed_share: int = -1):
    """The multi-GPU leg with rank 0's tile share chosen by a warm-up auto-tune: each candidate share gets its own context and
    communicator, 8 untimed + 32 timed frames (maximum over ranks, so every rank sees the same numbers and picks the same
    winner); the winner's context is kept, the others are destroyed.  --root-share N (>= 0) skips the tune."""
    args, world = env.args, env.world
    if args.root_share >= 0 or fixed_share >= 0 or not (2 <= world <= 8) or not want_native:
        share = args.root_share if args.root_share >= 0 else (fixed_share if fixed_share >= 0 else 100)
        return Leg(env, W, w, grid, sharded=True, want_native=want_native, batch=batch, root_share=share), {"tuned": False, "root_share": share}
    legs, ms = {}, {}
    for share in root_share_candidates(world):
        leg = Leg(env, W, w, grid, sharded=True, want_native=True, batch=batch, root_share=share)
        if not leg.native:     # every rank fell back to the torch path together: nothing to tune
            for other in legs.values():
                other.close()
            return leg, {"tuned": False, "root_share": 100}
        leg.run(max(8, batch))
        ms[share] = leg.timed(32) / 32 * 1e3
        legs[share] = leg
    best = min(sorted(ms), key=lambda k: ms[k])
    best = env.bcast(best)   # (identical on every rank already — the times are maxima over ranks — but a broadcast costs nothing)
    for share, leg in legs.items():
        if share != best:
            leg.close()
    return legs[best], {"tuned": True, "root_share": best, "candidates_ms_per_frame": {str(k): ms[k] for k in sorted(ms)},
                        "method": "per candidate: own context + communicator, 8 untimed + 32 timed frames, max over ranks"}


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
    ap.add_argument("--precondition-ms", type=float, default=0.0,
                    help="untimed GPU work before the warm-up steps of `value` (0: the protocol is W warm-up steps and K timed steps, nothing else; "
                         "`value_sustained` always reports the same steps behind 150 ms of frames)")
    ap.add_argument("--wall-budget", type=float, default=WALL_BUDGET_S, help="N > 1: seconds the whole run may take; the secondary leg is skipped if it would not fit")
    ap.add_argument("--frames-in-flight", type=int, default=2, help="1: frames strictly one after another; 2: two frames in flight")
    ap.add_argument("--pmc", choices=["auto", "live", "profile", "off"], default="auto",
                    help="HBM traffic / instruction counters of the roofline object: live = rocprofv3 passes over a child run now; "
                         "profile = the committed profiles/*_pmc.json; auto = live, falling back to profile")
    ap.add_argument("--pmc-frames", type=int, default=None, help="frames per view of a PMC child run; default: up to 24, about 0.6 s")
    ap.add_argument("--pmc-timeout", type=float, default=90.0, help="seconds per rocprofv3 --pmc pass (a pass takes 5-10 s; one that does not return is killed and the traffic is read from profiles/ instead)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dist", choices=["native", "torch"], default="native",
                    help="N>1 frame gather: native = RCCL send/recv inside libvrt_hip.so (pipelined); torch = torch.distributed.gather (fallback)")
    ap.add_argument("--dist-frames", type=int, default=4, help="launches in flight per rank of the native multi-GPU pipeline")
    ap.add_argument("--hw-queues", type=int, default=0, help="N > 1: GPU_MAX_HW_QUEUES of the HIP runtime (0: leave it; the environment wins)")
    ap.add_argument("--literal-launches", type=int, default=8, help="launches in flight per rank of the one-gather-per-frame leg (at most 16)")
    ap.add_argument("--dist-comms", type=int, default=0,
                    help="native multi-GPU pipeline: RCCL communicators the launch slots issue their gathers on (0: one per slot, at most 8 — RCCL runs "
                         "the operations of one communicator in issue order; 1: all on one, the round-5 pipeline)")
    ap.add_argument("--dist-batch", type=int, default=8,
                    help="frames traced by one launch and carried by one collective in the batched leg when world > 1 (every frame is gathered "
                         "once); the north_star-literal leg, one collective per frame, is always timed as well")
    ap.add_argument("--dist-submit", choices=["call", "frame"], default="call",
                    help="native multi-GPU pipeline: call = the consecutive frames of a view submitted by one vrt_dist_frames call (a compiled host's "
                         "frame loop); frame = one vrt_dist_frame call per frame from Python")
    ap.add_argument("--root-share", type=int, default=-1,
                    help="native multi-GPU pipeline: rank 0's share of the tiles in percent of an equal share; -1: chosen per leg by a warm-up "
                         "auto-tune over three candidates around 100 - 70 (world - 2) / 6")
    ap.add_argument("--no-secondary", action="store_true", help="N > 1: skip the secondary leg on BASELINE's sharded config (4K, 1024^3)")
    ap.add_argument("--force-gather", action="store_true", help="run the shard/gather/assemble path even at world size 1")
    ap.add_argument("--no-native-probe", action="store_true",
                    help="N > 1: skip the child-process probe of the native RCCL pipeline that decides between it and the torch path")
    ap.add_argument("--probe-timeout", type=float, default=240.0,
                    help="N > 1: seconds the child-process probe of the native pipeline may take (two ncclCommInitRank + ten ncclCommSplit on N GPUs, three small scenes)")
    ap.add_argument("--dist-probe", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--probe-uid", default="", help=argparse.SUPPRESS)
    ap.add_argument("--probe-rank", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--probe-world", type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument("--probe-device", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)            # tests/test_bench_plumbing.py
    ap.add_argument("--stub-fail-native-on", type=int, default=-1, help=argparse.SUPPRESS)
    args = ap.parse_args(argv)

    # N > 1: a rank's launch of its 1/N of the tiles lasts about as long as a whole frame's (its longest wave), so the literal leg's frame
    # rate is the number of launches the GPU runs side by side — and the HIP runtime maps streams onto 4 hardware queues unless told
    # otherwise, BEFORE it initialises (tools/experiments/literal_leg_probe.py: rank 1 of 8, one frame per launch, 30.9 us per frame with 8
    # launches on 4 queues, 16.2 with 16 on 24).  A knob of the runtime, set by the host process; libvrt_hip.so reads no environment.
    # On the one-GPU emulation of BOTH sides (tools/dist_emulate.py) 16 launches on 24 queues made rank 0 — which posts N-1 receives and
    # an un-swizzle per frame — three times slower (51 -> 200 us per frame at 8 ranks), so neither is the default: --hw-queues /
    # --literal-launches are there for the first run on a real node.
    if (args.gpus > 1 or args.dist_probe) and args.hw_queues > 0:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(args.hw_queues))
    if args.pmc_child:
        pmc_child(args)
        return
    if args.dist_probe:
        dist_probe_child(args)
        return
    ensure_ranks(args, argv)
    t_run0 = time.perf_counter()
    phase_seconds = {}

    def phase(name, t_from):
        phase_seconds[name] = phase_seconds.get(name, 0.0) + (time.perf_counter() - t_from)
        return time.perf_counter()

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner through C stdio (flushed at exit, i.e.
    # after anything Python printed), so everything else written to fd 1 by any library is sent to stderr and the
    # JSON line goes to a private copy of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np  # noqa: F401
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
    env = Env(args, torch, dist, world, rank, local_rank, dev, stub, use_dist)

    # every rank announces itself: rank 0 reports who took part
    ranks_seen = [rank]
    if env.multi:
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
    all_views = VIEW_ORDER + EXTRA_VIEWS

    def stub_counts():
        return {v: {"rays": 1000, "bytes": 4000, "issued_bytes": 2000, "counters": {}, "issued": {}, "primary_hit_fraction": 0.5} for v in all_views}

    t_ph = phase("startup", t_run0)
    # ---- rays and bytes per view, counted by counting builds of the kernel (untimed) ----
    per_view, count_err = {}, None
    if stub:
        per_view = stub_counts()
    elif rank == 0:
        try:
            per_view = count_rays(W, w, grid, all_views, args.variant, local_rank)
        except Exception as e:  # noqa: BLE001 - the other ranks wait in the broadcast below: tell them
            count_err = f"{type(e).__name__}: {e}"
    per_view, count_err = env.bcast((per_view, count_err))
    if count_err:
        raise SystemExit(f"bench.py: counting the rays failed on rank 0: {count_err}")

    def rays_of(pv, n):
        return sum(pv[view_of(i, n)]["rays"] for i in range(n))

    t_ph = phase("count_rays", t_ph)
    # ---- the timed legs ----
    want_native = args.dist == "native"
    probe_report = None
    if sharded and world > 1 and want_native and not stub and not args.no_native_probe:
        want_native, probe_report = native_probe(env, args.probe_timeout)
    t_ph = phase("native_probe", t_ph)
    legs_out = {}
    tune_reports = {}
    if not sharded:
        leg = Leg(env, W, w, grid, sharded=False, want_native=False, frames_in_flight=args.frames_in_flight)
        plan = [("single", leg)]
    else:
        # north_star's literal "one RCCL gather per frame" first, then the batched default: both are reported,
        # `value` is the batched leg's
        batches = [1] if (world == 1 or not want_native or args.dist_batch <= 1) else [1, args.dist_batch]
        plan = []
        for b in batches:
            leg, rep = tuned_leg(env, W, w, grid, b, want_native)
            tune_reports[f"batch{b}"] = rep
            plan.append((f"batch{leg.batch}" if leg.native else "torch", leg))
            if not leg.native:
                break       # the torch fallback has one form only
    t_ph = phase("contexts_and_root_share_tuning", t_ph)

    # a first look at the frame time (two untimed frames): sizes the run when --steps / --warmup were left to the script
    # and the settling / PMC legs for workloads whose frames take milliseconds instead of microseconds
    frame_ms_est = plan[-1][1].estimate_frame_ms()
    if args.steps is None:
        args.steps = int(min(600, max(6, 2500.0 / frame_ms_est)))
    if args.warmup is None:
        args.warmup = min(30, args.steps // 2)
    if args.pmc_frames is None:
        args.pmc_frames = int(min(24, max(2, 600.0 / frame_ms_est)))
    settle_frames = int(min(SETTLE_FRAMES, max(3, 1500.0 / frame_ms_est)))

    # `value` is the protocol's: W untimed warm-up steps, then exactly K timed steps (round 3 ran ~150 ms of untimed frames
    # first; --precondition-ms brings that back for A/B, default 0).  The GPU reaches its sustained clocks only after tens of milliseconds
    # of continuous work (tools/experiments/short_run.py: 20 frames timed cold 0.083 ms per frame, right after 40 ms of frames 0.077, 600 frames
    # 0.071), which a renderer — it runs continuously — has behind it: `value_sustained` (N = 1) times the same K steps again behind
    # PRECONDITION_MS of frames, and says so.
    precondition = 0 if stub else int(min(4000, args.precondition_ms / frame_ms_est))
    # N > 1: `value` is north_star's literal leg, one RCCL gather per frame (the first of the plan); the batched leg is reported beside it
    primary_name = plan[0][0]
    device_ms = None
    for name, leg in plan:
        if precondition:
            leg.run(precondition)
        leg.run(args.warmup)
        dt = leg.timed(args.steps)
        if name == primary_name:
            device_ms = getattr(leg, "device_ms", None)
        legs_out[name] = {"value": rays_of(per_view, args.steps) / dt / 1e6, "unit": "Mrays/s", "ms_per_step": dt / args.steps * 1e3,
                          "frames_per_collective": leg.batch, "launches_in_flight": leg.launches_in_flight,
                          "root_share_percent": leg.root_share if leg.native else None,
                          "communicators": leg.comm_info if leg.native else None,
                          "dist_path": ("native" if leg.native else "torch") if sharded else None,
                          "breakdown": leg.breakdown(2 * args.dist_frames)}
    sustained = None
    if not sharded and not stub:
        lg = plan[0][1]
        pre_sustained = int(min(4000, PRECONDITION_MS / frame_ms_est))
        lg.run(pre_sustained)
        lg.run(args.warmup)
        dts = lg.timed(args.steps)
        sustained = {"value": rays_of(per_view, args.steps) / dts / 1e6, "ms_per_step": dts / args.steps * 1e3, "precondition_frames": pre_sustained,
                     "value_device_events": (rays_of(per_view, args.steps) / (lg.device_ms * 1e-3) / 1e6) if lg.device_ms else None,
                     "note": f"the same {args.steps} steps timed again behind {PRECONDITION_MS:.0f} ms of untimed frames + the warm-up: the rate of a renderer that runs "
                             "continuously (the clocks ramp for tens of milliseconds); `value` is the cold protocol's"}
    leg = plan[-1][1]          # (kept open for the N = 1 roofline leg / the N > 1 line's pipeline description)
    dt = legs_out[primary_name]["ms_per_step"] * args.steps * 1e-3
    native, rccl_world = leg.native, leg.rccl_world
    for name, other in plan[:-1]:
        other.close()
    t_ph = phase("timed_legs", t_ph)

    if rank == 0 and sharded:
        # (to stderr, before the secondary leg: should that leg hang in a collective, the headline legs are on record)
        print("[bench] headline legs: " + json.dumps({"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "legs": legs_out,
                                                      "root_share_tuning": tune_reports}), file=sys.stderr, flush=True)

    # ---- N > 1: the same pipeline on BASELINE.json's sharded configurations: configs[3] (3840x2160, 1024^3, 4 rays per pixel: the one
    # BASELINE names for 2 -> 4 -> 8 GPUs) and configs[4] (3840x2160, 2048^3 sparse, 16 spp path trace, "8 MI355X") ----
    def secondary_leg(name, batches, tune, budget_note):
        """One BASELINE configuration through the native pipeline: per entry of `batches` a timed leg (1 = north_star's literal one RCCL
        gather per frame, whose rate is the leg's `value`; n = n frames per launch and per collective).  The root share is tuned on
        the first leg only (three candidates) and kept for the others; tune=False: an equal share (frames of milliseconds, beside
        which rank 0's receives and un-swizzle are nothing)."""
        w2 = W.WORKLOADS[name]
        t0 = time.perf_counter()
        grid2 = None if stub else W.build_grid(w2)
        # (rank 0 counts alone: a failure there must reach every rank, or the others would wait in the broadcast for ever)
        pv2, err2 = None, None
        if stub:
            pv2 = stub_counts()
        elif rank == 0:
            try:
                pv2 = count_rays(W, w2, grid2, VIEW_ORDER, args.variant, local_rank)
            except Exception as e:  # noqa: BLE001
                err2 = f"{type(e).__name__}: {e}"
        pv2, err2 = env.bcast((pv2, err2))
        if err2:
            raise RuntimeError(f"counting the rays of {w2.name} failed on rank 0: {err2}")
        out2 = {"workload": w2.name, "metric": metric_name(w2), "unit": "Mrays/s", "legs": {}, "budget": budget_note}
        share = -1 if tune else 100
        for b in batches:
            leg2, rep2 = tuned_leg(env, W, w2, grid2, b, True, fixed_share=share)
            if not leg2.native:
                leg2.close()
                raise RuntimeError("the native pipeline was not available on every rank")
            share = leg2.root_share
            if not stub and w2.max_bounce > 0:
                leg2.rt.reserve_samples(w2.spp)      # (the persistent kernels' sample buffers: made now, not inside the first frames)
            est2 = leg2.estimate_frame_ms()
            steps2 = int(min(args.steps, max(6, 2000.0 / est2)))
            leg2.run(min(args.warmup, steps2))
            dt2 = leg2.timed(steps2)
            out2["legs"][f"batch{leg2.batch}"] = {"value": rays_of(pv2, steps2) / dt2 / 1e6, "unit": "Mrays/s", "steps": steps2, "ms_per_step": dt2 / steps2 * 1e3,
                                                 "frames_per_collective": leg2.batch, "launches_in_flight": leg2.launches_in_flight, "root_share": rep2,
                                                 "kernel": None if stub else leg2.rt.kernel_name(), "breakdown": leg2.breakdown(2 * args.dist_frames)}
            leg2.close()
        first = f"batch{batches[0]}"
        out2.update({"value": out2["legs"][first]["value"], "ms_per_step": out2["legs"][first]["ms_per_step"], "steps": out2["legs"][first]["steps"],
                     "frames_per_collective": batches[0], "value_batched": out2["legs"][f"batch{batches[-1]}"]["value"] if len(batches) > 1 else None,
                     "seconds": round(time.perf_counter() - t0, 1)})
        del grid2
        return out2

    secondary, secondary_cfg4 = None, None
    run_secondary = sharded and world > 1 and native and not args.no_secondary and (args.workload or W.HEADLINE) == W.HEADLINE
    # (budgets, decided by every rank alike through an all-reduce.  configs[3]: a 1024^3 grid built per rank — ~13 s —, three root-share
    # candidates and two legs of up to `steps` 4K frames: estimated at twice what the headline's contexts + tuning + legs took + 60 s.
    # configs[4]: a 2048^3 sparse grid per rank, rank 0's ray count (~5 s of counting-build frames), one leg at an equal share, frames of
    # ~90 ms / N: 90 s.  A leg that would take the run past --wall-budget is skipped, and says so.)
    for key, name, batches, tune, estimate in (
            ("secondary", "cfg3_4k_1024c_b8", [1] + ([args.dist_batch] if args.dist_batch > 1 else []), True,
             2.0 * (phase_seconds.get("contexts_and_root_share_tuning", 0.0) + phase_seconds.get("timed_legs", 0.0)) + 60.0),
            ("secondary_cfg4", "cfg4_4k_2048c_b8_sparse", [1], False, 90.0)):
        if not run_secondary:
            break
        spent = time.perf_counter() - t_run0
        fits = bool(env.all_min_int(int(spent + estimate <= args.wall_budget)))
        note = f"{spent:.0f} s spent + ~{estimate:.0f} s estimated against --wall-budget {args.wall_budget:.0f} s"
        if not fits:
            result = {"skipped": "wall budget: " + note}
        else:
            try:
                result = secondary_leg(name, batches, tune, note)
            except Exception as e:  # noqa: BLE001 - a secondary leg must never take the headline line down
                print(f"[bench rank {rank}] {key} leg failed: {type(e).__name__}: {e}", file=sys.stderr)
                result = {"error": f"{type(e).__name__}: {e}"}
        if key == "secondary":
            secondary = result
        else:
            secondary_cfg4 = result
        if rank == 0:
            print(f"[bench] {key}: " + json.dumps(result), file=sys.stderr, flush=True)
        t_ph = phase(key + "_leg", t_ph)

    t_ph = phase("secondary_leg", t_ph) if "secondary_leg" not in phase_seconds else t_ph
    # ---- N = 1: the same steps strictly one frame after another, and the dominant kernel by HIP events ----
    roofline = None
    single = None
    if rank == 0 and not sharded and not stub:
        rt = leg.rt
        if args.frames_in_flight != 1:
            leg.close()
            leg = Leg(env, W, w, grid, sharded=False, want_native=False, frames_in_flight=1)
        rt1 = leg.rt
        leg.run(args.warmup)
        single = leg.timed(args.steps) / args.steps * 1e3
        # >= 100 timed frames per view, whatever --steps says — as long as a view's leg stays within about ten
        # seconds (frames of the 2048^3 path trace take 0.15 s each: 100 of them per view and pass would be minutes)
        reps = max(args.steps // len(VIEW_ORDER), min(100, max(8, int(5000.0 / frame_ms_est))))
        kernel_ms_view, frame_stats = {}, {}
        for v in all_views:
            leg.set_cam(v)
            # untimed: the camera has just jumped to this view, and the launch order follows the measured tile costs with
            # a lag (re-sorted from a running mean): let it settle as it would under a moving camera
            rt1.draw(frames=settle_frames)
            rt1.draw(frames=reps)                           # back to back, one event pair around all of them
            kernel_ms_view[v] = rt1.last_kernel_ms()
            frame_stats[v] = percentiles(rt1.draw_timed(min(reps, 512)))   # an event pair around every frame
        kernel_ran = rt1.kernel_name()   # the symbol of the launches timed above (vrt_kernel_name reports what ran)
        avg_ms = sum(kernel_ms_view[v] for v in VIEW_ORDER) / len(VIEW_ORDER)
        avg_bytes = sum(per_view[v]["bytes"] for v in VIEW_ORDER) / len(VIEW_ORDER)
        avg_issued = sum(per_view[v]["issued_bytes"] for v in VIEW_ORDER) / len(VIEW_ORDER)
        achieved = avg_bytes / (avg_ms * 1e-3) / 1e9
        try:
            hbm = hbm_achievable(dev)
        except Exception as e:  # noqa: BLE001
            hbm = {"error": f"{type(e).__name__}: {e}"}
        pmc, pmc_note = (None, "--pmc off")
        if args.pmc in ("auto", "live"):
            rt1.wait()
            pmc, pmc_note = pmc_live(args, w)
        if pmc is None and args.pmc in ("auto", "profile"):
            why = pmc_note
            pmc, pmc_note = pmc_from_profile(w.brick_dimension)
            if args.pmc == "auto":
                pmc_note = f"{pmc_note} (live measurement failed: {why})"
        traffic = issue_ipc = valu_frac = issue_slots_frac = lane_util = None
        insts = None
        clock_hz = None
        if pmc is not None:
            # FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE on gfx950 tallies 64 B per 128-byte line fetched
            # (MI355X_MICROARCH.md; tools/ubench/fetch_calib.hip confirms it for scattered dword reads): doubled
            traffic = 2.0 * pmc["FETCH_SIZE"] * 1024.0 + pmc["WRITE_SIZE"] * 1024.0
            keys = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_LDS")
            if all(k in pmc for k in keys[:3]):
                import ctypes as C
                insts = {k: pmc.get(k, 0.0) for k in keys}
                from zig_vulkan_amd import _lib as VL
                di = (C.c_int64 * 4)()
                VL.check(VL.lib.vrt_device_info(local_rank, di))
                clock_hz = di[0] * 1e3
                simds = int(di[1]) * 4
                issue_ipc = sum(insts.values()) / (simds * avg_ms * 1e-3 * clock_hz)
                valu_frac = insts["SQ_INSTS_VALU"] * 2.0 / (simds * avg_ms * 1e-3 * clock_hz)
                # every instruction takes one issue slot of its SIMD, a wave64 vector instruction two ("issue slots")
                issued = pmc.get("SQ_INSTS") or (sum(insts.values()) + pmc.get("SQ_INSTS_BRANCH", 0.0))
                issue_slots_frac = (issued + insts["SQ_INSTS_VALU"]) / (simds * avg_ms * 1e-3 * clock_hz)
                for k in ("SQ_INSTS_BRANCH", "SQ_INSTS"):
                    if k in pmc:
                        insts[k] = pmc[k]
            if pmc.get("SQ_ACTIVE_INST_VALU") and pmc.get("SQ_THREAD_CYCLES_VALU"):
                # how many of its 64 lanes the average vector instruction serves (EXEC's population, weighted by the instruction's cycles)
                lane_util = pmc["SQ_THREAD_CYCLES_VALU"] / (64.0 * pmc["SQ_ACTIVE_INST_VALU"])
        # the present / denoise pass that follows every trace in the reference's frame (image.frag:18-78 at GraphicsPipeline.zig:34-39's
        # defaults; Pipeline.draw, Pipeline.zig:432-541): the app's 1024x576 image goes to a 1920x1080 window, other workloads 1:1
        present = None
        try:
            pw, ph = (1920, 1080) if (w.width, w.height) == (1024, 576) else (w.width, w.height)
            leg.set_cam("V1")
            p_ms, t_ms = [], []
            for _ in range(64):
                rt1.draw()
                rt1.present(pw, ph)
                t_ms.append(rt1.last_kernel_ms())
                p_ms.append(rt1.last_denoise_ms())
            p_ms, t_ms = sorted(p_ms), sorted(t_ms)
            p_med = p_ms[len(p_ms) // 2]
            p_bytes = ((20 + 2) * 16 + 4) * pw * ph
            present = {"kernel": "vrt_denoise_kernel", "from": [w.width, w.height], "to": [pw, ph], "samples": 20, "us_median": p_med * 1e3, "us_min": p_ms[0] * 1e3,
                       "algorithmic_bytes": p_bytes, "bytes_note": "(samples + 2) bilinear taps of 4 texels x 4 B + 4 B written, per output pixel",
                       "achieved_GBps": p_bytes / (p_med * 1e-3) / 1e9, "frac": p_bytes / (p_med * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                       # (frac > 1: the byte model is no bound — the staged kernel reads a texel once into LDS and serves the taps from there)
                       "frac_model_valid": bool(p_bytes / (p_med * 1e-3) / 1e9 / HBM_PEAK_GBPS <= 1.0), "bound": "issue (taps served from LDS)",
                       "trace_us_median_same_frames": t_ms[len(t_ms) // 2] * 1e3, "frame_trace_plus_present_us": (t_ms[len(t_ms) // 2] + p_med) * 1e3,
                       "note": "view V1, 64 frames, one frame at a time, HIP events around each launch; the source image stays in L2: bound by its gathers and pow(), not by HBM"}
        except Exception as e:  # noqa: BLE001
            present = {"error": f"{type(e).__name__}: {e}"}
        # frac > 1: the kernel is not doing the modelled work (it never asks for the status words of cells it knows to be empty), so the
        # HBM model does not describe it: the governing figure is then the share of the issue slots the launch fills (VERDICT r03 #5a)
        frac = achieved / HBM_PEAK_GBPS
        model_valid = frac <= 1.0
        roofline = {
            "bound": "hbm" if model_valid else "issue", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": frac,
            "frac_model_valid": model_valid,
            "governing_frac": frac if model_valid else issue_slots_frac,
            "governing_note": ("frac (algorithmic bytes / kernel time / HBM peak)" if model_valid else
                               "issue_slots_frac: frac > 1 means the timed kernel does not request the bytes the model counts (skip-to-box), so the HBM "
                               "model is void for this workload; the kernel is bound by instruction issue"),
            "lane_util": lane_util,
            "lane_util_counters": ({k: pmc[k] for k in ("SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU")} if lane_util is not None else None),
            "lane_util_note": "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): the share of its 64 lanes the average vector instruction serves",
            "traffic": traffic, "traffic_note": pmc_note,
            "definition": "achieved = bytes the REFERENCE algorithm loads for these frames (4 S + 4 K + V + 25 H per ray + 4 B per pixel, "
                          "SURVEY.md 8(d), counted by the counting build that walks to the grid's face like the shader) / kernel time: a rate "
                          "of useful work, not of bytes moved — the product kernel never asks for the status words of cells it knows to "
                          "be empty (it jumps to the near face of the occupied-cell box and stops at its far face), so frac can exceed 1 on "
                          "views from outside the box.  issued_bytes = what the product kernel's lanes request (4 B per brick-level trip "
                          "taken, 12 per brick entered, 4 per voxel trip, 21 per hit, 4 per pixel). "
                          "traffic = HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE): the touched scene data lives in L2 / MALL.",
            "kernel": kernel_ran, "settle_frames": settle_frames, "timed_frames_per_view": reps,
            "kernel_ms_avg": avg_ms, "kernel_ms_per_view": kernel_ms_view,
            "frame_ms_percentiles_per_view": frame_stats,
            "algorithmic_bytes_per_launch": avg_bytes,
            "issued_bytes": avg_issued, "issued_GBps": avg_issued / (avg_ms * 1e-3) / 1e9,
            "frac_issued": avg_issued / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "issue_slots_frac": issue_slots_frac,
            "issue_slots_note": "(SQ_INSTS + SQ_INSTS_VALU) / (SIMDs x kernel_ms_avg x engine clock): a SIMD issues about one instruction per clock "
                                "in total, a wave64 vector instruction taking two slots (tools/ubench/issue_probe.hip; every same-box A/B of the walk "
                                "loops agrees with cost = 2 V + S + rest, DESIGN.md 4) - the share of its issue slots the launch fills is the "
                                "ceiling this path runs against, not HBM",
            "valu_frac": valu_frac,
            "valu_frac_note": "SQ_INSTS_VALU x 2 cycles (a wave64 vector instruction holds its SIMD's pipe for two) / (SIMDs x kernel_ms_avg x "
                              "engine clock): the share of the vector pipe the launch keeps busy (the ceiling is issue_slots_frac: vector instructions share "
                              "their SIMD's one issue slot per clock with everything else); "
                              "frac_issued = issued_GBps / HBM peak; hbm_traffic_GBps = traffic / kernel time: what HBM really carries",
            "hbm_traffic_GBps": (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None,
            "hbm_achievable": hbm,
            "issue_ipc": issue_ipc,
            "issue_ipc_note": "wave-instructions per launch (SQ_INSTS_VALU + SALU + VMEM + SMEM + LDS) / (1024 SIMDs x kernel_ms_avg x "
                              "the device's engine clock): the kernel is bound by instruction issue, not by HBM",
            "insts_per_launch": insts, "clock_hz": clock_hz,
        }

    t_ph = phase("roofline_leg", t_ph)
    if rank == 0:
        par = (f"1 GPU, whole frame, {args.frames_in_flight} frame(s) in flight" if not sharded else
               f"image tiles 16x16 interleaved over {world} GPU(s), every frame gathered once to rank 0, "
               + (f"native RCCL pipeline, ONE collective per frame (`value` = legs.{primary_name}; the leg with {leg.batch if world > 1 else 1} frame(s) per launch and per "
                  f"collective is `value_batched`), launches in flight: legs.*.launches_in_flight, rank 0 owns {leg.root_share} % of an equal share of the tiles"
                  if native else "torch.distributed gather per frame, frame f overlaps the kernel of f+1"))
        out = {
            "metric": metric_name(w),
            "value": legs_out[primary_name]["value"],
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "precondition_frames": precondition,
            "precondition_note": "untimed frames before the warm-up steps (--precondition-ms, default 0: the protocol is `warmup` untimed + `steps` timed steps)",
            "value_device_events": (rays_of(per_view, args.steps) / (device_ms * 1e-3) / 1e6) if device_ms else None,
            "device_ms_timed_region": device_ms, "rays_timed_region": rays_of(per_view, args.steps),
            "value_device_events_note": "the same timed steps by the device's clock: hipEventElapsedTime from the earlier of the two streams' begin events to the "
                                        "later of their end events (vrt_region_begin / _end, SURVEY.md 8(d)); `value` is the host's wall clock around them",
            "value_sustained": sustained,
            "phase_seconds": {k: round(v, 3) for k, v in phase_seconds.items()},
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
        if roofline is not None:
            out["present_pass"] = present
        if sharded:
            out["legs"] = legs_out
            last = plan[-1][0]
            out["value_batched"] = legs_out[last]["value"] if last != primary_name else None
            out["legs_note"] = ("batch1 = north_star's literal one RCCL gather per frame = `value`; the batched leg traces N frames per launch and gathers them "
                                "with one collective (N frames of latency for 2-3x the frame rate at 8 ranks in the one-GPU emulation: a rank's 1/R of the tiles "
                                "does not fill a GPU) = `value_batched`")
            out["wall_budget_s"] = args.wall_budget
            out["root_share_tuning"] = tune_reports
            out["native_probe"] = probe_report   # None: not run (one rank, --dist torch, --no-native-probe)
            out["secondary"] = secondary            # configs[3]: legs.batch1 (= its `value`) and legs.batch<N>
            out["secondary_cfg4"] = secondary_cfg4  # configs[4]: the path trace, one gather per frame
        if world == 1 and not sharded and not args.no_cpu_baseline and not stub:
            # frames of this library for the parity certificate: the product build, and its fused-arithmetic twin where built
            hip_frames = {}
            try:
                from zig_vulkan_amd import _lib as VL
                builds = {"product": None}
                if os.path.exists(VL.FUSED_LIB_PATH):
                    builds["fused"] = VL.FUSED_LIB_PATH
                for build, path in builds.items():
                    rtf = W.make_renderer(w, grid, device_id=local_rank, want_float_output=True, kernel_variant=args.variant,
                                          **({"library": path} if path else {}))
                    hip_frames[build] = {}
                    for v in VIEW_ORDER:
                        W.set_view(rtf, v)
                        rtf.draw()
                        hip_frames[build][v] = (rtf.read_rgba32f().copy(), rtf.read_rgba8().copy())
                    rtf.deinit()
            except Exception as e:  # noqa: BLE001
                print(f"[bench] parity frames unavailable: {type(e).__name__}: {e}", file=sys.stderr)
                hip_frames = {}
            port = cpu_baseline_port(w, grid, args.cpu_seconds, per_view)
            ref, parity, why = cpu_baseline_reference(w, grid, args.cpu_seconds, per_view, hip_frames)
            if ref is not None:
                out["cpu_baseline"] = ref
                out["cpu_baseline_port"] = port
                if parity is not None:
                    parity["note"] = ("product = libvrt_hip.so: its arithmetic contract is the reference shader's as Mesa gallivm executes it, so every pixel "
                                      "is expected bit-equal to the reference frame; fused = the same kernel source with fma fused and dot as an fma "
                                      "chain (libvrt_hip_fused.so, test infrastructure): what GLSL would also allow, and how far it moves the frame")
                    out["parity_vs_reference"] = parity
            else:
                port["reference_unavailable"] = why
                out["cpu_baseline"] = port
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    leg.close()
    if sharded and not stub:
        try:      # (every rank, at the same point: the pooled communicators are destroyed by a collective)
            from zig_vulkan_amd import VoxelRT
            VoxelRT.dist_release_communicators()
        except Exception as e:  # noqa: BLE001
            print(f"[bench rank {rank}] vrt_dist_release_communicators: {type(e).__name__}: {e}", file=sys.stderr)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
