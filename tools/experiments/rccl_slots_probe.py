#!/usr/bin/env python3
"""Real RCCL on ONE GPU (world 1): every launch slot of the multi-GPU pipeline issues a busy kernel and then its gather — a grouped self
send + recv of one shard — on its own stream (vrt_dist_selftest_slots), with the slots on ONE communicator (round 5's pipeline) and with
a communicator per slot (round 6).  What it shows: no deadlock at 16 streams, the bytes arrive, and whether RCCL runs the operations of
one communicator in issue order (us per launch with one communicator ~ the sum; with one per slot ~ the sum over the overlap).
Also: what making the communicators costs at world 1 (a lower bound of the 8-GPU cost).
usage: rccl_slots_probe.py [width height]      -> lines for profiles/r06_rccl_slots_world1.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from zig_vulkan_amd import VoxelRT
from zig_vulkan_amd import workloads as W

width, height = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
w = W.Workload("t", width, height, 64, 4, 1, 0, True, 0.0)
grid = W.build_grid(w)
print(f"# real RCCL (PyTorch's librccl), world 1, shard = the whole {width}x{height} frame as packed RGB; per launch: a one-wave kernel busy for B us, then the "
      "grouped self send + recv on the slot's communicator and stream; 40 rounds after one untimed round")
for slots in (4, 8, 16):
    for comms in (1, 0):
        rt = W.make_renderer(w, grid, shard_rank=0, shard_count=1)
        t0 = time.perf_counter()
        rt.dist_init(VoxelRT.dist_unique_id(), 0, 1, frames_in_flight=slots, communicators=comms)
        t_init = time.perf_counter() - t0
        info = rt.dist_comm_info()
        shard = rt.dist_stats()["shard_bytes_per_frame"]
        for busy in (0, 50, 200):
            st = rt.dist_selftest_slots(busy_us=busy, rounds=40)
            print(f"{slots:2d} slots on {info['communicators']:2d} communicator(s) (made in {1e3 * t_init:7.1f} ms, agreed by all-reduce: {info['agreed_by_all_reduce']}), "
                  f"busy {busy:3d} us, shard {shard} B: {st['us_per_launch']:7.1f} us per launch; last round, kernel start -> gather end: "
                  f"{1e3 * st['last_round_launch_ms']:7.1f} us", flush=True)
        rt.deinit()
