#!/usr/bin/env python3
"""Frame rate of the native multi-GPU pipeline with R ranks as contexts of ONE process on ONE GPU over the test-only RCCL
stand-in (tests/fake_rccl): the GPU does the same tracing work as a single context plus the staging copies, so the
difference to the plain frame rate is what the pipeline's launches, events and ordering cost (plus the stand-in's own
hipMalloc per send, which dominates at 8 ranks).  Measured: 92.5 us single context; 101 / 103 / 161 us per frame at 2 / 4 / 8
ranks.  usage: dist_overhead.py [frames]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from zig_vulkan_amd import workloads as W
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
w = W.WORKLOADS[W.HEADLINE]
grid = W.build_grid(w)
rt = W.make_renderer(w, grid, frames_in_flight=2)
W.set_view(rt, "V0")
for _ in range(20): rt.draw()
rt.wait(); t0 = time.perf_counter()
for _ in range(frames): rt.draw()
rt.wait(); plain = (time.perf_counter() - t0) / frames
rt.deinit()
print(f"single context, 2 frames in flight: {plain*1e6:.1f} us per frame (view V0)")
for world in (2, 4, 8):
    uid = b"overhead" + bytes([world]) + os.urandom(16) + bytes(128 - 25)
    ranks = [W.make_renderer(w, grid, shard_rank=r, shard_count=world) for r in range(world)]
    for r, x in enumerate(ranks):
        W.set_view(x, "V0")
        x.dist_init(uid, r, world, frames_in_flight=8, rccl_path=FAKE)
    def drive(x, n):
        for _ in range(n): x.dist_frame()
        x.dist_wait()
    for n in (20, frames):
        th = [threading.Thread(target=drive, args=(x, n)) for x in ranks]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = (time.perf_counter() - t0) / n
    print(f"{world} ranks on one GPU, 8 frames in flight each: {dt*1e6:.1f} us per frame")
    for x in ranks: x.deinit()
