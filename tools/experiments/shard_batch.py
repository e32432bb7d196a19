#!/usr/bin/env python3
"""What one non-root rank of an 8-GPU run achieves on its own GPU: rank 1 of 8 over the test-only RCCL stand-in (its
sends never wait for a receiver), frames per launch 1 / 2 / 4 / 8.  Compare with an eighth of the whole-frame time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from zig_vulkan_amd import workloads as W
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
w = W.WORKLOADS[W.HEADLINE]
grid = W.build_grid(w)
full = W.make_renderer(w, grid, frames_in_flight=2)
whole = {}
for view in ["V0", "V1", "V2"]:
    W.set_view(full, view)
    for _ in range(10): full.draw()
    full.wait(); t0 = time.perf_counter()
    for _ in range(200): full.draw()
    full.wait(); whole[view] = (time.perf_counter() - t0) / 200
full.deinit()
print("whole frame, 2 in flight (us):", {v: round(t * 1e6, 1) for v, t in whole.items()}, " an eighth:", {v: round(t * 1e6 / 8, 1) for v, t in whole.items()})
# usage: shard_batch.py [world] [root share in % (0 = equal)]: rank 1 of `world`, whose share grows as the root's shrinks
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
weight = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for batch, slots in ((1, 8), (2, 4), (4, 4), (8, 2), (8, 4)):
    rt = W.make_renderer(w, grid, shard_rank=1, shard_count=world, shard_root_weight=weight)
    rt.dist_init(b"shard-batch" + bytes([batch, slots]) + os.urandom(16) + bytes(128 - 29), 1, world, frames_in_flight=slots, rccl_path=FAKE, frames_per_launch=batch)
    out = {}
    for view in ["V0", "V1", "V2"]:
        W.set_view(rt, view)
        for _ in range(32): rt.dist_frame()
        rt.dist_wait(); t0 = time.perf_counter()
        n = 480
        for _ in range(n): rt.dist_frame()
        rt.dist_wait(); out[view] = round((time.perf_counter() - t0) / n * 1e6, 1)
    print(f"rank 1 of {world} (root share {weight or 100} %), {batch} frame(s) per launch, {slots} launches in flight: us per frame", out, " mean", round(sum(out.values()) / 3, 1))
    rt.deinit()
