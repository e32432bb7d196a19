#!/usr/bin/env python3
"""What one rank of an 8-GPU run sees: 1/8 of the headline frame's tiles per kernel (4 080 waves, fewer than the
6 144 wave slots), kernel duration set by its longest wave.  How many such kernels must be in flight, on how many
streams, to keep one GPU busy?  K contexts x 2 streams each, all rendering shard 0 of 8."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W
w = W.WORKLOADS[W.HEADLINE]
grid = W.build_grid(w)
full = W.make_renderer(w, grid, frames_in_flight=2)
for view in ["V0", "V1", "V2"]:
    W.set_view(full, view)
    for _ in range(10): full.draw()
    full.wait(); t0 = time.perf_counter()
    for _ in range(200): full.draw()
    full.wait(); tf = (time.perf_counter() - t0) / 200
    out = [f"{view}: whole frame {tf*1e6:.1f} us (an eighth: {tf*1e6/8:.1f});  an eighth of the tiles per kernel with"]
    for K in (1, 2, 4):
        rts = [W.make_renderer(w, grid, shard_rank=0, shard_count=8, frames_in_flight=2) for _ in range(K)]
        for rt in rts: W.set_view(rt, view)
        for _ in range(10):
            for rt in rts: rt.draw()
        for rt in rts: rt.wait()
        n = 400
        t0 = time.perf_counter()
        for _ in range(n // K):
            for rt in rts: rt.draw()
        for rt in rts: rt.wait()
        out.append(f"{2*K} streams {(time.perf_counter() - t0) / n * 1e6:.1f} us")
        for rt in rts: rt.deinit()
    print(" ".join(out[:1]) + " " + ", ".join(out[1:]))
full.deinit()
