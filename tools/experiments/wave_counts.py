#!/usr/bin/env python3
"""Wave-level execution counts for the headline workload: how often a wave runs the voxel-level walk."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W
w = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else W.HEADLINE]
grid = W.build_grid(w)
rt = W.make_renderer(w, grid, enable_counters=True)
for view in ["V0", "V1", "V2"]:
    W.set_view(rt, view); rt.draw()
    c, wc = rt.counters(), rt.wave_counters()
    nw = rt.shard_info().owned_tiles * 4
    print(view, {k: round(v / nw, 1) for k, v in wc.items()}, "per wave;  per-lane totals / wave:",
          {k: round(c[k] / nw, 1) for k in ("grid_steps", "bricks_entered", "voxel_steps", "rays")})
rt.deinit()
