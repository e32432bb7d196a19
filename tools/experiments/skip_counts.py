#!/usr/bin/env python3
"""Trips of the grid-level walk per view with and without the skip to the occupied-cell box (enable_counters = 2: what the
product kernel walks).  usage: skip_counts.py <workload> [tuning_flags]   (0: with the jump; 1 = VRT_TUNE_NO_SKIP_TO_BOX: every cell walked)"""
import sys
sys.path.insert(0, ".")
from zig_vulkan_amd import workloads as W
w = W.WORKLOADS[sys.argv[1]]
grid = W.build_grid(w)
rt = W.make_renderer(w, grid, enable_counters=2, tuning_flags=int(sys.argv[2], 0) if len(sys.argv) > 2 else 0)
for v in ["V0", "V1", "V2", "V1x", "VG"]:
    W.set_view(rt, v)
    rt.draw()
    c = rt.counters()
    print(v, {k: c[k] for k in ("rays", "grid_steps", "bricks_entered", "voxel_steps", "hits") if k in c}, flush=True)
rt.deinit()
