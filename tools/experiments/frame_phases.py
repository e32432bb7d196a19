#!/usr/bin/env python3
"""Whole-frame phase profile of vrt_trace_kernel (library built with make EXTRA=-DVRT_DEV_PROFILE): core-clock cycles per
phase summed over all waves.  usage: frame_phases.py [workload] [views]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from zig_vulkan_amd import workloads as W

w = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else W.HEADLINE]
views = sys.argv[2].split(",") if len(sys.argv) > 2 else ["V0", "V1", "V2", "VG"]
grid = W.build_grid(w)
rt = W.make_renderer(w, grid, kernel_variant=0x30000)   # reverse raster: no split tiles / spare workgroups
names = ["grid loop", "brick walks (entry + voxel loop + material)", "voxel loops", "grid_hit setup", "material test", "-", "-", "whole wave"]
for v in views:
    W.set_view(rt, v)
    rt.draw(3); rt.wait()
    pr = rt.wave_timeline(raw=True).reshape(-1, 8).astype(np.float64).sum(axis=0)
    tot = pr[7]
    other = tot - pr[0] - pr[1] - pr[3]
    print(f"{w.name} {v}: whole {tot/1e9:.3f} G wave-cycles | grid loop {100*pr[0]/tot:.1f} % | bricks {100*pr[1]/tot:.1f} % (voxel loops {100*pr[2]/tot:.1f}, material {100*pr[4]/tot:.1f}) "
          f"| grid_hit setup {100*pr[3]/tot:.1f} % (slab test {100*pr[6]/tot:.1f}, skip to box {100*pr[5]/tot:.1f}) | rest (ray gen, shading, sun jitter, tone-map, store) {100*other/tot:.1f} %")
rt.deinit()
