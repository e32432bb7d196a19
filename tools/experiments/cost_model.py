#!/usr/bin/env python3
"""Where the traversal kernel's time goes, by differential experiments on the headline frame size:
   - no rays at all (device max_bounce 0): ray generation + background + store only
   - empty grid: slab test + brick-level walk only (no brick entered, no shadow ray)
   - terrain, sun off: primary rays
   - terrain, sun on: primary + shadow (the headline)
Prints kernel ms (HIP events, mean of 20 launches) beside the wave-level trip counts of the same frame."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dataclasses
from zig_vulkan_amd import workloads as W
from zig_vulkan_amd.voxel_rt import BrickGrid

base = W.WORKLOADS[W.HEADLINE]
variant = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0


def run(label, w, grid):
    rt = W.make_renderer(w, grid, kernel_variant=variant)
    rc = W.make_renderer(w, grid, enable_counters=True, kernel_variant=variant)
    for view in ["V0", "V1", "V2"]:
        W.set_view(rt, view); W.set_view(rc, view)
        rt.draw(3); rt.wait()
        rt.draw(20); rt.wait()
        ms = rt.last_kernel_ms()
        rc.draw(); rc.wait()
        c, wc = rc.counters(), rc.wave_counters()
        nw = rc.shard_info().owned_tiles * 4
        print(f"{label:28s} {view} {ms*1000:8.1f} us  rays/px {c['rays']/(w.width*w.height):.2f}  per wave: grid trips {wc['wave_grid_iters']/nw:6.1f} "
              f"brick walks {wc['wave_brick_walks']/nw:5.2f} voxel trips {wc['wave_voxel_iters']/nw:6.1f}   lanes/trip {c['grid_steps']/max(1,wc['wave_grid_iters']):.1f}")
    rt.deinit(); rc.deinit()


terrain = W.build_grid(base)
n = base.voxels // base.brick_dimension
empty = BrickGrid(n, n, n, min_point=(-32.0, -32.0, -32.0), scale=64.0 / n, brick_dimension=base.brick_dimension)
run("no rays (max_bounce 0)", dataclasses.replace(base, max_bounce=-1), terrain)
run("empty grid, sun off", dataclasses.replace(base, sun_enabled=False), empty)
run("terrain, sun off", dataclasses.replace(base, sun_enabled=False), terrain)
run("terrain, sun on (headline)", base, terrain)
