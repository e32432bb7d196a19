#!/usr/bin/env python3
"""Latency of ONE 16x16 tile rendered alone (the frame sharded so that this context owns a single tile): the
isolated-wave cost behind the kernel's tail.  usage: one_tile.py VIEW TILE_X TILE_Y [variant ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W

view, tx, ty = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
profile = "--profile" in sys.argv
variants = [int(v, 0) for v in sys.argv[4:] if v != "--profile"] or [0]
w = W.WORKLOADS[W.HEADLINE]
grid = W.build_grid(w)
tiles_x, tiles_y = (w.width + 15) // 16, (w.height + 15) // 16
tile = ty * tiles_x + tx
for variant in variants:
    rt = W.make_renderer(w, grid, shard_rank=tile, shard_count=tiles_x * tiles_y, kernel_variant=variant)
    rc = W.make_renderer(w, grid, shard_rank=tile, shard_count=tiles_x * tiles_y, kernel_variant=variant, enable_counters=True)
    W.set_view(rt, view); W.set_view(rc, view)
    rt.draw(3); rt.wait(); rt.draw(20); rt.wait()
    rc.draw(); rc.wait()
    c, wc = rc.counters(), rc.wave_counters()
    print(f"variant {variant:#x} {rt.kernel_name()}: tile ({tx},{ty}) alone {rt.last_kernel_ms()*1000:.1f} us; 4 waves: grid trips {wc['wave_grid_iters']} brick walks {wc['wave_brick_walks']} "
          f"voxel trips {wc['wave_voxel_iters']}; lanes: grid steps {c['grid_steps']} bricks {c['bricks_entered']} voxel steps {c['voxel_steps']} hits {c['hits']} rays {c['rays']}")
    if profile:  # library built with make EXTRA=-DVRT_DEV_PROFILE: core-clock cycles per phase, summed over the 4 waves
        pr = rt.wave_timeline(raw=True).reshape(-1)[:8]
        names = ["grid loop", "brick walks (all)", "voxel loops", "grid_hit setup", "material test", "-", "-", "whole wave"]
        print("   cycles summed over 4 waves:", ", ".join(f"{n} {int(v)}" for n, v in zip(names, pr) if n != "-"))
    rt.deinit(); rc.deinit()
