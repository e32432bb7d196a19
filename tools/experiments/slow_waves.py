#!/usr/bin/env python3
"""The slowest waves of one frame: where they are in the image and what their 64 rays do (oracle counters per pixel)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from helpers import oracle_scene_from_grid, push_for, O

view = sys.argv[1] if len(sys.argv) > 1 else "V1"
w = W.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else W.HEADLINE]
grid = W.build_grid(w)
rt = W.make_renderer(w, grid, kernel_variant=0x30000, tuning_flags=16384)   # reverse raster, whole tiles: wave -> pixel block is the plain map
W.set_view(rt, view)
rt.draw(); rt.wait()
t = rt.wave_timeline().astype(np.int64)
dur = (t[:, 1] - t[:, 0]) / 100.0
order = np.argsort(-dur)[:8]
tiles_x = (w.width + 15) // 16
ntiles = rt.shard_info().owned_tiles
scene = oracle_scene_from_grid(grid)
pc = push_for(rt.camera, rt.sun)
for wid in order:
    block, wave = divmod(int(wid), 4)
    tile = ntiles - 1 - block
    ty, tx = divmod(tile, tiles_x)
    x0, y0 = tx * 16 + (wave & 1) * 8, ty * 16 + (wave >> 1) * 8
    per = []
    for yy in range(8):
        for xx in range(8):
            _, _, c = O.render_pixels(scene, pc, np.array([[x0 + xx, y0 + yy]]))
            per.append((c["rays"], c["grid_steps"], c["bricks_entered"], c["voxel_steps"], c["hits"]))
    per = np.array(per)
    print(f"wave {wid}: {dur[wid]:.1f} us  pixel block ({x0},{y0})  per-lane rays {per[:,0].min()}..{per[:,0].max()}  grid steps mean {per[:,1].mean():.0f} max {per[:,1].max()}  "
          f"bricks entered mean {per[:,2].mean():.1f} max {per[:,2].max()} sum {per[:,2].sum()}  voxel steps mean {per[:,3].mean():.0f} max {per[:,3].max()} sum {per[:,3].sum()}")
rt.deinit()
