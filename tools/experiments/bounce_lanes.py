#!/usr/bin/env python3
"""Where the lanes of the reference app's bounce frames go: the same scene and views traced with device max_bounce 1 / 2 / 3
by a counting context; the differences are what each bounce ADDS — rays, lane-steps of the brick-level loop, wave-trips of that loop
(counted once per wave however many lanes took part), voxel-level steps and trips.  The image is a pure function of its inputs and a
path's first k bounces do not depend on max_bounce, so the differences are exact per-bounce totals.
  lanes per wave-trip = lane-steps / wave-trips: how many of a wave's 64 lanes do useful work in an average trip of the walk loop;
  rays per wave       = rays of the bounce / waves of the frame: how many lanes of a wave HAVE a ray in that bounce at all.
Compaction at bounce boundaries can only raise the second towards 64; what the lockstep kernel loses is the first.
usage: bounce_lanes.py [workload]      (the counting kernels walk every cell: no skip to the box; lockstep, one pixel per lane)"""
import dataclasses
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "refapp_1024x576_128x64x128_b4"
base = W.WORKLOADS[name]
grid = W.build_grid(base)
KEYS = ("rays", "grid_steps", "wave_grid_trips", "voxel_steps", "wave_voxel_trips", "bricks_entered", "wave_brick_walks", "hits")
for spp in (1, base.spp) if base.spp > 1 else (1,):
    per = {}
    for mb in range(0, base.max_bounce + 1):
        w = dataclasses.replace(base, spp=spp, max_bounce=mb)
        cnt = W.make_renderer(w, grid, enable_counters=True)
        for v in ("V0", "V1", "V2"):
            W.set_view(cnt, v)
            cnt.draw()
            c, wc = cnt.counters(), cnt.wave_counters()
            per[(mb, v)] = dict(rays=c["rays"], grid_steps=c["grid_steps"], voxel_steps=c["voxel_steps"], bricks_entered=c["bricks_entered"], hits=c["hits"],
                                wave_grid_trips=wc["wave_grid_iters"], wave_brick_walks=wc["wave_brick_walks"], wave_voxel_trips=wc["wave_voxel_iters"])
        cnt.deinit()
    waves = ((base.width + 15) // 16) * ((base.height + 15) // 16) * 4
    print(f"# {name}, {spp} sample(s) per pixel, {waves} waves of 64 pixels; per bounce = counters(max_bounce b) - counters(max_bounce b - 1); every bounce = its ray + its shadow ray")
    for v in ("V0", "V1", "V2"):
        prev = {k: 0 for k in KEYS}
        for mb in range(0, base.max_bounce + 1):
            cur = per[(mb, v)]
            d = {k: cur[k] - prev[k] for k in KEYS}
            prev = cur
            what = "primary + shadow" if mb == 0 else f"bounce {mb} + shadow"
            print(f"{v} {what:18s}: {d['rays'] / 1e6:6.2f} M rays = {d['rays'] / spp / waves:5.1f} rays per wave and sample; brick-level loop {d['grid_steps'] / 1e6:8.1f} M lane-steps in "
                  f"{d['wave_grid_trips'] / 1e6:7.2f} M wave-trips = {d['grid_steps'] / max(1, d['wave_grid_trips']):5.1f} lanes per trip; voxel loop "
                  f"{d['voxel_steps'] / 1e6:7.1f} M lane-steps in {d['wave_voxel_trips'] / 1e6:6.2f} M wave-trips = {d['voxel_steps'] / max(1, d['wave_voxel_trips']):5.1f} lanes per trip; "
                  f"{d['bricks_entered'] / max(1, d['wave_brick_walks']):5.1f} lanes per brick walk", flush=True)
