#!/usr/bin/env python3
"""Lane utilisation of the shadow-ray pass of the headline workload (counting build, sun on minus sun off): would compacting
shadow rays across waves pay?  (No: 59-62 of 64 lanes are active per brick-level trip of the shadow pass.)"""
import os
import sys, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W
base = W.WORKLOADS[W.HEADLINE]
grid = W.build_grid(base)
res = {}
for sun in (False, True):
    w = dataclasses.replace(base, sun_enabled=sun)
    rc = W.make_renderer(w, grid, enable_counters=True)
    for view in ["V0", "V1", "V2"]:
        W.set_view(rc, view); rc.draw(); rc.wait()
        c, wc = rc.counters(), rc.wave_counters()
        res[(sun, view)] = (c["grid_steps"], wc["wave_grid_iters"], c["voxel_steps"], wc["wave_voxel_iters"], c["rays"])
    rc.deinit()
for view in ["V0", "V1", "V2"]:
    a, b = res[(False, view)], res[(True, view)]
    print(view, "shadow pass: lane steps", b[0]-a[0], "wave trips", b[1]-a[1], "-> lanes/trip %.1f" % ((b[0]-a[0])/max(1,b[1]-a[1])),
          "| voxel lanes/trip %.1f" % ((b[2]-a[2])/max(1,b[3]-a[3])), "| shadow rays", b[4]-a[4], "| primary wave trips", a[1])
