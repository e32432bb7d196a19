#!/usr/bin/env python3
"""Small bounce frames through vrt_path_kernel with a tuning flag set, against the oracle (development aid).
usage: flag_check.py <flags> [b] [sparse]     (sparse: a field of spheres whose occupied cells reach the grid's faces)"""
import sys
sys.path.insert(0, ".")
import numpy as np
from tests.helpers import O, oracle_scene_from_grid
from zig_vulkan_amd import workloads as W
flags = int(sys.argv[1], 0)
b = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sparse = len(sys.argv) > 3 and sys.argv[3] == "sparse"
for (wd, ht, vox, spp, bounce) in (((96, 64, 256, 1, 2), (250, 131, 512, 2, 2)) if sparse else ((64, 48, 64, 1, 1), (250, 131, 128, 2, 2))):
    w = W.Workload("t", wd, ht, vox, b, spp, bounce, True, 5.0, "sparse", 0.08, 60000) if sparse else W.Workload("t", wd, ht, vox, b, spp, bounce, True, 5.0)
    grid = W.build_grid(w)
    rt = W.make_renderer(w, grid, want_float_output=True, kernel_variant=1 << 23, tuning_flags=flags)
    for view in ("V0", "V2", "V1x", "V0", "V1"):
        W.set_view(rt, view)
        rt.draw()
        f = rt.read_rgba32f()
        fo, uo, _ = O.render(oracle_scene_from_grid(grid), O.push_constants(rt.camera.blob(), rt.sun.blob()))
        bad = int((f.view(np.uint32) != fo.view(np.uint32)).any(axis=2).sum())
        print(f"{wd}x{ht} {vox}^3 b{b} spp{spp} bounce{bounce} {view} {rt.kernel_name()}: {bad} of {wd * ht} pixels differ", flush=True)
    rt.deinit()
