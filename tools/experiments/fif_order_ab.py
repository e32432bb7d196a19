#!/usr/bin/env python3
"""Two frames in flight: the tile order of the frames.  The library's choice for such frames (reverse raster, no split tiles) against
the cost schedule on both streams (kernel_variant 0x70000), wall clock per frame over back-to-back draws.
usage: fif_order_ab.py [workload] [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W
w = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "refapp_1024x576_128x64x128_b4"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
grid = W.build_grid(w)
for rep in range(2):
    for fif, variant in ((1, 0), (2, 0), (2, 0x70000)):
        rt = W.make_renderer(w, grid, frames_in_flight=fif, kernel_variant=variant)
        out = []
        for view in ["V0", "V1", "V2"]:
            W.set_view(rt, view)
            for _ in range(80):
                rt.draw()
            rt.wait()
            t0 = time.perf_counter()
            for i in range(n):
                rt.draw()
            rt.wait()
            out.append(f"{view} {(time.perf_counter() - t0) / n * 1e3:.4f}")
        print(f"frames_in_flight {fif} variant {variant:#x} {rt.kernel_name()}: ms per frame  " + "  ".join(out), flush=True)
        rt.deinit()
