#!/usr/bin/env python3
"""Host -> HBM rate of the boundary's upload path (pinned staging ring over PCIe) for the headline scene,
and what a frame costs when the whole scene is re-uploaded before it (never part of bench.py's `value`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import _lib as L
from zig_vulkan_amd import workloads as W
from zig_vulkan_amd._lib import lib, check
w = W.WORKLOADS[W.HEADLINE]
grid = W.build_grid(w)
rt = W.make_renderer(w, grid)
W.set_view(rt, "V2")
rt.draw(); rt.wait()
nbytes = sum(rt.buffer_size(b) for b in range(7))
for _ in range(2):
    t0 = time.perf_counter()
    check(lib.vrt_upload_grid(rt._h, grid._h), rt._h)
    rt.draw(); rt.wait()
    dt = time.perf_counter() - t0
print(f"full scene upload + one frame: {dt * 1e3:.2f} ms for {nbytes / 2**20:.1f} MiB -> {nbytes / dt / 1e9:.1f} GB/s PCIe-inclusive; "
      f"frame alone {rt.last_kernel_ms():.3f} ms")
for y in range(40, 200):
    grid.insert(100, y, 100, 7)
t0 = time.perf_counter(); rt.update_grid_delta(); rt.draw(); rt.wait(); dt = time.perf_counter() - t0
print(f"delta upload of a 160-voxel edit + one frame: {dt * 1e3:.3f} ms")
rt.deinit()
