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
# The other direction (round 6): frames handed back to the host through the boundary's vrt_read_rgba8 (8.3 MB per 1080p frame over PCIe),
# the hand-off a host WITHOUT device interop uses — never part of bench.py's `value`.
rt = W.make_renderer(w, grid, frames_in_flight=2)
rays = {}
cnt = W.make_renderer(w, grid, enable_counters=True)
for v in ("V0", "V1", "V2"):
    W.set_view(cnt, v); cnt.draw(); rays[v] = cnt.counters()["rays"]
cnt.deinit()
for _ in range(20):
    rt.draw(); rt.read_rgba8()
n, total = 300, 0
t0 = time.perf_counter()
for i in range(n):
    v = ("V0", "V1", "V2")[i % 3]
    W.set_view(rt, v)
    rt.draw()
    rt.read_rgba8()          # waits for the frame, copies it to host memory
    total += rays[v]
dt = time.perf_counter() - t0
print(f"{n} frames each read back to the host (vrt_read_rgba8): {dt / n * 1e3:.3f} ms per frame = {total / dt / 1e9:.2f} Grays/s PCIe-inclusive "
      f"({w.width * w.height * 4 / (dt / n) / 1e9:.1f} GB/s of pixels)")
rt.deinit()
