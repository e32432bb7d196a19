#!/usr/bin/env python3
"""What a frame of nothing but sky costs (the fixed part of every wave: launch, kernel arguments, ray set-up, slab test, background,
store): the camera above the grid looking away from it.  Kernel time by HIP events, single stream, and the per-wave timeline.
usage: sky_frame.py [workload]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W  # noqa: E402

w = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else W.HEADLINE]
grid = W.build_grid(w)
for variant in (0, 0x30000, 0x130000):
    rt = W.make_renderer(w, grid, kernel_variant=variant)
    rt.camera.look_at((0.0, -40.0, 0.0), (0.0, -80.0, 1.0))   # (world is Y-down: above the grid, looking up)
    rt.draw(frames=80)
    ts = []
    for _ in range(200):
        rt.draw(); ts.append(rt.last_kernel_ms())
    t = rt.wave_timeline().astype(np.int64)
    t0 = t[:, 0].min()
    start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
    dur = end - start
    nz = np.count_nonzero(rt.read_rgba8().view(np.uint32) != rt.read_rgba8().view(np.uint32).flat[0])
    print(f"{w.name} variant {variant:#x} {rt.kernel_name()}: kernel median {np.median(ts) * 1e3:.1f} us; timeline span {end.max():.1f} us, {len(t)} waves, "
          f"wave duration mean {dur.mean():.2f} p90 {np.percentile(dur, 90):.2f} max {dur.max():.2f} us; waves started per us (median over the span) "
          f"{np.median(np.histogram(start, bins=max(1, int(end.max())))[0]):.0f}; mean resident waves {dur.sum() / end.max():.0f}; pixels differing from the first: {nz}")
    rt.deinit()
