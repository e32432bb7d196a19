#!/usr/bin/env python3
"""Occupancy over time of one frame of the headline workload from per-wave timestamps
(vrt_trace_wave_timeline): how many waves are resident in each 5 % slice of the kernel's duration."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W  # noqa: E402

variant = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
w = W.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else W.HEADLINE]
grid = W.build_grid(w)
rt = W.make_renderer(w, grid, kernel_variant=variant)
for view in ["V0", "V1", "V2"]:
    W.set_view(rt, view)
    rt.draw(frames=40)  # lets the cost-feedback tile schedule (default order) settle on this view
    rt.draw(); rt.wait()
    t = rt.wave_timeline().astype(np.int64)
    t0, t1 = t[:, 0].min(), t[:, 1].max()
    span = (t1 - t0) / 100.0  # us
    dur = (t[:, 1] - t[:, 0]) / 100.0
    edges = np.linspace(t0, t1, 21)
    occ = []
    for a, b in zip(edges[:-1], edges[1:]):
        overlap = np.clip(np.minimum(t[:, 1], b) - np.maximum(t[:, 0], a), 0, None).sum() / (b - a)
        occ.append(overlap)
    print(f"{view}: span {span:.1f} us, waves {len(t)}, wave duration mean {dur.mean():.1f} us p50 {np.median(dur):.1f} p99 {np.percentile(dur, 99):.1f} max {dur.max():.1f}; "
          f"mean resident waves {dur.sum() / span:.0f} (capacity 6144 at 6 waves/SIMD)")
    print("   resident waves per 5% slice:", " ".join(f"{o:.0f}" for o in occ))
rt.deinit()
