#!/usr/bin/env python3
"""The tail of one frame under the cost schedule: when the last-finishing waves started and how long they ran, and how long the
longest waves waited for their slot (vrt_trace_wave_timeline).  usage: timeline_tail.py [workload] [views]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W  # noqa: E402

w = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "refapp_1024x576_128x64x128_b4"]
views = sys.argv[2].split(",") if len(sys.argv) > 2 else ["V0", "V1", "V2"]
grid = W.build_grid(w)
rt = W.make_renderer(w, grid)
for view in views:
    W.set_view(rt, view)
    rt.draw(frames=80)
    rt.draw(); rt.wait()
    t = rt.wave_timeline().astype(np.int64)
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0) / 100.0
    end = (t[:, 1] - t0) / 100.0
    dur = end - start
    span = end.max()
    last = np.argsort(-end)[:12]
    longest = np.argsort(-dur)[:12]
    print(f"{view}: span {span:.1f} us, {len(t)} waves, sum of durations {dur.sum() / 1e3:.1f} wave-ms")
    print("   last to finish   (start -> end, us): " + "  ".join(f"{start[i]:.0f}->{end[i]:.0f}" for i in last))
    print("   longest          (start -> end, us): " + "  ".join(f"{start[i]:.0f}->{end[i]:.0f}" for i in longest))
    edges = [0.0, 1.0] + [span * k / 8 for k in range(1, 9)]
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (start >= lo) & (start < hi + (1e-9 if hi == span else 0))
        if m.any():
            print(f"   started in [{lo:6.1f}, {hi:6.1f}) us: {m.sum():6d} waves, duration mean {dur[m].mean():6.1f} p90 {np.percentile(dur[m], 90):6.1f} max {dur[m].max():6.1f}, last end {end[m].max():6.1f}")
    res = [np.clip(np.minimum(end, b) - np.maximum(start, a), 0, None).sum() / (b - a) for a, b in zip(np.linspace(0, span, 11)[:-1], np.linspace(0, span, 11)[1:])]
    print("   resident waves per tenth of the span: " + " ".join(f"{x:.0f}" for x in res))
rt.deinit()
