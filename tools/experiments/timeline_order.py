#!/usr/bin/env python3
"""Launch position against start time and duration of the workgroups of one frame under the cost schedule: where in the launch order
the last-finishing workgroups stood.  usage: timeline_order.py [workload] [views]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W  # noqa: E402

w = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else W.HEADLINE]
views = sys.argv[2].split(",") if len(sys.argv) > 2 else ["V1", "V2"]
grid = W.build_grid(w)
rt = W.make_renderer(w, grid)
for view in views:
    W.set_view(rt, view)
    rt.draw(frames=80)
    rt.draw(); rt.wait()
    t = rt.wave_timeline(raw=True).astype(np.int64).reshape(-1, 4, 2)   # [workgroup][wave][begin, end]
    live = t[:, :, 1].max(axis=1) != 0
    pos = np.nonzero(live)[0]
    t = t[live]
    t0 = t[:, :, 0].min()
    start = (t[:, :, 0].min(axis=1) - t0) / 100.0
    end = (t[:, :, 1].max(axis=1) - t0) / 100.0
    dur = end - start
    span = end.max()
    print(f"{view}: span {span:.1f} us, {len(pos)} workgroups launched of {len(live)} entries")
    last = np.argsort(-end)[:8]
    print("   last to finish: " + "  ".join(f"pos {pos[i]} (xcd {pos[i] % 8}) {start[i]:.0f}->{end[i]:.0f}" for i in last))
    # rank of each workgroup's duration against its launch position
    rank = np.argsort(np.argsort(-dur))
    for i in last[:4]:
        print(f"   pos {pos[i]}: duration {dur[i]:.1f} us is rank {rank[i]} of {len(dur)}; workgroups launched before it with a shorter duration: {(dur[:i] < dur[i]).sum()}")
    q = np.linspace(0, len(pos), 9).astype(int)
    print("   by launch position (eighths): start mean / duration mean / duration max: " +
          "  ".join(f"{start[a:b].mean():.0f}/{dur[a:b].mean():.0f}/{dur[a:b].max():.0f}" for a, b in zip(q[:-1], q[1:])))
    for x in range(8):
        m = (pos % 8) == x
        print(f"   xcd {x}: {m.sum()} workgroups, sum of durations {dur[m].sum() / 1e3:.2f} ms, last end {end[m].max():.1f}")
rt.deinit()
