#!/usr/bin/env python3
"""Where a short timed region loses against a long one (20 steps under the driver's protocol gave 13 % less
than 600): the bench's timed region for several step counts, with the host's submission time split out, on one box.
usage: short_run.py [workload] [frames_in_flight]"""
import sys
import time
sys.path.insert(0, ".")
import numpy as np
import torch
from zig_vulkan_amd import workloads as W

name = sys.argv[1] if len(sys.argv) > 1 else W.HEADLINE
fif = int(sys.argv[2]) if len(sys.argv) > 2 else 2
w = W.WORKLOADS[name]
grid = W.build_grid(w)
rt = W.make_renderer(w, grid, frames_in_flight=fif)
views = ["V0", "V1", "V2"]


def region(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    per_call = []
    for i in range(n):
        W.set_view(rt, views[min(2, i * 3 // n)])
        a = time.perf_counter()
        rt.draw()
        per_call.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    rt.wait()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    return (t3 - t0) / n * 1e3, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, np.array(per_call) * 1e6


for _ in range(30):
    rt.draw()
rt.wait()
for n in (20, 20, 20, 60, 200, 600, 20):
    ms, sub, wait, sync, pc = region(n)
    print(f"n={n:4d}: {ms:.4f} ms/step  total {ms * n:.3f} ms = submit {sub:.3f} + wait {wait:.3f} + sync {sync:.3f}; "
          f"host us per dispatch: first {pc[0]:.1f}, median {np.median(pc):.1f}, max {pc.max():.1f}")
rt.deinit()
