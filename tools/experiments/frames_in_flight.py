#!/usr/bin/env python3
"""Frame time of the headline workload per view against the number of frames in flight: one context with
frames_in_flight 1 and 2, and two such contexts side by side (four frames in flight)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W
w = W.WORKLOADS[W.HEADLINE]
grid = W.build_grid(w)
for fif, nctx in ((1, 1), (2, 1), (2, 2)):
    try:
        rts = [W.make_renderer(w, grid, frames_in_flight=fif) for _ in range(nctx)]
    except Exception as e:
        print(f"frames_in_flight {fif}: {e}")
        continue
    out = []
    for view in ["V0", "V1", "V2"]:
        for rt in rts: W.set_view(rt, view)
        for _ in range(10):
            for rt in rts: rt.draw()
        for rt in rts: rt.wait()
        t0 = time.perf_counter()
        n = 240
        for i in range(n // nctx):
            for rt in rts: rt.draw()
        for rt in rts: rt.wait()
        out.append(f"{view} {(time.perf_counter() - t0) / n * 1e6:.1f}")
    print(f"{nctx} context(s) x frames_in_flight {fif}: us per frame  " + "  ".join(out))
    for rt in rts: rt.deinit()
