#!/usr/bin/env python3
"""What each bounce of the reference app's run costs: the same scene and views at max_bounce 0 / 1 / 2 and 1 / 2 samples per pixel,
kernel time by HIP events (settled: 40 frames first), rays counted by a counting context.
usage: bounce_cost.py [workload]"""
import dataclasses
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "refapp_1024x576_128x64x128_b4"
base = W.WORKLOADS[name]
grid = W.build_grid(base)
for spp in (1, 2):
    for mb in (0, 1, 2):
        w = dataclasses.replace(base, spp=spp, max_bounce=mb)
        rt = W.make_renderer(w, grid)
        cnt = W.make_renderer(w, grid, enable_counters=True)
        out = []
        for v in ("V0", "V1", "V2"):
            W.set_view(rt, v)
            W.set_view(cnt, v)
            cnt.draw()
            rays = cnt.counters()["rays"]
            for _ in range(40):
                rt.draw()
            rt.wait()
            ts = []
            for _ in range(8):
                rt.draw()
                ts.append(rt.last_kernel_ms())
            out.append(f"{v}: {min(ts) * 1e3:7.1f} us, {rays / 1e6:5.2f} M rays, {rays / min(ts) / 1e6:6.2f} Grays/s")
        print(f"spp {spp} max_bounce {mb} [{rt.kernel_name()}]  " + " | ".join(out), flush=True)
        rt.deinit()
        cnt.deinit()
