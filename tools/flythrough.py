#!/usr/bin/env python3
"""The reference's own benchmark protocol on the HIP path: the scripted 60 s fly-through of
src/modules/voxel_rt/Benchmark.zig replayed at a fixed simulated frame rate; reports the
min / max / avg frame time like Benchmark.Report.print (Benchmark.zig:109-136), from HIP events.

    python tools/flythrough.py [workload] [simulated_fps]
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zig_vulkan_amd import workloads as W  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else W.HEADLINE
fps = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
w = W.WORKLOADS[name]
grid = W.build_grid(w)
rt = W.make_renderer(w, grid, enable_counters=True)
bench = rt.create_benchmark()
times, rays = [], 0
done = False
while not done:
    rt.draw()
    times.append(rt.last_kernel_ms())
    rays += rt.counters()["rays"]
    done = bench.update(1.0 / fps)
rt.deinit()
# the counters build is slow; time the same path again without counters
rt = W.make_renderer(w, grid)
bench = rt.create_benchmark()
times = []
done = False
while not done:
    rt.draw()
    times.append(rt.last_kernel_ms())
    done = bench.update(1.0 / fps)
rt.deinit()
print(json.dumps({"workload": w.name, "frames": len(times), "simulated_fps": fps, "min_frame_ms": min(times), "max_frame_ms": max(times),
                  "avg_frame_ms": sum(times) / len(times), "rays": rays, "Mrays_per_s": rays / (sum(times) * 1e-3) / 1e6}))
