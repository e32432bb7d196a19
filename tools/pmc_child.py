#!/usr/bin/env python3
"""A few frames of one workload for a rocprofv3 --pmc pass (tools/pmc_path.sh).
usage: pmc_child.py <workload> <variant> <tuning_flags> [frames] [view]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zig_vulkan_amd import workloads as W
w = W.WORKLOADS[sys.argv[1]]
variant, flags = int(sys.argv[2], 0), int(sys.argv[3], 0)
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 2
view = sys.argv[5] if len(sys.argv) > 5 else "V0"
grid = W.build_grid(w)
lib = os.environ.get("VRT_SWEEP_LIB")
rt = W.make_renderer(w, grid, kernel_variant=variant, tuning_flags=flags, **({"library": lib} if lib else {}))
W.set_view(rt, view)
for _ in range(frames):
    rt.draw()
    rt.wait()   # (behind a finished frame the library knows the box of the occupied cells: the next frame takes the grid-exit kernel)
print(rt.kernel_name(), rt.last_kernel_ms())
rt.deinit()
