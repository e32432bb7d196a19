#!/usr/bin/env python3
"""Same-box A/B of vrt_pool_kernel (round 4: a pool of 128 rays per wave) against vrt_path_kernel (a ray per lane) on one workload:
alternating frames of two contexts over the same grid, HIP-event kernel time, frames compared bit for bit.
usage: pool_ab.py [workload] [view ...]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W  # noqa: E402

NO_POOL = 1 << 13
name = sys.argv[1] if len(sys.argv) > 1 else "cfg4_4k_2048c_b8_sparse"
views = sys.argv[2:] or ["V0", "V1x"]
w = W.WORKLOADS[name]
grid = W.build_grid(w)
kw = dict(width=int(os.environ.get("AB_WIDTH", w.width)), height=int(os.environ.get("AB_HEIGHT", w.height)))
a = W.make_renderer(w, grid, tuning_flags=0, **kw)
b = W.make_renderer(w, grid, tuning_flags=NO_POOL, **kw)
for v in views:
    for rt in (a, b):
        W.set_view(rt, v)
        rt.draw(); rt.wait()   # the library learns the box of the occupied cells behind this frame
        rt.draw(); rt.wait()
    ta, tb = [], []
    for _ in range(int(os.environ.get("AB_REPS", "3"))):
        a.draw(); ta.append(a.last_kernel_ms())
        b.draw(); tb.append(b.last_kernel_ms())
    ha = hashlib.sha256(a.read_rgba8().tobytes()).hexdigest()[:16]
    hb = hashlib.sha256(b.read_rgba8().tobytes()).hexdigest()[:16]
    print(f"{name} {v}: {a.kernel_name()} {min(ta):.2f} ms (all {[round(t, 2) for t in ta]}) | {b.kernel_name()} {min(tb):.2f} ms (all {[round(t, 2) for t in tb]}) | "
          f"frames {'EQUAL' if ha == hb else 'DIFFER'} {ha} {hb}", flush=True)
a.deinit(); b.deinit()
