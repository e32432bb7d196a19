#!/usr/bin/env python3
"""Same-box A/B of two tuning_flags settings on one workload: alternating frames of two contexts over the same grid, HIP-event kernel
time per frame (trace + resolve where there is one), frames compared bit for bit.
usage: flags_ab.py <flags A> <flags B> [workload] [view ...]        (AB_REPS frames each, default 4; AB_VARIANT kernel_variant)"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zig_vulkan_amd import workloads as W  # noqa: E402

fa, fb = int(sys.argv[1], 0), int(sys.argv[2], 0)
name = sys.argv[3] if len(sys.argv) > 3 else "cfg4_4k_2048c_b8_sparse"
views = sys.argv[4:] or ["V0", "V1", "V2", "V1x"]
w = W.WORKLOADS[name]
grid = W.build_grid(w)
variant = int(os.environ.get("AB_VARIANT", "0"), 0)
a = W.make_renderer(w, grid, tuning_flags=fa, kernel_variant=variant)
b = W.make_renderer(w, grid, tuning_flags=fb, kernel_variant=variant)
reps = int(os.environ.get("AB_REPS", "4"))
sa = sb = 0.0
for v in views:
    for rt in (a, b):
        W.set_view(rt, v)
        rt.draw(); rt.wait()   # the library learns the box of the occupied cells behind this frame
        rt.draw(); rt.wait()
    ta, tb = [], []
    for _ in range(reps):
        a.draw(); ta.append(a.last_kernel_ms())
        b.draw(); tb.append(b.last_kernel_ms())
    ha = hashlib.sha256(a.read_rgba8().tobytes()).hexdigest()[:16]
    hb = hashlib.sha256(b.read_rgba8().tobytes()).hexdigest()[:16]
    sa += min(ta); sb += min(tb)
    print(f"{name} {v}: flags {fa:#x} {a.kernel_name()} {min(ta):.3f} ms (all {[round(t, 3) for t in ta]}) | flags {fb:#x} {b.kernel_name()} {min(tb):.3f} ms "
          f"(all {[round(t, 3) for t in tb]}) | {100.0 * (min(ta) / min(tb) - 1.0):+.1f} % | frames {'EQUAL' if ha == hb else 'DIFFER'}", flush=True)
print(f"{name} mean over {len(views)} views: {sa / len(views):.3f} vs {sb / len(views):.3f} ms ({100.0 * (sa / sb - 1.0):+.1f} %)")
a.deinit(); b.deinit()
