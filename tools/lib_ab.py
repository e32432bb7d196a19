#!/usr/bin/env python3
"""Same-box, same-process A/B of two builds of libvrt_hip on one workload: two contexts over the same grid, alternating frames,
HIP-event kernel time (min and all), frames compared bit for bit.
usage: lib_ab.py <libA.so> <libB.so> [workload] [view ...]        env: AB_REPS (3), AB_FLAGS_A / AB_FLAGS_B (tuning flags), AB_VARIANT_A / AB_VARIANT_B (kernel_variant),
AB_WIDTH / AB_HEIGHT / AB_SPP (another frame size / sample count than the workload's)"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zig_vulkan_amd import workloads as W  # noqa: E402

la, lb = os.path.abspath(sys.argv[1]), os.path.abspath(sys.argv[2])
name = sys.argv[3] if len(sys.argv) > 3 else "cfg4_4k_2048c_b8_sparse"
views = sys.argv[4:] or ["V0"]
w = W.WORKLOADS[name]
grid = W.build_grid(w)
size = {k: int(os.environ[e]) for k, e in (("width", "AB_WIDTH"), ("height", "AB_HEIGHT")) if e in os.environ}
a = W.make_renderer(w, grid, library=la, **size, tuning_flags=int(os.environ.get("AB_FLAGS_A", "0"), 0), kernel_variant=int(os.environ.get("AB_VARIANT_A", "0"), 0))
b = W.make_renderer(w, grid, library=lb, **size, tuning_flags=int(os.environ.get("AB_FLAGS_B", "0"), 0), kernel_variant=int(os.environ.get("AB_VARIANT_B", "0"), 0))
reps = int(os.environ.get("AB_REPS", "3"))
if "AB_SPP" in os.environ:
    a.camera.d_camera.samples_per_pixel = b.camera.d_camera.samples_per_pixel = int(os.environ["AB_SPP"])
for v in views:
    for rt in (a, b):
        W.set_view(rt, v)
        rt.draw(); rt.wait()   # the library learns the box of the occupied cells behind this frame
        rt.draw(); rt.wait()
    ta, tb = [], []
    for _ in range(reps):
        a.draw(); a.wait(); ta.append(a.last_kernel_ms())
        b.draw(); b.wait(); tb.append(b.last_kernel_ms())
    ha = hashlib.sha256(a.read_rgba8().tobytes()).hexdigest()[:16]
    hb = hashlib.sha256(b.read_rgba8().tobytes()).hexdigest()[:16]
    sa, sb = sorted(ta), sorted(tb)
    print(f"{name} {v}: A {os.path.basename(la)} {a.kernel_name()} min {sa[0]:.4f} med {sa[len(sa) // 2]:.4f} ms | "
          f"B {os.path.basename(lb)} {b.kernel_name()} min {sb[0]:.4f} med {sb[len(sb) // 2]:.4f} ms | "
          f"B/A {sb[len(sb) // 2] / sa[len(sa) // 2]:.4f} | frames {'EQUAL' if ha == hb else 'DIFFER'} {ha} {hb}", flush=True)
a.deinit(); b.deinit()
