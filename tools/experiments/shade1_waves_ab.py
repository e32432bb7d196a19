#!/usr/bin/env python3
"""The several-samples kernel (vrt_trace_kernel<B, false, MODE, W, 1>) at W = 6 (the product's) against W = 5 / 7 waves per SIMD on the
development build (VRT_DEV_SHADE1_WAVES is read when a context is made): two contexts over the same grid, alternating frames, HIP-event
kernel time, frames hashed.   usage: VRT_HIP_LIB=zig_vulkan_amd/libvrt_hip_dev.so shade1_waves_ab.py [workload] [waves ...]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W  # noqa: E402
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3_4k_1024c_b8"
others = [int(x) for x in sys.argv[2:]] or [7]
w = W.WORKLOADS[name]
grid = W.build_grid(w)
os.environ["VRT_DEV_SHADE1_WAVES"] = "6"
a = W.make_renderer(w, grid)
for waves in others:
    os.environ["VRT_DEV_SHADE1_WAVES"] = str(waves)
    b = W.make_renderer(w, grid)
    for v in ("V0", "V1", "V2", "V1x", "VG"):
        for rt in (a, b):
            W.set_view(rt, v)
            rt.draw(); rt.wait(); rt.draw(); rt.wait()
        ta, tb = [], []
        for _ in range(int(os.environ.get("AB_REPS", "9"))):
            a.draw(); a.wait(); ta.append(a.last_kernel_ms())
            b.draw(); b.wait(); tb.append(b.last_kernel_ms())
        ha, hb = (hashlib.sha256(r.read_rgba8().tobytes()).hexdigest()[:12] for r in (a, b))
        ta.sort(); tb.sort()
        print(f"{name} {v}: 6 waves [{a.kernel_name()}] med {ta[len(ta) // 2]:.4f} ms | {waves} waves [{b.kernel_name()}] med {tb[len(tb) // 2]:.4f} ms | "
              f"ratio {tb[len(tb) // 2] / ta[len(ta) // 2]:.4f} | frames {'EQUAL' if ha == hb else 'DIFFER'}", flush=True)
    b.deinit()
a.deinit()
