#!/usr/bin/env python3
"""vrt_pool_kernel's phase rule and occupancy on one workload (development build: the VRT_DEV_POOL_* knobs are read at vrt_create):
kernel time of one view per setting, single stream, HIP events, the frame's hash beside it.
usage: pool_sweep.py <workload> <view> K:B:T:W[,K:B:T:W...] [waves:slots:stages,...]
       (walk_k : brick_thr : trans_thr : walk_min; waves per SIMD : LDS slots of another compiled vrt_pool_kernel)"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W  # noqa: E402

name, view = sys.argv[1], sys.argv[2]
settings = [tuple(int(x) for x in s.split(":")) for s in sys.argv[3].split(",")]
kernels = sys.argv[4].split(",") if len(sys.argv) > 4 else [""]
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "zig_vulkan_amd", "libvrt_hip_dev.so")
w = W.WORKLOADS[name]
grid = W.build_grid(w)
for kern in kernels:
    for k, b, t, wm in settings:
        if kern:
            os.environ["VRT_DEV_POOL_KERNEL"] = kern
        os.environ.update(VRT_DEV_POOL_WALK_K=str(k), VRT_DEV_POOL_BRICK_THR=str(b), VRT_DEV_POOL_TRANS_THR=str(t), VRT_DEV_POOL_WALK_MIN=str(wm))
        rt = W.make_renderer(w, grid, library=lib, kernel_variant=int(os.environ.get("SWEEP_VARIANT", "0"), 0))
        W.set_view(rt, view)
        rt.draw(); rt.wait()
        rt.draw(); rt.wait()
        ts = []
        for _ in range(2):
            rt.draw(); ts.append(rt.last_kernel_ms())
        h = hashlib.sha256(rt.read_rgba8().tobytes()).hexdigest()[:8]
        print(f"walk_k {k:2d} brick_thr {b:2d} trans_thr {t:2d} walk_min {wm:2d}: {min(ts):8.2f} ms  {rt.kernel_name()} [{h}]", flush=True)
        rt.deinit()
