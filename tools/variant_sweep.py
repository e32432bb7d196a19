#!/usr/bin/env python3
"""Kernel time per view of one workload under several kernel_variant values (single stream, HIP events).
usage: variant_sweep.py <workload> <variant,variant,...> [frames] [views]"""
import sys
sys.path.insert(0, ".")
from zig_vulkan_amd import workloads as W

name = sys.argv[1]
variants = [int(v, 0) for v in sys.argv[2].split(",")]
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 4
views = sys.argv[4].split(",") if len(sys.argv) > 4 else ["V0", "V1", "V2", "V1x", "VG"]
w = W.WORKLOADS[name]
grid = W.build_grid(w)
for variant in variants:
    try:
        rt = W.make_renderer(w, grid, kernel_variant=variant)
    except Exception as e:  # noqa: BLE001
        print(f"variant {variant:#x}: {e}")
        continue
    out = []
    for v in views:
        W.set_view(rt, v)
        rt.draw(frames=max(2, frames // 2))
        rt.draw(frames=frames)
        out.append(f"{v} {rt.last_kernel_ms():9.3f}")
    print(f"variant {variant:#010x} {rt.kernel_name():70s} " + "  ".join(out), flush=True)
    rt.deinit()
