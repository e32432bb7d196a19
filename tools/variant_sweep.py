#!/usr/bin/env python3
"""Kernel time per view of one workload under several kernel_variant values (single stream, HIP events).
usage: variant_sweep.py <workload> <variant[/tuning_flags],variant,...> [frames] [views]      (e.g. 0,0/0x20,0/0x60)
environment: VRT_SWEEP_LIB = path of another build of the library (the development build for its variants)"""
import sys
sys.path.insert(0, ".")
from zig_vulkan_amd import workloads as W

name = sys.argv[1]
import os
variants = [tuple(int(x, 0) for x in (v.split("/") + ["0"])[:2]) for v in sys.argv[2].split(",")]
library = os.environ.get("VRT_SWEEP_LIB")
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 4
views = sys.argv[4].split(",") if len(sys.argv) > 4 else ["V0", "V1", "V2", "V1x", "VG"]
w = W.WORKLOADS[name]
grid = W.build_grid(w)
for variant, flags in variants:
    try:
        rt = W.make_renderer(w, grid, kernel_variant=variant, tuning_flags=flags, **({"library": library} if library else {}))
    except Exception as e:  # noqa: BLE001
        print(f"variant {variant:#x}: {e}")
        continue
    out = []
    W.set_view(rt, views[0])
    rt.draw()
    rt.wait()   # (behind the first frame the library knows the box of the occupied cells: kernel choices that depend on it have settled)
    for v in views:
        W.set_view(rt, v)
        rt.draw(frames=max(2, frames // 2))
        rt.draw(frames=frames)
        import hashlib
        out.append(f"{v} {rt.last_kernel_ms():9.4f} [{hashlib.sha256(rt.read_rgba8().tobytes()).hexdigest()[:8]}]")
    print(f"variant {variant:#010x} flags {flags:#04x} {rt.kernel_name():44s} " + "  ".join(out), flush=True)
    rt.deinit()
