#!/usr/bin/env python3
"""Two frames in flight on builds of the lockstep bounce kernel at other wave counts (development library: kernel_variant bits 8-15 =
waves per SIMD asked for), wall clock per frame over back-to-back draws.  usage: fif_waves_ab.py [workload] [frames]
env: FIF_LIB (another library), FIF_VARIANTS (comma-separated kernel_variant values)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W
w = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "refapp_1024x576_128x64x128_b4"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
lib = os.environ.get("FIF_LIB") or os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "zig_vulkan_amd", "libvrt_hip_dev.so")
variants = [int(v, 0) for v in os.environ.get("FIF_VARIANTS", "0,0x500,0x600,0x800").split(",")]
grid = W.build_grid(w)
for rep in range(2):
    for fif in (1, 2):
        for variant in variants:
            try:
                rt = W.make_renderer(w, grid, frames_in_flight=fif, kernel_variant=variant, library=lib)
            except Exception as e:  # noqa: BLE001
                print(f"variant {variant:#x}: {e}")
                continue
            out = []
            for view in ["V0", "V1", "V2"]:
                W.set_view(rt, view)
                for _ in range(80):
                    rt.draw()
                rt.wait()
                t0 = time.perf_counter()
                for i in range(n):
                    rt.draw()
                rt.wait()
                out.append(f"{view} {(time.perf_counter() - t0) / n * 1e3:.4f}")
            print(f"frames_in_flight {fif} variant {variant:#x} {rt.kernel_name()}: ms per frame  " + "  ".join(out), flush=True)
            rt.deinit()
