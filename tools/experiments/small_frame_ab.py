#!/usr/bin/env python3
"""Small frames (configs[0]): kernel time per view, single stream, with and without the split of every tile into several
workgroups of narrower waves (VRT_TUNE_NO_SMALL_FRAME_SPLIT), frames hashed.  usage: small_frame_ab.py [workload] [library ...]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg0_256x256_64c_b4"
libs = sys.argv[2:] or [None]
w = W.WORKLOADS[name]
g = W.build_grid(w)
for lib in libs:
    for flags in (0, 16384):
        rt = W.make_renderer(w, g, tuning_flags=flags, frames_in_flight=1, **({"library": lib} if lib else {}))
        out = []
        for v in ("V0", "V1", "V2", "V1x", "VG"):
            W.set_view(rt, v)
            rt.draw(frames=50)
            rt.draw(frames=200)
            ms = rt.last_kernel_ms()
            out.append(f"{v} {ms * 1e3:.2f}us [{hashlib.sha256(rt.read_rgba8().tobytes()).hexdigest()[:6]}]")
        print(name, os.path.basename(lib or "libvrt_hip.so"), "no split" if flags else "split", rt.kernel_name(), " ".join(out), flush=True)
        rt.deinit()
