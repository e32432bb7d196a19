#!/usr/bin/env python3
"""How many frames in flight fill the GPU?  The plain two-stream dispatch against the multi-GPU pipeline's launch slots with ONE rank
(world 1: no collective; each slot its own stream, plus the un-swizzle of the packed tiles) at 2 / 3 / 4 / 6 slots.
usage: streams_probe.py [workload]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from zig_vulkan_amd import workloads as W  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else W.HEADLINE
w = W.WORKLOADS[name]
grid = W.build_grid(w)
FAKE = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "fake_rccl", "libfake_rccl.so")


def timed(submit, wait, n=600, warm=150):
    out = {}
    for v in ("V0", "V1", "V2"):
        submit(v, warm)
        wait()
        t0 = time.perf_counter()
        submit(v, n)
        wait()
        out[v] = (time.perf_counter() - t0) / n * 1e6
    return out


rt = W.make_renderer(w, grid, frames_in_flight=2)
def sub(v, k):
    W.set_view(rt, v)
    for _ in range(k):
        rt.draw()
r = timed(sub, rt.wait)
print(f"{name} plain dispatch, 2 frames in flight [{rt.kernel_name()}]: " + " | ".join(f"{v} {t:.1f} us" for v, t in r.items()) + f" | mean {sum(r.values()) / 3:.1f}", flush=True)
rt.deinit()
for slots in (1, 2, 3, 4, 6, 8):
    rt = W.make_renderer(w, grid, shard_rank=0, shard_count=1)
    rt.dist_init(b"streams-probe" + bytes([slots]) + os.urandom(16) + bytes(128 - 30), 0, 1, frames_in_flight=slots, rccl_path=FAKE, frames_per_launch=1)
    def subd(v, k):
        W.set_view(rt, v)
        for _ in range(k):
            rt.dist_frame()
    r = timed(subd, rt.dist_wait)
    print(f"{name} pipeline, 1 rank, {slots} launch slots [{rt.kernel_name()}]: " + " | ".join(f"{v} {t:.1f} us" for v, t in r.items()) + f" | mean {sum(r.values()) / 3:.1f}", flush=True)
    rt.deinit()
