#!/usr/bin/env python3
"""Determinism soak (usage: soak.py [frames] [workload] [frames_in_flight]): many frames of a workload (two in flight by default; with 1 the
frames go through the cost-ordered launch with split tiles, and a few extra frames are drawn between reads so that it re-sorts), every frame's
bytes must equal the first frame of its view.  A timing-dependent fault in the hand-written loops (a missed hazard, a stale wait count)
would show up here as a differing frame."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zig_vulkan_amd import workloads as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
w = W.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else W.HEADLINE]
grid = W.build_grid(w)
fif = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rt = W.make_renderer(w, grid, frames_in_flight=fif)
ref, bad = {}, 0
for i in range(n):
    view = ["V0", "V1", "V2"][i % 3]
    W.set_view(rt, view)
    rt.draw()
    if fif == 1:
        rt.draw(frames=1 + i % 13)  # not read back: the schedule re-sorts every 32 frames
    elif i % 2 == 1:
        rt.draw()  # an extra frame of the same view on the other stream, not read back
    h = hashlib.sha1(rt.read_rgba8().tobytes()).hexdigest()
    if ref.setdefault(view, h) != h:
        bad += 1
        print(f"frame {i} ({view}) differs")
print(f"{n} frames, {bad} differing; digests {ref}")
rt.deinit()
sys.exit(1 if bad else 0)
