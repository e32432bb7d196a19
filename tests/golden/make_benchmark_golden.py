#!/usr/bin/env python3
"""Generates tests/golden/benchmark/benchmark_path.npz: camera states of the reference's scripted fly-through
(src/modules/voxel_rt/Benchmark.zig:22-74, path :141-172) as raw Camera.Device bytes, so that the restatement of
the zalgebra helpers (un-vendored upstream) in zig_vulkan_amd/csrc/host_benchmark.cpp cannot drift unnoticed.

  keys     : 11 blobs — a fresh Benchmark advanced by ONE update of k * 60/11 s + 1 ms, k = 0..10 (just past key k)
  sampled  : the 60 s path stepped at 30 fps (dt = 1/30 s), every 150th frame: frames 0, 150, ..., 1800
Each blob is the 96 bytes of Camera.Device followed by the 32 bytes of the default Sun.Device (the 128 push-constant
bytes of a frame).  tests/test_benchmark_path.py also checks them against an independent float64 restatement.

    python tests/golden/make_benchmark_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from zig_vulkan_amd import Camera, CameraConfig, Sun, SunConfig  # noqa: E402
from zig_vulkan_amd.voxel_rt import Benchmark  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "benchmark", "benchmark_path.npz")
W, H = 320, 180
SEG = 60.0 / 11.0


def camera():
    return Camera(75.0, W, H, CameraConfig(samples_per_pixel=1, max_bounce=0))


def blob(cam, sun) -> np.ndarray:
    return np.frombuffer(cam.blob() + sun.blob(), dtype=np.uint8).copy()


def key_blobs():
    sun = Sun(SunConfig(enabled=True, radius=0.0))
    out = []
    for k in range(11):
        cam = camera()
        b = Benchmark(cam)
        b.update(k * SEG + 1e-3)
        out.append(blob(cam, sun))
    return np.stack(out)


def sampled_blobs(every=150, fps=30.0):
    sun = Sun(SunConfig(enabled=True, radius=0.0))
    cam = camera()
    b = Benchmark(cam)
    frames, blobs = [0], [blob(cam, sun)]
    f = 0
    done = False
    while not done:
        done = b.update(1.0 / fps)
        f += 1
        if f % every == 0:
            frames.append(f)
            blobs.append(blob(cam, sun))
    return np.array(frames), np.stack(blobs), f, b.report()


def main():
    keys = key_blobs()
    frames, sampled, total, report = sampled_blobs()
    np.savez_compressed(OUT, width=np.int32(W), height=np.int32(H), keys=keys, sampled_frames=frames, sampled=sampled,
                        total_frames=np.int32(total))
    print(f"{len(keys)} key blobs, {len(frames)} sampled frames {frames.tolist()} of {total}; report {report}")


if __name__ == "__main__":
    main()
