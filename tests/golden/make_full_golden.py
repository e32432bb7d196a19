#!/usr/bin/env python3
"""Whole-frame digests of the BASELINE.json configurations that had none: configs[1] (1920x1080, 256^3, primary
rays), configs[3] (3840x2160, 1024^3, 2 samples x (primary + shadow)) and configs[4] (3840x2160, 2048^3 sparse, 16 spp, 2 bounces,
soft sun) rendered WHOLE by the parity oracle (oracle/vrt_oracle.c, all cores of this container) -> tests/golden/full/<cfg>_<view>.npz:
SHA-256 of the float frame and of the RGBA8 frame, SHA-256 per band of 16 rows (a mismatch is located without the frame), eight
float crops, the frame's mean colour, the oracle's counters, the 128 push-constant bytes and the scene's digest.  The frames
themselves are far too large to keep (133 MB of floats at 4K).

tests/test_fullsize_gpu.py renders the same frames with libvrt_hip.so at full size and compares the digests: every pixel of every
configuration is then pinned, not a sample of them.  (configs[2] has had its whole-frame fixtures since round 1 —
make_golden.py — and the reference shader's own since round 3 — make_ref_golden.py.)

    python tests/golden/make_full_golden.py [cfg1 cfg3 cfg4]        # cfg4: ~0.5 G rays per frame, minutes on 8 cores
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from tests.golden.make_golden import scene_digest  # noqa: E402
from tests.helpers import oracle_scene_from_grid  # noqa: E402
from zig_vulkan_amd import workloads as W  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "full")
BAND, CROP = 16, 48

# (the views of each configuration that tests/test_fullsize_gpu.py renders; the sun radius is the workload's own — the soft sun and
# the bounces go through the sin-based RNG, whose lowering is part of the contract since round 3)
CASES = {
    "cfg1": ("cfg1_1080p_256c_b4", ["V0", "V1", "V2"]),
    "cfg3": ("cfg3_4k_1024c_b8", ["V1", "V2"]),
    "cfg4": ("cfg4_4k_2048c_b8_sparse", ["V1", "V1x"]),
}


def crop_origins(w, h):
    """Eight crops spread over the frame (fixed positions: sky, horizon, terrain)."""
    return [(min(int(h * fy) // 8 * 8, h - CROP), min(int(w * fx) // 8 * 8, w - CROP))
            for fy, fx in ((0.15, 0.2), (0.35, 0.7), (0.5, 0.45), (0.55, 0.1), (0.65, 0.8), (0.75, 0.3), (0.85, 0.6), (0.93, 0.05))]


def digest(f: np.ndarray, u: np.ndarray) -> dict:
    """What is kept of a frame (float RGBA, RGBA8); the same function digests the library's frame in the test."""
    h, w = u.shape[:2]
    origins = crop_origins(w, h)
    return dict(
        rgba8_sha256=np.array(hashlib.sha256(np.ascontiguousarray(u).tobytes()).hexdigest()),
        float_sha256=np.array(hashlib.sha256(np.ascontiguousarray(f).tobytes()).hexdigest()),
        band_rows=np.int32(BAND),
        band_sha256=np.array([hashlib.sha256(np.ascontiguousarray(f[y:y + BAND]).tobytes()).hexdigest() for y in range(0, h, BAND)]),
        crop_origins=np.array(origins), float_crops=np.stack([f[y:y + CROP, x:x + CROP, :3] for y, x in origins]),
        mean_rgb=f[:, :, :3].mean(axis=(0, 1)).astype(np.float64))


def main(which):
    os.makedirs(OUT, exist_ok=True)
    for key in which:
        name, views = CASES[key]
        w = W.WORKLOADS[name]
        grid = W.build_grid(w)
        scene = oracle_scene_from_grid(grid)
        sha = scene_digest(grid)
        for view in views:
            pc = O.push_constants(W.camera_for(w, view).blob(), W.sun_for(w).blob())
            t0 = time.perf_counter()
            f, u, c = O.render(scene, pc)
            dt = time.perf_counter() - t0
            np.savez_compressed(
                os.path.join(OUT, f"{key}_{view}.npz"),
                provenance=np.array(f"oracle/vrt_oracle.c (gallivm lowering), whole frame, {os.cpu_count()} threads, {dt:.0f} s; tests/golden/make_full_golden.py"),
                workload=np.array(w.name), view=np.array(view), size=np.array([w.width, w.height]), push_constants=pc, scene_sha256=np.array(sha),
                counters=np.array([c[k] for k in ("rays", "status_loads", "bricks_entered", "voxel_steps", "hits", "grid_steps")], dtype=np.uint64),
                **digest(f, u))
            print(f"{key}_{view}: {w.width}x{w.height}, {c['rays']} rays in {dt:.1f} s = {c['rays'] / dt / 1e6:.2f} Mrays/s, mean colour {f[:, :, :3].mean(axis=(0, 1))}", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or list(CASES))
