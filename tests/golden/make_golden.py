#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the parity oracle (oracle/vrt_oracle.c) in this container.

The reference holds no golden vectors for this path and cannot be run here, so
these fixtures pin the ORACLE's output — and with it every later build of the oracle and of the HIP
kernels — on fixed inputs.  Inputs are stored as raw data (scene array digests, the 128 push-constant
bytes), outputs as RGBA8 frames plus float32 crops and a SHA-256 of the full float frame.

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from tests.helpers import oracle_scene_from_grid, push_for  # noqa: E402
from zig_vulkan_amd import _lib as L  # noqa: E402
from zig_vulkan_amd import workloads as W  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # BASELINE.json configs[0]: 256x256, 64^3 dense, primary rays only
    "cfg0_V0": (W.WORKLOADS["cfg0_256x256_64c_b4"], "V0"),
    "cfg0_V1": (W.WORKLOADS["cfg0_256x256_64c_b4"], "V1"),
    "cfg0_V2": (W.WORKLOADS["cfg0_256x256_64c_b4"], "V2"),
    # configs[2] shape at fixture size: 8^3 bricks, primary + shadow, hard sun (deterministic)
    "shadow_b8_V2": (W.Workload("shadow_b8", 192, 128, 128, 8, 1, 0, True, 0.0), "V2"),
    # reference default shading: spp 2, bounces, soft sun -> scatter functions and sin-based RNG
    "path_b4_V0": (W.Workload("path_b4", 128, 96, 64, 4, 2, 2, True, 5.0), "V0"),
}


# Full-size frames of the headline workload (BASELINE.json configs[2]): too large to store, so SHA-256 of the whole
# RGBA8 and float frames plus crops; written to tests/golden/full/.
FULL_CASES = {f"cfg2_{v}": (W.WORKLOADS[W.HEADLINE], v) for v in ("V0", "V1", "V2", "V1x")}


def scene_digest(grid) -> str:
    h = hashlib.sha256()
    h.update(bytes(grid.device_state))
    for bid in (L.BUF_BRICK_STATUS, L.BUF_BRICK_INDEX, L.BUF_BRICK_OCCUPANCY, L.BUF_BRICK_START_INDEX, L.BUF_MATERIAL_INDEX):
        h.update(grid.array(bid).tobytes())
    return h.hexdigest()


def main():
    for name, (w, view) in CASES.items():
        grid = W.build_grid(w)
        cam, sun = W.camera_for(w, view), W.sun_for(w)
        pc = push_for(cam, sun)
        f, u, c = O.render(oracle_scene_from_grid(grid), pc)
        cy, cx = w.height // 2, w.width // 2
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"),
            workload=np.array([w.width, w.height, w.voxels, w.brick_dimension, w.spp, w.max_bounce, int(w.sun_enabled)], dtype=np.int64),
            sun_radius=np.float32(w.sun_radius), view=np.array(view),
            push_constants=pc, scene_sha256=np.array(scene_digest(grid)),
            rgba8=u, float_sha256=np.array(hashlib.sha256(f.tobytes()).hexdigest()),
            float_crop=f[cy - 24:cy + 24, cx - 32:cx + 32].copy(), crop_origin=np.array([cy - 24, cx - 32]),
            counters=np.array([c[k] for k in ("rays", "status_loads", "bricks_entered", "voxel_steps", "hits", "grid_steps")], dtype=np.uint64))
        print(name, c)


def main_full():
    out = os.path.join(OUT, "full")
    os.makedirs(out, exist_ok=True)
    grids = {}
    for name, (w, view) in FULL_CASES.items():
        grid = grids.setdefault(w.name, W.build_grid(w))
        pc = push_for(W.camera_for(w, view), W.sun_for(w, 0.0))   # hard sun: the frame does not depend on sin()
        f, u, c = O.render(oracle_scene_from_grid(grid), pc)
        cy, cx = 2 * w.height // 3, w.width // 2
        np.savez_compressed(
            os.path.join(out, name + ".npz"),
            workload=np.array(w.name), view=np.array(view), push_constants=pc, scene_sha256=np.array(scene_digest(grid)),
            rgba8_sha256=np.array(hashlib.sha256(u.tobytes()).hexdigest()), float_sha256=np.array(hashlib.sha256(f.tobytes()).hexdigest()),
            rgba8_crop=u[cy - 32:cy + 32, cx - 48:cx + 48].copy(), float_crop=f[cy - 32:cy + 32, cx - 48:cx + 48].copy(),
            crop_origin=np.array([cy - 32, cx - 48]),
            counters=np.array([c[k] for k in ("rays", "status_loads", "bricks_entered", "voxel_steps", "hits", "grid_steps")], dtype=np.uint64))
        print(name, c)


if __name__ == "__main__":
    main()
    main_full()
