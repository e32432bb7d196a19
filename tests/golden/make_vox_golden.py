#!/usr/bin/env python3
"""Generates tests/golden/vox/*.npz from the reference's own .vox assets (the only reference-held data fixtures of
any row of SURVEY.md §8: assets/models/doom.vox, loaded by src/main.zig:84, and assets/models/monu10.vox).

    python tests/golden/make_vox_golden.py        # needs /root/reference (build container only)

doom_scene.npz: the grid of src/main.zig:77-81 (128 x 64 x 128 bricks of 4^3, min_point (-32,-16,-32), scale 0.5, dense
allocation) with doom.vox inserted exactly as src/main.zig:109-117 does (voxel (x, y, z) -> grid (x + 200, z + 50, y + 150),
material = color_index + 8) and the material table of src/main.zig:87-106 (8 terrain materials, then the palette).  Stored:
the parse results (sizes, voxel count, SHA-256 of the XYZI records and of the palette), a SHA-256 of each of the seven scene
buffers (NOT the buffers: they are the asset's voxel data re-encoded, and the reference states no licence for its assets), one
camera (128 push-constant bytes, reference defaults: 2 samples, 2 bounces, sun on), and two frames of it — the oracle's and the
reference shader's own under llvmpipe (oracle/_ref).  Tests that need the buffers rebuild them from /root/reference.
The terrain that main.zig adds afterwards is not part of it (irreproducible in the reference: SURVEY.md §5).
This pins the .vox loader and the palette mapping to reference-held data; the traversal is pinned by tests/golden/ref.
monu10.npz: parse results only.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from tests.helpers import oracle_scene_from_grid  # noqa: E402
from zig_vulkan_amd import BrickGrid, Camera, CameraConfig, Sun, SunConfig, default_materials, vox  # noqa: E402

MODELS = "/root/reference/assets/models"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vox")
TERRAIN_MATERIALS = 8   # terrain.materials.len, src/modules/voxel_rt/terrain/terrain.zig:130-196


def parse_summary(v):
    xyzi = v.xyzi(0)
    return dict(num_models=np.int32(v.num_models), size=np.array(v.size(0), dtype=np.int32), voxels=np.int64(xyzi.shape[0]),
                xyzi_sha256=np.array(hashlib.sha256(xyzi.tobytes()).hexdigest()), rgba_sha256=np.array(hashlib.sha256(v.rgba.tobytes()).hexdigest()))


def buffer_digests(scene):
    """SHA-256 of each of the seven scene buffers (bindings 1..7)."""
    bufs = (("grid_state", scene.grid_state), ("materials", scene.materials.view(np.uint8).reshape(-1)), ("brick_status", scene.brick_status),
            ("brick_index", scene.brick_index), ("brick_occupancy", scene.brick_occupancy), ("brick_start_index", scene.brick_start_index),
            ("material_index", scene.material_index))
    return {k + "_sha256": np.array(hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()) for k, a in bufs}


def doom_scene():
    """(grid, materials) as src/main.zig:77-117 builds them before the terrain."""
    v = vox.load(os.path.join(MODELS, "doom.vox"), strict=False)          # main.zig:84: vox.load(false, ...)
    grid = BrickGrid(128, 64, 128, min_point=(-32.0, -16.0, -32.0), scale=0.5, brick_dimension=4)   # main.zig:77-81
    v.insert_into(grid, 0, offset=(200, 50, 150), material_offset=TERRAIN_MATERIALS)              # main.zig:109-117
    materials = default_materials(256).copy()                                                     # main.zig:87-91
    materials[TERRAIN_MATERIALS:] = v.materials(256 - TERRAIN_MATERIALS)                          # main.zig:93-106
    return v, grid, materials


def doom_camera():
    cam = Camera(75.0, 256, 144, CameraConfig(samples_per_pixel=2, max_bounce=2))   # main.zig:126-129 at fixture size
    cam.look_at((2.0, -1.0, 7.0), (0.5, 2.0, -5.0))
    return cam, Sun(SunConfig(enabled=True))


def main():
    os.makedirs(OUT, exist_ok=True)
    v, grid, materials = doom_scene()
    scene = oracle_scene_from_grid(grid, materials)
    cam, sun = doom_camera()
    pc = O.push_constants(cam.blob(), sun.blob())
    f, u, c = O.render(scene, pc)
    extra = {}
    from oracle import ref_gl
    if ref_gl.available() is None:
        rf, ru = ref_gl.ReferenceShader(4).render(scene, pc)
        fl, ul, _ = O.render(scene, pc, lowering="llvmpipe")
        assert np.array_equal(rf.view(np.uint32), fl.view(np.uint32)) and np.array_equal(ru, ul)
        extra = dict(ref_rgba8=ru, ref_rgb32f=np.ascontiguousarray(rf[:, :, :3]))
    np.savez_compressed(os.path.join(OUT, "doom_scene.npz"), **parse_summary(v), **buffer_digests(scene),
                        active_bricks=np.int64(grid.active_bricks),
                        push_constants=pc, oracle_rgba8=u, oracle_rgb32f=np.ascontiguousarray(f[:, :, :3]), **extra)
    print("doom.vox", v.size(0), v.xyzi(0).shape[0], "voxels;", grid.active_bricks, "bricks;", c)
    m = vox.load(os.path.join(MODELS, "monu10.vox"), strict=False)
    np.savez_compressed(os.path.join(OUT, "monu10.npz"), **parse_summary(m))
    print("monu10.vox", m.size(0), m.xyzi(0).shape[0], "voxels")


if __name__ == "__main__":
    main()
