"""Shared test helpers: build oracle inputs from the product's host objects."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from zig_vulkan_amd import _lib as L  # noqa: E402
from zig_vulkan_amd import default_materials  # noqa: E402


def oracle_scene_from_grid(grid, materials=None) -> O.OracleScene:
    materials = default_materials(256) if materials is None else materials
    return O.OracleScene(
        bytes(grid.device_state), materials,
        grid.array(L.BUF_BRICK_STATUS), grid.array(L.BUF_BRICK_INDEX), grid.array(L.BUF_BRICK_OCCUPANCY),
        grid.array(L.BUF_BRICK_START_INDEX), grid.array(L.BUF_MATERIAL_INDEX), grid.brick_dimension)


def push_for(camera, sun) -> np.ndarray:
    return O.push_constants(camera.blob(), sun.blob())


# ---- which build of the library holds a kernel_variant ---------------------------------------------------------------
PRODUCT_MODES = (0, 5, 9)          # default, the shader's words one request per trip, byte-per-cell status


def needs_dev_build(variant: int) -> bool:
    """True when the variant is compiled only with -DVRT_DEV_VARIANTS (variants that lost their A/B measurement)."""
    mode, min_waves = variant & 0xFF, (variant >> 8) & 0xFF
    if mode not in PRODUCT_MODES or (variant & (1 << 22)):      # other status modes; the block-skipping path kernel
        return True
    return min_waves not in (0, 5)                               # forced wave counts (5 is vrt_path_kernel's own)


def dev_library_or_none():
    if os.environ.get("VRT_NO_DEV_LIB"):   # (run the suite as a box without the development build would)
        return None
    return L.DEV_LIB_PATH if os.path.exists(L.DEV_LIB_PATH) else None


def variant_kwargs(variant: int) -> dict:
    """make_renderer keywords for a kernel_variant: development variants run on libvrt_hip_dev.so (make -C zig_vulkan_amd/csrc
    dev; not built by __graft_entry__.build()) and are skipped where it is absent."""
    import pytest
    if not needs_dev_build(variant):
        return {"kernel_variant": variant}
    dev = dev_library_or_none()
    if dev is None:
        pytest.skip("development variant: needs libvrt_hip_dev.so (make -C zig_vulkan_amd/csrc dev)")
    return {"kernel_variant": variant, "library": dev}


def available_variants(variants):
    """The variants of a sweep this box can run: all of them with the development build, the product ones without."""
    dev = dev_library_or_none()
    return [v for v in variants if dev is not None or not needs_dev_build(v)]
