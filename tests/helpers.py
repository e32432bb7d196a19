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
