"""MagicaVoxel .vox input, mirroring src/modules/voxel_rt/vox/loader.zig of the reference
(`load`, `parseBuffer`, `validateHeader`) on top of the C ABI (vrt_vox_*)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import VrtError, lib
from .voxel_rt import MATERIAL_DTYPE, BrickGrid


def _check(rc: int) -> None:
    if rc != L.VRT_OK:
        raise VrtError(rc, "vox")


def validate_header(buffer: bytes) -> None:
    """validateHeader, loader.zig:231-245; raises VrtError(InvalidId / UnexpectedVersion / InvalidFileContent)."""
    _check(lib.vrt_vox_validate_header(buffer, len(buffer)))


class Vox:
    """types.zig Vox: pack_chunk.num_models, size_chunks, xyzi_chunks, rgba_chunk."""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            lib.vrt_vox_destroy(self._h)
            self._h = None

    @property
    def num_models(self) -> int:
        return lib.vrt_vox_num_models(self._h)

    def size(self, model: int = 0):
        out = (C.c_int32 * 3)()
        _check(lib.vrt_vox_model_size(self._h, model, C.byref(out)))
        return tuple(out)

    def xyzi(self, model: int = 0) -> np.ndarray:
        n = C.c_uint64()
        ptr = lib.vrt_vox_model_voxels(self._h, model, C.byref(n))
        if not n.value:
            return np.zeros((0, 4), dtype=np.uint8)
        return np.frombuffer(bytes((C.c_uint8 * (4 * n.value)).from_address(ptr)), dtype=np.uint8).reshape(-1, 4).copy()

    @property
    def rgba(self) -> np.ndarray:
        return np.frombuffer(bytes((C.c_uint8 * 1024).from_address(lib.vrt_vox_palette(self._h))), dtype=np.uint8).reshape(256, 4).copy()

    def materials(self, count: int = 256) -> np.ndarray:
        arr = (L.Material * count)()
        _check(lib.vrt_vox_materials(self._h, arr, count))
        return np.frombuffer(bytes(arr), dtype=MATERIAL_DTYPE).copy()

    def insert_into(self, grid: BrickGrid, model: int = 0, offset=(0, 0, 0), material_offset: int = 0) -> None:
        _check(lib.vrt_vox_insert(grid._h, self._h, model, offset[0], offset[1], offset[2], material_offset))


def parse_buffer(buffer: bytes, strict: bool = True) -> Vox:
    h = C.c_void_p()
    _check(lib.vrt_vox_parse(buffer, len(buffer), 1 if strict else 0, C.byref(h)))
    return Vox(h)


def load(path: str, strict: bool = True) -> Vox:
    with open(path, "rb") as f:
        return parse_buffer(f.read(), strict)
