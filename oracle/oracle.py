"""ctypes wrapper of the parity oracle (oracle/libvrt_oracle.so, built from vrt_oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of vrt_oracle.c.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by zig_vulkan_amd.
PARITY PINNED against the reference's own shader run under Mesa llvmpipe (oracle/ref_gl, tests/test_ref_gl.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvrt_oracle.so")                # the oracle: the reference shader as Mesa llvmpipe executes it
LIB_PATH_FUSED = os.path.join(_HERE, "libvrt_oracle_fused.so")    # fma fused, dot as an fma chain (libvrt_hip_fused.so's counterpart)


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("vrt_oracle.c", "denoise_oracle.c")]
    if force or any(not os.path.exists(p) or os.path.getmtime(p) < max(os.path.getmtime(s) for s in srcs)
                    for p in (LIB_PATH, LIB_PATH_FUSED)):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return LIB_PATH


class GridState(C.Structure):
    _fields_ = [("voxel_dim_x", C.c_uint32), ("voxel_dim_y", C.c_uint32), ("voxel_dim_z", C.c_uint32),
                ("dim_x", C.c_uint32), ("dim_y", C.c_uint32), ("dim_z", C.c_uint32),
                ("padding1", C.c_uint32), ("padding2", C.c_uint32),
                ("min_point_base_t", C.c_float * 4), ("max_point_scale", C.c_float * 4)]


class Scene(C.Structure):
    _fields_ = [("grid", C.c_void_p), ("materials", C.c_void_p), ("brick_type_bits", C.c_void_p),
                ("brick_indices", C.c_void_p), ("brick_solid_mask", C.c_void_p), ("brick_type_and_index", C.c_void_p),
                ("material_indices", C.c_void_p), ("brick_bytes", C.c_uint32), ("brick_dimensions", C.c_int32),
                ("brick_voxel_scale", C.c_float)]


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("status_loads", C.c_uint64), ("bricks_entered", C.c_uint64),
                ("voxel_steps", C.c_uint64), ("hits", C.c_uint64), ("grid_steps", C.c_uint64)]

    def as_dict(self) -> dict:
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


_lib = None
_libs = {}


_LOWERINGS = {"ref": "ref", "llvmpipe": "ref", "fused": "fused", "hw": "fused"}


def lib(lowering: Optional[str] = None) -> C.CDLL:
    """lowering "ref" (default; alias "llvmpipe"): the oracle — the GLSL built-ins lowered as Mesa llvmpipe lowers them, bit-equal
    to the reference's own shader.  "fused" (alias "hw"): fma fused, dot as an fma chain (see vrt_oracle.c)."""
    global _lib
    lowering = _LOWERINGS[lowering or "ref"]
    if lowering not in _libs:
        build()
        L = C.CDLL({"ref": LIB_PATH, "fused": LIB_PATH_FUSED}[lowering])
        L.oracle_render_rows.restype = None
        L.oracle_render_rows.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_render_pixels.restype = None
        L.oracle_render_pixels.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_grid_hit.restype = C.c_int
        L.oracle_grid_hit.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p]
        L.oracle_grid_hit_raw.restype = C.c_int
        L.oracle_grid_hit_raw.argtypes = [C.POINTER(Scene), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_void_p, C.c_void_p,
                                          C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p]
        L.oracle_sinf.restype = C.c_float
        L.oracle_sinf.argtypes = [C.c_float]
        L.oracle_hash12.restype = C.c_float
        L.oracle_hash12.argtypes = [C.c_float, C.c_float]
        L.oracle_rand2.restype = C.c_float
        L.oracle_rand2.argtypes = [C.c_float, C.c_float]
        L.oracle_rand3.restype = C.c_float
        L.oracle_rand3.argtypes = [C.c_float, C.c_float, C.c_float]
        L.oracle_randvec3.restype = None
        L.oracle_randvec3.argtypes = [C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.oracle_adv_norm_intersect.restype = C.c_int
        L.oracle_adv_norm_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.oracle_denoise_rows.restype = None
        L.oracle_denoise_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _libs[lowering] = L
        if lowering == "ref":
            _lib = L
    return _libs[lowering]


class OracleScene:
    """The seven shader inputs (bindings 1..7) as numpy arrays, plus the specialization constants."""

    def __init__(self, grid_state_bytes: bytes, materials: np.ndarray, brick_status: np.ndarray, brick_index: np.ndarray,
                 brick_occupancy: np.ndarray, brick_start_index: np.ndarray, material_index: np.ndarray, brick_dimension: int):
        assert len(grid_state_bytes) == 64
        self.grid_state = np.frombuffer(grid_state_bytes, dtype=np.uint8).copy()
        self.materials = np.ascontiguousarray(materials)
        assert self.materials.dtype.itemsize == 20 or self.materials.dtype == np.uint8
        self.brick_status = np.ascontiguousarray(brick_status, dtype=np.uint32)
        self.brick_index = np.ascontiguousarray(brick_index, dtype=np.uint32)
        self.brick_occupancy = np.ascontiguousarray(brick_occupancy, dtype=np.uint8)
        self.brick_start_index = np.ascontiguousarray(brick_start_index, dtype=np.uint32)
        self.material_index = np.ascontiguousarray(material_index, dtype=np.uint8)
        self.brick_dimension = brick_dimension
        s = Scene()
        s.grid = self.grid_state.ctypes.data
        s.materials = self.materials.ctypes.data
        s.brick_type_bits = self.brick_status.ctypes.data
        s.brick_indices = self.brick_index.ctypes.data
        s.brick_solid_mask = self.brick_occupancy.ctypes.data
        s.brick_type_and_index = self.brick_start_index.ctypes.data
        s.material_indices = self.material_index.ctypes.data
        s.brick_bytes = brick_dimension ** 3 // 8         # Pipeline.zig:310
        s.brick_dimensions = brick_dimension              # Pipeline.zig:311
        s.brick_voxel_scale = 1.0 / brick_dimension       # Pipeline.zig:312
        self.c = s


def push_constants(camera_blob: bytes, sun_blob: bytes) -> np.ndarray:
    assert len(camera_blob) == 96 and len(sun_blob) == 32
    return np.frombuffer(camera_blob + sun_blob, dtype=np.uint8).copy()


def render(scene: OracleScene, pc: np.ndarray, *, rows: Optional[Tuple[int, int]] = None, threads: int = 0,
           want_counters: bool = True, lowering: Optional[str] = None):
    """Render image rows [y0,y1) (default: all) with the oracle.  Returns
    (rgba32f[H,W,4], rgba8[H,W,4], counters dict) — rows outside the range stay zero."""
    L = lib(lowering)
    w, h = np.frombuffer(pc[:8].tobytes(), dtype=np.uint32)
    w, h = int(w), int(h)
    y0, y1 = rows if rows else (0, h)
    f32 = np.zeros((h, w, 4), dtype=np.float32)
    u8 = np.zeros((h, w, 4), dtype=np.uint8)
    threads = threads or os.cpu_count() or 1
    nrows = y1 - y0
    # interleaved 4-row bands: terrain rows are far more expensive than sky rows
    bands = [(y, min(y + 4, y1)) for y in range(y0, y1, 4)]
    groups = [bands[i::threads] for i in range(threads)]
    groups = [g for g in groups if g]

    def work(group):
        c = Counters()
        for (a, b) in group:
            L.oracle_render_rows(C.byref(scene.c), pc.ctypes.data, a, b, f32.ctypes.data, u8.ctypes.data,
                                 C.byref(c) if want_counters else None)
        return c

    total = Counters()
    if len(groups) <= 1 or nrows <= 4:
        cs = [work(g) for g in groups]
    else:
        with ThreadPoolExecutor(max_workers=len(groups)) as ex:
            cs = list(ex.map(work, groups))
    for c in cs:
        for k, _ in Counters._fields_:
            setattr(total, k, getattr(total, k) + getattr(c, k))
    return f32, u8, total.as_dict()


def render_pixels(scene: OracleScene, pc: np.ndarray, xy: np.ndarray):
    L = lib()
    xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
    n = xy.shape[0]
    f32 = np.zeros((n, 4), dtype=np.float32)
    u8 = np.zeros((n, 4), dtype=np.uint8)
    c = Counters()
    L.oracle_render_pixels(C.byref(scene.c), pc.ctypes.data, xy.ctypes.data, n, f32.ctypes.data, u8.ctypes.data, C.byref(c))
    return f32, u8, c.as_dict()


def grid_hit(scene: OracleScene, pc: np.ndarray, origin, direction):
    """Single GridHit (comp:271) probe.  Returns (hit, point, normal, t, material_index, counters)."""
    L = lib()
    o = np.asarray(origin, dtype=np.float32)
    d = np.asarray(direction, dtype=np.float32)
    point = np.zeros(3, dtype=np.float32)
    normal = np.zeros(3, dtype=np.float32)
    t = C.c_float()
    idx = C.c_uint32()
    c = Counters()
    ok = L.oracle_grid_hit(C.byref(scene.c), pc.ctypes.data, o.ctypes.data, d.ctypes.data, point.ctypes.data, normal.ctypes.data,
                           C.byref(t), C.byref(idx), C.byref(c))
    return bool(ok), point, normal, float(t.value), int(idx.value), c.as_dict()


def grid_hit_raw(scene: OracleScene, pc: np.ndarray, origin, direction, ignore_type_material: int = 3, internal_reflection: float = 1.0):
    """GridHit (comp:271) for a ray used as given (direction NOT normalised).  Returns (hit, point, normal, t, material_index)."""
    L = lib()
    o = np.asarray(origin, dtype=np.float32)
    d = np.asarray(direction, dtype=np.float32)
    point = np.zeros(3, dtype=np.float32)
    normal = np.zeros(3, dtype=np.float32)
    t = C.c_float()
    idx = C.c_uint32()
    c = Counters()
    ok = L.oracle_grid_hit_raw(C.byref(scene.c), pc.ctypes.data, o.ctypes.data, d.ctypes.data, int(ignore_type_material),
                               float(internal_reflection), point.ctypes.data, normal.ctypes.data, C.byref(t), C.byref(idx), C.byref(c))
    return bool(ok), point, normal, np.float32(t.value), int(idx.value)


def algorithmic_bytes(counters: dict, pixels: int) -> int:
    """4*S + 4*K + 1*V + 25*H per ray, summed, + 4 B per pixel stored."""
    return (4 * counters["status_loads"] + 4 * counters["bricks_entered"] + counters["voxel_steps"] + 25 * counters["hits"]
            + 4 * pixels)


def denoise(image_rgba8: np.ndarray, out_w: int, out_h: int, samples=20, distribution_bias=0.6, pixel_multiplier=1.5,
            inverse_hue_tolerance=20.0):
    """image.frag over an out_w x out_h target (defaults: GraphicsPipeline.Config, GraphicsPipeline.zig:34-39).
    Returns (rgba32f, rgba8)."""
    L = lib()
    img = np.ascontiguousarray(image_rgba8, dtype=np.uint8)
    h, w = img.shape[:2]
    pc = np.zeros(1, dtype=np.dtype([("samples", np.int32), ("b", np.float32), ("m", np.float32), ("t", np.float32)]))
    pc[0] = (samples, distribution_bias, pixel_multiplier, inverse_hue_tolerance)
    f32 = np.zeros((out_h, out_w, 4), dtype=np.float32)
    u8 = np.zeros((out_h, out_w, 4), dtype=np.uint8)
    threads = os.cpu_count() or 1
    bands = [(y, min(y + 8, out_h)) for y in range(0, out_h, 8)]
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(lambda ab: L.oracle_denoise_rows(img.ctypes.data, w, h, pc.ctypes.data, out_w, out_h, ab[0], ab[1], f32.ctypes.data,
                                                     u8.ctypes.data), bands))
    return f32, u8
