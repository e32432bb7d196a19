"""Host-side mirror of the reference's scene / camera / renderer API for the
traversal path, on top of the C ABI (include/vrt_hip.h).

Names follow the reference so tests read like its call sites:
  BrickGrid.init / insert            src/modules/voxel_rt/brick/Grid.zig:36,129
  Camera.init                        src/modules/voxel_rt/Camera.zig:36
  Sun.init                           src/modules/voxel_rt/Sun.zig:35
  VoxelRT.init / push_materials / update_grid_delta / draw
                                     src/modules/VoxelRT.zig:39,85,107,76
All computation happens inside libvrt_hip.so (C++ host code + HIP kernels);
this module only marshals arguments.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib as L
from ._lib import lib, check

_GRID_ARRAY_DTYPES = {
    L.BUF_BRICK_STATUS: np.uint32,
    L.BUF_BRICK_INDEX: np.uint32,
    L.BUF_BRICK_OCCUPANCY: np.uint8,
    L.BUF_BRICK_START_INDEX: np.uint32,
    L.BUF_MATERIAL_INDEX: np.uint8,
}


class BrickGrid:
    """BrickGrid (Grid.zig).  `init` arguments are Grid.zig:36 + Grid.Config (Grid.zig:13-20),
    plus brick_dimension (4 in the reference, State.zig:5)."""

    def __init__(self, dim_x: int, dim_y: int, dim_z: int, *, brick_alloc: Optional[int] = None, base_t: float = 0.01,
                 min_point: Sequence[float] = (0.0, 0.0, 0.0), scale: float = 1.0, brick_dimension: int = 4):
        cfg = L.GridConfig()
        cfg.brick_alloc = int(brick_alloc or 0)
        cfg.base_t = base_t
        cfg.min_point[:] = list(min_point)
        cfg.scale = scale
        cfg.brick_dimension = brick_dimension
        h = C.c_void_p()
        check(lib.vrt_grid_create(dim_x, dim_y, dim_z, C.byref(cfg), C.byref(h)))
        self._h = h
        self.brick_dimension = brick_dimension
        self.dim = (dim_x, dim_y, dim_z)
        self.brick_alloc = int(brick_alloc) if brick_alloc else dim_x * dim_y * dim_z

    init = classmethod(lambda cls, *a, **k: cls(*a, **k))

    def deinit(self) -> None:
        if self._h:
            lib.vrt_grid_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.deinit()
        except Exception:
            pass

    def insert(self, x: int, y: int, z: int, material_index: int) -> None:
        check(lib.vrt_grid_insert(self._h, x, y, z, material_index))

    def insert_many(self, xyz: np.ndarray, materials: np.ndarray) -> None:
        xyz = np.ascontiguousarray(xyz, dtype=np.uint32).reshape(-1, 3)
        materials = np.ascontiguousarray(materials, dtype=np.uint8).reshape(-1)
        assert xyz.shape[0] == materials.shape[0]
        check(lib.vrt_grid_insert_many(self._h, xyz.ctypes.data, materials.ctypes.data, xyz.shape[0]))

    @property
    def device_state(self) -> L.GridState:
        return lib.vrt_grid_device_state(self._h).contents

    @property
    def active_bricks(self) -> int:
        return lib.vrt_grid_active_bricks(self._h)

    def array(self, buf_id: int) -> np.ndarray:
        """Copy of host array `buf_id` (BUF_BRICK_STATUS .. BUF_MATERIAL_INDEX)."""
        n = C.c_uint64()
        ptr = lib.vrt_grid_data(self._h, buf_id, C.byref(n))
        raw = (C.c_uint8 * n.value).from_address(ptr)
        return np.frombuffer(bytes(raw), dtype=_GRID_ARRAY_DTYPES[buf_id]).copy()

    def array_view(self, buf_id: int) -> np.ndarray:
        """Zero-copy view (valid while the grid lives and is not resized)."""
        n = C.c_uint64()
        ptr = lib.vrt_grid_data(self._h, buf_id, C.byref(n))
        raw = (C.c_uint8 * n.value).from_address(ptr)
        return np.frombuffer(raw, dtype=_GRID_ARRAY_DTYPES[buf_id])

    def delta(self, buf_id: int) -> Tuple[bool, int, int]:
        a, b = C.c_uint64(), C.c_uint64()
        active = lib.vrt_grid_delta(self._h, buf_id, C.byref(a), C.byref(b))
        return bool(active), a.value, b.value

    def reset_delta(self, buf_id: int) -> None:
        lib.vrt_grid_reset_delta(self._h, buf_id)

    def synth_terrain(self, seed: int = 420) -> None:
        check(lib.vrt_synth_terrain(self._h, seed))

    def synth_sparse(self, seed: int = 420, p: float = 0.05) -> None:
        check(lib.vrt_synth_sparse(self._h, seed, p))


@dataclass
class CameraConfig:  # Camera.zig:5-14
    viewport_height: float = 2.0
    origin: Sequence[float] = (0.0, 0.0, 0.0)
    samples_per_pixel: int = 2
    max_bounce: int = 2


class Camera:
    def __init__(self, vertical_fov: float, image_width: int, image_height: int, config: Optional[CameraConfig] = None):
        config = config or CameraConfig()
        cc = L.CameraConfig()
        cc.viewport_height = config.viewport_height
        cc.origin[:] = list(config.origin)
        cc.samples_per_pixel = config.samples_per_pixel
        cc.max_bounce = config.max_bounce
        self.vertical_fov = vertical_fov
        self.viewport_height_cfg = config.viewport_height
        self.d_camera = L.CameraDevice()
        check(lib.vrt_camera_init(vertical_fov, image_width, image_height, C.byref(cc), C.byref(self.d_camera)))
        self._forward = (0.0, 0.0, 1.0)

    init = classmethod(lambda cls, *a, **k: cls(*a, **k))

    def set_forward(self, forward: Sequence[float]) -> None:
        f = (C.c_float * 3)(*forward)
        check(lib.vrt_camera_set_forward(C.byref(self.d_camera), self.vertical_fov, self.viewport_height_cfg, C.byref(f)))
        self._forward = tuple(forward)

    def set_origin(self, origin: Sequence[float]) -> None:  # Camera.setOrigin, Camera.zig:89-92
        self.d_camera.origin[:] = list(origin)
        self.set_forward(self._forward)

    def look_at(self, origin: Sequence[float], target: Sequence[float]) -> None:
        """Place the camera at `origin` viewing `target`.  Rays leave along -forward
        (lower_left_corner = origin - h/2 - v/2 - forward, Camera.zig:177-180)."""
        self.d_camera.origin[:] = list(origin)
        self.set_forward([o - t for o, t in zip(origin, target)])

    def blob(self) -> bytes:
        return bytes(self.d_camera)


@dataclass
class SunConfig:  # Sun.zig:4-11
    enabled: bool = True
    color: Sequence[float] = (1.0, 1.1, 1.0)
    radius: float = 5.0
    sun_distance: float = 1000.0


class Sun:
    def __init__(self, config: Optional[SunConfig] = None):
        config = config or SunConfig()
        sc = L.SunConfig()
        sc.enabled = 1 if config.enabled else 0
        sc.color[:] = list(config.color)
        sc.radius = config.radius
        sc.sun_distance = config.sun_distance
        self.device_data = L.SunDevice()
        check(lib.vrt_sun_init(C.byref(sc), C.byref(self.device_data)))

    init = classmethod(lambda cls, *a, **k: cls(*a, **k))

    def blob(self) -> bytes:
        return bytes(self.device_data)


def default_materials(capacity: int = 256) -> np.ndarray:
    """The reference's terrain material table (terrain.zig:130-196) padded to `capacity`
    20-byte records; returned as a (capacity, 5) uint32 view-compatible structured array."""
    arr = (L.Material * capacity)()
    lib.vrt_default_materials(arr, capacity)
    return np.frombuffer(bytes(arr), dtype=MATERIAL_DTYPE).copy()


MATERIAL_DTYPE = np.dtype([("type", np.uint32), ("albedo_r", np.float32), ("albedo_g", np.float32),
                           ("albedo_b", np.float32), ("type_data", np.float32)])
assert MATERIAL_DTYPE.itemsize == 20


class Benchmark:
    """Benchmark (src/modules/voxel_rt/Benchmark.zig): scripted 60 s fly-through driving a Camera."""

    def __init__(self, camera: Camera):
        self.camera = camera
        h = C.c_void_p()
        check(lib.vrt_benchmark_create(C.byref(camera.d_camera), camera.vertical_fov, camera.viewport_height_cfg, C.byref(h)))
        self._h = h

    def __del__(self):
        if getattr(self, "_h", None):
            lib.vrt_benchmark_destroy(self._h)
            self._h = None

    def update(self, dt: float) -> bool:
        rc = lib.vrt_benchmark_update(self._h, dt, C.byref(self.camera.d_camera))
        if rc < 0:
            check(rc)
        return rc == 1

    def report(self) -> dict:
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        check(lib.vrt_benchmark_report(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"min_frame_ms": a.value, "max_frame_ms": b.value, "avg_frame_ms": c.value}


@dataclass
class Config:  # VoxelRT.Config, VoxelRT.zig:22-28 (+ the knobs of this implementation)
    internal_resolution_width: int = 1280
    internal_resolution_height: int = 720
    camera: CameraConfig = field(default_factory=CameraConfig)
    sun: SunConfig = field(default_factory=SunConfig)
    material_buffer: int = 256  # Pipeline.Config.material_buffer, Pipeline.zig:30
    want_float_output: bool = False
    enable_counters: int = 0   # 1/True: the reference algorithm's counts; 2: the loads the product kernel issues (vrt_hip.h)
    device_id: int = -1
    shard_rank: int = 0
    shard_count: int = 1
    shard_root_weight: int = 0  # rank 0's share of the tiles in percent of an equal share (0: equal)
    kernel_variant: int = 0
    frames_in_flight: int = 1
    stream: int = 0
    external_target_rgba8: int = 0
    external_target_rgba32f: int = 0
    tuning_flags: int = 0      # VRT_TUNE_* (include/vrt_hip.h): A/B switches, every setting renders the same frame
    library: Optional[str] = None  # path of another build of libvrt_hip (fused / dev twins); None: the product library


class VoxelRT:
    """VoxelRT (src/modules/VoxelRT.zig): owns camera, sun and the device pipeline for one grid."""

    def __init__(self, brick_grid: BrickGrid, config: Optional[Config] = None, upload_grid: bool = True):
        config = config or Config()
        self.config = config
        self.brick_grid = brick_grid
        self.camera = Camera(75.0, config.internal_resolution_width, config.internal_resolution_height, config.camera)  # VoxelRT.zig:42
        self.sun = Sun(config.sun)
        cfg = L.Config()
        cfg.struct_size = C.sizeof(L.Config)
        cfg.abi_version = L.VRT_ABI_VERSION
        cfg.width = config.internal_resolution_width
        cfg.height = config.internal_resolution_height
        cfg.brick_dimension = brick_grid.brick_dimension
        cfg.dim_x, cfg.dim_y, cfg.dim_z = brick_grid.dim
        cfg.brick_alloc = brick_grid.brick_alloc
        cfg.material_capacity = config.material_buffer
        cfg.device_id = config.device_id
        cfg.want_float_output = 1 if config.want_float_output else 0
        cfg.enable_counters = int(config.enable_counters)
        cfg.shard_rank = config.shard_rank
        cfg.shard_count = config.shard_count
        cfg.shard_root_weight = config.shard_root_weight
        cfg.kernel_variant = config.kernel_variant
        cfg.frames_in_flight = config.frames_in_flight
        cfg.stream = config.stream or None
        cfg.external_target_rgba8 = config.external_target_rgba8 or None
        cfg.external_target_rgba32f = config.external_target_rgba32f or None
        cfg.tuning_flags = config.tuning_flags
        # (config.library: another build of the same library — the reference-lowering twin or the development build;
        # host objects such as BrickGrid stay with the default build: same sources, same layout)
        self._lib = L.load_library(config.library) if config.library else lib
        h = C.c_void_p()
        rc = self._lib.vrt_create(C.byref(cfg), C.byref(h))
        if rc != L.VRT_OK:
            raise L.VrtError(rc, (self._lib.vrt_last_error(None) or b"").decode())
        self._h = h
        self.width, self.height = cfg.width, cfg.height
        if upload_grid:
            self._check(self._lib.vrt_upload_grid(self._h, brick_grid._h))  # VoxelRT.zig:62 (+ first full delta)

    init = classmethod(lambda cls, *a, **k: cls(*a, **k))

    def _check(self, rc: int) -> None:
        if rc != L.VRT_OK:
            raise L.VrtError(rc, (self._lib.vrt_last_error(self._h) or b"").decode())

    def deinit(self) -> None:
        if getattr(self, "_h", None):
            self._lib.vrt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.deinit()
        except Exception:
            pass

    # -- uploads ------------------------------------------------------------
    def push_materials(self, materials: np.ndarray) -> None:  # VoxelRT.zig:85-87
        materials = np.ascontiguousarray(materials)
        self._check(self._lib.vrt_upload(self._h, L.BUF_MATERIALS, 0, materials.ctypes.data, materials.nbytes))

    def update_grid_delta(self) -> None:  # VoxelRT.zig:107-172
        self._check(self._lib.vrt_update_grid_delta(self._h, self.brick_grid._h))

    def upload(self, buf_id: int, byte_offset: int, data: np.ndarray) -> None:  # Pipeline.transfer*, Pipeline.zig:560-652
        data = np.ascontiguousarray(data)
        self._check(self._lib.vrt_upload(self._h, buf_id, byte_offset, data.ctypes.data, data.nbytes))

    def buffer_size(self, buf_id: int) -> int:
        return self._lib.vrt_buffer_size(self._h, buf_id)

    # -- frame --------------------------------------------------------------
    def draw(self, frames: int = 1) -> None:  # VoxelRT.draw -> Pipeline.draw -> compute dispatch (Pipeline.zig:441)
        if frames == 1:
            self._check(self._lib.vrt_dispatch(self._h, C.byref(self.camera.d_camera), C.byref(self.sun.device_data)))
        else:
            self._check(self._lib.vrt_dispatch_repeat(self._h, C.byref(self.camera.d_camera), C.byref(self.sun.device_data), frames))

    def draw_timed(self, frames: int) -> np.ndarray:
        """`frames` frames one after another on the primary stream; returns the hipEvent time of each in ms."""
        ms = np.zeros(frames, dtype=np.float32)
        self._check(self._lib.vrt_dispatch_timed(self._h, C.byref(self.camera.d_camera), C.byref(self.sun.device_data), frames,
                                     ms.ctypes.data_as(C.POINTER(C.c_float))))
        return ms

    def wait(self) -> None:
        self._check(self._lib.vrt_wait(self._h))

    def region_begin(self) -> None:
        self._check(self._lib.vrt_region_begin(self._h))

    def region_end(self) -> float:
        """Milliseconds the dispatches since region_begin() took on the device (HIP events on both streams)."""
        ms = C.c_double()
        self._check(self._lib.vrt_region_end(self._h, C.byref(ms)))
        return float(ms.value)

    def last_kernel_ms(self) -> float:
        return self._lib.vrt_last_kernel_ms(self._h)

    def shard_info(self) -> L.ShardInfo:
        s = L.ShardInfo()
        self._check(self._lib.vrt_get_shard_info(self._h, C.byref(s)))
        return s

    def target_bytes_rgba8(self) -> int:
        return self._lib.vrt_target_bytes_rgba8(self._h)

    def read_rgba8(self) -> np.ndarray:
        n = self.target_bytes_rgba8()
        out = np.empty(n, dtype=np.uint8)
        self._check(self._lib.vrt_read_rgba8(self._h, out.ctypes.data, n))
        if self.config.shard_count <= 1:
            return out.reshape(self.height, self.width, 4)
        return out.reshape(-1, 16, 16, 4)

    def read_rgba32f(self) -> np.ndarray:
        n = self.target_bytes_rgba8() * 4
        out = np.empty(n // 4, dtype=np.float32)
        self._check(self._lib.vrt_read_rgba32f(self._h, out.ctypes.data, n))
        if self.config.shard_count <= 1:
            return out.reshape(self.height, self.width, 4)
        return out.reshape(-1, 16, 16, 4)

    def set_target(self, rgba8_ptr: int, rgba32f_ptr: int = 0) -> None:
        self._check(self._lib.vrt_set_target(self._h, rgba8_ptr, rgba32f_ptr or None))

    def denoise(self, out_w: int, out_h: int, *, samples: int = 20, distribution_bias: float = 0.6, pixel_multiplier: float = 1.5,
                inverse_hue_tolerance: float = 20.0, want_float: bool = False):
        """The present/denoise pass (image.frag) over the most recent frame; returns rgba8 (and rgba32f)."""
        dc = L.DenoiseConfig(samples, distribution_bias, pixel_multiplier, inverse_hue_tolerance)
        self._check(self._lib.vrt_denoise(self._h, C.byref(dc), out_w, out_h, 1 if want_float else 0))
        u8 = np.empty((out_h, out_w, 4), dtype=np.uint8)
        self._check(self._lib.vrt_read_denoised_rgba8(self._h, u8.ctypes.data, u8.nbytes))
        if not want_float:
            return u8
        f32 = np.empty((out_h, out_w, 4), dtype=np.float32)
        self._check(self._lib.vrt_read_denoised_rgba32f(self._h, f32.ctypes.data, f32.nbytes))
        return u8, f32

    def present(self, out_w: int, out_h: int, *, samples: int = 20, distribution_bias: float = 0.6, pixel_multiplier: float = 1.5,
                inverse_hue_tolerance: float = 20.0) -> None:
        """The present/denoise pass without the read-back (GraphicsPipeline.zig:27-39 after every trace, Pipeline.zig:432-541)."""
        dc = L.DenoiseConfig(samples, distribution_bias, pixel_multiplier, inverse_hue_tolerance)
        self._check(self._lib.vrt_denoise(self._h, C.byref(dc), out_w, out_h, 0))

    def last_denoise_ms(self) -> float:
        return float(self._lib.vrt_last_denoise_ms(self._h))

    def device_target_rgba8(self) -> int:
        return self._lib.vrt_device_target_rgba8(self._h)

    def assemble_frame(self, gathered_ptr: int, dst_ptr: int, bytes_per_pixel: int = 4) -> None:
        self._check(self._lib.vrt_assemble_frame(self._h, gathered_ptr, dst_ptr, bytes_per_pixel))

    def counters(self) -> dict:
        c = L.Counters()
        self._check(self._lib.vrt_get_counters(self._h, C.byref(c)))
        return {k: getattr(c, k) for k, _ in L.Counters._fields_}

    def wave_timeline(self, raw: bool = False) -> np.ndarray:
        """One frame with per-wave [begin, end] wall-clock ticks (100 MHz); shape (waves, 2).  The cost-ordered launch has up to
        one spare workgroup per tile (second halves of split tiles); the rows of those that stayed idle are dropped unless raw."""
        n = (2 * self.shard_info().owned_tiles + 1024) * 4
        out = np.zeros((n, 2), dtype=np.uint64)
        got = C.c_uint64()
        self._check(self._lib.vrt_trace_wave_timeline(self._h, C.byref(self.camera.d_camera), C.byref(self.sun.device_data), out.ctypes.data, n,
                                          C.byref(got)))
        out = out[:got.value]
        return out if raw else out[out[:, 1] != 0]

    def wave_counters(self) -> dict:
        out = (C.c_uint64 * 3)()
        self._check(self._lib.vrt_get_wave_counters(self._h, C.byref(out)))
        return {"wave_grid_iters": out[0], "wave_brick_walks": out[1], "wave_voxel_iters": out[2]}

    # -- multi-GPU frame pipeline (native RCCL; see include/vrt_hip.h vrt_dist_*) --------------------
    @staticmethod
    def dist_unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        check(lib.vrt_dist_unique_id(L.rccl_library_path().encode(), buf))
        return bytes(buf)

    def dist_init(self, unique_id: bytes, rank: int, world: int, frames_in_flight: int = 4, rccl_path: Optional[str] = None,
                  frames_per_launch: int = 1, communicators: int = 0) -> None:
        """frames_in_flight: launches in flight; frames_per_launch: consecutive frames traced by one launch and gathered by
        one collective; communicators: how many RCCL communicators the launch slots issue their gathers on (0: one per slot, at most 8;
        1: all on one — vrt_dist_init_ex).  rccl_path: the library to bind (default: the RCCL PyTorch ships; tests pass a
        single-process stand-in)."""
        assert len(unique_id) == 128
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        opt = L.DistOptions()
        opt.struct_size = C.sizeof(L.DistOptions)
        opt.frames_in_flight, opt.frames_per_launch, opt.communicators = frames_in_flight, frames_per_launch, communicators
        self._check(self._lib.vrt_dist_init_ex(self._h, (rccl_path or L.rccl_library_path()).encode(), buf, rank, world, C.byref(opt)))

    def dist_comm_info(self) -> dict:
        out = (C.c_int32 * 4)()
        self._check(self._lib.vrt_dist_comm_info(self._h, out))
        return {"communicators": out[0], "made_by_this_init": out[1], "library_splits": bool(out[2]), "agreed_by_all_reduce": bool(out[3])}

    def dist_selftest_slots(self, busy_us: int = 50, rounds: int = 20) -> dict:
        """vrt_dist_selftest_slots: every launch slot's kernel + self send / recv at once on the bound library."""
        out = (C.c_double * 4)()
        self._check(self._lib.vrt_dist_selftest_slots(self._h, busy_us, rounds, out))
        return {"wall_ms": out[0], "launches": int(out[1]), "us_per_launch": out[0] * 1e3 / max(1.0, out[1]), "last_round_launch_ms": out[2],
                "communicators": int(out[3])}

    @staticmethod
    def dist_keep_communicators(keep: bool = True) -> bool:
        return bool(lib.vrt_dist_keep_communicators(1 if keep else 0))

    @staticmethod
    def dist_release_communicators() -> int:
        return int(lib.vrt_dist_release_communicators())

    def dist_frame(self) -> None:
        self._check(self._lib.vrt_dist_frame(self._h, C.byref(self.camera.d_camera), C.byref(self.sun.device_data)))

    def dist_frames(self, cameras, sun=None) -> None:
        """vrt_dist_frames: the frames of `cameras` (96-byte Camera.Device blobs, or a ctypes array of CameraDevice) submitted by ONE
        call across the ABI, all with this renderer's sun."""
        if not isinstance(cameras, C.Array):
            arr = (L.CameraDevice * len(cameras))()
            for i, blob in enumerate(cameras):
                C.memmove(C.byref(arr[i]), bytes(blob), 96)
            cameras = arr
        self._check(self._lib.vrt_dist_frames(self._h, cameras, C.byref(sun if sun is not None else self.sun.device_data), len(cameras), 0))

    def reserve_samples(self, max_samples_per_pixel: int) -> None:
        """vrt_reserve_samples: the persistent kernels' sample buffers made now instead of by the first dispatch that needs them."""
        self._check(self._lib.vrt_reserve_samples(self._h, int(max_samples_per_pixel)))

    def bounce_autotune_info(self) -> dict:
        """vrt_bounce_autotune_info: what the library's timing of the lockstep kernel against vrt_pool_kernel has said so far."""
        out = (C.c_double * 4)()
        self._check(self._lib.vrt_bounce_autotune_info(self._h, out))
        return {"state": ("not applicable", "trials", "lockstep", "pool")[int(out[0])], "trials_launched": int(out[1]), "lockstep_ms": out[2], "pool_ms": out[3]}

    def dist_wait(self) -> None:
        self._check(self._lib.vrt_dist_wait(self._h))

    def dist_read_frame(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        self._check(self._lib.vrt_dist_read_frame(self._h, out.ctypes.data, out.nbytes))
        return out

    def dist_info(self) -> dict:
        """rank / world as the RCCL communicator reports them, frames per launch, launches in flight."""
        out = (C.c_int32 * 4)()
        self._check(self._lib.vrt_dist_info(self._h, out))
        return {"rank": out[0], "world": out[1], "frames_per_launch": out[2], "launches_in_flight": out[3]}

    def dist_profile(self, enable: bool = True) -> None:
        """Start (and clear) / stop the per-launch stage timing of the pipeline (vrt_dist_profile)."""
        self._check(self._lib.vrt_dist_profile(self._h, 1 if enable else 0))

    def dist_stats(self) -> dict:
        out = (C.c_double * 8)()
        self._check(self._lib.vrt_dist_stats(self._h, out))
        return {"launches_sampled": int(out[0]), "frames_sampled": int(out[1]), "kernel_ms_per_launch": out[2], "collective_ms_per_launch": out[3],
                "unswizzle_ms_per_launch": out[4], "owned_tiles": int(out[5]), "shard_bytes_per_frame": int(out[6]), "frames_per_launch": int(out[7])}

    def dist_selftest(self) -> None:
        self._check(self._lib.vrt_dist_selftest(self._h))

    # element sizes of the five buffers BrickGrid tracks deltas for (Grid.zig:129-194)
    _DELTA_BUFFERS = ((L.BUF_BRICK_STATUS, 4), (L.BUF_BRICK_INDEX, 4), (L.BUF_BRICK_OCCUPANCY, 1), (L.BUF_BRICK_START_INDEX, 4),
                      (L.BUF_MATERIAL_INDEX, 1))

    def dist_broadcast(self, buf_id: int, byte_offset: int, nbytes: int, root: int = 0) -> None:
        """Collective: the byte range of scene buffer `buf_id` on every rank becomes rank `root`'s (vrt_dist_broadcast)."""
        self._check(self._lib.vrt_dist_broadcast(self._h, buf_id, byte_offset, nbytes, root))

    def grid_delta_ranges(self) -> list:
        """[(buffer id, byte offset, bytes)] of this host's dirty ranges (what update_grid_delta is about to upload)."""
        out = []
        for buf_id, es in self._DELTA_BUFFERS:
            active, a, b = self.brick_grid.delta(buf_id)
            if active and b > a:
                out.append((buf_id, a * es, (b - a) * es))
        return out

    def dist_broadcast_grid_delta(self, root: int = 0, ranges: Optional[list] = None) -> None:
        """Incremental edits made on ONE rank's host reach every replica: rank `root` uploads its dirty
        ranges (update_grid_delta) and every rank takes part in one broadcast per range.  Collective.  `ranges`: root's
        grid_delta_ranges() as every rank knows them; None: they are shared over torch.distributed first."""
        rank = self.dist_info()["rank"]
        if ranges is None:
            import torch.distributed as dist
            box = [self.grid_delta_ranges() if rank == root else None]
            dist.broadcast_object_list(box, src=root)
            ranges = box[0]
        if rank == root:
            self.update_grid_delta()
        for buf_id, off, nbytes in ranges:
            self.dist_broadcast(buf_id, off, nbytes, root)

    def create_benchmark(self) -> "Benchmark":  # VoxelRT.createBenchmark, VoxelRT.zig:72-74
        return Benchmark(self.camera)

    def kernel_name(self) -> str:
        return self._lib.vrt_kernel_name(self._h).decode()
