"""ctypes binding of libvrt_hip.so — the C ABI declared in include/vrt_hip.h.

This is the only place the shared library is loaded.  There is no fallback: if
the library is missing or a symbol is absent, importing fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (VRT_HIP_LIB: development builds of the same library, e.g. the phase-profile build of tools/experiments/frame_phases.py)
LIB_PATH = os.environ.get("VRT_HIP_LIB") or os.path.join(_HERE, "libvrt_hip.so")

VRT_ABI_VERSION = 4

VRT_OK = 0
VRT_E_INVALID_ARG = -1
VRT_E_OOM = -2
VRT_E_OUT_OF_RANGE = -3
VRT_E_HIP = -4
VRT_E_NO_DEVICE = -5
VRT_E_STATE = -6
VRT_E_RCCL = -7

ERROR_NAMES = {
    VRT_E_INVALID_ARG: "VRT_E_INVALID_ARG",
    VRT_E_OOM: "VRT_E_OOM",
    VRT_E_OUT_OF_RANGE: "VRT_E_OUT_OF_RANGE",
    VRT_E_HIP: "VRT_E_HIP",
    VRT_E_NO_DEVICE: "VRT_E_NO_DEVICE",
    VRT_E_STATE: "VRT_E_STATE",
    VRT_E_RCCL: "VRT_E_RCCL",
}

# vrt_config.tuning_flags (VRT_TUNE_*): A/B switches, every setting renders the same frame
TUNE_NO_SKIP_TO_BOX, TUNE_NO_PATH_BRICK_LDS, TUNE_NO_PATH_HALFBLOCKS, TUNE_PATH_EAGER_START, TUNE_DIST_NO_BROADCAST = 1, 2, 4, 8, 16
TUNE_NO_CELL_OCCUPANCY, TUNE_NO_START_SHORTCUT, TUNE_PATH_AHEAD, TUNE_PATH_DISTANCE, TUNE_NO_PATH_DILATED, TUNE_NO_PATH_GRID_EXIT, TUNE_PATH_BLOCKS64, TUNE_PATH_TWO_AHEAD = 32, 64, 128, 256, 512, 1024, 2048, 4096
TUNE_NO_SMALL_FRAME_SPLIT = 16384
TUNE_NO_BOUNCE_WAVE_GROUPS = 32768
TUNE_NO_SAMPLE_UNITS = 65536
TUNE_NO_DEFERRED_MATERIAL = 131072
TUNE_NO_CELL_MATERIAL = 262144
TUNE_GRID_EXIT_ANY_BOX = 524288
TUNE_NO_BOUNCE_AUTOTUNE = 1048576
TUNE_PRESENT_OWN_STREAM = 2097152
TUNE_NO_PATH_POOL = 8192  # frames with bounces on scenes larger than the caches: vrt_path_kernel (a ray per lane) instead of vrt_pool_kernel

# vrt_buffer_id — shader bindings 1..7
BUF_GRID_STATE, BUF_MATERIALS, BUF_BRICK_STATUS, BUF_BRICK_INDEX, BUF_BRICK_OCCUPANCY, BUF_BRICK_START_INDEX, BUF_MATERIAL_INDEX = range(7)
BUF_COUNT = 7


class VrtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {message}")
        self.code = code


class GridState(C.Structure):  # State.Device, State.zig:60-79
    _fields_ = [
        ("voxel_dim_x", C.c_uint32), ("voxel_dim_y", C.c_uint32), ("voxel_dim_z", C.c_uint32),
        ("dim_x", C.c_uint32), ("dim_y", C.c_uint32), ("dim_z", C.c_uint32),
        ("padding1", C.c_uint32), ("padding2", C.c_uint32),
        ("min_point_base_t", C.c_float * 4),
        ("max_point_scale", C.c_float * 4),
    ]


class Material(C.Structure):  # gpu_types.Material, gpu_types.zig:16-32
    _fields_ = [("type", C.c_uint32), ("albedo_r", C.c_float), ("albedo_g", C.c_float), ("albedo_b", C.c_float),
                ("type_data", C.c_float)]


class CameraDevice(C.Structure):  # Camera.Device, Camera.zig:183-193 (96 bytes)
    _fields_ = [
        ("image_width", C.c_uint32), ("image_height", C.c_uint32), ("_pad0", C.c_uint32 * 2),
        ("horizontal", C.c_float * 3), ("_pad1", C.c_float),
        ("vertical", C.c_float * 3), ("_pad2", C.c_float),
        ("lower_left_corner", C.c_float * 3), ("_pad3", C.c_float),
        ("origin", C.c_float * 3), ("_pad4", C.c_float),
        ("samples_per_pixel", C.c_int32), ("max_bounce", C.c_int32), ("_pad5", C.c_uint32 * 2),
    ]


class SunDevice(C.Structure):  # Sun.Device, Sun.zig:13-18 (32 bytes)
    _fields_ = [("position", C.c_float * 3), ("enabled", C.c_uint32), ("color", C.c_float * 3), ("radius", C.c_float)]


class Config(C.Structure):  # vrt_config
    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32),
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("brick_dimension", C.c_uint32),
        ("dim_x", C.c_uint32), ("dim_y", C.c_uint32), ("dim_z", C.c_uint32),
        ("brick_alloc", C.c_uint64),
        ("material_capacity", C.c_uint32),
        ("device_id", C.c_int32),
        ("want_float_output", C.c_uint32),
        ("enable_counters", C.c_uint32),
        ("shard_rank", C.c_uint32), ("shard_count", C.c_uint32),
        ("tile_w", C.c_uint32), ("tile_h", C.c_uint32),
        ("external_target_rgba8", C.c_void_p),
        ("external_target_rgba32f", C.c_void_p),
        ("stream", C.c_void_p),
        ("kernel_variant", C.c_uint32),
        ("frames_in_flight", C.c_uint32),
        ("shard_root_weight", C.c_uint32),
        ("tuning_flags", C.c_uint32),
        ("_reserved", C.c_uint32 * 4),
    ]


class DistOptions(C.Structure):  # vrt_dist_options
    _fields_ = [("struct_size", C.c_uint32), ("frames_in_flight", C.c_uint32), ("frames_per_launch", C.c_uint32), ("communicators", C.c_uint32),
                ("reserved", C.c_uint32 * 4)]


class ShardInfo(C.Structure):
    _fields_ = [("tiles_x", C.c_uint32), ("tiles_y", C.c_uint32), ("tile_w", C.c_uint32), ("tile_h", C.c_uint32),
                ("shard_rank", C.c_uint32), ("shard_count", C.c_uint32), ("owned_tiles", C.c_uint32),
                ("tiles_per_rank", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("status_loads", C.c_uint64), ("bricks_entered", C.c_uint64),
                ("voxel_steps", C.c_uint64), ("hits", C.c_uint64), ("grid_steps", C.c_uint64)]


class DenoiseConfig(C.Structure):  # GraphicsPipeline.PushConstant / Config, GraphicsPipeline.zig:27-39
    _fields_ = [("samples", C.c_int32), ("distribution_bias", C.c_float), ("pixel_multiplier", C.c_float),
                ("inverse_hue_tolerance", C.c_float)]


class GridConfig(C.Structure):  # Grid.zig:13-20
    _fields_ = [("brick_alloc", C.c_uint64), ("base_t", C.c_float), ("min_point", C.c_float * 3), ("scale", C.c_float),
                ("brick_dimension", C.c_uint32)]


class CameraConfig(C.Structure):  # Camera.zig:5-14
    _fields_ = [("viewport_height", C.c_float), ("origin", C.c_float * 3), ("samples_per_pixel", C.c_int32),
                ("max_bounce", C.c_int32)]


class SunConfig(C.Structure):  # Sun.zig:4-11
    _fields_ = [("enabled", C.c_uint32), ("color", C.c_float * 3), ("radius", C.c_float), ("sun_distance", C.c_float)]


assert C.sizeof(GridState) == 64 and C.sizeof(Material) == 20
assert C.sizeof(CameraDevice) == 96 and C.sizeof(SunDevice) == 32

_P = C.POINTER
_ctx = C.c_void_p
_grid = C.c_void_p

# name -> (restype, argtypes).  Every function declared in include/vrt_hip.h.
SIGNATURES = {
    "vrt_abi_version": (C.c_uint32, []),
    "vrt_last_error": (C.c_char_p, [_ctx]),
    "vrt_kernel_name": (C.c_char_p, [_ctx]),
    "vrt_compiled_kernel_count": (C.c_int, []),
    "vrt_create": (C.c_int, [_P(Config), _P(_ctx)]),
    "vrt_destroy": (None, [_ctx]),
    "vrt_upload": (C.c_int, [_ctx, C.c_int, C.c_uint64, C.c_void_p, C.c_uint64]),
    "vrt_upload_device": (C.c_int, [_ctx, C.c_int, C.c_uint64, C.c_void_p, C.c_uint64]),
    "vrt_buffer_size": (C.c_uint64, [_ctx, C.c_int]),
    "vrt_dispatch": (C.c_int, [_ctx, _P(CameraDevice), _P(SunDevice)]),
    "vrt_dispatch_repeat": (C.c_int, [_ctx, _P(CameraDevice), _P(SunDevice), C.c_uint32]),
    "vrt_dispatch_timed": (C.c_int, [_ctx, _P(CameraDevice), _P(SunDevice), C.c_uint32, _P(C.c_float)]),
    "vrt_wait": (C.c_int, [_ctx]),
    "vrt_read_rgba8": (C.c_int, [_ctx, C.c_void_p, C.c_uint64]),
    "vrt_read_rgba32f": (C.c_int, [_ctx, C.c_void_p, C.c_uint64]),
    "vrt_set_target": (C.c_int, [_ctx, C.c_void_p, C.c_void_p]),
    "vrt_device_target_rgba8": (C.c_void_p, [_ctx]),
    "vrt_device_target_rgba32f": (C.c_void_p, [_ctx]),
    "vrt_target_bytes_rgba8": (C.c_uint64, [_ctx]),
    "vrt_denoise": (C.c_int, [_ctx, _P(DenoiseConfig), C.c_uint32, C.c_uint32, C.c_uint32]),
    "vrt_read_denoised_rgba8": (C.c_int, [_ctx, C.c_void_p, C.c_uint64]),
    "vrt_read_denoised_rgba32f": (C.c_int, [_ctx, C.c_void_p, C.c_uint64]),
    "vrt_device_denoised_rgba8": (C.c_void_p, [_ctx]),
    "vrt_last_denoise_ms": (C.c_double, [_ctx]),
    "vrt_get_shard_info": (C.c_int, [_ctx, _P(ShardInfo)]),
    "vrt_assemble_frame": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_uint32]),
    "vrt_dist_unique_id": (C.c_int, [C.c_char_p, C.c_void_p]),
    "vrt_dist_init": (C.c_int, [_ctx, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_uint32]),
    "vrt_dist_init_batched": (C.c_int, [_ctx, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32]),
    "vrt_dist_frame": (C.c_int, [_ctx, _P(CameraDevice), _P(SunDevice)]),
    "vrt_dist_frames": (C.c_int, [_ctx, _P(CameraDevice), _P(SunDevice), C.c_uint32, C.c_uint32]),
    "vrt_reserve_samples": (C.c_int, [_ctx, C.c_uint32]),
    "vrt_bounce_autotune_info": (C.c_int, [_ctx, _P(C.c_double)]),
    "vrt_dist_wait": (C.c_int, [_ctx]),
    "vrt_dist_read_frame": (C.c_int, [_ctx, C.c_void_p, C.c_uint64]),
    "vrt_dist_selftest": (C.c_int, [_ctx]),
    "vrt_dist_selftest_slots": (C.c_int, [_ctx, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]),
    "vrt_dist_init_ex": (C.c_int, [_ctx, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(DistOptions)]),
    "vrt_dist_keep_communicators": (C.c_int, [C.c_int]),
    "vrt_dist_release_communicators": (C.c_int, []),
    "vrt_dist_comm_info": (C.c_int, [_ctx, C.POINTER(C.c_int32)]),
    "vrt_dist_profile": (C.c_int, [_ctx, C.c_uint32]),
    "vrt_dist_stats": (C.c_int, [_ctx, _P(C.c_double)]),
    "vrt_dist_broadcast": (C.c_int, [_ctx, C.c_int, C.c_uint64, C.c_uint64, C.c_int]),
    "vrt_dist_info": (C.c_int, [_ctx, _P(C.c_int32)]),
    "vrt_device_info": (C.c_int, [C.c_int, _P(C.c_int64)]),
    "vrt_last_kernel_ms": (C.c_double, [_ctx]),
    "vrt_region_begin": (C.c_int, [_ctx]),
    "vrt_region_end": (C.c_int, [_ctx, _P(C.c_double)]),
    "vrt_get_counters": (C.c_int, [_ctx, _P(Counters)]),
    "vrt_get_wave_counters": (C.c_int, [_ctx, _P(C.c_uint64 * 3)]),
    "vrt_trace_wave_timeline": (C.c_int, [_ctx, _P(CameraDevice), _P(SunDevice), C.c_void_p, C.c_uint64, _P(C.c_uint64)]),
    "vrt_grid_create": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, _P(GridConfig), _P(_grid)]),
    "vrt_grid_destroy": (None, [_grid]),
    "vrt_grid_insert": (C.c_int, [_grid, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint8]),
    "vrt_grid_insert_many": (C.c_int, [_grid, C.c_void_p, C.c_void_p, C.c_uint64]),
    "vrt_grid_device_state": (_P(GridState), [_grid]),
    "vrt_grid_data": (C.c_void_p, [_grid, C.c_int, _P(C.c_uint64)]),
    "vrt_grid_active_bricks": (C.c_uint32, [_grid]),
    "vrt_grid_brick_dimension": (C.c_uint32, [_grid]),
    "vrt_grid_delta": (C.c_int, [_grid, C.c_int, _P(C.c_uint64), _P(C.c_uint64)]),
    "vrt_grid_reset_delta": (None, [_grid, C.c_int]),
    "vrt_upload_grid": (C.c_int, [_ctx, _grid]),
    "vrt_update_grid_delta": (C.c_int, [_ctx, _grid]),
    "vrt_camera_init": (C.c_int, [C.c_float, C.c_uint32, C.c_uint32, _P(CameraConfig), _P(CameraDevice)]),
    "vrt_camera_set_forward": (C.c_int, [_P(CameraDevice), C.c_float, C.c_float, _P(C.c_float * 3)]),
    "vrt_sun_init": (C.c_int, [_P(SunConfig), _P(SunDevice)]),
    "vrt_default_materials": (C.c_uint32, [_P(Material), C.c_uint32]),
    "vrt_synth_terrain": (C.c_int, [_grid, C.c_uint64]),
    "vrt_synth_sparse": (C.c_int, [_grid, C.c_uint64, C.c_float]),
    "vrt_benchmark_create": (C.c_int, [_P(CameraDevice), C.c_float, C.c_float, _P(C.c_void_p)]),
    "vrt_benchmark_destroy": (None, [C.c_void_p]),
    "vrt_benchmark_update": (C.c_int, [C.c_void_p, C.c_float, _P(CameraDevice)]),
    "vrt_benchmark_report": (C.c_int, [C.c_void_p, _P(C.c_float), _P(C.c_float), _P(C.c_float)]),
    "vrt_vox_validate_header": (C.c_int, [C.c_char_p, C.c_uint64]),
    "vrt_vox_parse": (C.c_int, [C.c_char_p, C.c_uint64, C.c_int, _P(C.c_void_p)]),
    "vrt_vox_destroy": (None, [C.c_void_p]),
    "vrt_vox_num_models": (C.c_uint32, [C.c_void_p]),
    "vrt_vox_model_size": (C.c_int, [C.c_void_p, C.c_uint32, _P(C.c_int32 * 3)]),
    "vrt_vox_model_voxels": (C.c_void_p, [C.c_void_p, C.c_uint32, _P(C.c_uint64)]),
    "vrt_vox_palette": (C.c_void_p, [C.c_void_p]),
    "vrt_vox_materials": (C.c_int, [C.c_void_p, _P(Material), C.c_uint32]),
    "vrt_vox_insert": (C.c_int, [_grid, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
}

# loader.zig:33-41 ParseError
VOX_E_INVALID_ID, VOX_E_EXPECTED_SIZE_HEADER, VOX_E_EXPECTED_XYZI_HEADER, VOX_E_EXPECTED_RGBA_HEADER = -100, -101, -102, -103
VOX_E_UNEXPECTED_VERSION, VOX_E_INVALID_FILE_CONTENT, VOX_E_MULTIPLE_PACK_CHUNKS = -104, -105, -106
ERROR_NAMES.update({VOX_E_INVALID_ID: "InvalidId", VOX_E_EXPECTED_SIZE_HEADER: "ExpectedSizeHeader",
                    VOX_E_EXPECTED_XYZI_HEADER: "ExpectedXyziHeader", VOX_E_EXPECTED_RGBA_HEADER: "ExpectedRgbaHeader",
                    VOX_E_UNEXPECTED_VERSION: "UnexpectedVersion", VOX_E_INVALID_FILE_CONTENT: "InvalidFileContent",
                    VOX_E_MULTIPLE_PACK_CHUNKS: "MultiplePackChunks"})


def load_library(path: str) -> C.CDLL:
    """Load one build of the library and bind every function include/vrt_hip.h declares (AttributeError if one is absent)."""
    # torch bundles its own libamdhip64.so.7; whichever HIP runtime is mapped first serves the whole
    # process.  Load torch's first so libvrt_hip.so (NEEDED libamdhip64.so.7) shares it — the other
    # order leaves torch with "No HIP GPUs are available".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = os.path.abspath(path)
    if path in _loaded:
        return _loaded[path]
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `make -C zig_vulkan_amd/csrc` (or __graft_entry__.build()). "
            "zig_vulkan_amd has no non-HIP implementation of the traversal path.")
    lib = C.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.vrt_abi_version() != VRT_ABI_VERSION:
        raise ImportError(f"{path}: ABI version mismatch")
    _loaded[path] = lib
    return lib


_loaded: dict = {}
lib = load_library(LIB_PATH)
# test-infrastructure twins of the product library (zig_vulkan_amd/csrc/Makefile); absent unless built
FUSED_LIB_PATH = os.path.join(_HERE, "libvrt_hip_fused.so")    # fma fused, dot as an fma chain (make fused): how far fusing moves the frames
DEV_LIB_PATH = os.path.join(_HERE, "libvrt_hip_dev.so")        # + the variants that lost their A/B measurement (make dev)


def rccl_library_path() -> str:
    """The RCCL the process already uses: PyTorch's bundled librccl.so (backend "nccl" on ROCm)."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        origin = sys.modules["torch"].__file__
    else:   # (located, not imported: a process that only needs the library's path does not pay for `import torch`)
        spec = importlib.util.find_spec("torch")
        origin = spec.origin if spec is not None else None
    path = os.path.join(os.path.dirname(origin), "lib", "librccl.so") if origin else ""
    return path if path and os.path.exists(path) else "librccl.so"


def check(rc: int, ctx=None) -> None:
    if rc != VRT_OK:
        msg = lib.vrt_last_error(ctx)
        raise VrtError(rc, msg.decode() if msg else "")
