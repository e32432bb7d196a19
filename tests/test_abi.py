"""The C-ABI library loads and exports every symbol include/vrt_hip.h declares (no compute calls
without a GPU), and the data-contract structs have the reference's sizes and offsets."""
import ctypes as C
import os
import re

import pytest

from zig_vulkan_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "vrt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vrt_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared_functions()
    assert len(names) >= 35
    raw = C.CDLL(L.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in vrt_hip.h but not exported by libvrt_hip.so"
        assert n in L.SIGNATURES, f"{n} has no ctypes signature in zig_vulkan_amd/_lib.py"
    assert sorted(L.SIGNATURES) == names


def test_abi_version():
    assert L.lib.vrt_abi_version() == L.VRT_ABI_VERSION == 1


def test_struct_layouts_match_reference():
    # State.Device, State.zig:60-79
    assert C.sizeof(L.GridState) == 64
    assert L.GridState.min_point_base_t.offset == 32 and L.GridState.max_point_scale.offset == 48
    # gpu_types.Material, gpu_types.zig:16-32 (std430 stride 20)
    assert C.sizeof(L.Material) == 20
    # Camera.Device, Camera.zig:183-193 / push constants comp:58-69
    cd = L.CameraDevice
    assert C.sizeof(cd) == 96
    assert (cd.image_width.offset, cd.image_height.offset) == (0, 4)
    assert (cd.horizontal.offset, cd.vertical.offset, cd.lower_left_corner.offset, cd.origin.offset) == (16, 32, 48, 64)
    assert (cd.samples_per_pixel.offset, cd.max_bounce.offset) == (80, 84)
    # Sun.Device, Sun.zig:13-18; pushed at byte 96 (ComputePipeline.zig:488-505)
    sd = L.SunDevice
    assert C.sizeof(sd) == 32
    assert (sd.position.offset, sd.enabled.offset, sd.color.offset, sd.radius.offset) == (0, 12, 16, 28)


def test_buffer_ids_follow_shader_bindings():
    # bindings 1..7, comp:79,105,112,117,124,128,132
    assert [L.BUF_GRID_STATE, L.BUF_MATERIALS, L.BUF_BRICK_STATUS, L.BUF_BRICK_INDEX, L.BUF_BRICK_OCCUPANCY,
            L.BUF_BRICK_START_INDEX, L.BUF_MATERIAL_INDEX] == list(range(7))


def test_create_rejects_bad_arguments_before_touching_a_device():
    h = C.c_void_p()
    assert L.lib.vrt_create(None, C.byref(h)) == L.VRT_E_INVALID_ARG
    cfg = L.Config()
    assert L.lib.vrt_create(C.byref(cfg), C.byref(h)) == L.VRT_E_INVALID_ARG  # struct_size/abi 0
    assert b"ABI" in L.lib.vrt_last_error(None)
    cfg.struct_size, cfg.abi_version = C.sizeof(L.Config), L.VRT_ABI_VERSION
    cfg.width, cfg.height, cfg.brick_dimension = 64, 64, 5
    cfg.dim_x = cfg.dim_y = cfg.dim_z = 4
    assert L.lib.vrt_create(C.byref(cfg), C.byref(h)) == L.VRT_E_INVALID_ARG  # brick_dimension
    cfg.brick_dimension = 8
    cfg.dim_x = cfg.dim_y = cfg.dim_z = 2048  # 2048^3 bricks of 512 bits: u31 start index overflows
    assert L.lib.vrt_create(C.byref(cfg), C.byref(h)) == L.VRT_E_OUT_OF_RANGE
    cfg.dim_x = cfg.dim_y = cfg.dim_z = 4
    cfg.shard_rank, cfg.shard_count = 3, 2
    assert L.lib.vrt_create(C.byref(cfg), C.byref(h)) == L.VRT_E_INVALID_ARG
    assert not h.value


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="this check is for a box without a GPU")
def test_no_device_fails_loudly_no_cpu_fallback():
    cfg = L.Config()
    cfg.struct_size, cfg.abi_version = C.sizeof(L.Config), L.VRT_ABI_VERSION
    cfg.width, cfg.height, cfg.brick_dimension = 64, 64, 4
    cfg.dim_x = cfg.dim_y = cfg.dim_z = 4
    h = C.c_void_p()
    assert L.lib.vrt_create(C.byref(cfg), C.byref(h)) == L.VRT_E_NO_DEVICE
    assert not h.value
    assert b"no CPU path" in L.lib.vrt_last_error(None)


def test_product_never_references_the_oracle():
    """The oracle is test infrastructure: nothing under zig_vulkan_amd/ may import, link or load it."""
    pkg = os.path.join(ROOT, "zig_vulkan_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "libvrt_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
                assert "oracle/" not in text.replace("parity oracle", ""), f
