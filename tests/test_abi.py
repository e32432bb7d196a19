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
    assert L.lib.vrt_abi_version() == L.VRT_ABI_VERSION == 4
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert "#define VRT_ABI_VERSION 4u" in open(os.path.join(root, "include", "vrt_hip.h")).read()
    assert "abi_version: u32 = 4," in open(os.path.join(root, "bindings", "vrt_hip.zig")).read()   # the Zig host's default


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


# ---- bindings/vrt_hip.zig: zig is not in the image, so the binding is checked against the header mechanically ----
def _zig():
    return open(os.path.join(ROOT, "bindings", "vrt_hip.zig")).read()


def _header():
    return open(os.path.join(ROOT, "include", "vrt_hip.h")).read()


def test_zig_binding_extern_block_is_generated_from_the_header():
    import subprocess
    import sys
    rc = subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_zig_binding.py"), "--check"])
    assert rc == 0, "bindings/vrt_hip.zig is out of date: run python tools/gen_zig_binding.py"
    declared = set(re.findall(r"pub extern fn (vrt_\w+)\(", _zig()))
    assert declared == set(_declared_functions())


def test_zig_binding_lists_every_tuning_flag_of_the_header():
    """ABI 3 added three VRT_TUNE_* bits the Zig binding lacked (a Zig host could not switch the auto-tune off)."""
    header = open(os.path.join(ROOT, "include", "vrt_hip.h")).read()
    want = {n: int(b) for n, b in re.findall(r"#define\s+VRT_(TUNE_\w+)\s+\(1u\s*<<\s*(\d+)\)", header)}
    have = {n: int(b) for n, b in re.findall(r"pub const (TUNE_\w+): u32 = 1 << (\d+);", _zig())}
    assert len(want) >= 21 and have == want


def test_zig_status_enum_lists_every_status_code_of_the_header():
    """Every value of the header's two status enums has a name in `Status`, `Status` is non-exhaustive (`_`), and check()
    handles each name: a code the binding does not know can never be illegal behaviour (VERDICT r01 Weak #7)."""
    hdr = re.sub(r"/\*.*?\*/", "", _header(), flags=re.S)
    codes = {name: int(val) for name, val in re.findall(r"\b(VRT_(?:OK|E_\w+|VOX_E_\w+))\s*=\s*(-?\d+)", hdr)}
    assert codes["VRT_E_RCCL"] == -7 and codes["VRT_VOX_E_MULTIPLE_PACK_CHUNKS"] == -106 and len(codes) == 15
    zig = _zig()
    body = zig[zig.index("pub const Status = enum(c_int) {"):]
    body = body[:body.index("};")]
    zig_codes = {name: int(val) for name, val in re.findall(r"^\s*(\w+)\s*=\s*(-?\d+),", body, flags=re.M)}
    assert sorted(zig_codes.values()) == sorted(codes.values())
    assert re.search(r"^\s*_,\s*$", body, flags=re.M), "Status must be a non-exhaustive enum"
    check = zig[zig.index("pub fn check(rc: c_int)"):]
    check = check[:check.index("\n}\n")]
    for name in zig_codes:
        assert f".{name} =>" in check, f"check() does not handle .{name}"
    assert "else => error.VrtFailure" in check
    # the same codes are the ones the ctypes binding names
    assert {getattr(L, n) for n in dir(L) if n.startswith(("VRT_E_", "VOX_E_"))} | {0} == set(codes.values())


def _c_struct_fields(hdr, name):
    m = re.search(r"typedef struct " + name + r" \{(.*?)\} " + name + r";", hdr, flags=re.S)
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        names = decl.split(",")
        first = names[0].rsplit(" ", 1)[1] if " " in names[0] else names[0]
        for n in [first] + [x.strip() for x in names[1:]]:
            fields.append(re.sub(r"\[\d+\]|\*", "", n).strip())
    return fields


def test_zig_extern_structs_have_the_headers_fields_in_order():
    hdr, zig = _header(), _zig()
    pairs = {"vrt_grid_state": "GridState", "vrt_material": "Material", "vrt_camera_device": "CameraDevice", "vrt_sun_device": "SunDevice",
             "vrt_config": "Config", "vrt_shard_info": "ShardInfo", "vrt_counters": "Counters", "vrt_grid_config": "GridConfig",
             "vrt_camera_config": "CameraConfig", "vrt_sun_config": "SunConfig", "vrt_denoise_config": "DenoiseConfig",
             "vrt_vox_xyzi": "VoxXyzi", "vrt_vox_rgba": "VoxRgba"}
    for cname, zname in pairs.items():
        m = re.search(r"pub const " + zname + r" = extern struct \{(.*?)\};", zig, flags=re.S)
        assert m, zname
        zfields = re.findall(r"(?:^|[\{,\n])\s*(\w+):", re.sub(r"//[^\n]*", "", m.group(1)))
        assert zfields == _c_struct_fields(hdr, cname), (cname, zfields, _c_struct_fields(hdr, cname))
