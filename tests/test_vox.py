""".vox loader.  The first three tests are the reference's own three unit tests
(src/modules/voxel_rt/vox/loader.zig:265-281) — the only tests the reference has — carried over with
the same inputs and expectations; the rest pin parse semantics on hand-built buffers."""
import hashlib
import struct

import numpy as np
import pytest

from zig_vulkan_amd import BrickGrid, vox
from zig_vulkan_amd import _lib as L
from zig_vulkan_amd._lib import VrtError


def test_validate_header_valid_header_accepted():  # loader.zig:265-269
    vox.validate_header(b"VOX " + bytes([150, 0, 0, 0]) + b"MAIN")


def test_validate_header_invalid_id_detected():  # loader.zig:271-275
    with pytest.raises(VrtError) as e:
        vox.validate_header(b"!VOX" + bytes([150, 0, 0, 0]) + b"MAIN")
    assert e.value.code == L.VOX_E_INVALID_ID


def test_validate_header_invalid_version_detected():  # loader.zig:277-281
    with pytest.raises(VrtError) as e:
        vox.validate_header(b"VOX " + bytes([169, 0, 0, 0]) + b"MAIN")
    assert e.value.code == L.VOX_E_UNEXPECTED_VERSION


def chunk(tag: bytes, content: bytes, children: bytes = b"") -> bytes:
    return tag + struct.pack("<ii", len(content), len(children)) + content + children


def make_vox(models, rgba=None, pack=None, extra=b"") -> bytes:
    body = b""
    if pack is not None:
        body += chunk(b"PACK", struct.pack("<i", pack))
    for size, voxels in models:
        body += chunk(b"SIZE", struct.pack("<iii", *size))
        body += chunk(b"XYZI", struct.pack("<i", len(voxels)) + b"".join(bytes(v) for v in voxels))
    body += extra
    if rgba is not None:
        body += chunk(b"RGBA", rgba)
    return b"VOX " + struct.pack("<i", 150) + chunk(b"MAIN", b"", body)


def test_single_model_default_palette():
    buf = make_vox([((3, 4, 5), [(0, 1, 2, 7), (2, 3, 4, 255)])])
    v = vox.parse_buffer(buf)
    assert v.num_models == 1 and v.size() == (3, 4, 5)
    assert v.xyzi().tolist() == [[0, 1, 2, 7], [2, 3, 4, 255]]
    pal = v.rgba
    # MagicaVoxel default palette: spot values and a digest of all 1024 bytes (0xAABBGGRR words, little endian)
    assert pal[0].tolist() == [0, 0, 0, 0] and pal[1].tolist() == [255, 255, 255, 255]
    assert pal[2].tolist() == [255, 255, 0xcc, 255] and pal[37].tolist() == [0xcc, 255, 255, 255]
    assert pal[216].tolist() == [0xee, 0, 0, 255] and pal[255].tolist() == [0x11, 0x11, 0x11, 255]
    assert hashlib.sha256(pal.tobytes()).hexdigest() == "cc9800abac4eea5f0dc2b29399d7d6440165026723622ca30ffd77dd71653605"


def test_pack_two_models_rgba_and_unknown_chunks():
    rgba = bytes(range(256)) * 4  # 1024 bytes: entries (0,1,2,3), (4,5,6,7), ...
    extra = chunk(b"nTRN", b"\x00" * 8)  # an extension chunk: skipped 4 bytes at a time (loader.zig:190-193)
    buf = make_vox([((1, 1, 1), [(0, 0, 0, 1)]), ((2, 2, 2), [(1, 1, 1, 2), (0, 1, 0, 3)])], rgba=rgba, pack=2, extra=extra)
    v = vox.parse_buffer(buf)
    assert v.num_models == 2 and v.size(1) == (2, 2, 2) and len(v.xyzi(1)) == 2
    pal = v.rgba
    assert pal[0].tolist() == [0, 0, 0, 1]              # loader.zig:169-174
    assert pal[1].tolist() == [0, 1, 2, 3] and pal[254].tolist() == [244, 245, 246, 247]  # file colour i-1 -> entry i
    assert pal[255].tolist() == [0, 0, 0, 0]            # never written by the reference's `while (i < 255)` loop


def test_strict_mode_errors_and_truncation():
    good = make_vox([((1, 1, 1), [(0, 0, 0, 1)])])
    bad_size = good.replace(b"SIZE", b"SIZF")
    with pytest.raises(VrtError) as e:
        vox.parse_buffer(bad_size)
    assert e.value.code == L.VOX_E_EXPECTED_SIZE_HEADER
    with pytest.raises(VrtError) as e:
        vox.parse_buffer(good.replace(b"XYZI", b"XYZJ"))
    assert e.value.code == L.VOX_E_EXPECTED_XYZI_HEADER
    vox.parse_buffer(bad_size, strict=False)  # non-strict skips the tag checks like the reference
    for cut in range(12, len(good)):  # every truncation is an error, never an out-of-bounds read
        with pytest.raises(VrtError) as e:
            vox.parse_buffer(good[:cut], strict=False)
        assert e.value.code == L.VOX_E_INVALID_FILE_CONTENT
        with pytest.raises(VrtError) as e:
            vox.parse_buffer(good[:cut], strict=True)
        assert e.value.code in (L.VOX_E_INVALID_FILE_CONTENT, L.VOX_E_EXPECTED_SIZE_HEADER, L.VOX_E_EXPECTED_XYZI_HEADER)
    with pytest.raises(VrtError) as e:
        vox.parse_buffer(b"VOX " + struct.pack("<i", 150) + b"NOPE" + b"\0" * 40)
    assert e.value.code == L.VOX_E_INVALID_FILE_CONTENT


def test_palette_to_materials_and_grid_insert_like_main_zig():
    rgba = bytearray(1024)
    rgba[0:4] = bytes([255, 128, 0, 255])   # palette entry 1: opaque -> lambertian
    rgba[4:8] = bytes([10, 20, 30, 100])    # palette entry 2: alpha 100/255 < 0.8 -> dielectric 1.52
    v = vox.parse_buffer(make_vox([((4, 4, 4), [(1, 2, 3, 1), (0, 0, 0, 2)])], rgba=bytes(rgba)))
    m = v.materials(8)
    assert m["type"][1] == 0 and np.allclose([m["albedo_r"][1], m["albedo_g"][1], m["albedo_b"][1]], [1.0, 128 / 255, 0.0])
    assert m["type"][2] == 2 and np.isclose(m["type_data"][2], 1.52)
    g = BrickGrid(4, 4, 4, brick_dimension=4)
    v.insert_into(g, 0, offset=(5, 6, 7), material_offset=8)  # main.zig:109-117: (x+ox, z+oy, y+oz), index + 8
    h = BrickGrid(4, 4, 4, brick_dimension=4)
    h.insert(1 + 5, 3 + 6, 2 + 7, 1 + 8)
    h.insert(0 + 5, 0 + 6, 0 + 7, 2 + 8)
    for bid in (L.BUF_BRICK_STATUS, L.BUF_BRICK_INDEX, L.BUF_BRICK_OCCUPANCY, L.BUF_BRICK_START_INDEX, L.BUF_MATERIAL_INDEX):
        assert np.array_equal(g.array(bid), h.array(bid))
