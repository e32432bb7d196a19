"""libvrt_hip.so == the reference shader's own frames, BIT FOR BIT — bounces, soft sun and every scatter function included.

The kernels' arithmetic contract (zig_vulkan_amd/csrc/vrt_math.h) is the reference shader's arithmetic as the only executable
implementation of the reference performs it — Mesa's gallivm / llvmpipe, the back end of lavapipe: fma as a*b + c, dot reduced from
the last channel, Mesa's two rewrites of hash12, gallivm's sin.  Everything else is the shader's operations one by one.  So the
product must reproduce, with zero differing bits (north_star asks for 1e-4 per channel),

  * tests/golden/ref/*.npz      all eleven frames the reference shader rendered under llvmpipe (160x90 ... 256x256), and
  * tests/golden/ref_full/*.npz the headline workload at its full 1920x1080 / 512^3 / 8^3 bricks (hard and soft sun, V0/V1/V2)
                                and the reference app's own default run (1024x576, 128x64x128 bricks of 4^3, 2 spp, 2 bounces),
                                as SHA-256 of the whole frame + per-band hashes + float crops,

made by tests/golden/make_ref_golden.py from /root/reference/assets/shaders/brick_raytracer.comp:153-596 and rand.comp:3-26.
That pins every statement of the kernels — ray set-up, both DDA levels, the skip to the occupied-cell box, scatter functions, RNG,
shading, tone-map, store, the path kernel's lane scheduling — to the reference itself, not only to the oracle.

libvrt_hip_fused.so (make fused; built by __graft_entry__.build()) is the same source with fma fused and dot as an fma chain —
what a GPU driver's compiler would emit for the same GLSL, and this repo's contract until round 3.  The last tests say how far that
moves the frames: last bits everywhere, whole pixels where a DDA tie flips or the sin-hash RNG is reached.
"""
import glob
import hashlib
import os

import numpy as np
import pytest

from tests.test_ref_gl import FIXTURES, IDS, _hip_render
from zig_vulkan_amd import _lib as L
from zig_vulkan_amd import workloads as W

pytestmark = pytest.mark.gpu

FULL = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ref_full", "*.npz")))
FULL_IDS = [os.path.basename(p)[:-4] for p in FULL]
PATH = 1 << 23       # force vrt_path_kernel (persistent lanes) on scenes the library would give to the lockstep kernel
LOCKSTEP = 1 << 21


def _fused():
    if not os.path.exists(L.FUSED_LIB_PATH):
        pytest.skip("libvrt_hip_fused.so not built (make -C zig_vulkan_amd/csrc fused)")
    return L.FUSED_LIB_PATH


def _assert_is_fixture(f, u, z):
    diff = int((f[:, :, :3].view(np.uint32) != z["rgb32f"].view(np.uint32)).any(axis=2).sum())
    assert diff == 0, f"{diff} of {f.shape[0] * f.shape[1]} pixels differ from the reference shader's frame"
    assert np.array_equal(u, z["rgba8"])
    assert np.all(f[:, :, 3] == 1.0)
    assert hashlib.sha256(np.ascontiguousarray(f).tobytes()).hexdigest() == str(z["float_sha256"])


@pytest.mark.parametrize("path", FIXTURES, ids=IDS)
def test_kernels_reproduce_the_reference_shaders_frame_bit_for_bit(path):
    z = np.load(path)
    f, u = _hip_render(z)
    _assert_is_fixture(f, u, z)


@pytest.mark.parametrize("path", [p for p in FIXTURES if "path_" in p or "sparse_" in p], ids=lambda p: os.path.basename(p)[:-4])
@pytest.mark.parametrize("variant", [PATH, LOCKSTEP], ids=["path-kernel", "lockstep-kernel"])
def test_bounce_frames_on_both_bounce_kernels(path, variant):
    """The fixtures with bounces (all scatter functions, soft sun, several samples) through vrt_path_kernel and through the lockstep
    kernel: the reference's frame either way."""
    z = np.load(path)
    f, u = _hip_render(z, kernel_variant=variant)
    _assert_is_fixture(f, u, z)


def test_fused_twin_differs_only_where_lowering_can():
    """Sanity of the set-up itself: the fused twin is NOT the product (on a stochastic fixture the two builds must differ somewhere
    — if they did not, the flag would not have reached the kernels), and on a deterministic fixture they agree to the last few bits."""
    z = np.load([p for p in FIXTURES if "path_b4_spp3" in p][0])
    ff, _ = _hip_render(z, library=_fused())
    fp, _ = _hip_render(z)
    assert not np.array_equal(ff, fp)
    z = np.load([p for p in FIXTURES if "cfg0_V1" in p][0])
    ff, _ = _hip_render(z, library=_fused())
    fp, _ = _hip_render(z)
    d = np.abs(ff - fp).max(axis=2)
    assert float((d <= 1e-6).mean()) > 0.999


@pytest.mark.parametrize("path", FIXTURES, ids=IDS)
def test_fused_twin_is_the_fused_oracle_and_within_tolerance_of_the_reference(path):
    """libvrt_hip_fused.so == oracle (fused lowering) bit for bit on the fixture's own inputs, and within north_star's tolerance of the
    reference's frame as far as fused arithmetic can be (deterministic frames: all but a handful of pixels; stochastic: as images)."""
    from oracle import oracle as O
    from tests.test_ref_gl import _compare_hw_lowering, _scene
    z = np.load(path)
    f, u = _hip_render(z, library=_fused())
    fo, uo, _ = O.render(_scene(z), z["push_constants"].copy(), lowering="fused")
    assert np.array_equal(f.view(np.uint32), fo.view(np.uint32)) and np.array_equal(u, uo)
    _compare_hw_lowering(os.path.basename(path), f, u, z)


# ---------------------------------------------------------------------------------------------- full size
_GRIDS = {}


def _full_render(z, **config):
    from tests.golden.make_golden import scene_digest
    import ctypes as C
    w = W.WORKLOADS[str(z["workload"])]
    grid = _GRIDS.get(w.name)
    if grid is None:
        _GRIDS.clear()      # (one 146 MiB grid at a time)
        grid = _GRIDS.setdefault(w.name, W.build_grid(w))
    assert scene_digest(grid) == str(z["scene_sha256"]), "the synthetic scene generator no longer produces the fixture's scene"
    width, height = (int(v) for v in z["size"])
    rt = W.make_renderer(w, grid, width=width, height=height, want_float_output=True, **config)
    pc = z["push_constants"].tobytes()
    C.memmove(C.byref(rt.camera.d_camera), pc[:96], 96)
    C.memmove(C.byref(rt.sun.device_data), pc[96:], 32)
    rt.draw()
    f, u, name = rt.read_rgba32f(), rt.read_rgba8(), rt.kernel_name()
    rt.deinit()
    return f, u, name


@pytest.mark.parametrize("path", FULL, ids=FULL_IDS)
def test_kernels_reproduce_the_reference_shader_at_full_size(path):
    z = np.load(path)
    f, u, name = _full_render(z)
    width, height = (int(v) for v in z["size"])
    band = int(z["band_rows"])
    bad = [i for i, y in enumerate(range(0, height, band))
           if hashlib.sha256(np.ascontiguousarray(f[y:y + band]).tobytes()).hexdigest() != str(z["band_sha256"][i])]
    crop = z["float_crops"].shape[1]
    for (y, x), want in zip(z["crop_origins"], z["float_crops"]):
        got = f[y:y + crop, x:x + crop, :3]
        n = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert n == 0, f"{name}: crop at ({x},{y}): {n} pixels differ from the reference shader's"
    assert not bad, f"{name}: bands of {band} rows that differ from the reference shader's frame: {bad[:20]} ({len(bad)} of {len(z['band_sha256'])})"
    assert hashlib.sha256(f.tobytes()).hexdigest() == str(z["float_sha256"])
    assert hashlib.sha256(u.tobytes()).hexdigest() == str(z["rgba8_sha256"])


def _assert_is_full_fixture(f, u, z, name):
    width, height = (int(v) for v in z["size"])
    band = int(z["band_rows"])
    crop = z["float_crops"].shape[1]
    for (y, x), want in zip(z["crop_origins"], z["float_crops"]):
        got = f[y:y + crop, x:x + crop, :3]
        n = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
        assert n == 0, f"{name}: crop at ({x},{y}): {n} pixels differ from the reference shader's"
    bad = [i for i, y in enumerate(range(0, height, band))
           if hashlib.sha256(np.ascontiguousarray(f[y:y + band]).tobytes()).hexdigest() != str(z["band_sha256"][i])]
    assert not bad, f"{name}: bands of {band} rows that differ from the reference shader's frame: {bad[:20]} ({len(bad)} of {len(z['band_sha256'])})"
    assert hashlib.sha256(f.tobytes()).hexdigest() == str(z["float_sha256"])
    assert hashlib.sha256(u.tobytes()).hexdigest() == str(z["rgba8_sha256"])


BIG_PATH = [p for p in FULL if "big_path_" in p]
BIG_SHADOW = [p for p in FULL if "big_shadow_" in p]


@pytest.mark.parametrize("path", BIG_PATH, ids=lambda p: os.path.basename(p)[:-4])
@pytest.mark.parametrize("kernels", ["pool", "dil2", "lockstep"])
def test_the_kernels_of_configs4_reproduce_the_reference_shader_at_4k(path, kernels):
    """vrt_pool_kernel — the kernel BASELINE configs[4] settles on — and its neighbours against the REFERENCE SHADER's own
    frame, not only the oracle's: 3840x2160 on a 1024^3 sparse scene of configs[4]'s kind (the largest whose buffers fit GL's 128 MiB
    storage blocks), 4 samples, 3 bounces, soft sun, rendered by brick_raytracer.comp under llvmpipe (tests/golden/make_ref_golden.py
    big).  `pool`: the persistent kernels forced (the scene would stay with the lockstep kernel by size): the first frame by
    vrt_path_kernel<..., DIL 1>, the second — the box of the occupied cells known — by vrt_pool_kernel<8, 6, 60, 2>; BOTH frames must be
    the reference's, bit for bit (float and RGBA8, every band).  `dil2`: the pool kernel switched off, vrt_path_kernel<..., DIL 2>.
    `lockstep`: vrt_trace_kernel<8, false, 4, 5, 0, 256>."""
    import ctypes as C
    from tests.golden.make_golden import scene_digest
    z = np.load(path)
    w = W.WORKLOADS[str(z["workload"])]
    grid = _GRIDS.get(w.name)
    if grid is None:
        _GRIDS.clear()
        grid = _GRIDS.setdefault(w.name, W.build_grid(w))
    assert scene_digest(grid) == str(z["scene_sha256"])
    cfg = dict(pool=dict(kernel_variant=PATH), dil2=dict(kernel_variant=PATH, tuning_flags=L.TUNE_NO_PATH_POOL), lockstep=dict(kernel_variant=LOCKSTEP))[kernels]
    rt = W.make_renderer(w, grid, want_float_output=True, **cfg)
    pc = z["push_constants"].tobytes()
    C.memmove(C.byref(rt.camera.d_camera), pc[:96], 96)
    C.memmove(C.byref(rt.sun.device_data), pc[96:], 32)
    names = []
    for _ in range(2):
        rt.draw()
        f, u = rt.read_rgba32f(), rt.read_rgba8()       # (the read waits: the box of the occupied cells has reached the host by the second frame)
        names.append(rt.kernel_name())
        _assert_is_full_fixture(f, u, z, names[-1])
    rt.deinit()
    if kernels == "pool":
        assert names[0] in ("vrt_path_kernel<8, 5, false, false, false, false, 1>", "vrt_pool_kernel<8, 6, 60, 2>") and names[1] == "vrt_pool_kernel<8, 6, 60, 2>", names
    elif kernels == "dil2":
        assert names[1] == "vrt_path_kernel<8, 5, false, false, false, false, 2>", names
    else:
        assert names == ["vrt_trace_kernel<8, false, 4, 5, 0, 256>"] * 2, names


@pytest.mark.parametrize("path", BIG_SHADOW, ids=lambda p: os.path.basename(p)[:-4])
def test_the_kernel_of_configs3_reproduces_the_reference_shader_at_4k(path):
    """... and vrt_trace_kernel<8, false, 4, 7, 1, 256> — what BASELINE configs[3] runs (4K, 2 samples x (primary + shadow ray), soft sun;
    128^3 cells: the status bits by words) — against the reference shader's frame of the same 4K / 1024^3 scene."""
    z = np.load(path)
    f, u, name = _full_render(z)
    assert name == "vrt_trace_kernel<8, false, 4, 7, 1, 256>", name
    _assert_is_full_fixture(f, u, z, name)


def test_full_size_fixtures_exist():
    assert len(BIG_PATH) == 2 and len(BIG_SHADOW) == 2
    names = set(FULL_IDS)
    assert {f"cfg2_r{r}_{v}" for r in (0, 5) for v in ("V0", "V1", "V2")} <= names
    assert {"refapp_V0", "refapp_256x144_V0", "refapp_256x144_V2"} <= names
    assert {f"cfg1_{v}" for v in ("V0", "V1", "V2")} <= names      # (round 5: BASELINE configs[1], whole frames by the reference shader)
    for p in FULL:
        assert "brick_raytracer.comp" in str(np.load(p)["provenance"])


@pytest.mark.parametrize("path", [p for p in FULL if "cfg2_r0_" in p], ids=lambda p: os.path.basename(p)[:-4])
def test_fused_twin_within_tolerance_of_the_reference_shader_at_full_size(path):
    """The fused twin against the reference shader's own headline frames, hard sun (no pixel depends on sin): within north_star's
    1e-4 per channel on the crops except at isolated pixels where a last-bit difference flips a DDA tie or a hit / miss (VERDICT r02
    measured 0 / 11 / 29 such pixels of 2 073 600 on V0 / V1 / V2)."""
    z = np.load(path)
    f, u, name = _full_render(z, library=_fused())
    crop = z["float_crops"].shape[1]
    flipped = total = 0
    for (y, x), want in zip(z["crop_origins"], z["float_crops"]):
        d = np.abs(f[y:y + crop, x:x + crop, :3] - want).max(axis=2)
        flipped += int((d > 1e-4).sum())
        total += d.size
        assert float(d[d <= 1e-4].max()) <= 1e-6
    assert flipped <= max(2, total // 2000), (name, flipped, total)
    assert np.abs(f[:, :, :3].mean(axis=(0, 1)) - z["mean_rgb"]).max() <= 1e-5
