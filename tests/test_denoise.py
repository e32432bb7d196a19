"""Present/denoise pass: oracle KATs on CPU, HIP kernel vs oracle on the GPU.
Floating-point kernel with pow(): tolerance 1e-4 per channel (north_star), RGBA8 within 1 LSB."""
import numpy as np
import pytest

from tests.helpers import O
from zig_vulkan_amd import workloads as W


def test_oracle_constant_image_is_a_fixed_point():
    # every tap returns the same colour c: sum(c*w)/sum(w) == c whatever the weights (image.frag:70)
    img = np.zeros((32, 48, 4), dtype=np.uint8)
    img[...] = (200, 120, 40, 255)
    f, u = O.denoise(img, 48, 32)
    want = np.array([200, 120, 40], dtype=np.float32) / np.float32(255)
    assert np.abs(f[..., :3] - want).max() < 2e-6 and (f[..., 3] == 1).all()
    assert (u[..., :3] == [200, 120, 40]).all() and (u[..., 3] == 255).all()
    f2, _ = O.denoise(img, 96, 64)  # other output resolution (window size differs from the traced image)
    assert np.abs(f2[..., :3] - want).max() < 2e-6


def test_oracle_black_taps_poison_the_pixel_like_the_shader():
    # normalize(vec3(0)) is NaN in the shader (image.frag:40,61): a black centre or tap makes the sum NaN,
    # which the UNORM store turns into 0
    img = np.zeros((16, 16, 4), dtype=np.uint8)
    img[..., :3] = 180
    img[8, 8, :3] = 0
    f, u = O.denoise(img, 16, 16)
    assert np.isnan(f[8, 8, :3]).all() and (u[8, 8, :3] == 0).all()
    assert not np.isnan(f[0, 0, :3]).any()  # far away (spiral radius ~ 0.75*sqrt(20) px) stays clean


def test_oracle_edge_preservation_direction():
    # two flat regions: pixels well inside a region keep its colour (hue/saturation filter rejects the other)
    img = np.zeros((32, 64, 4), dtype=np.uint8)
    img[:, :32, :3] = (220, 40, 40)
    img[:, 32:, :3] = (40, 40, 220)
    f, _ = O.denoise(img, 64, 32)
    assert np.abs(f[16, 8, :3] * 255 - [220, 40, 40]).max() < 0.5
    assert np.abs(f[16, 56, :3] * 255 - [40, 40, 220]).max() < 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("out_size", [(320, 200), (400, 260)])
def test_hip_denoise_matches_oracle(out_size):
    w = W.Workload("t", 320, 200, 64, 4, 2, 2, False, 0.0)  # sun off: no pitch-black shadow pixels, few NaNs
    grid = W.build_grid(w)
    rt = W.make_renderer(w, grid)
    W.set_view(rt, "V2")
    rt.draw()
    traced = rt.read_rgba8()
    u, f = rt.denoise(out_size[0], out_size[1], want_float=True)
    rt.deinit()
    fo, uo = O.denoise(traced, out_size[0], out_size[1])
    nan_o, nan_k = np.isnan(fo[..., :3]).any(axis=-1), np.isnan(f[..., :3]).any(axis=-1)
    assert np.array_equal(nan_o, nan_k)
    ok = ~nan_o
    assert np.abs(f[ok] - fo[ok]).max() <= 1e-4
    assert np.abs(u.astype(int) - uo.astype(int)).max() <= 1
    assert (u[..., 3] == 255).all()
    assert not np.array_equal(u[:, :, :3], 0 * u[:, :, :3])


@pytest.mark.gpu
def test_hip_denoise_with_shadows_nan_pattern():
    w = W.Workload("t", 256, 160, 64, 4, 1, 0, True, 0.0)  # hard shadows: exactly black pixels exist
    grid = W.build_grid(w)
    rt = W.make_renderer(w, grid)
    W.set_view(rt, "V1")
    rt.draw()
    traced = rt.read_rgba8()
    u, f = rt.denoise(256, 160, samples=12, pixel_multiplier=2.0, want_float=True)
    rt.deinit()
    fo, uo = O.denoise(traced, 256, 160, samples=12, pixel_multiplier=2.0)
    assert np.array_equal(np.isnan(fo), np.isnan(f))
    ok = ~np.isnan(fo)
    assert np.abs(f[ok] - fo[ok]).max() <= 1e-4
    assert np.abs(u.astype(int) - uo.astype(int)).max() <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("frame, out_size, kw", [
    ((320, 200), (320, 200), dict(samples=20)),                                   # the workgroup's box of texels fits LDS: the staged path
    ((320, 200), (320, 200), dict(samples=20, inverse_hue_tolerance=7.0)),        # staged, a whole hue exponent other than the default's 20
    ((320, 200), (200, 130), dict(samples=60, pixel_multiplier=6.0)),             # a spiral of 23 texels: taps from global memory, one conditional wrap
    ((320, 200), (640, 400), dict(samples=7, inverse_hue_tolerance=12.5, distribution_bias=0.9)),   # up-scaling; a non-integer hue exponent
    ((24, 16), (96, 64), dict(samples=40, pixel_multiplier=9.0)),                 # the spiral reaches across the whole image: general wrap
    ((320, 200), (320, 200), dict(samples=300)),                                  # more samples than the per-sample table holds
])
def test_hip_denoise_paths_match_oracle(frame, out_size, kw):
    """The present pass's three ways to a texel (round 4: a box staged in LDS — vrt_denoise_tile_kernel, round 5 — / global with one conditional wrap / global with the
    general wrap) and its per-sample table, each against the oracle within the pass's tolerance."""
    w = W.Workload("t", frame[0], frame[1], 64, 4, 2, 2, False, 0.0)
    grid = W.build_grid(w)
    rt = W.make_renderer(w, grid)
    W.set_view(rt, "V2")
    rt.draw()
    traced = rt.read_rgba8()
    u, f = rt.denoise(out_size[0], out_size[1], want_float=True, **kw)
    rt.deinit()
    fo, uo = O.denoise(traced, out_size[0], out_size[1], **kw)
    nan_o, nan_k = np.isnan(fo[..., :3]).any(axis=-1), np.isnan(f[..., :3]).any(axis=-1)
    assert np.array_equal(nan_o, nan_k)
    ok = ~nan_o
    assert ok.any() and np.abs(f[ok] - fo[ok]).max() <= 1e-4
    assert np.abs(u.astype(int) - uo.astype(int))[ok].max() <= 1


@pytest.mark.gpu
def test_present_pass_overlapping_the_next_frames_trace_reads_the_right_frame():
    """Round 6: with two frames in flight the present pass of frame N overlaps the trace of frame N + 1 (other stream, other
    target); with VRT_TUNE_PRESENT_OWN_STREAM it runs on a stream of its own behind an event of the frame it reads, and the frame after next
    waits for the pass before it overwrites that target.  A moving camera, twelve frames submitted without a wait, the presented image
    read after every frame or only at the end: each must equal the image of the same frame presented one frame at a time."""
    from zig_vulkan_amd import _lib as L
    from zig_vulkan_amd import workloads as W
    w = W.Workload("t", 480, 270, 128, 8, 1, 0, True, 5.0)
    grid = W.build_grid(w)
    views = ["V0", "V1", "V2", "V1x", "VG", "V2", "V0", "V1", "V1", "VG", "V0", "V2"]
    serial = W.make_renderer(w, grid, frames_in_flight=1)
    want = []
    for v in views:
        W.set_view(serial, v)
        serial.draw()
        want.append(serial.denoise(640, 360).copy())
    serial.deinit()
    for reads, flags in (("every", 0), ("last", 0), ("every", L.TUNE_PRESENT_OWN_STREAM), ("last", L.TUNE_PRESENT_OWN_STREAM)):
        rt = W.make_renderer(w, grid, frames_in_flight=2, tuning_flags=flags)
        got = []
        for i, v in enumerate(views):
            W.set_view(rt, v)
            rt.draw()
            if reads == "every":
                got.append(rt.denoise(640, 360).copy())
            else:
                rt.present(640, 360)
        if reads == "last":
            rt.wait()
            last = np.empty((360, 640, 4), dtype=np.uint8)
            L.check(L.lib.vrt_read_denoised_rgba8(rt._h, last.ctypes.data, last.nbytes))
            got = [None] * (len(views) - 1) + [last]
        rt.deinit()
        for i, (g, x) in enumerate(zip(got, want)):
            if g is not None:
                assert np.array_equal(g, x), f"frame {i} ({views[i]}), reads {reads}: the overlapped present pass read another frame"
