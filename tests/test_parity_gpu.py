"""Parity tests proper: the HIP path (through the C ABI) against the oracle on
the same inputs.  Bit-exact on the float target and on RGBA8, counters equal.
Tolerance for the float target per BASELINE.json north_star is 1e-4 per
channel; this suite demands max-abs-err == 0."""
import os
import numpy as np
import pytest

from tests.helpers import O, available_variants, oracle_scene_from_grid, push_for, variant_kwargs
from zig_vulkan_amd import _lib as L
from zig_vulkan_amd import workloads as W

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star tolerance; asserted as an upper bound next to exact equality


def _run_hip(w, grid, view, *, width=None, height=None, variant=0, sun_radius=None, counters=True):
    rt = W.make_renderer(w, grid, width=width or w.width, height=height or w.height, want_float_output=True,
                         enable_counters=counters, **variant_kwargs(variant),
                         **({"sun_radius": sun_radius} if sun_radius is not None else {}))
    W.set_view(rt, view)
    rt.draw()
    f = rt.read_rgba32f()
    u = rt.read_rgba8()
    c = rt.counters() if counters else None
    cam_blob, sun_blob = rt.camera.blob(), rt.sun.blob()
    rt.deinit()
    return f, u, c, O.push_constants(cam_blob, sun_blob)


def _compare(f, u, c, fo, uo, co):
    err = np.abs(f.astype(np.float64) - fo.astype(np.float64))
    assert err.max() <= TOL, f"max abs err {err.max()} at {np.unravel_index(err.argmax(), err.shape)}"
    assert np.array_equal(f.view(np.uint32), fo.view(np.uint32)), f"float target not bit-exact: {np.count_nonzero(f != fo)} values differ"
    assert np.array_equal(u, uo)
    if c is not None:
        assert c == co


# shipped kernels (words / bytes) under every tile order (bits 16-19) and one-wave workgroups (bit 20) ...
SHIPPED = [0, 5, 9, 0x70009, 0x10005, 0x20009, 0x30005, 0x60009, 0x100005]
# ... and the variants that lost their A/B measurement: only in the development build (skipped where it is absent)
DEV_ONLY = [1, 2, 3, 4, 6, 7, 8, 0x10002, 0x20003, 0x30001, 0x60001, 0x140001, 0x409, 0x809]


@pytest.mark.parametrize("view", ["V0", "V1", "V2"])
@pytest.mark.parametrize("variant", SHIPPED + DEV_ONLY)
def test_config0_primary_rays(view, variant):
    w = W.WORKLOADS["cfg0_256x256_64c_b4"]
    grid = W.build_grid(w)
    f, u, c, pc = _run_hip(w, grid, view, variant=variant)
    fo, uo, co = O.render(oracle_scene_from_grid(grid), pc)
    assert co["rays"] == 256 * 256
    _compare(f, u, c, fo, uo, co)


@pytest.mark.parametrize("b", [4, 8])
@pytest.mark.parametrize("view", ["V0", "V1", "V2"])
def test_primary_plus_shadow_deterministic(b, view):
    """configs[2] shape at test size: 128^3 voxels, sun on, radius 0 (RandVec3 collapses to 0)."""
    w = W.Workload("t", 320, 192, 128, b, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    f, u, c, pc = _run_hip(w, grid, view)
    fo, uo, co = O.render(oracle_scene_from_grid(grid), pc)
    assert co["rays"] > 320 * 192  # shadow rays were cast
    _compare(f, u, c, fo, uo, co)


@pytest.mark.parametrize("b", [4, 8])
def test_stochastic_path_bit_exact(b):
    """Reference default shading (spp 2, max_bounce 2, soft sun radius 5): every scatter function and
    the sin-based RNG run; oracle and kernel share the specified sin, so this is still bit-exact."""
    w = W.Workload("t", 192, 128, 128, b, 2, 2, True, 5.0)
    grid = W.build_grid(w)
    for view in ["V0", "V2"]:
        f, u, c, pc = _run_hip(w, grid, view)
        fo, uo, co = O.render(oracle_scene_from_grid(grid), pc)
        _compare(f, u, c, fo, uo, co)


def test_ragged_image_and_counters_off():
    """Image size not a multiple of the 16x16 tile; counters disabled build."""
    w = W.Workload("t", 250, 131, 64, 4, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    f, u, c, pc = _run_hip(w, grid, "V1", counters=False)
    fo, uo, _ = O.render(oracle_scene_from_grid(grid), pc)
    _compare(f, u, None, fo, uo, None)


def test_empty_grid_is_all_background():
    w = W.Workload("t", 64, 64, 64, 4, 1, 0, True, 0.0)
    from zig_vulkan_amd import BrickGrid
    grid = BrickGrid(16, 16, 16, min_point=(-32, -32, -32), scale=4.0, brick_dimension=4)
    f, u, c, pc = _run_hip(w, grid, "V1")
    fo, uo, co = O.render(oracle_scene_from_grid(grid), pc)
    assert co["hits"] == 0 and co["rays"] == 64 * 64
    _compare(f, u, c, fo, uo, co)


def test_sharded_tiles_reassemble_to_full_frame():
    """Image-tile sharding: every rank's packed shard, gathered rank-major and un-swizzled on the
    device, equals the single-context frame."""
    import ctypes as C
    w = W.Workload("t", 200, 120, 64, 4, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    _, u_full, _, _ = _run_hip(w, grid, "V2", counters=False)
    R = 3
    shards = []
    for r in range(R):
        rt = W.make_renderer(w, grid, shard_rank=r, shard_count=R)
        W.set_view(rt, "V2")
        rt.draw()
        shards.append(rt.read_rgba8().reshape(-1))
        info = rt.shard_info()
        assert info.tiles_per_rank * 256 * 4 == shards[-1].size
        rt.deinit()
    gathered = np.concatenate(shards)
    # un-swizzle on the device through the C ABI
    import torch
    g_dev = torch.from_numpy(gathered).cuda()
    out = torch.zeros(w.height * w.width * 4, dtype=torch.uint8, device="cuda")
    rt = W.make_renderer(w, grid, shard_rank=0, shard_count=R)
    rt.assemble_frame(g_dev.data_ptr(), out.data_ptr(), 4)
    rt.wait()
    torch.cuda.synchronize()
    rt.deinit()
    assert np.array_equal(out.cpu().numpy().reshape(w.height, w.width, 4), u_full)


@pytest.mark.parametrize("frames_in_flight", [1, 2])
def test_amortised_tile_schedule_keeps_every_frame_identical(frames_in_flight):
    """kernel_variant 0x..70000: tiles launched in the order of their measured cost, re-sorted every 8 frames here (0x3 << 28;
    32 by default) into the other of two schedule buffers while frames of the second stream may be running; the heaviest
    tiles are split into two half-tile workgroups (on this small frame every re-sort splits as many as it has spare entries
    for).  Order and splitting must never change a pixel: 70 overlapping frames without a read in between, then every frame
    of a second sequence is compared."""
    w = W.Workload("t", 320, 200, 64, 4, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    ref = {}
    for view in ["V0", "V1", "V2"]:
        _, ref[view], _, _ = _run_hip(w, grid, view, counters=False)
    rt = W.make_renderer(w, grid, frames_in_flight=frames_in_flight, kernel_variant=0x30070000)
    for i in range(70):
        W.set_view(rt, ["V0", "V1", "V2"][(i // 5) % 3])
        rt.draw()
    assert np.array_equal(rt.read_rgba8(), ref[["V0", "V1", "V2"][(69 // 5) % 3]])
    for i in range(40):
        view = ["V2", "V0", "V1"][i % 3]
        W.set_view(rt, view)
        rt.draw()
        assert np.array_equal(rt.read_rgba8(), ref[view]), (i, view)
    W.set_view(rt, "V1")
    rt.draw(frames=40)  # back-to-back launches of one call re-sort inside the loop
    assert np.array_equal(rt.read_rgba8(), ref["V1"])
    rt.deinit()


@pytest.mark.parametrize("width, height, variant, frames_in_flight, warm", [(400, 300, 0x30070000, 1, 20), (400, 300, 0x30070000, 2, 20), (1024, 576, 0, 2, 70)])
def test_tile_schedule_of_two_sample_frames_splits_freely_and_follows_the_sample_count(width, height, variant, frames_in_flight, warm):
    """Round 4: under the cost schedule, frames of two samples per pixel may split EVERY tile whose slowest wave is above a lower bar
    (the halves trace the second sample on their idle lanes), frames of any other sample count keep round 3's rule (an eighth of the
    tiles, a higher bar).  The rule follows the camera of each dispatch: a change re-sorts at once into a list laid out for either.
    One context, the sample count going 2 -> 1 -> 3 -> 2 with re-sorts every 8 frames in between: every frame read equals the frame of
    a context that launches its tiles in plain reverse raster (no schedule, no splits).  Also with two frames in flight, and with the
    library's own choice of order at the app's frame size (1024 x 576: the cost schedule; with two frames in flight reverse raster —
    except for frames of two samples, which keep the schedule on both streams)."""
    w = W.Workload("t", width, height, 128, 4, 2, 2, True, 5.0)
    grid = W.build_grid(w)
    ref = {}
    for spp in (1, 2, 3):
        rt = W.make_renderer(w, grid, kernel_variant=0x30000)
        rt.camera.d_camera.samples_per_pixel = spp
        for v in ("V0", "V2"):
            W.set_view(rt, v)
            rt.draw()
            ref[(spp, v)] = rt.read_rgba8().copy()
        rt.deinit()
    assert not np.array_equal(ref[(1, "V2")], ref[(2, "V2")]) and not np.array_equal(ref[(3, "V2")], ref[(2, "V2")])
    rt = W.make_renderer(w, grid, kernel_variant=variant, frames_in_flight=frames_in_flight)
    for spp in (2, 1, 3, 2, 2):
        rt.camera.d_camera.samples_per_pixel = spp
        for i in range(warm):   # two re-sorts on measured costs without a read in between
            W.set_view(rt, ("V0", "V2")[(i // 3) % 2])
            rt.draw()
        for v in ("V2", "V0", "V2"):
            W.set_view(rt, v)
            rt.draw()
            assert np.array_equal(rt.read_rgba8(), ref[(spp, v)]), (spp, v)
    # ... and the first frame after a change of the sample count (sorted on the costs of the other rule's frames)
    for spp, v in ((1, "V0"), (2, "V2"), (3, "V0"), (2, "V0"), (1, "V2")):
        rt.camera.d_camera.samples_per_pixel = spp
        W.set_view(rt, v)
        rt.draw()
        assert np.array_equal(rt.read_rgba8(), ref[(spp, v)]), (spp, v)
    rt.deinit()


@pytest.mark.parametrize("variant", [0, 0x50000, 0x70000, 0x100000])
def test_two_frames_in_flight_give_the_same_frames_and_respect_uploads(variant):
    """frames_in_flight = 2: frames alternate between two streams/targets; every frame still equals
    the single-stream result, and a grid edit between frames is seen by the next frame on either stream.
    (0x50000: the cost-feedback tile schedule, which is re-sorted before every frame; 0x100000: one wave per workgroup.)"""
    w = W.Workload("t", 320, 200, 64, 4, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    ref = {}
    for view in ["V0", "V1", "V2"]:
        _, ref[view], _, _ = _run_hip(w, grid, view, counters=False)
    rt = W.make_renderer(w, grid, frames_in_flight=2, kernel_variant=variant)
    seq = ["V0", "V1", "V2", "V1", "V0", "V2", "V2"]
    for view in seq:  # queue without reading: frames overlap
        W.set_view(rt, view)
        rt.draw()
    assert np.array_equal(rt.read_rgba8(), ref[seq[-1]])
    for view in seq:  # read every frame
        W.set_view(rt, view)
        rt.draw()
        assert np.array_equal(rt.read_rgba8(), ref[view])
    # edit the grid between frames: both slots must see it
    for y in range(20, 60):
        grid.insert(31, y, 31, 7)
        grid.insert(32, y, 31, 7)
    rt.update_grid_delta()
    W.set_view(rt, "V1")
    rt.draw()
    a = rt.read_rgba8().copy()
    rt.draw()
    b = rt.read_rgba8().copy()
    rt.deinit()
    _, want, _, _ = _run_hip(w, grid, "V1", counters=False)
    assert np.array_equal(a, want) and np.array_equal(b, want)
    assert not np.array_equal(want, ref["V1"])


def test_vox_model_scene_like_main_zig():
    """The reference app's scene build (src/main.zig:84-138): a .vox model inserted into the grid at an
    offset, its palette appended after the 8 terrain materials, then terrain — rendered with the
    reference defaults (spp 2, max_bounce 2, sun) and compared with the oracle."""
    import struct
    from tests.test_vox import make_vox
    from zig_vulkan_amd import BrickGrid, default_materials, vox
    voxels = [(x, y, z, 1 + ((x + 2 * y + 3 * z) % 5)) for x in range(12) for y in range(12) for z in range(20)
              if (x - 6) ** 2 + (y - 6) ** 2 + ((z - 10) * 0.6) ** 2 < 30]
    rgba = bytearray(1024)
    for i, (r, g, b, a) in enumerate([(200, 30, 30, 255), (30, 200, 30, 255), (40, 40, 220, 120), (250, 250, 60, 255), (90, 90, 90, 255)]):
        rgba[4 * i:4 * i + 4] = bytes([r, g, b, a])  # file colour i -> palette entry i+1
    model = vox.parse_buffer(make_vox([((12, 12, 20), voxels)], rgba=bytes(rgba)))
    grid = BrickGrid(16, 16, 16, min_point=(-32, -32, -32), scale=4.0, brick_dimension=4)
    model.insert_into(grid, 0, offset=(26, 34, 26), material_offset=8)
    grid.synth_terrain(420)
    materials = default_materials(256)
    pal = model.materials(248)
    materials[8:] = pal[:248]
    w = W.Workload("t", 256, 160, 64, 4, 2, 2, True, 5.0)
    rt = W.make_renderer(w, grid, want_float_output=True, enable_counters=True)
    rt.push_materials(materials)
    W.set_view(rt, "V2")
    rt.draw()
    f, u, c = rt.read_rgba32f(), rt.read_rgba8(), rt.counters()
    pc = O.push_constants(rt.camera.blob(), rt.sun.blob())
    rt.deinit()
    fo, uo, co = O.render(oracle_scene_from_grid(grid, materials), pc)
    _compare(f, u, c, fo, uo, co)
    # palette entry 0 is (0,0,0,1) (alpha 1/255: glass, loader.zig:169-174); entry 3 has alpha 120: glass
    assert (materials["type"][8:14] == [2, 0, 0, 2, 0, 0]).all()


def test_reference_app_default_scene_shape_non_cubic_grid():
    """The reference app's own defaults (src/main.zig:77-81,122-135): 128x64x128 bricks of 4^3 at
    min (-32,-16,-32), scale 0.5, 1024x576, spp 2, max_bounce 2, sun on — a non-cubic grid whose status
    bitmap is 128 KiB (test_dev_lds_variant_falls_back_to_global_memory asks the development build to stage it in LDS)."""
    from zig_vulkan_amd import BrickGrid, Config, CameraConfig, SunConfig, VoxelRT, default_materials
    grid = BrickGrid(128, 64, 128, min_point=(-32.0, -16.0, -32.0), scale=0.5, brick_dimension=4)
    grid.synth_terrain(420)
    rt = VoxelRT(grid, Config(internal_resolution_width=1024, internal_resolution_height=576, camera=CameraConfig(samples_per_pixel=2, max_bounce=2),
                              sun=SunConfig(enabled=True), want_float_output=True, enable_counters=True))
    rt.push_materials(default_materials(256))
    rt.camera.set_origin((0.0, 0.0, 0.0))  # Camera.Config default origin, looking -Z
    rt.draw()
    assert rt.kernel_name() == "vrt_trace_kernel<4, false, 4, 5, 0, 256>"  # what ran: the lockstep bounce kernel on the shader's words
    f, u, c = rt.read_rgba32f(), rt.read_rgba8(), rt.counters()
    pc = O.push_constants(rt.camera.blob(), rt.sun.blob())
    rt.deinit()
    rng = np.random.default_rng(5)
    xy = np.stack([rng.integers(0, 1024, 20000), rng.integers(0, 576, 20000)], axis=-1).astype(np.int32)
    fo, uo, _ = O.render_pixels(oracle_scene_from_grid(grid), pc, xy)
    assert np.array_equal(f[xy[:, 1], xy[:, 0]].view(np.uint32), fo.view(np.uint32))
    assert np.array_equal(u[xy[:, 1], xy[:, 0]], uo)
    assert c["hits"] > 0


@pytest.mark.parametrize("b", [4, 8])
@pytest.mark.parametrize("scale", [0.3, 3.0, 0.5])
def test_grid_scale_not_a_power_of_two(b, scale):
    """The kernel replaces `x / scale` by `x * (1 / scale)` when the grid and voxel scales are powers of two
    (exact); any other scale must take the IEEE divisions of the shader (comp:288,392)."""
    from zig_vulkan_amd import BrickGrid
    w = W.Workload("t", 200, 120, 16 * b, b, 1, 0, True, 2.0)
    n = 16
    grid = BrickGrid(n, n, n, min_point=(-0.5 * n * scale, -0.5 * n * scale, -0.5 * n * scale), scale=scale, brick_dimension=b)
    grid.synth_terrain(420)
    for variant in available_variants((0, 4)):
        rt = W.make_renderer(w, grid, want_float_output=True, enable_counters=True, **variant_kwargs(variant))
        rt.camera.look_at((0.9 * n * scale, -0.8 * n * scale, 1.1 * n * scale), (0.0, 0.1 * n * scale, 0.0))
        rt.draw()
        f, u, c = rt.read_rgba32f(), rt.read_rgba8(), rt.counters()
        pc = O.push_constants(rt.camera.blob(), rt.sun.blob())
        rt.deinit()
        fo, uo, co = O.render(oracle_scene_from_grid(grid), pc)
        _compare(f, u, c, fo, uo, co)
    assert co["hits"] > 0 and co["rays"] > 200 * 120


@pytest.mark.parametrize("dims", [(5, 3, 7), (1, 1, 1), (2, 9, 2)])
def test_odd_grid_dimensions(dims):
    """Grids whose dimensions are not multiples of the 4x4x4 status blocks (and a single-cell grid)."""
    from zig_vulkan_amd import BrickGrid
    w = W.Workload("t", 160, 100, 64, 4, 1, 0, True, 0.0)
    grid = BrickGrid(*dims, min_point=(-10.0, -6.0, -14.0), scale=4.0, brick_dimension=4)
    rng = np.random.default_rng(sum(dims))
    n = 40 * dims[0] * dims[1] * dims[2]
    xyz = np.stack([rng.integers(0, 4 * d, n) for d in dims], axis=-1)
    grid.insert_many(xyz, rng.integers(0, 8, n))
    for variant in available_variants((0, 1, 2, 3, 5, 6, 7, 8, 9)):
        rt = W.make_renderer(w, grid, want_float_output=True, enable_counters=True, **variant_kwargs(variant))
        rt.camera.look_at((30.0, -25.0, 40.0), (0.0, 5.0, 0.0))
        rt.draw()
        f, u, c = rt.read_rgba32f(), rt.read_rgba8(), rt.counters()
        pc = O.push_constants(rt.camera.blob(), rt.sun.blob())
        rt.deinit()
        fo, uo, co = O.render(oracle_scene_from_grid(grid), pc)
        _compare(f, u, c, fo, uo, co)
    assert co["hits"] > 0


def test_native_rccl_pipeline_single_rank():
    """vrt_dist_* on one GPU (world = 1): RCCL bound through dlopen of PyTorch's librccl, communicator
    created, a shard sent to and received from this rank itself, then the pipelined frame loop (4 frames
    in flight, packed tile-major targets, un-swizzle) must reproduce the plain frames — also across a
    grid edit.  Peers other than self cannot be exercised on a one-GPU box."""
    from zig_vulkan_amd import VoxelRT
    w = W.Workload("t", 330, 210, 64, 4, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    ref = {}
    for view in ["V0", "V1", "V2"]:
        _, ref[view], _, _ = _run_hip(w, grid, view, counters=False)
    rt = W.make_renderer(w, grid, shard_rank=0, shard_count=1)
    rt.dist_init(VoxelRT.dist_unique_id(), 0, 1, frames_in_flight=4)
    rt.dist_selftest()
    seq = ["V0", "V1", "V2", "V2", "V1", "V0", "V1"]
    for view in seq:
        W.set_view(rt, view)
        rt.dist_frame()
    assert np.array_equal(rt.dist_read_frame(), ref[seq[-1]])
    for view in seq:
        W.set_view(rt, view)
        rt.dist_frame()
        assert np.array_equal(rt.dist_read_frame(), ref[view])
    for y in range(20, 60):
        grid.insert(31, y, 31, 7)
    # (the replica update of a one-process-edits host: upload on the root + ncclBroadcast of each dirty range — RCCL's own,
    # here with a single rank)
    ranges = rt.grid_delta_ranges()
    assert ranges
    rt.dist_broadcast_grid_delta(root=0, ranges=ranges)
    W.set_view(rt, "V1")
    for _ in range(5):
        rt.dist_frame()
    got = rt.dist_read_frame()
    rt.dist_wait()
    rt.deinit()
    _, want, _, _ = _run_hip(w, grid, "V1", counters=False)
    assert np.array_equal(got, want) and not np.array_equal(want, ref["V1"])


def _parity_for(grid, w, view_setup, materials=None, counters=True):
    rt = W.make_renderer(w, grid, want_float_output=True, enable_counters=counters)
    if materials is not None:
        rt.push_materials(materials)
    view_setup(rt)
    rt.draw()
    f, u = rt.read_rgba32f(), rt.read_rgba8()
    c = rt.counters() if counters else None
    pc = O.push_constants(rt.camera.blob(), rt.sun.blob())
    rt.deinit()
    fo, uo, co = O.render(oracle_scene_from_grid(grid, materials), pc)
    return f, u, c, fo, uo, co


def test_quirk_unknown_material_type_and_glass_and_metal():
    """SURVEY.md §8 quirks 7-8 and the scatter paths: a scene of glass, metal and an out-of-enum material
    type (loop_count -= 1, comp:235-238), traced with bounces."""
    from zig_vulkan_amd import BrickGrid, default_materials
    mats = default_materials(256)
    mats[9] = (7, 0.9, 0.2, 0.9, 1.0)      # unknown type 7
    mats[10] = (3, 0.3, 0.9, 0.3, 1.0)     # type == MAT_NONE with type_data == 1: the ignore test of comp:427 matches it
    mats[11] = (2, 0.9, 0.9, 1.0, 1.52)    # glass
    mats[12] = (1, 0.8, 0.8, 0.8, 0.05)    # polished metal
    grid = BrickGrid(16, 16, 16, min_point=(-32, -32, -32), scale=4.0, brick_dimension=4)
    rng = np.random.default_rng(11)
    for cx, cz, m in [(12, 12, 9), (30, 14, 10), (48, 12, 11), (14, 40, 12), (34, 40, 0), (50, 44, 7)]:
        for x in range(cx - 5, cx + 6):
            for z in range(cz - 5, cz + 6):
                for y in range(24, 40):
                    if (x - cx) ** 2 + (z - cz) ** 2 <= 30 or y < 27:
                        grid.insert(x, y, z, m)
    for x in range(64):
        for z in range(64):
            grid.insert(x, 20, z, 1 + int(rng.integers(0, 6)))
    w = W.Workload("t", 240, 160, 64, 4, 2, 3, True, 5.0)
    f, u, c, fo, uo, co = _parity_for(grid, w, lambda rt: rt.camera.look_at((10.0, -14.0, 36.0), (0.0, 4.0, 0.0)), mats)
    _compare(f, u, c, fo, uo, co)
    assert co["hits"] > 0 and co["rays"] > 240 * 160 * 2


def test_camera_inside_solid_and_axis_aligned_rays():
    """Rays that start inside an occupied brick / voxel, and a camera whose centre column and row hold
    rays with exactly zero direction components (safeInverse(0) = 1e12, sign(0) = 0)."""
    from zig_vulkan_amd import BrickGrid
    grid = BrickGrid(8, 8, 8, min_point=(-16, -16, -16), scale=4.0, brick_dimension=8)
    for x in range(20, 44):
        for y in range(20, 44):
            for z in range(20, 44):
                if (x + y + z) % 3:
                    grid.insert(x, y, z, 1 + (x % 6))
    w = W.Workload("t", 129, 97, 64, 8, 1, 0, True, 0.0)  # odd sizes: pixel (64,48) is the exact image centre
    f, u, c, fo, uo, co = _parity_for(grid, w, lambda rt: rt.camera.set_origin((0.3, 0.2, 0.1)))  # inside the block, looking -Z
    _compare(f, u, c, fo, uo, co)
    f, u, c, fo, uo, co = _parity_for(grid, w, lambda rt: rt.camera.set_origin((0.0, 0.0, 40.0)))  # outside, centre ray = (0,0,-1)
    _compare(f, u, c, fo, uo, co)
    assert co["hits"] > 0


def test_degenerate_one_pixel_image_nan_rays():
    """image_width - 1 == 0 makes u = 0/0 (comp:168): every ray direction is NaN.  The shader then misses
    the slab test and the colour is NaN, stored as 0; kernel and oracle must agree on that too."""
    from zig_vulkan_amd import BrickGrid
    grid = BrickGrid(4, 4, 4, min_point=(-8, -8, -8), scale=4.0)
    grid.insert(8, 8, 8, 1)
    w = W.Workload("t", 1, 1, 16, 4, 1, 0, True, 0.0)
    f, u, c, fo, uo, co = _parity_for(grid, w, lambda rt: rt.camera.set_origin((0.0, 0.0, 20.0)))
    assert np.array_equal(np.isnan(f), np.isnan(fo)) and np.isnan(fo[0, 0, :3]).all()
    assert np.array_equal(u, uo) and u[0, 0].tolist() == [0, 0, 0, 255]
    assert c == co and co["rays"] == 1


def test_sparse_allocation_fewer_slots_than_cells():
    """brick_alloc below the cell count (the 2048^3 configuration's case): slots are handed out in
    insertion order, unset start indices stay 0xFFFFFFFF for unused slots."""
    from zig_vulkan_amd import BrickGrid
    grid = BrickGrid(32, 32, 32, min_point=(-32, -32, -32), scale=2.0, brick_dimension=8, brick_alloc=6000)
    grid.synth_sparse(420, 0.08)
    assert 0 < grid.active_bricks <= 6000 < 32 ** 3
    w = W.Workload("t", 200, 150, 256, 8, 1, 1, True, 5.0, "sparse", 0.08, 6000)
    f, u, c, fo, uo, co = _parity_for(grid, w, lambda rt: W.set_view(rt, "V2"))
    _compare(f, u, c, fo, uo, co)
    assert co["hits"] > 0


def test_random_scenes_fuzz_slice():
    """A fixed slice of tools/fuzz_parity.py (60 random scenes / cameras / ray budgets, seed 5): whole frames of the product
    kernel against the oracle, bit for bit."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.fuzz(60, 5, verbose=False) == 0
    # power-of-two grids, frames with bounces: the path kernel's walk loop on half-block words
    assert mod.fuzz(24, 11, big=True, verbose=False, pow2=True) == 0
    from tests.helpers import dev_library_or_none
    if dev_library_or_none():   # development build: the block-skipping walk (skip_empty_block) and the other variants join the draw
        assert mod.fuzz(24, 11, big=True, verbose=False, pow2=True, library=dev_library_or_none()) == 0
        assert mod.fuzz(30, 5, big=True, verbose=False, library=dev_library_or_none()) == 0


def test_walk_ends_at_the_bounding_box_of_the_occupied_cells():
    """The product kernel ends a brick-level walk where the ray leaves the bounding box of the occupied cells (derived from
    binding 3 on upload) instead of at the grid's face.  Pixels must not change: an island of bricks in a larger empty grid seen
    from inside the box, from outside it on every side, looking away from it and past it (odd frame size: the centre ray is
    (nearly) axis-aligned), with shadow rays; an empty grid; and an edit that grows the box, seen by the next frame."""
    from zig_vulkan_amd import BrickGrid, VoxelRT, Config, CameraConfig, SunConfig, default_materials
    mats = default_materials(256)

    def island():
        g = BrickGrid(12, 10, 14, min_point=(-24.0, -20.0, -28.0), scale=4.0, brick_dimension=4)
        for x in range(20, 29):
            for y in range(18, 25):
                for z in range(30, 41):
                    if (x + y + z) % 3:
                        g.insert(x, y, z, 1 + (x % 5))
        return g

    def check(rt, grid, origin, target, what):
        rt.camera.look_at(list(origin), list(target))
        rt.draw()
        f, u = rt.read_rgba32f(), rt.read_rgba8()
        fo, uo, _ = O.render(oracle_scene_from_grid(grid, mats), O.push_constants(rt.camera.blob(), rt.sun.blob()))
        assert np.array_equal(f.view(np.uint32), fo.view(np.uint32)) and np.array_equal(u, uo), what

    cfg = Config(internal_resolution_width=161, internal_resolution_height=99, camera=CameraConfig(samples_per_pixel=1, max_bounce=0),
                 sun=SunConfig(enabled=True, radius=0.0), want_float_output=True)
    grid = island()
    rt = VoxelRT(grid, cfg)
    rt.push_materials(mats)
    centre = (0.2, -3.0, 3.5)  # inside the island (cells 5..7, 4..6, 7..10 of the grid)
    check(rt, grid, centre, (5.0, -2.0, 9.0), "from inside the box")
    for k, off in enumerate([(-20, 0, 0), (22, 1, 0), (0, -15, 0.5), (0.5, 16, 0), (0, 0.5, -24), (1, 0, 26), (-18, -14, -20)]):
        o = tuple(c + d for c, d in zip(centre, off))
        check(rt, grid, o, centre, f"towards the box from side {k}")
        check(rt, grid, o, tuple(2 * a - b for a, b in zip(o, centre)), f"away from the box from side {k}")
        check(rt, grid, o, (o[0], o[1], o[2] - 10.0), f"along -z past the box from side {k}")
        check(rt, grid, o, (o[0] + 10.0, o[1], o[2]), f"along +x past the box from side {k}")
    # an edit that grows the box: one voxel in a far corner of the grid, then one right above the camera's usual spot
    grid.insert(1, 38, 2, 4)
    rt.update_grid_delta()
    check(rt, grid, (-20.0, 14.0, -25.0), (-23.0, 17.5, -27.0), "new corner voxel, seen from nearby")
    check(rt, grid, centre, (5.0, -2.0, 9.0), "after the edit, from inside")
    grid.insert(46, 1, 54, 2)
    rt.update_grid_delta()
    check(rt, grid, (20.0, -14.0, 25.0), (22.5, -17.5, 27.0), "second corner voxel")
    check(rt, grid, (-22.0, 18.0, -26.0), (22.5, -17.5, 27.0), "across the whole grid")
    rt.deinit()
    # no occupied cell at all
    empty = BrickGrid(5, 4, 6, min_point=(-10.0, -8.0, -12.0), scale=4.0, brick_dimension=8)
    rt = VoxelRT(empty, cfg)
    rt.push_materials(mats)
    check(rt, empty, (0.0, 0.0, 0.0), (1.0, 0.5, 3.0), "empty grid, camera inside")
    check(rt, empty, (0.0, -30.0, 0.0), (0.0, 0.0, 0.0), "empty grid, camera outside")
    rt.deinit()


# ---- vrt_path_kernel (frames with bounces, persistent lanes): chosen by the library for scenes larger than the caches, forced
# ---- here (kernel_variant bit 23) on small scenes so that whole frames can be compared with the oracle
PATH = 1 << 23
PATH_FILTER = PATH | (1 << 22)      # ... with the block-skipping walk: lanes in empty 4x4x4 blocks of cells jump to the block's exit face (x, z dimensions powers of two)
PATH_5WAVES = PATH | (5 << 8)


@pytest.mark.parametrize("variant", [PATH, PATH_FILTER, PATH_5WAVES, PATH | (2 << 24), PATH_FILTER | (12 << 24)])
@pytest.mark.parametrize("b", [4, 8])
def test_path_kernel_stochastic_path_bit_exact(b, variant):
    """The reference's default shading (2 samples, 2 bounces, soft sun) through the persistent-lane kernel, at several
    batch thresholds; 250 x 131 is not a multiple of the tile size, so some fetched pixels lie outside the image."""
    w = W.Workload("t", 250, 131, 128, b, 2, 2, True, 5.0)
    grid = W.build_grid(w)
    for view in ["V0", "V2"]:
        f, u, c, pc = _run_hip(w, grid, view, variant=variant)
        fo, uo, co = O.render(oracle_scene_from_grid(grid), pc)
        _compare(f, u, c, fo, uo, co)


@pytest.mark.parametrize("variant", [PATH, PATH_FILTER])
def test_path_kernel_all_material_types_many_samples(variant):
    """Glass, metal, lambertian and an unknown material type, 5 samples, 3 bounces, sparse allocation."""
    from zig_vulkan_amd import BrickGrid, default_materials
    mats = default_materials(256)
    mats[1] = (7, 0.9, 0.2, 0.9, 1.0)
    mats[2] = (2, 0.9, 0.9, 1.0, 1.52)
    mats[3] = (1, 0.8, 0.8, 0.8, 0.05)
    grid = BrickGrid(32, 32, 32, min_point=(-32, -32, -32), scale=2.0, brick_dimension=8, brick_alloc=9000)
    grid.synth_sparse(7, 0.1)
    w = W.Workload("t", 200, 120, 256, 8, 5, 3, True, 5.0, "sparse", 0.1, 9000)
    rt = W.make_renderer(w, grid, want_float_output=True, enable_counters=True, **variant_kwargs(variant))
    rt.push_materials(mats)
    for view in ["V1", "V1x"]:
        W.set_view(rt, view)
        rt.draw()
        f, u, c = rt.read_rgba32f(), rt.read_rgba8(), rt.counters()
        fo, uo, co = O.render(oracle_scene_from_grid(grid, mats), O.push_constants(rt.camera.blob(), rt.sun.blob()))
        _compare(f, u, c, fo, uo, co)
        assert co["hits"] > 0
    rt.deinit()


def test_path_kernel_sharded_context_and_two_frames_in_flight():
    """A shard (rank 1 of 3) rendered by the path kernel holds the same packed tiles as the lockstep kernel's shard; and with
    two frames in flight (two streams, one pixel counter each) every frame equals the single-stream frame."""
    w = W.Workload("t", 208, 112, 128, 8, 2, 2, True, 5.0)
    grid = W.build_grid(w)
    shards = {}
    for variant in (PATH, 1 << 21):
        rt = W.make_renderer(w, grid, shard_rank=1, shard_count=3, kernel_variant=variant)
        W.set_view(rt, "V2")
        rt.draw()
        shards[variant] = rt.read_rgba8().copy()
        rt.deinit()
    assert np.array_equal(shards[PATH], shards[1 << 21]) and shards[PATH].any()
    one = W.make_renderer(w, grid, kernel_variant=PATH)
    two = W.make_renderer(w, grid, kernel_variant=PATH, frames_in_flight=2)
    for view in ["V0", "V2", "V1", "V2", "V0"]:
        for rt in (one, two):
            W.set_view(rt, view)
            rt.draw()
        assert np.array_equal(one.read_rgba8(), two.read_rgba8()), view
    one.deinit()
    two.deinit()


def test_large_scene_takes_the_path_kernel_small_scene_the_lockstep_kernel():
    """The library's choice (vrt_create): bindings 3-5 above 192 MiB -> persistent lanes.  Checked through the frame only: both
    kernels must give the oracle's pixels, whichever runs (kernel names are not part of the ABI)."""
    w = W.WORKLOADS["refapp_1024x576_512c_b4"]
    grid = W.build_grid(w)
    rt = W.make_renderer(w, grid, want_float_output=True)
    W.set_view(rt, "V2")
    rt.draw()
    f, u = rt.read_rgba32f(), rt.read_rgba8()
    pc = O.push_constants(rt.camera.blob(), rt.sun.blob())
    rt.deinit()
    rng = np.random.default_rng(3)
    xy = np.stack([rng.integers(0, w.width, 6000), rng.integers(0, w.height, 6000)], axis=-1).astype(np.int32)
    fo, uo, _ = O.render_pixels(oracle_scene_from_grid(grid), pc, xy)
    assert np.array_equal(f[xy[:, 1], xy[:, 0]].view(np.uint32), fo.view(np.uint32)) and np.array_equal(u[xy[:, 1], xy[:, 0]], uo)


@pytest.mark.parametrize("variant,bounces", [(0, 0), (0, 2), (PATH, 2), (PATH_FILTER, 2)])
def test_axis_aligned_rays_jump_to_the_box(variant, bounces):
    """Rays with one or two direction components exactly 0 (odd image size, camera on a grid axis looking along it: the centre
    row, the centre column and the centre pixel), started in front of an island of bricks in a larger empty grid: the jump to
    the occupied-cell box has to leave the axes the ray does not move along alone (they carry the hang-guard budget, not a
    distance), from every side of the box."""
    from zig_vulkan_amd import BrickGrid, default_materials
    b = 8
    grid = BrickGrid(16, 16, 16, min_point=(-8.0, -8.0, -8.0), scale=1.0, brick_dimension=b, brick_alloc=600)
    rng = np.random.default_rng(3)
    grid.insert_many(np.stack([rng.integers(40, 88, 6000) for _ in range(3)], axis=-1), rng.integers(0, 6, 6000))  # cells 5..10 of 16
    w = W.Workload("axis", 65, 33, 128, b, 2 if bounces else 1, bounces, True, 0.0)
    rt = W.make_renderer(w, grid, want_float_output=True, enable_counters=True, **variant_kwargs(variant))
    rt.push_materials(default_materials(256))
    scene = oracle_scene_from_grid(grid)
    hits = 0
    for origin, target in [((0.0, 0.0, -7.5), (0.0, 0.0, 0.0)), ((0.0, 0.0, 7.5), (0.0, 0.0, 0.0)), ((-7.5, 0.0, 0.0), (0.0, 0.0, 0.0)),
                           ((7.5, 0.0, 0.0), (0.0, 0.0, 0.0)), ((0.0, -7.5, 0.0), (0.0, 0.0, 0.001)), ((0.5, 7.5, 0.5), (0.5, 0.0, 0.501)),
                           ((-7.5, -7.5, 0.0), (0.0, 0.0, 0.0)), ((-12.0, 0.0, 0.0), (0.0, 0.0, 0.0))]:
        rt.camera.look_at(origin, target)
        rt.draw()
        f, u, c = rt.read_rgba32f(), rt.read_rgba8(), rt.counters()
        fo, uo, co = O.render(scene, O.push_constants(rt.camera.blob(), rt.sun.blob()))
        _compare(f, u, c, fo, uo, co)
        hits += co["hits"]
    rt.deinit()
    assert hits > 0


def test_dev_lds_variant_falls_back_to_global_memory():
    """Development build: the LDS-staged status bitmap (variant 6) of the reference app's 128x64x128 grid is 128 KiB, above the
    64 KiB budget — vrt_create selects the global-memory kernel and says so."""
    from zig_vulkan_amd import BrickGrid, Config, SunConfig, VoxelRT
    kw = variant_kwargs(6)
    grid = BrickGrid(128, 64, 128, min_point=(-32.0, -16.0, -32.0), scale=0.5, brick_dimension=4)
    rt = VoxelRT(grid, Config(internal_resolution_width=64, internal_resolution_height=64, sun=SunConfig(enabled=True), **kw))
    assert "global-memory variant" in rt.kernel_name()
    rt.deinit()


def test_product_build_refuses_development_variants():
    """The product library holds only the kernels it chooses itself: a development variant is an argument error, not a silent
    substitution."""
    w = W.Workload("t", 64, 64, 64, 4, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    for variant in (1, 2, 3, 4, 6, 7, 8, 0x805, (1 << 23) | (1 << 22)):
        with pytest.raises(L.VrtError) as e:
            W.make_renderer(w, grid, kernel_variant=variant)
        assert e.value.code == L.VRT_E_INVALID_ARG and "development" in str(e.value)
    assert L.lib.vrt_compiled_kernel_count() <= 40


@pytest.mark.parametrize("variant,bounces", [(0, 0), (0, 2), (PATH, 2)])
def test_start_index_shortcut_only_when_the_pattern_holds(variant, bounces):
    """comp:422 reads a brick's start index from binding 6.  The reference's allocator hands out B^3 material entries per brick in
    slot order (MaterialAllocator.zig:39), so the entry is slot * B^3 — the library checks that on every upload of binding 6 and
    then multiplies instead of loading (TraceParams::start_is_slot).  A host is free to upload anything else: here the material
    blocks of random pairs of bricks are swapped (start indices and material bytes; the scene means the same), some entries carry
    the ignored top bit (quirk 7) — the check must fail and the look-up must be taken: frames equal the oracle's on the uploaded
    buffers, and equal the unswapped scene's frames."""
    b = 8
    w = W.Workload("t", 200, 120, 128, b, 2 if bounces else 1, bounces, True, 0.0)
    grid = W.build_grid(w)
    start = grid.array(L.BUF_BRICK_START_INDEX).copy()
    mat = grid.array(L.BUF_MATERIAL_INDEX).copy()
    used = np.flatnonzero(start != 0xFFFFFFFF)
    assert used.size > 100 and np.array_equal(start[used] & 0x7FFFFFFF, used.astype(np.uint32) * b ** 3)   # the canonical pattern
    rng = np.random.default_rng(9)
    perm = rng.permutation(used)[:200].reshape(-1, 2)
    for i, j in perm:
        si, sj = int(start[i] & 0x7FFFFFFF), int(start[j] & 0x7FFFFFFF)
        bi, bj = mat[si:si + b ** 3].copy(), mat[sj:sj + b ** 3].copy()
        mat[si:si + b ** 3], mat[sj:sj + b ** 3] = bj, bi
        start[i], start[j] = sj | 0x80000000, si           # (one of the two with the LOD bit the shader masks off)
    frames = {}
    for name in ("canonical", "swapped"):
        rt = W.make_renderer(w, grid, want_float_output=True, **variant_kwargs(variant))
        if name == "swapped":
            rt.upload(L.BUF_BRICK_START_INDEX, 0, start)
            rt.upload(L.BUF_MATERIAL_INDEX, 0, mat)
        W.set_view(rt, "V2")
        rt.draw()
        frames[name] = (rt.read_rgba32f().copy(), rt.read_rgba8().copy())
        pc = O.push_constants(rt.camera.blob(), rt.sun.blob())
        rt.deinit()
    scene = oracle_scene_from_grid(grid)
    swapped = O.OracleScene(bytes(grid.device_state), scene.materials, scene.brick_status, scene.brick_index, scene.brick_occupancy, start, mat, b)
    fo, uo, co = O.render(swapped, pc)
    assert co["hits"] > 1000
    for name in frames:
        assert np.array_equal(frames[name][0].view(np.uint32), fo.view(np.uint32)) and np.array_equal(frames[name][1], uo), name


@pytest.mark.parametrize("slots,communicators", [(8, 1), (8, 0), (16, 0), (16, 16)])
def test_every_launch_slot_at_once_on_real_rccl(slots, communicators):
    """VERDICT r05 #2a, on the real library (world 1, PyTorch's librccl): every launch slot issues a kernel and its gather — a grouped
    self send + recv of one shard — on its own stream, 20 rounds, with the slots on ONE communicator (round 5's pipeline) and with a
    communicator per slot (ncclCommSplit duplicates; the ranks' agreement on their number by ncclAllReduce is exercised too).  Must
    complete (a deadlock is the test's time limit), deliver the bytes, and report the communicators it used.  The overlap it measures is
    printed for profiles/r06_rccl_slots_world1.txt (tools/experiments/rccl_slots_probe.py runs the same call at more settings)."""
    from zig_vulkan_amd import VoxelRT
    w = W.Workload("t", 1920, 1080, 64, 4, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    rt = W.make_renderer(w, grid, shard_rank=0, shard_count=1)
    rt.dist_init(VoxelRT.dist_unique_id(), 0, 1, frames_in_flight=slots, communicators=communicators)
    info = rt.dist_comm_info()
    want = communicators if communicators else min(slots, 8)
    assert info["library_splits"], "PyTorch's RCCL (2.27) has ncclCommSplit"
    assert info["communicators"] == want and info["made_by_this_init"] == want, info
    assert info["agreed_by_all_reduce"] == (want > 1)
    rt.dist_selftest()
    st = rt.dist_selftest_slots(busy_us=50, rounds=20)
    assert st["launches"] == 20 * slots and st["communicators"] == want and st["wall_ms"] > 0
    print(f"real RCCL, world 1, {slots} slots on {want} communicator(s): {st['us_per_launch']:.1f} us per launch (50 us kernel + self send/recv of "
          f"{rt.dist_stats()['shard_bytes_per_frame']} B), last round {1e3 * st['last_round_launch_ms']:.1f} us from kernel start to gather end")
    # ... and the frame loop still reproduces the plain frame on the slots' own communicators (world 1: no peers, but the streams are these)
    W.set_view(rt, "V1")
    for _ in range(2 * slots + 1):
        rt.dist_frame()
    got = rt.dist_read_frame().copy()
    rt.deinit()
    plain = W.make_renderer(w, grid)
    W.set_view(plain, "V1")
    plain.draw()
    assert np.array_equal(got, plain.read_rgba8())
    plain.deinit()


def test_bench_probe_child_runs_the_native_pipeline_on_real_rccl():
    """bench.py --dist-probe, the child process by which the N > 1 bench tries the native RCCL pipeline before it trusts it: here
    with one rank (RCCL refuses two ranks on one device) — one and eight frames per collective, and (round 5) a small path trace whose
    bounce frames the persistent kernels trace inside the pipeline; every assembled frame compared with a single context's.  Round 6:
    the cases share ONE unique id through the library's communicator pool the way the timed legs do (a context made while another holds
    its communicators, one that takes everything from the pool), a communicator per launch slot."""
    import subprocess
    import sys
    from zig_vulkan_amd import VoxelRT
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    uids = VoxelRT.dist_unique_id() + VoxelRT.dist_unique_id() + VoxelRT.dist_unique_id()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dist-probe", "--probe-uid", uids.hex(), "--probe-rank", "0",
                        "--probe-world", "1", "--probe-device", "0"], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-2000:]   # (stdout: RCCL's version banner; the parent discards it)


def test_small_frames_split_every_tile_and_keep_the_frame():
    """configs[0] (256 tiles: one wave per SIMD) goes to two workgroups of 32-lane waves per tile (round 4); a frame whose pixels do not
    fill its last tiles and several samples per pixel as well: the same bytes as one workgroup per tile, and the oracle's."""
    from zig_vulkan_amd import _lib as L
    # (two samples per pixel in a half-tile workgroup: lanes 32-63 trace the second sample of the pixels of lanes 0-31 — bounces and
    # soft sun, both brick sizes, and without bounces)
    for w in (W.WORKLOADS["cfg0_256x256_64c_b4"], W.Workload("small_odd", 203, 121, 64, 8, 3, 0, True, 5.0),
              W.Workload("small_dual_b4", 203, 121, 64, 4, 2, 2, True, 5.0), W.Workload("small_dual_b8", 190, 131, 128, 8, 2, 1, True, 5.0),
              W.Workload("small_dual_nobounce", 120, 75, 64, 4, 2, 0, True, 0.0)):
        grid = W.build_grid(w)
        frames = {}
        for flags in (0, L.TUNE_NO_SMALL_FRAME_SPLIT):
            rt = W.make_renderer(w, grid, tuning_flags=flags, want_float_output=True)
            for v in ("V0", "V2"):
                W.set_view(rt, v)
                rt.draw()
                frames[(flags, v)] = (rt.read_rgba32f().copy(), rt.read_rgba8().copy())
            if flags == 0:
                pc = O.push_constants(rt.camera.blob(), rt.sun.blob())   # (view V2)
            rt.deinit()
        for v in ("V0", "V2"):
            assert np.array_equal(frames[(0, v)][0].view(np.uint32), frames[(L.TUNE_NO_SMALL_FRAME_SPLIT, v)][0].view(np.uint32)), (w.name, v)
            assert np.array_equal(frames[(0, v)][1], frames[(L.TUNE_NO_SMALL_FRAME_SPLIT, v)][1]), (w.name, v)
        fo, uo, _ = O.render(oracle_scene_from_grid(grid), pc)
        assert np.array_equal(frames[(0, "V2")][0].view(np.uint32), fo.view(np.uint32)) and np.array_equal(frames[(0, "V2")][1], uo), w.name


def test_pool_kernel_shards_two_frames_in_flight_and_frames_smaller_than_a_wave():
    """vrt_pool_kernel (round 4: a pool of 128 rays per wave; taken where the counter-free dilated-index walk would run on 8^3 bricks — a
    scene whose occupied cells reach the grid's faces, once the host knows the box): a shard (rank 1 of 3) holds the lockstep kernel's
    packed tiles; two frames in flight (two streams, a pixel counter and a block of path records each) equal the single-stream frames;
    a frame of fewer pixels than ONE wave's 128 paths, odd frame sizes, one sample and sixteen: all the oracle's / the lockstep kernel's."""
    w = W.Workload("t", 208, 112, 256, 8, 2, 2, True, 5.0, "sparse", 0.08, 30000)
    grid = W.build_grid(w)

    def frames(views, **kw):
        rt = W.make_renderer(w, grid, **kw)
        W.set_view(rt, views[0])
        rt.draw()
        rt.wait()      # (the box of the occupied cells has reached the host)
        out = []
        for v in views:
            W.set_view(rt, v)
            rt.draw()
            out.append(rt.read_rgba8().copy())
        name = rt.kernel_name()
        rt.deinit()
        return out, name

    a, name = frames(["V2"], shard_rank=1, shard_count=3, kernel_variant=PATH)
    assert name.startswith("vrt_pool_kernel<8,"), name
    b, name_b = frames(["V2"], shard_rank=1, shard_count=3, kernel_variant=1 << 21)
    assert name_b.startswith("vrt_trace_kernel<8,")
    assert np.array_equal(a[0], b[0]) and a[0].any()
    views = ["V0", "V2", "V1x", "V2", "V0"]
    one, n1 = frames(views, kernel_variant=PATH)
    two, n2 = frames(views, kernel_variant=PATH, frames_in_flight=2)
    assert n1.startswith("vrt_pool_kernel<8,") and n2.startswith("vrt_pool_kernel<8,")
    for v, x, y in zip(views, one, two):
        assert np.array_equal(x, y), v
    for (width, height, spp, bounce) in ((9, 7, 1, 3), (37, 3, 16, 2), (130, 95, 3, 1)):
        ws = W.Workload("s", width, height, 256, 8, spp, bounce, True, 5.0, "sparse", 0.08, 30000)
        rt = W.make_renderer(ws, grid, kernel_variant=PATH, want_float_output=True)
        W.set_view(rt, "V0")
        rt.draw()
        rt.wait()
        rt.draw()
        f, u = rt.read_rgba32f(), rt.read_rgba8()
        assert rt.kernel_name().startswith("vrt_pool_kernel<8,")
        pc = O.push_constants(rt.camera.blob(), rt.sun.blob())
        rt.deinit()
        fo, uo, _ = O.render(oracle_scene_from_grid(grid), pc)
        assert np.array_equal(f.view(np.uint32), fo.view(np.uint32)) and np.array_equal(u, uo), (width, height, spp, bounce)


def test_pool_kernel_sample_buffer_follows_the_sample_count():
    """vrt_pool_kernel takes SAMPLES from its counter and leaves their terms of the sample sum to vrt_pool_resolve_kernel in a buffer
    sized by the frames asked for: one context, two frames in flight, the sample count going 2 -> 16 -> 3 -> 1 -> 5 between frames (the
    buffer grows behind both streams); every frame equals vrt_path_kernel's, which sums a pixel's samples in its own lane."""
    w = W.Workload("t", 208, 112, 256, 8, 2, 2, True, 5.0, "sparse", 0.08, 30000)
    grid = W.build_grid(w)
    seq = [("V0", 2), ("V2", 16), ("V2", 16), ("V1x", 3), ("V0", 1), ("V2", 5), ("V0", 5)]

    def frames(**kw):
        rt = W.make_renderer(w, grid, kernel_variant=PATH, **kw)
        W.set_view(rt, "V0")
        rt.draw()
        rt.wait()      # (the box of the occupied cells has reached the host)
        out, names = [], set()
        for v, spp in seq:
            W.set_view(rt, v)
            rt.camera.d_camera.samples_per_pixel = spp
            rt.draw()
            out.append(rt.read_rgba8().copy())
            names.add(rt.kernel_name().split("<")[0])
        rt.deinit()
        return out, names

    pool, n_pool = frames(frames_in_flight=2)
    path, n_path = frames(tuning_flags=L.TUNE_NO_PATH_POOL, frames_in_flight=2)     # (vrt_path_kernel, samples as units as well)
    pixels, n_pixels = frames(tuning_flags=L.TUNE_NO_SAMPLE_UNITS)                  # (vrt_path_kernel, a pixel per lane: the sum in its lane)
    assert n_pool == {"vrt_pool_kernel"} and n_path == {"vrt_path_kernel"} and n_pixels == {"vrt_path_kernel"}, (n_pool, n_path, n_pixels)
    for (v, spp), a, b, c in zip(seq, pool, path, pixels):
        assert np.array_equal(a, c) and np.array_equal(b, c) and a.any(), (v, spp)


def test_bounce_frames_as_one_wave_workgroups_keep_the_frame():
    """The lockstep bounce kernel is launched as one-wave workgroups (round 4: a tile's four waves end at different times, and a 256-thread
    workgroup waits for four free slots of one CU): the reference app's shape at a size whose tiles the cost schedule splits, two and three
    samples per pixel, an odd frame, two frames in flight — the same bytes as with a 256-thread workgroup per tile
    (VRT_TUNE_NO_BOUNCE_WAVE_GROUPS), and the oracle's on the last view."""
    from zig_vulkan_amd import _lib as L
    for w, fif in ((W.Workload("app_like", 509, 283, 256, 4, 2, 2, True, 5.0), 1), (W.Workload("app_like_3spp", 320, 200, 128, 8, 3, 2, True, 5.0), 2), (W.Workload("two_spp_one_bounce", 640, 360, 256, 8, 2, 1, True, 5.0), 1),
                   (W.Workload("one_sample_two_in_flight", 640, 360, 256, 8, 1, 1, True, 5.0), 2)):   # (reverse raster on both streams: one-wave workgroups too)
        grid = W.build_grid(w)
        frames = {}
        for flags in (0, L.TUNE_NO_BOUNCE_WAVE_GROUPS):
            rt = W.make_renderer(w, grid, tuning_flags=flags, want_float_output=True, frames_in_flight=fif, kernel_variant=1 << 21)   # (lockstep)
            for v in ("V0", "V2", "V1"):
                W.set_view(rt, v)
                for _ in range(40):   # (past a sort of the cost schedule: split tiles, half-tile waves, the second sample on the idle lanes)
                    rt.draw()
                frames[(flags, v)] = (rt.read_rgba32f().copy(), rt.read_rgba8().copy())
            assert rt.kernel_name().startswith("vrt_trace_kernel<"), rt.kernel_name()
            if flags == 0:
                pc = O.push_constants(rt.camera.blob(), rt.sun.blob())   # (view V1)
            rt.deinit()
        for v in ("V0", "V2", "V1"):
            assert np.array_equal(frames[(0, v)][0].view(np.uint32), frames[(L.TUNE_NO_BOUNCE_WAVE_GROUPS, v)][0].view(np.uint32)), (w.name, v)
            assert np.array_equal(frames[(0, v)][1], frames[(L.TUNE_NO_BOUNCE_WAVE_GROUPS, v)][1]), (w.name, v)
        fo, uo, _ = O.render(oracle_scene_from_grid(grid), pc)
        assert np.array_equal(frames[(0, "V1")][0].view(np.uint32), fo.view(np.uint32)) and np.array_equal(frames[(0, "V1")][1], uo), w.name


def test_pool_kernel_defers_the_hit_material_only_where_no_record_can_be_ignored():
    """vrt_pool_kernel leaves comp:337 / :422-427's look-ups of a hit to the round of transitions that shades it while no material record
    has the type MAT_NONE (TraceParams::materials_plain): the same bytes as with the look-ups in the brick round
    (VRT_TUNE_NO_DEFERRED_MATERIAL).  A table with a MAT_NONE record whose type_data is 1 — which camera and shadow rays DO ignore
    (comp:427) — uploaded between two frames of one context turns the deferral off: that frame is the oracle's, and differs."""
    from zig_vulkan_amd import _lib as L
    from zig_vulkan_amd import default_materials
    w = W.Workload("t", 208, 112, 256, 8, 2, 2, True, 5.0, "sparse", 0.08, 30000)
    grid = W.build_grid(w)
    mats = default_materials(256)
    mats[3] = (3, 0.3, 0.9, 0.3, 1.0)     # MAT_NONE, type_data == the rays' refraction index: ignored by comp:427

    def frames(**kw):
        rt = W.make_renderer(w, grid, kernel_variant=PATH, want_float_output=True, **kw)
        W.set_view(rt, "V0")
        rt.draw()
        rt.wait()      # (the box of the occupied cells has reached the host)
        out = []
        for v in ("V0", "V2"):
            W.set_view(rt, v)
            rt.draw()
            out.append((rt.read_rgba32f().copy(), rt.read_rgba8().copy()))
        rt.push_materials(mats)
        rt.draw()
        out.append((rt.read_rgba32f().copy(), rt.read_rgba8().copy()))
        assert rt.kernel_name().startswith("vrt_pool_kernel<8,"), rt.kernel_name()
        pc = O.push_constants(rt.camera.blob(), rt.sun.blob())
        rt.deinit()
        return out, pc

    a, pc = frames()
    b, _ = frames(tuning_flags=L.TUNE_NO_DEFERRED_MATERIAL)
    for (fa, ua), (fb, ub) in zip(a, b):
        assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32)) and np.array_equal(ua, ub) and ua.any()
    scene = oracle_scene_from_grid(grid)
    fo, uo, _ = O.render(scene, pc)                      # (the default table)
    assert np.array_equal(a[1][0].view(np.uint32), fo.view(np.uint32)) and np.array_equal(a[1][1], uo)
    scene2 = oracle_scene_from_grid(grid, materials=mats)
    fo2, uo2, _ = O.render(scene2, pc)
    assert np.array_equal(a[2][0].view(np.uint32), fo2.view(np.uint32)) and np.array_equal(a[2][1], uo2)
    assert not np.array_equal(uo, uo2)


def test_frames_the_pool_kernel_cannot_take_keep_the_counter_free_path_kernel():
    """vrt_pool_kernel packs the bounce count into 4 bits; a frame of more than 15 bounces used to fall back to
    vrt_path_kernel<..., DIL 1> (steps-left counters in the walk loop) although the box of the occupied cells is the grid — now the DIL-2
    twin is kept beside the pool kernel and takes such frames.  Same bytes as the lockstep kernel's frame."""
    w = W.Workload("t", 208, 112, 256, 8, 2, 2, True, 5.0, "sparse", 0.08, 30000)
    grid = W.build_grid(w)
    out = {}
    for variant in (PATH, 1 << 21):
        rt = W.make_renderer(w, grid, kernel_variant=variant)
        W.set_view(rt, "V0")
        rt.draw()
        rt.wait()      # (the box of the occupied cells has reached the host)
        W.set_view(rt, "V2")
        rt.draw()
        first = rt.kernel_name()
        rt.camera.d_camera.max_bounce = 17
        rt.draw()
        out[variant] = (rt.read_rgba8().copy(), first, rt.kernel_name())
        rt.deinit()
    assert out[PATH][1] == "vrt_pool_kernel<8, 6, 60, 2>" and out[PATH][2] == "vrt_path_kernel<8, 5, false, false, false, false, 2>", out[PATH][1:]
    assert out[1 << 21][2].startswith("vrt_trace_kernel<8,")
    assert np.array_equal(out[PATH][0], out[1 << 21][0]) and out[PATH][0].any()


def test_sample_buffers_reserved_up_front_and_kept_when_growth_fails():
    """vrt_reserve_samples sizes the persistent kernels' sample buffers once (no allocation, no drain inside a dispatch);
    a request no memory can serve — 65 535 samples per pixel of a 4K frame — is VRT_E_OOM / VRT_E_OUT_OF_RANGE and leaves the context
    as it was: the next frames are still vrt_pool_kernel's and still right.  Contexts without persistent kernels have nothing to reserve."""
    from zig_vulkan_amd import _lib as L
    w = W.Workload("t", 208, 112, 256, 8, 2, 2, True, 5.0, "sparse", 0.08, 30000)
    grid = W.build_grid(w)
    ref = W.make_renderer(w, grid, kernel_variant=1 << 21)
    W.set_view(ref, "V2")
    ref.camera.d_camera.samples_per_pixel = 7
    ref.draw()
    want = ref.read_rgba8().copy()
    ref.reserve_samples(64)          # (the lockstep kernel: nothing to reserve, VRT_OK)
    ref.deinit()
    for fif in (1, 2):
        rt = W.make_renderer(w, grid, kernel_variant=PATH, frames_in_flight=fif)
        rt.reserve_samples(7)
        W.set_view(rt, "V0")
        rt.draw()
        rt.wait()
        W.set_view(rt, "V2")
        rt.camera.d_camera.samples_per_pixel = 7
        for _ in range(3):
            rt.draw()
        assert rt.kernel_name() == "vrt_pool_kernel<8, 6, 60, 2>" and np.array_equal(rt.read_rgba8(), want)
        with pytest.raises(L.VrtError):
            rt.reserve_samples(70000)
        big = W.make_renderer(W.Workload("big", 3840, 2160, 256, 8, 2, 2, True, 5.0, "sparse", 0.08, 30000), grid, kernel_variant=PATH)
        with pytest.raises(L.VrtError) as e:
            big.reserve_samples(65535)      # 8.3 M pixels x 65 535 samples: beyond the 32-bit unit counter
        assert e.value.code in (L.VRT_E_OOM, L.VRT_E_OUT_OF_RANGE)
        big.deinit()
        rt.draw()
        rt.draw()
        assert rt.kernel_name() == "vrt_pool_kernel<8, 6, 60, 2>" and np.array_equal(rt.read_rgba8(), want)
        rt.deinit()


def test_wave_timeline_across_a_change_of_the_schedule_rule():
    """vrt_trace_wave_timeline sized its device buffer by the spare workgroups of the CURRENT schedule rule and
    then dispatched a frame that may change the rule (two samples per pixel: every tile may be split) — the launch then wrote rows past
    the buffer.  The buffer now holds a row for every wave any rule can launch, and the row count is that of the launch: timelines of
    1-, 2-, 1-sample frames in turn, each followed by a plain frame that must still be the fresh context's."""
    w = W.Workload("t", 1024, 576, 256, 4, 1, 2, True, 5.0)
    grid = W.build_grid(w)
    fresh = {}
    for spp in (1, 2):
        rt = W.make_renderer(w, grid, kernel_variant=1 << 21)
        W.set_view(rt, "V1")
        rt.camera.d_camera.samples_per_pixel = spp
        rt.draw()
        fresh[spp] = rt.read_rgba8().copy()
        rt.deinit()
    rt = W.make_renderer(w, grid, kernel_variant=1 << 21)
    tiles = rt.shard_info().owned_tiles
    W.set_view(rt, "V1")
    for spp in (1, 2, 1, 2, 2, 1):
        rt.camera.d_camera.samples_per_pixel = spp
        rt.draw(frames=40)           # (past a sort of the cost schedule under this rule)
        rows = rt.wave_timeline(raw=True)
        assert tiles * 4 <= len(rows) <= tiles * 8, (spp, len(rows), tiles)
        assert (rows[:, 1] >= rows[:, 0]).all()
        rt.draw()
        assert np.array_equal(rt.read_rgba8(), fresh[spp]), spp
    rt.deinit()


def test_pool_kernel_takes_the_material_of_one_material_bricks_from_a_byte_per_cell():
    """Round 5 (a third of the 2048^3 path trace's fabric traffic was a hit's two dependent look-ups, brick_index then one
    byte of the 2 GiB material_index): a derived byte per cell names the material ALL solid voxels of the cell's brick share, 0xFF where
    they do not (TraceParams::cell_material).  A scene with both kinds — spheres of one material each, plus 6 000 stray voxels of random
    materials that make the bricks they fall into mixed: frames with and without the structure (VRT_TUNE_NO_CELL_MATERIAL) and the
    oracle's are the same bytes.  Then the host re-inserts 500 solid voxels with ANOTHER material — only material_index entries change,
    uniform bricks become mixed — and the delta upload must refresh exactly those bytes: the next frame is the oracle's of the edited grid."""
    w = W.Workload("t", 208, 112, 256, 8, 2, 2, True, 5.0, "sparse", 0.08, 30000)
    rng = np.random.default_rng(11)
    stray = rng.integers(0, 256, (6000, 3))
    stray_m = rng.integers(1, 8, 6000)

    def build():
        g = W.build_grid(w)
        g.insert_many(stray, stray_m)
        return g

    def frames(grid, flags, edit=None):
        rt = W.make_renderer(w, grid, kernel_variant=PATH, want_float_output=True, tuning_flags=flags)
        W.set_view(rt, "V0")
        rt.draw()
        rt.wait()      # (the box of the occupied cells has reached the host)
        out = []
        for v in ("V0", "V2"):
            W.set_view(rt, v)
            rt.draw()
            out.append((rt.read_rgba32f().copy(), rt.read_rgba8().copy()))
        assert rt.kernel_name() == "vrt_pool_kernel<8, 6, 60, 2>", rt.kernel_name()
        if edit is not None:
            edit(grid)
            rt.update_grid_delta()
            # (insert() rewrites the cell's status word as well: the box of the occupied cells is unknown until its copy has come back, and
            # the first frame behind the edit is vrt_path_kernel<..., DIL 1>'s; the one behind that vrt_pool_kernel's again — both checked)
            rt.draw()
            first = (rt.read_rgba32f().copy(), rt.read_rgba8().copy())
            rt.draw()
            out.append((rt.read_rgba32f().copy(), rt.read_rgba8().copy()))
            assert rt.kernel_name() == "vrt_pool_kernel<8, 6, 60, 2>", rt.kernel_name()
            assert np.array_equal(first[0].view(np.uint32), out[-1][0].view(np.uint32)) and np.array_equal(first[1], out[-1][1])
        pc = O.push_constants(rt.camera.blob(), rt.sun.blob())
        rt.deinit()
        return out, pc

    # solid voxels of the spheres, re-inserted with another material (insert() flips y: use voxels this scene is known to hold — the strays)
    def edit(g):
        g.insert_many(stray[:500], (stray_m[:500] % 7) + 1)

    grid = build()
    a, pc = frames(grid, 0, edit)
    b, _ = frames(build(), L.TUNE_NO_CELL_MATERIAL, edit)
    for (fa, ua), (fb, ub) in zip(a, b):
        assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32)) and np.array_equal(ua, ub) and ua.any()
    fo, uo, _ = O.render(oracle_scene_from_grid(grid), pc)        # (the edited grid, view V2)
    assert np.array_equal(a[2][0].view(np.uint32), fo.view(np.uint32)) and np.array_equal(a[2][1], uo)
    fo1, uo1, _ = O.render(oracle_scene_from_grid(build()), pc)   # (before the edit)
    assert np.array_equal(a[1][0].view(np.uint32), fo1.view(np.uint32)) and np.array_equal(a[1][1], uo1)
    assert not np.array_equal(uo, uo1), "the edit changed no pixel: the test does not test the refresh"


def test_pool_kernel_on_bricks_of_4_and_on_a_box_smaller_than_the_grid():
    """Round 5: vrt_pool_kernel for the reference's own brick size (4^3: a brick is two words, read as its walk goes —
    no staging in LDS) and, with VRT_TUNE_GRID_EXIT_ANY_BOX, on scenes whose occupied cells do NOT reach the grid's faces (a terrain: rays
    that leave its box walk the empty cells up to the grid's face).  The reference app's shape — 128 x 64 x 128 bricks of 4^3, 2 samples,
    2 bounces, soft sun — at test size, and a sparse field of 4^3 bricks: frames equal the lockstep kernel's and the oracle's."""
    cases = [(W.Workload("app_like", 320, 180, 512, 4, 2, 2, True, 5.0, dims=(128, 64, 128)), L.TUNE_GRID_EXIT_ANY_BOX, ("V0", "V2", "V1")),
             (W.Workload("sparse_b4", 208, 112, 128, 4, 3, 2, True, 5.0, "sparse", 0.08, 30000), L.TUNE_GRID_EXIT_ANY_BOX, ("V0", "V1x")),
             (W.Workload("terrain_b8", 256, 144, 256, 8, 2, 2, True, 5.0), L.TUNE_GRID_EXIT_ANY_BOX, ("V1", "V0"))]
    for w, flags, views in cases:
        grid = W.build_grid(w)
        out = {}
        for variant, fl in ((PATH, flags), (1 << 21, 0)):
            rt = W.make_renderer(w, grid, kernel_variant=variant, tuning_flags=fl, want_float_output=True)
            W.set_view(rt, views[0])
            rt.draw()
            rt.wait()      # (the box of the occupied cells has reached the host)
            frames = []
            for v in views:
                W.set_view(rt, v)
                rt.draw()
                frames.append((rt.read_rgba32f().copy(), rt.read_rgba8().copy()))
            out[variant] = (frames, rt.kernel_name(), O.push_constants(rt.camera.blob(), rt.sun.blob()))
            rt.deinit()
        assert out[PATH][1] == ("vrt_pool_kernel<8, 6, 60, 2>" if w.brick_dimension == 8 else "vrt_pool_kernel<4, 6, 64, 0>"), (w.name, out[PATH][1])
        assert out[1 << 21][1].startswith("vrt_trace_kernel<"), out[1 << 21][1]
        for v, (fa, ua), (fb, ub) in zip(views, out[PATH][0], out[1 << 21][0]):
            assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32)) and np.array_equal(ua, ub) and ua.any(), (w.name, v)
        fo, uo, _ = O.render(oracle_scene_from_grid(grid), out[PATH][2])
        assert np.array_equal(out[PATH][0][-1][0].view(np.uint32), fo.view(np.uint32)) and np.array_equal(out[PATH][0][-1][1], uo), w.name


def test_bounce_kernel_is_chosen_by_timing_where_both_kernels_apply():
    """Round 5: the size rule gives bounce frames of scenes that stay in the caches to the lockstep kernel — right for a terrain, wrong
    by 1.6-2.2 x for a sparse field (profiles/r05_pool_generalised_ab.txt).  Where vrt_pool_kernel applies as well the library times
    four single-frame dispatches (lockstep / pool / lockstep / pool) and keeps the faster.  A sparse field of 4^3 bricks whose spheres
    reach the grid's faces: the trial frames and every frame behind them are the lockstep kernel's bytes, and the verdict is `pool`;
    VRT_TUNE_NO_BOUNCE_AUTOTUNE and kernel_variant bit 21 keep the lockstep kernel without trials; the reference app's shape (a terrain:
    its box is not the grid) and an odd-sized grid never start any."""
    # (a 1024^3 field of 4^3 bricks at the reference app's frame size: lockstep 1.8 ms, pool 1.1 at 2 spp — the same spheres in a 512^3 grid
    # are closer together, their rays shorter, and there the lockstep kernel wins the trials: 0.58 against 0.74 ms)
    w = W.Workload("sparse_auto", 1024, 576, 1024, 4, 4, 2, True, 5.0, "sparse", 0.08, 2_000_000)
    grid = W.build_grid(w)
    ref = W.make_renderer(w, grid, kernel_variant=1 << 21)
    assert ref.bounce_autotune_info()["state"] == "not applicable"
    want = {}
    for v in ("V0", "V2"):
        W.set_view(ref, v)
        ref.draw()
        want[v] = ref.read_rgba8().copy()
    ref.deinit()
    for fif in (1, 2):
        rt = W.make_renderer(w, grid, frames_in_flight=fif)
        names = []
        for i in range(14):
            v = ("V0", "V2")[(i // 3) % 2]
            W.set_view(rt, v)
            rt.draw()
            assert np.array_equal(rt.read_rgba8(), want[v]), (fif, i, rt.kernel_name())
            names.append(rt.kernel_name().split("<")[0])
        info = rt.bounce_autotune_info()
        assert info["state"] == "pool" and info["trials_launched"] == 4 and 0 < info["pool_ms"] < info["lockstep_ms"], info
        assert names[-1] == "vrt_pool_kernel" and names[0] == "vrt_trace_kernel", names
        rt.push_materials(default_materials_for_test())     # (not a status upload: the verdict stands)
        rt.draw()
        assert rt.bounce_autotune_info()["state"] == "pool"
        rt.deinit()
    off = W.make_renderer(w, grid, tuning_flags=L.TUNE_NO_BOUNCE_AUTOTUNE)
    W.set_view(off, "V0")
    for _ in range(8):
        off.draw()
    assert off.bounce_autotune_info()["state"] == "not applicable" and off.kernel_name().startswith("vrt_trace_kernel<") and np.array_equal(off.read_rgba8(), want["V0"])
    off.deinit()
    for wl in (W.Workload("app_like", 320, 180, 512, 4, 2, 2, True, 5.0, dims=(128, 64, 128)), W.Workload("odd", 200, 120, 96, 8, 2, 2, True, 5.0)):
        g = W.build_grid(wl)
        rt = W.make_renderer(wl, g)
        W.set_view(rt, "V1")
        for _ in range(8):
            rt.draw()
        rt.wait()
        info = rt.bounce_autotune_info()
        assert info["trials_launched"] == 0 and rt.kernel_name().startswith("vrt_trace_kernel<"), (wl.name, info)
        rt.deinit()


def default_materials_for_test():
    from zig_vulkan_amd import default_materials
    return default_materials(256)
