"""BASELINE.json configurations at their FULL sizes: the HIP frame is rendered whole, the oracle
renders a seeded random sample of its pixels (any pixel is independent), compared bit-for-bit; plus
size-independent properties (ray count bounds, alpha, determinism across launches and across shards)."""
import numpy as np
import pytest

from tests.helpers import O, dev_library_or_none, oracle_scene_from_grid, variant_kwargs
from zig_vulkan_amd import _lib as L
from zig_vulkan_amd import workloads as W

pytestmark = pytest.mark.gpu


def _sampled_parity(name, view, n_pixels, seed=1234, kernels=None, **overrides):
    w = W.WORKLOADS[name]
    grid = W.build_grid(w)
    # A counting context runs the counting build once, overwrites both targets with 0xCD and then lets the PRODUCT kernel
    # render the frame that is read back (vrt_hip.h): a pixel the product kernel leaves unwritten reads back as poison,
    # which the whole-frame alpha checks below catch for every pixel, not only the sampled ones.
    rt = W.make_renderer(w, grid, want_float_output=True, enable_counters=True, **overrides)
    W.set_view(rt, view)
    rt.draw()
    f = rt.read_rgba32f()
    u = rt.read_rgba8()
    c1 = rt.counters()
    first = rt.kernel_name()
    rt.draw()  # determinism: a second launch gives the same bytes
    assert np.array_equal(rt.read_rgba8(), u)
    if kernels is not None:  # (the product kernels of the two frames: the second may be another twin, see the caller)
        assert (first, rt.kernel_name()) == kernels
    pc = O.push_constants(rt.camera.blob(), rt.sun.blob())
    rt.deinit()
    rng = np.random.default_rng(seed)
    xy = np.stack([rng.integers(0, w.width, n_pixels), rng.integers(0, w.height, n_pixels)], axis=-1).astype(np.int32)
    fo, uo, co = O.render_pixels(oracle_scene_from_grid(grid), pc, xy)
    got_f = f[xy[:, 1], xy[:, 0]]
    got_u = u[xy[:, 1], xy[:, 0]]
    assert np.abs(got_f - fo).max() <= 1e-4
    assert np.array_equal(got_f.view(np.uint32), fo.view(np.uint32))
    assert np.array_equal(got_u, uo)
    assert (u[..., 3] == 255).all() and (f[..., 3] == 1.0).all()
    px = w.width * w.height
    assert px * w.spp <= c1["rays"] <= px * w.spp * 2 * (w.max_bounce + 1)
    return c1


def test_cfg1_1080p_256c_primary():
    c = _sampled_parity("cfg1_1080p_256c_b4", "V2", 30000)
    assert c["rays"] == 1920 * 1080  # sun disabled: exactly one ray per pixel


@pytest.mark.parametrize("name", ["cfg2_1080p_512c_b8", "cfg2_1080p_512c_b4"])
def test_cfg2_headline_primary_plus_shadow(name):
    _sampled_parity(name, "V0", 30000)


def test_cfg3_4k_1024c_four_rays_per_pixel():
    _sampled_parity("cfg3_4k_1024c_b8", "V2", 20000)


def test_cfg4_4k_2048c_sparse_path_trace():
    # the first frame is traced on the dilated cell index with steps-left counters and compared with the oracle; behind it the host
    # knows that the spheres fill the grid, and the second frame — vrt_pool_kernel's (round 4: a pool of 128 rays per wave) — must be the same bytes
    _sampled_parity("cfg4_4k_2048c_b8_sparse", "V1", 3000,
                    kernels=("vrt_path_kernel<8, 5, false, false, false, false, 1>", "vrt_pool_kernel<8, 6, 60, 2>"))


def _whole_frame_cases():
    import glob
    import os
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "full", "cfg[134]_*.npz")))
    return files, [os.path.basename(p)[:-4] for p in files]


_WF_FILES, _WF_IDS = _whole_frame_cases()


def test_whole_frame_fixtures_exist():
    """every BASELINE configuration has whole-frame digests (configs[2]: tests/golden/full/cfg2_* and ref_full/)."""
    assert {"cfg1_V0", "cfg1_V1", "cfg1_V2", "cfg3_V1", "cfg3_V2", "cfg4_V1", "cfg4_V1x"} <= set(_WF_IDS)


@pytest.mark.parametrize("path", _WF_FILES, ids=_WF_IDS)
def test_whole_frames_of_the_baseline_configurations_are_the_oracles(path):
    """The WHOLE frame of BASELINE configs[1], [3] and [4] at full size — every pixel, float bits and RGBA8 — against the oracle's
    digests (tests/golden/make_full_golden.py: SHA-256 of the frame, per band of 16 rows, eight float crops).  A product-only context
    (no counting build ever touches its target).  configs[4]: BOTH frames — the first by vrt_path_kernel<..., DIL 1>, the second, once
    the host knows the box of the occupied cells, by vrt_pool_kernel (the kernel that ships for this configuration) — until round 4 the
    pool kernel's full-size frame was compared as RGBA8 only, and on 3 000 sampled pixels of 8.3 M."""
    import hashlib
    from tests.golden.make_full_golden import digest
    from tests.golden.make_golden import scene_digest
    z = np.load(path)
    w = W.WORKLOADS[str(z["workload"])]
    grid = W.build_grid(w)
    assert scene_digest(grid) == str(z["scene_sha256"]), "the synthetic scene generator no longer produces the fixture's scene"
    rt = W.make_renderer(w, grid, want_float_output=True)
    W.set_view(rt, str(z["view"]))
    assert O.push_constants(rt.camera.blob(), rt.sun.blob()).tobytes() == z["push_constants"].tobytes()
    frames = 2 if w.max_bounce > 0 else 1
    names = []
    for k in range(frames):
        rt.draw()
        f, u = rt.read_rgba32f(), rt.read_rgba8()    # (waits: by the next frame the box of the occupied cells has reached the host)
        names.append(rt.kernel_name())
        d = digest(f, u)
        bad = [i for i, (a, b) in enumerate(zip(d["band_sha256"], z["band_sha256"])) if str(a) != str(b)]
        for (y, x), got, want in zip(z["crop_origins"], d["float_crops"], z["float_crops"]):
            n = int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum())
            assert n == 0, f"{names[-1]}: crop at ({x},{y}): {n} pixels differ from the oracle's"
        assert not bad, f"{names[-1]}: bands of 16 rows that differ from the oracle's frame: {bad[:20]} ({len(bad)} of {len(z['band_sha256'])})"
        assert str(d["float_sha256"]) == str(z["float_sha256"]) and str(d["rgba8_sha256"]) == str(z["rgba8_sha256"]), names[-1]
        assert (u[..., 3] == 255).all() and (f[..., 3] == 1.0).all()
    rt.deinit()
    if str(z["workload"]).startswith("cfg4"):
        assert names == ["vrt_path_kernel<8, 5, false, false, false, false, 1>", "vrt_pool_kernel<8, 6, 60, 2>"], names


def test_grid_edits_reach_the_next_dispatch():
    """insert() after the first frame, vrt_update_grid_delta uploads only the dirty
    ranges, and the very next dispatch sees them (VoxelRT.updateGridDelta, VoxelRT.zig:107-172)."""
    w = W.Workload("t", 320, 200, 64, 4, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    rt = W.make_renderer(w, grid, want_float_output=True, enable_counters=True)
    W.set_view(rt, "V1")
    rt.draw()
    before = rt.read_rgba8().copy()
    # build a tower of iron in the middle of the map and punch a second one in a fresh brick column
    for y in range(20, 60):
        for dx in range(3):
            for dz in range(3):
                grid.insert(30 + dx, y, 30 + dz, 7)
                grid.insert(5 + dx, y, 50 + dz, 5)
    from zig_vulkan_amd import _lib as L
    active, a, b = grid.delta(L.BUF_BRICK_OCCUPANCY)
    assert active and b - a < grid.array(L.BUF_BRICK_OCCUPANCY).size  # a partial range, not everything
    rt.update_grid_delta()
    assert grid.delta(L.BUF_BRICK_OCCUPANCY)[0] is False  # deltas were reset
    rt.draw()
    f, u, c = rt.read_rgba32f(), rt.read_rgba8(), rt.counters()
    pc = O.push_constants(rt.camera.blob(), rt.sun.blob())
    rt.deinit()
    assert not np.array_equal(u, before)
    fo, uo, co = O.render(oracle_scene_from_grid(grid), pc)
    assert np.array_equal(f.view(np.uint32), fo.view(np.uint32)) and np.array_equal(u, uo) and c == co


def test_grid_edits_reach_the_path_kernels_derived_structures():
    """The same for a context that traces its frames with vrt_path_kernel (bounces; forced here): its walk loop reads a derived
    copy of the status bits (half-block words) and the occupied-cell box, both rebuilt on upload — a clump inserted into an empty
    corner of the grid, outside the old box, must be seen by the next frame."""
    from zig_vulkan_amd import BrickGrid
    w = W.Workload("edit_path", 320, 180, 256, 8, 2, 2, True, 5.0)
    grid = BrickGrid(32, 32, 32, min_point=(-16.0, -16.0, -16.0), scale=1.0, brick_dimension=8, brick_alloc=4000)
    rng = np.random.default_rng(5)
    grid.insert_many(np.stack([rng.integers(96, 160, 4000) for _ in range(3)], axis=-1), rng.integers(0, 6, 4000))   # a blob in the middle
    rt = W.make_renderer(w, grid, kernel_variant=1 << 23)
    rt.camera.look_at((-8.0, 8.0, -8.0), (-13.0, 13.0, -13.0))  # from inside the grid towards the corner that is edited below (insert() flips y)
    rt.draw()
    before = rt.read_rgba8().copy()
    grid.insert_many(np.stack([rng.integers(8, 40, 3000) for _ in range(3)], axis=-1), rng.integers(0, 6, 3000))     # the corner: outside the blob's box
    rt.update_grid_delta()
    rt.draw()
    after = rt.read_rgba8().copy()
    pc = O.push_constants(rt.camera.blob(), rt.sun.blob())
    rt.deinit()
    assert (before != after).any()
    _, uo, _ = O.render(oracle_scene_from_grid(grid), pc)
    assert np.array_equal(after, uo)


def test_headline_frames_are_deterministic_with_two_in_flight():
    """90 frames of the headline workload, two in flight, views cycled: every frame's bytes equal the first frame of
    its view (a timing-dependent fault in the hand-written loops would show up as a differing frame;
    tools/soak.py runs the same for thousands of frames)."""
    import hashlib
    w = W.WORKLOADS[W.HEADLINE]
    grid = W.build_grid(w)
    rt = W.make_renderer(w, grid, frames_in_flight=2)
    ref = {}
    for i in range(90):
        view = ["V0", "V1", "V2"][i % 3]
        W.set_view(rt, view)
        rt.draw()
        if i % 2 == 1:
            rt.draw()
        h = hashlib.sha1(rt.read_rgba8().tobytes()).hexdigest()
        assert ref.setdefault(view, h) == h, f"frame {i} ({view}) differs"
    rt.deinit()
    assert len(set(ref.values())) == 3


def test_headline_cost_ordered_launch_and_split_tiles_keep_the_frame():
    """One frame at a time (the default tile order of such a context: measured cost, re-sorted every 32 frames, heaviest
    tiles split into two half-tile workgroups): the first frame of a view (reverse raster, nothing measured yet) and the
    frame after 70 more are the same bytes, for all views; on V1x the settled launch really contains split tiles
    (more waves than 4 per tile)."""
    import hashlib
    w = W.WORKLOADS[W.HEADLINE]
    grid = W.build_grid(w)
    fresh = {}
    for view in ["V0", "V1", "V2", "V1x"]:
        rt = W.make_renderer(w, grid)
        W.set_view(rt, view)
        rt.draw()
        fresh[view] = hashlib.sha1(rt.read_rgba8().tobytes()).hexdigest()
        rt.deinit()
    rt = W.make_renderer(w, grid)
    tiles = rt.shard_info().owned_tiles
    for view in ["V1", "V0", "V2", "V1x"]:
        W.set_view(rt, view)
        rt.draw(frames=70)
        rt.draw()
        assert hashlib.sha1(rt.read_rgba8().tobytes()).hexdigest() == fresh[view], view
    waves = len(rt.wave_timeline())
    assert waves > tiles * 4, "no tile was split on V1x"
    assert hashlib.sha1(rt.read_rgba8().tobytes()).hexdigest() == fresh["V1x"]
    rt.deinit()


def _golden_full(view):
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "full", f"cfg2_{view}.npz"))


@pytest.mark.parametrize("view", ["V0", "V1", "V2", "V1x"])
def test_headline_settled_frame_is_the_committed_golden_frame(view):
    """PRODUCT-ONLY context (no counting build ever touches its target), headline size, hard sun: after 70 frames the
    launch order has been re-sorted twice from measured costs and heavy tiles are split — the frame must still be the
    oracle's frame, checked three ways: SHA-256 of the whole RGBA8 and float frames against tests/golden/full (made by
    the oracle in tests/golden/make_golden.py), the committed crops, and a fresh oracle sample of 20 000 pixels."""
    import ctypes as C
    import hashlib
    z = _golden_full(view)
    w = W.WORKLOADS[str(z["workload"])]
    grid = W.build_grid(w)
    from tests.golden.make_golden import scene_digest
    assert scene_digest(grid) == str(z["scene_sha256"])
    rt = W.make_renderer(w, grid, want_float_output=True, sun_radius=0.0)
    pc = z["push_constants"].tobytes()
    C.memmove(C.byref(rt.camera.d_camera), pc[:96], 96)
    C.memmove(C.byref(rt.sun.device_data), pc[96:], 32)
    first = None
    for frames in (1, 70, 1):
        rt.draw(frames=frames)
        u = rt.read_rgba8()
        f = rt.read_rgba32f()
        assert hashlib.sha256(u.tobytes()).hexdigest() == str(z["rgba8_sha256"]), f"after {frames} more frame(s)"
        assert hashlib.sha256(f.tobytes()).hexdigest() == str(z["float_sha256"])
        first = first if first is not None else u
    rt.deinit()
    y0, x0 = z["crop_origin"].tolist()
    h, wd = z["rgba8_crop"].shape[:2]
    assert np.array_equal(u[y0:y0 + h, x0:x0 + wd], z["rgba8_crop"])
    assert np.array_equal(f[y0:y0 + h, x0:x0 + wd].view(np.uint32), z["float_crop"].view(np.uint32))
    rng = np.random.default_rng(99)
    xy = np.stack([rng.integers(0, w.width, 20000), rng.integers(0, w.height, 20000)], axis=-1).astype(np.int32)
    fo, uo, _ = O.render_pixels(oracle_scene_from_grid(grid), z["push_constants"].copy(), xy)
    assert np.array_equal(f[xy[:, 1], xy[:, 0]].view(np.uint32), fo.view(np.uint32))
    assert np.array_equal(u[xy[:, 1], xy[:, 0]], uo)


def test_counting_context_reads_back_the_product_kernels_frame():
    """The 0xCD poison between the counting build and the product kernel: with the product launch suppressed the frame
    would be poison, so a frame equal to the oracle's proves the product kernel wrote every pixel of it.  Also: the
    counters are those of ONE frame however many frames the call renders."""
    w = W.Workload("t", 640, 360, 128, 8, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    rt = W.make_renderer(w, grid, want_float_output=True, enable_counters=True)
    W.set_view(rt, "V2")
    rt.draw()
    c1 = rt.counters()
    rt.draw(frames=5)
    c5 = rt.counters()
    f, u = rt.read_rgba32f(), rt.read_rgba8()
    pc = O.push_constants(rt.camera.blob(), rt.sun.blob())
    rt.deinit()
    assert c1 == c5
    fo, uo, co = O.render(oracle_scene_from_grid(grid), pc)
    assert c1 == co
    assert np.array_equal(f.view(np.uint32), fo.view(np.uint32)) and np.array_equal(u, uo)
    assert not (u == 0xCD).all(axis=2).any()


def test_issued_counters_walk_to_the_occupied_box():
    """enable_counters = 2: the counting build ends its brick-level walk where the product kernel does.  Hits, bricks
    entered and voxel steps are the reference algorithm's (the box only removes empty cells); grid steps shrink."""
    w = W.Workload("t", 640, 360, 128, 8, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    got = {}
    for mode in (1, 2):
        rt = W.make_renderer(w, grid, enable_counters=mode)
        W.set_view(rt, "V0")
        rt.draw()
        got[mode] = rt.counters()
        rt.deinit()
    for k in ("rays", "bricks_entered", "voxel_steps", "hits"):
        assert got[1][k] == got[2][k], k
    assert got[2]["grid_steps"] < got[1]["grid_steps"]
    assert got[2]["status_loads"] <= got[1]["status_loads"]


def _frames_with_flags(name, views, tuning_flags, kernel_names=None, **overrides):
    """Whole RGBA8 frames of a workload with vrt_config.tuning_flags set (VRT_TUNE_*: the library reads no environment)."""
    w = W.WORKLOADS[name] if isinstance(name, str) else name
    grid = _GRIDS.setdefault(w.name, W.build_grid(w))
    rt = W.make_renderer(w, grid, tuning_flags=tuning_flags, **overrides)
    out = []
    # (a first frame and a wait: the library learns the box of the occupied cells behind the scene upload without ever waiting for
    # it, and from then on may trace bounce frames of a scene that fills its grid with the counter-free walk loop)
    W.set_view(rt, views[0])
    rt.draw()
    rt.wait()
    for v in views:
        W.set_view(rt, v)
        rt.draw()
        out.append(rt.read_rgba8().copy())
        if kernel_names is not None:
            kernel_names.append(rt.kernel_name())
    rt.deinit()
    return out


_GRIDS = {}


@pytest.mark.parametrize("name", ["cfg2_1080p_512c_b8", "cfg3_4k_1024c_b8"])
def test_skip_to_the_box_changes_no_pixel_at_full_size(name):
    """Size-independent property of the jump to the occupied-cell box (skip_to_box): whole full-size frames with the jump and
    with every cell walked (VRT_TUNE_NO_SKIP_TO_BOX) are the same bytes — from inside the box, above it, and far outside."""
    views = ["V0", "V1", "V2", "V1x"]
    walked = _frames_with_flags(name, views, L.TUNE_NO_SKIP_TO_BOX)
    jumped = _frames_with_flags(name, views, 0)
    for v, a, b in zip(views, walked, jumped):
        assert np.array_equal(a, b), v
        assert a[..., :3].any()


def test_path_kernel_memory_layouts_change_no_pixel():
    """vrt_path_kernel on a scene of the path-trace configuration's kind (sparse 1024^3, 8^3 bricks, 4 samples, 3 bounces) at 720p:
    bricks staged in LDS or read through the L1, walk loop on half-block words (dilated or linear cell index) or on the linear words,
    jump to the box or not,
    brick bits reached by cell or through brick_index, start index multiplied or looked up — all the same frame; so is the
    lockstep kernel's."""
    w = W.Workload("path_layouts", 1280, 720, 1024, 8, 4, 2, True, 5.0, "sparse", 0.08, 200000)
    views = ["V0", "V1x"]
    path = 1 << 23
    names = []
    base = _frames_with_flags(w, views, 0, kernel_names=names, kernel_variant=path)
    # the spheres reach the grid's faces: the walk ends at the grid's face — round 4: a pool of rays per wave (vrt_pool_kernel);
    # without it vrt_path_kernel's dilated-index walk without steps-left counters; with the second flag, the one with them
    assert set(names) == {"vrt_pool_kernel<8, 6, 60, 2>"}, names
    names = []
    for v, a, b in zip(views, base, _frames_with_flags(w, views, L.TUNE_NO_PATH_POOL, kernel_names=names, kernel_variant=path)):
        assert np.array_equal(a, b), ("a ray per lane", v)
    assert set(names) == {"vrt_path_kernel<8, 5, false, false, false, false, 2>"}, names
    names = []
    _frames_with_flags(w, views[:1], L.TUNE_NO_PATH_GRID_EXIT, kernel_names=names, kernel_variant=path)
    assert set(names) == {"vrt_path_kernel<8, 5, false, false, false, false, 1>"}, names
    for flags in (L.TUNE_NO_PATH_GRID_EXIT, L.TUNE_NO_PATH_GRID_EXIT | L.TUNE_NO_SKIP_TO_BOX | L.TUNE_NO_PATH_BRICK_LDS,
                  L.TUNE_NO_PATH_DILATED, L.TUNE_NO_PATH_DILATED | L.TUNE_NO_SKIP_TO_BOX | L.TUNE_NO_CELL_OCCUPANCY, L.TUNE_NO_SKIP_TO_BOX,
                  L.TUNE_NO_PATH_BRICK_LDS, L.TUNE_NO_PATH_HALFBLOCKS, L.TUNE_PATH_EAGER_START, L.TUNE_NO_CELL_OCCUPANCY, L.TUNE_NO_START_SHORTCUT,
                  L.TUNE_NO_PATH_POOL | L.TUNE_NO_CELL_OCCUPANCY, L.TUNE_NO_PATH_POOL | L.TUNE_NO_SKIP_TO_BOX | L.TUNE_NO_START_SHORTCUT,
                  L.TUNE_NO_CELL_OCCUPANCY | L.TUNE_NO_START_SHORTCUT | L.TUNE_PATH_EAGER_START,
                  L.TUNE_NO_PATH_BRICK_LDS | L.TUNE_NO_PATH_HALFBLOCKS | L.TUNE_NO_SKIP_TO_BOX):
        for v, a, b in zip(views, base, _frames_with_flags(w, views, flags, kernel_variant=path)):
            assert np.array_equal(a, b), (flags, v)
    for v, a, b in zip(views, base, _frames_with_flags(w, views, 0, kernel_variant=1 << 21)):
        assert np.array_equal(a, b), ("lockstep", v)
    if dev_library_or_none():  # the two-trips-ahead walk loop lives in the development build
        for flags in (L.TUNE_PATH_AHEAD, L.TUNE_PATH_AHEAD | L.TUNE_NO_CELL_OCCUPANCY | L.TUNE_NO_PATH_BRICK_LDS | L.TUNE_NO_SKIP_TO_BOX):
            for v, a, b in zip(views, base, _frames_with_flags(w, views, flags, kernel_variant=path, library=dev_library_or_none())):
                assert np.array_equal(a, b), ("two trips ahead", flags, v)
    if dev_library_or_none():  # the two-cells-ahead dilated walk, the 4 x 4 x 4-cell words and the distance-field walk live in the development build
        for flags in (L.TUNE_PATH_TWO_AHEAD, L.TUNE_PATH_TWO_AHEAD | L.TUNE_NO_SKIP_TO_BOX | L.TUNE_NO_PATH_BRICK_LDS,
                      L.TUNE_PATH_BLOCKS64, L.TUNE_PATH_BLOCKS64 | L.TUNE_NO_SKIP_TO_BOX | L.TUNE_NO_CELL_OCCUPANCY,
                      L.TUNE_PATH_DISTANCE, L.TUNE_PATH_DISTANCE | L.TUNE_NO_SKIP_TO_BOX | L.TUNE_NO_PATH_BRICK_LDS):
            for v, a, b in zip(views, base, _frames_with_flags(w, views, flags, kernel_variant=path, library=dev_library_or_none())):
                assert np.array_equal(a, b), ("distance field", flags, v)
    if dev_library_or_none():  # the block-skipping walk lives in the development build
        for v, a, b in zip(views, base, _frames_with_flags(w, views, 0, **variant_kwargs(path | (1 << 22)))):
            assert np.array_equal(a, b), ("block-skipping walk", v)
    assert base[0][..., :3].any()
