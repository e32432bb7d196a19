"""BASELINE.json configurations at their FULL sizes: the HIP frame is rendered whole, the oracle
renders a seeded random sample of its pixels (any pixel is independent), compared bit-for-bit; plus
size-independent properties (ray count bounds, alpha, determinism across launches and across shards)."""
import numpy as np
import pytest

from tests.helpers import O, oracle_scene_from_grid
from zig_vulkan_amd import workloads as W

pytestmark = pytest.mark.gpu


def _sampled_parity(name, view, n_pixels, seed=1234, **overrides):
    w = W.WORKLOADS[name]
    grid = W.build_grid(w)
    rt = W.make_renderer(w, grid, want_float_output=True, enable_counters=True, **overrides)
    W.set_view(rt, view)
    rt.draw()
    f = rt.read_rgba32f()
    u = rt.read_rgba8()
    c1 = rt.counters()
    rt.draw()  # determinism: a second launch gives the same bytes
    assert np.array_equal(rt.read_rgba8(), u)
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
    _sampled_parity("cfg4_4k_2048c_b8_sparse", "V1", 3000)


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
    frame after 70 more are the same bytes, for all three views; on V1 the settled launch really contains split tiles
    (more waves than 4 per tile)."""
    import hashlib
    w = W.WORKLOADS[W.HEADLINE]
    grid = W.build_grid(w)
    fresh = {}
    for view in ["V0", "V1", "V2"]:
        rt = W.make_renderer(w, grid)
        W.set_view(rt, view)
        rt.draw()
        fresh[view] = hashlib.sha1(rt.read_rgba8().tobytes()).hexdigest()
        rt.deinit()
    rt = W.make_renderer(w, grid)
    tiles = rt.shard_info().owned_tiles
    for view in ["V1", "V0", "V2", "V1"]:
        W.set_view(rt, view)
        rt.draw(frames=70)
        rt.draw()
        assert hashlib.sha1(rt.read_rgba8().tobytes()).hexdigest() == fresh[view], view
    waves = len(rt.wave_timeline())
    assert waves > tiles * 4, "no tile was split on V1"
    assert hashlib.sha1(rt.read_rgba8().tobytes()).hexdigest() == fresh["V1"]
    rt.deinit()
