"""Product grid builder (C++ mirror of Grid.zig) against the pure-Python restatement
(oracle/grid_oracle.py): exact bytes of the five arrays, State.Device and the delta ranges."""
import numpy as np
import pytest

from oracle.grid_oracle import GridOracle
from zig_vulkan_amd import BrickGrid
from zig_vulkan_amd import _lib as L

IDS = {"statuses": L.BUF_BRICK_STATUS, "indices": L.BUF_BRICK_INDEX, "occupancy": L.BUF_BRICK_OCCUPANCY,
       "start_indices": L.BUF_BRICK_START_INDEX, "material_indices": L.BUF_MATERIAL_INDEX}


def _both(dims, b, **kw):
    return BrickGrid(*dims, brick_dimension=b, **kw), GridOracle(*dims, brick_dimension=b, **kw)


def _assert_same(g: BrickGrid, o: GridOracle):
    assert bytes(g.device_state) == o.device_state_bytes()
    assert np.array_equal(g.array(L.BUF_BRICK_STATUS), o.brick_statuses)
    assert np.array_equal(g.array(L.BUF_BRICK_INDEX), o.brick_indices)
    assert np.array_equal(g.array(L.BUF_BRICK_OCCUPANCY), o.brick_occupancy)
    assert np.array_equal(g.array(L.BUF_BRICK_START_INDEX), o.brick_start_indices)
    assert np.array_equal(g.array(L.BUF_MATERIAL_INDEX), o.material_indices)
    assert g.active_bricks == o.active_bricks
    for name, bid in IDS.items():
        active, a, b = g.delta(bid)
        d = o.deltas[name]
        assert active == (d.state == "active")
        assert (a, b) == (d.from_, d.to), name


@pytest.mark.parametrize("b", [4, 8])
def test_random_inserts_match_reference_semantics(b):
    rng = np.random.default_rng(7 + b)
    dims = (5, 3, 4)
    g, o = _both(dims, b, min_point=(-1.5, 2.0, 0.25), scale=0.5)
    vx, vy, vz = dims[0] * b, dims[1] * b, dims[2] * b
    for _ in range(1500):
        x, y, z, m = int(rng.integers(vx)), int(rng.integers(vy)), int(rng.integers(vz)), int(rng.integers(256))
        g.insert(x, y, z, m)
        o.insert(x, y, z, m)
    _assert_same(g, o)


def test_single_insert_layout_by_hand():
    """One voxel, everything derivable by hand (Grid.zig:129-194): y flip, x-z-y cell order,
    bit v%8 of byte v/8, bump-allocated start index."""
    g = BrickGrid(2, 3, 4, brick_dimension=4)
    g.insert(5, 2, 9, 77)  # voxel dims 8 x 12 x 16
    fy = 12 - 1 - 2  # = 9
    cell = (5 // 4) + 2 * ((9 // 4) + 4 * (fy // 4))  # x + dim_x*(z + dim_z*y) = 1 + 2*(2 + 4*2) = 21
    st = g.array(L.BUF_BRICK_STATUS)
    assert st[cell // 32] == 1 << (cell % 32) and st.sum() == st[cell // 32]
    assert g.array(L.BUF_BRICK_INDEX)[cell] == 0 and g.active_bricks == 1
    v = (5 % 4) + 4 * ((9 % 4) + 4 * (fy % 4))  # 1 + 4*(1 + 4*1) = 21
    occ = g.array(L.BUF_BRICK_OCCUPANCY)
    assert occ[v // 8] == 1 << (v % 8) and occ.sum() == occ[v // 8]
    start = g.array(L.BUF_BRICK_START_INDEX)
    assert start[0] == 0 and (start[1:] == 0xFFFFFFFF).all()
    mi = g.array(L.BUF_MATERIAL_INDEX)
    assert mi[v] == 77 and np.count_nonzero(mi) == 1
    # a second brick gets the next 64-entry block
    g.insert(0, 11, 0, 5)  # flipped y = 0 -> cell 0
    assert g.array(L.BUF_BRICK_INDEX)[0] == 1
    assert g.array(L.BUF_BRICK_START_INDEX)[1] == 64


def test_device_state_max_point_and_scale():
    g = BrickGrid(128, 64, 128, min_point=(-32.0, -16.0, -32.0), scale=0.5)  # the reference app's grid, main.zig:77-81
    d = g.device_state
    assert (d.voxel_dim_x, d.voxel_dim_y, d.voxel_dim_z) == (512, 256, 512)
    assert list(d.min_point_base_t) == [-32.0, -16.0, -32.0, np.float32(0.01)]
    assert list(d.max_point_scale) == [32.0, 16.0, 32.0, 0.5]


def test_first_delta_starts_at_zero_then_resets():
    """DeviceDataDelta.empty has from = 0 (State.zig:15-20): the first dirty range always starts at 0;
    after resetDelta it is tight."""
    g, o = _both((4, 4, 4), 4)
    g.insert(15, 0, 15, 1)
    o.insert(15, 0, 15, 1)
    active, a, b = g.delta(L.BUF_BRICK_INDEX)
    assert active and a == 0 and b == o.deltas["indices"].to
    for bid in IDS.values():
        g.reset_delta(bid)
    for d in o.deltas.values():
        d.reset()
    assert g.delta(L.BUF_BRICK_INDEX)[0] is False
    g.insert(15, 1, 15, 2)
    o.insert(15, 1, 15, 2)
    _assert_same(g, o)
    active, a, b = g.delta(L.BUF_BRICK_INDEX)
    assert active and b - a == 1


def test_errors():
    from zig_vulkan_amd._lib import VrtError
    g = BrickGrid(2, 2, 2, brick_dimension=4, brick_alloc=1)
    with pytest.raises(VrtError) as e:
        g.insert(8, 0, 0, 1)  # x == voxel_dim_x: the reference asserts (Grid.zig:130)
    assert e.value.code == L.VRT_E_OUT_OF_RANGE
    g.insert(0, 0, 0, 1)
    with pytest.raises(VrtError) as e:
        g.insert(7, 7, 7, 1)  # second brick but brick_alloc = 1
    assert e.value.code == L.VRT_E_OOM
    with pytest.raises(VrtError):
        BrickGrid(0, 1, 1)
    with pytest.raises(VrtError):
        BrickGrid(1, 1, 1, brick_dimension=3)


def test_bulk_insert_equals_single_inserts():
    rng = np.random.default_rng(3)
    xyz = rng.integers(0, 32, size=(4000, 3)).astype(np.uint32)
    mats = rng.integers(0, 256, size=4000).astype(np.uint8)
    a = BrickGrid(4, 4, 4, brick_dimension=8)
    b = BrickGrid(4, 4, 4, brick_dimension=8)
    a.insert_many(xyz, mats)
    for (x, y, z), m in zip(xyz.tolist(), mats.tolist()):
        b.insert(x, y, z, m)
    for bid in IDS.values():
        assert np.array_equal(a.array(bid), b.array(bid))
        assert a.delta(bid) == b.delta(bid)


def test_synth_terrain_is_deterministic_and_fits_dense_alloc():
    a = BrickGrid(16, 16, 16, min_point=(-32, -32, -32), scale=4.0, brick_dimension=4)
    b = BrickGrid(16, 16, 16, min_point=(-32, -32, -32), scale=4.0, brick_dimension=4)
    a.synth_terrain(420)
    b.synth_terrain(420)
    for bid in IDS.values():
        assert np.array_equal(a.array(bid), b.array(bid))
    assert 0 < a.active_bricks < 16 ** 3
    occ = a.array(L.BUF_BRICK_OCCUPANCY)
    mats = a.array(L.BUF_MATERIAL_INDEX)
    assert set(np.unique(mats).tolist()) <= set(range(8))
    assert np.unpackbits(occ).sum() > 1000
