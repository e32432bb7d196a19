"""oracle/vrt_oracle.c against the independent literal restatement in tests/literal_port.py: the same random
rays through random small brick grids must give the same hit flag, distance, point, normal and material index,
bit for bit.  (Neither is the reference — the reference cannot be built here — but the two restatements share no
code: the C one is what every parity test of the kernel leans on.)"""
import numpy as np
import pytest

from tests import literal_port as LP
from tests.helpers import O, oracle_scene_from_grid, push_for
from zig_vulkan_amd import BrickGrid, Camera, Sun, default_materials
from zig_vulkan_amd import _lib as L


def _scenes():
    rng = np.random.default_rng(2024)
    for b, dims, scale, min_point in [(4, (4, 3, 5), 1.0, (-2.0, -1.5, -2.5)), (8, (3, 3, 2), 0.75, (0.25, -1.0, 3.0)), (4, (2, 2, 2), 3.0, (-3.0, -3.0, -3.0))]:
        grid = BrickGrid(*dims, min_point=min_point, scale=scale, brick_dimension=b)
        n = int(0.06 * np.prod(dims) * b ** 3)
        xyz = np.stack([rng.integers(0, b * d, n) for d in dims], axis=-1)
        grid.insert_many(xyz, rng.integers(0, 12, n))
        mats = default_materials(256)
        mats[9] = (3, 0.5, 0.5, 0.5, 1.0)    # type MAT_NONE with type_data 1: ignored by rays from CreateRay (comp:427)
        mats[10] = (2, 0.9, 0.9, 1.0, 1.52)  # glass
        mats[11] = (2, 0.9, 0.9, 1.0, 1.0)
        yield grid, mats, rng


def _literal_scene(grid, mats):
    st = grid.device_state
    return LP.Scene([st.min_point_base_t[i] for i in range(3)], [st.max_point_scale[i] for i in range(3)], st.max_point_scale[3],
                    (st.dim_x, st.dim_y, st.dim_z), grid.brick_dimension, grid.array(L.BUF_BRICK_STATUS), grid.array(L.BUF_BRICK_INDEX),
                    grid.array(L.BUF_BRICK_OCCUPANCY), grid.array(L.BUF_BRICK_START_INDEX), grid.array(L.BUF_MATERIAL_INDEX), mats)


def _rays(grid, rng, n):
    st = grid.device_state
    lo = np.array([st.min_point_base_t[i] for i in range(3)], dtype=np.float64)
    hi = np.array([st.max_point_scale[i] for i in range(3)], dtype=np.float64)
    size = hi - lo
    for i in range(n):
        kind = i % 5
        target = lo + rng.random(3) * size
        if kind == 0:    # from outside towards the box
            origin = lo + (rng.random(3) * 3.0 - 1.0) * size
        elif kind == 1:  # from inside
            origin = lo + rng.random(3) * size
        elif kind == 2:  # axis-aligned (two zero direction components), on a cell boundary
            origin = lo + np.floor(rng.random(3) * 4) * size / 4 + np.array([0.0, -2.0 * size[1], 0.0])
            target = origin + np.array([0.0, 1.0, 0.0])
        elif kind == 3:  # one zero component
            origin = lo + (rng.random(3) * 3.0 - 1.0) * size
            target[rng.integers(0, 3)] = origin[rng.integers(0, 3)]
        else:            # grazing along a face of the box
            origin = lo + (rng.random(3) * 3.0 - 1.0) * size
            origin[1] = lo[1] + 1e-4
            target[1] = lo[1] + 2e-4
        d = (target - origin).astype(np.float32)
        if not np.any(d):
            d = np.array([0.3, -0.2, 0.9], dtype=np.float32)
        if kind != 4:
            d = (d / np.float32(np.sqrt(np.float32((d * d).sum())))).astype(np.float32)
        yield origin.astype(np.float32), d


@pytest.mark.parametrize("case", [0, 1, 2])
def test_c_oracle_equals_literal_restatement(case):
    grid, mats, rng = list(_scenes())[case]
    scene = oracle_scene_from_grid(grid, mats)
    pc = push_for(Camera(75.0, 16, 16), Sun())
    lit = _literal_scene(grid, mats)
    hits = 0
    for k, (origin, d) in enumerate(_rays(grid, rng, 400)):
        ignore, refl = [(3, 1.0), (2, 1.52), (2, 1.0), (3, 0.5)][k % 4]
        ok_c, point_c, normal_c, t_c, index_c = O.grid_hit_raw(scene, pc, origin, d, ignore, refl)
        ok_l, rec = LP.grid_hit(lit, origin, d, ignore, refl)
        assert ok_c == ok_l, f"ray {k}: hit flag {ok_c} vs {ok_l}"
        if ok_c:
            hits += 1
            assert np.float32(t_c).view(np.uint32) == np.float32(rec["t"]).view(np.uint32), f"ray {k}: t {t_c} vs {rec['t']}"
            assert np.array_equal(point_c.view(np.uint32), np.array(rec["point"], dtype=np.float32).view(np.uint32)), f"ray {k}: point"
            assert np.array_equal(normal_c, np.array(rec["normal"], dtype=np.float32)), f"ray {k}: normal"
            assert index_c == rec["index"], f"ray {k}: material index"
    assert hits > 40


@pytest.mark.parametrize("sun_on", [True, False])
def test_c_oracle_pixels_equal_literal_restatement(sun_on):
    """Whole deterministic pixels (ray generation, primary + shadow traversal, background, tone-map, gamma, UNORM8)."""
    from zig_vulkan_amd import CameraConfig, SunConfig
    grid, mats, rng = list(_scenes())[0]
    mats[9] = (7, 0.9, 0.2, 0.9, 1.0)  # an unknown material type: background is added on top (comp:235-238)
    scene = oracle_scene_from_grid(grid, mats)
    w, h = 48, 30
    cam = Camera(75.0, w, h, CameraConfig(samples_per_pixel=1, max_bounce=0))
    cam.look_at((2.6, -2.4, 3.4), (0.0, 0.2, 0.0))
    sun = Sun(SunConfig(enabled=sun_on, radius=0.0))
    pc = push_for(cam, sun)
    d, sd = cam.d_camera, sun.device_data
    cam_fields = {"image_width": d.image_width, "image_height": d.image_height, "max_bounce": d.max_bounce,
                  "horizontal": [np.float32(x) for x in d.horizontal[:3]], "vertical": [np.float32(x) for x in d.vertical[:3]],
                  "lower_left_corner": [np.float32(x) for x in d.lower_left_corner[:3]], "origin": [np.float32(x) for x in d.origin[:3]]}
    sun_fields = {"position": list(sd.position[:3]), "enabled": sd.enabled, "color": list(sd.color[:3])}
    lit = _literal_scene(grid, mats)
    xy = np.stack([rng.integers(0, w, 160), rng.integers(0, h, 160)], axis=-1).astype(np.int32)
    fo, uo, co = O.render_pixels(scene, pc, xy)
    for k, (x, y) in enumerate(xy):
        rgb, rgba8 = LP.pixel(lit, cam_fields, sun_fields, int(x), int(y))
        assert np.array_equal(np.array(rgb, dtype=np.float32).view(np.uint32), fo[k, :3].view(np.uint32)), f"pixel {x},{y}: {rgb} vs {fo[k]}"
        assert rgba8 == [int(c) for c in uo[k]], f"pixel {x},{y}"
    assert co["hits"] > 20
