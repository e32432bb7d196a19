"""Host-side mirror of the reference's Camera / Sun / materials (CPU only)."""
import math

import numpy as np
import pytest

from zig_vulkan_amd import Camera, CameraConfig, Sun, SunConfig, default_materials


def f32(x):
    return np.float32(x)


def test_camera_init_matches_camera_zig():
    # Camera.init(75, 1920, 1080, {}), Camera.zig:36-77 with zalgebra up=(0,1,0), forward=(0,0,1)
    cam = Camera(75.0, 1920, 1080, CameraConfig(origin=(1.0, 2.0, 3.0), samples_per_pixel=3, max_bounce=4))
    d = cam.d_camera
    vh = f32(2.0) * f32(math.tan(f32(f32(75.0) * f32(math.pi / 180.0)) * f32(0.5)))
    vw = f32(f32(1920.0) / f32(1080.0)) * vh
    assert (d.image_width, d.image_height) == (1920, 1080)
    assert d.samples_per_pixel == 3
    assert d.max_bounce == 5  # + 1, Camera.zig:74
    assert np.allclose(list(d.horizontal), [vw, 0, 0], rtol=1e-6)   # right = up x forward = (1,0,0)
    assert np.allclose(list(d.vertical), [0, vh, 0], rtol=1e-6)     # up = forward x right = (0,1,0)
    llc = [1.0 - vw / 2, 2.0 - vh / 2, 3.0 - 1.0]                    # origin - h/2 - v/2 - forward
    assert np.allclose(list(d.lower_left_corner), llc, rtol=1e-6)
    assert list(d.origin) == [1.0, 2.0, 3.0]
    assert len(cam.blob()) == 96


def test_camera_forward_change_keeps_orthogonal_basis():
    cam = Camera(75.0, 640, 480)
    cam.look_at((20.0, -20.0, 20.0), (0.0, 8.0, 0.0))
    h, v = np.array(list(cam.d_camera.horizontal)), np.array(list(cam.d_camera.vertical))
    o, llc = np.array(list(cam.d_camera.origin)), np.array(list(cam.d_camera.lower_left_corner))
    fwd = o - h / 2 - v / 2 - llc
    assert abs(np.dot(h, v)) < 1e-5 and abs(np.dot(h, fwd)) < 1e-5 and abs(np.dot(v, fwd)) < 1e-5
    assert abs(np.linalg.norm(fwd) - 1) < 1e-5  # recovered through float32 subtraction of ~20-unit coordinates
    # rays leave along -forward: the centre ray points from origin to the target
    centre = llc + h / 2 + v / 2 - o
    want = np.array([0.0, 8.0, 0.0]) - o
    assert np.allclose(centre / np.linalg.norm(centre), want / np.linalg.norm(want), atol=1e-5)


def test_sun_init_matches_sun_zig():
    s = Sun().device_data  # Sun.Config defaults, Sun.zig:4-11
    assert list(s.position) == [0.0, -1000.0, 0.0]  # static_pos_vec, Sun.zig:41
    assert s.enabled == 1
    assert np.allclose(list(s.color), [1.0, 1.1, 1.0])
    assert s.radius == 5.0
    s2 = Sun(SunConfig(enabled=False, radius=0.0, sun_distance=10.0)).device_data
    assert s2.enabled == 0 and list(s2.position) == [0.0, -10.0, 0.0]
    assert len(Sun().blob()) == 32


def test_default_materials_are_the_terrain_table():
    m = default_materials(256)  # terrain.zig:130-196
    assert m.shape == (256,) and m.dtype.itemsize == 20
    assert m["type"][:8].tolist() == [2, 0, 0, 0, 0, 0, 0, 1]
    assert np.isclose(m["type_data"][0], 1.333) and np.isclose(m["type_data"][7], 0.45)
    assert np.allclose([m["albedo_r"][0], m["albedo_g"][0], m["albedo_b"][0]], [0.117, 0.45, 0.85])
    assert not m.view(np.uint8).reshape(256, 20)[8:].any()
