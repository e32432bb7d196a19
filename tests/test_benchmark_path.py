"""Benchmark fly-through: hand-derived checks of the path restatement
(src/modules/voxel_rt/Benchmark.zig:22-74,141-172) and of the zalgebra quaternion helpers it needs."""
import math

import numpy as np

from zig_vulkan_amd import Camera, CameraConfig
from zig_vulkan_amd.voxel_rt import Benchmark


def _forward(cam):
    d = cam.d_camera
    o, h, v, llc = (np.array(list(x)) for x in (d.origin, d.horizontal, d.vertical, d.lower_left_corner))
    return o - h / 2 - v / 2 - llc


def test_start_state_and_first_segment():
    cam = Camera(75.0, 640, 360, CameraConfig(origin=(9.0, 9.0, 9.0)))
    b = Benchmark(cam)
    assert list(cam.d_camera.origin) == [0.0, 0.0, 0.0]              # path_points[0]
    assert np.allclose(_forward(cam), [0, 0, 1], atol=1e-6)           # identity orientation
    seg = 60.0 / 11.0
    assert b.update(seg / 2) is False
    assert np.allclose(list(cam.d_camera.origin), [1.0, 2.5, 0.0], atol=1e-5)  # halfway (0,0,0) -> (2,5,0)
    # halfway between identity and yaw 45 deg by component-wise lerp (not slerp), then normalised:
    q0 = np.array([1.0, 0, 0, 0])
    q1 = np.array([math.cos(math.radians(22.5)), 0, math.sin(math.radians(22.5)), 0])
    q = (q0 + q1) / 2
    q /= np.linalg.norm(q)
    ang = 2 * math.atan2(q[2], q[0])  # rotation about +y
    assert np.allclose(_forward(cam), [math.sin(ang), 0, math.cos(ang)], atol=1e-5)
    h, v = np.array(list(cam.d_camera.horizontal)), np.array(list(cam.d_camera.vertical))
    assert abs(np.dot(h, v)) < 1e-5 and abs(np.dot(h, _forward(cam))) < 1e-5


def test_key_orientations_and_completion():
    cam = Camera(75.0, 320, 200)
    b = Benchmark(cam)
    seg = 60.0 / 11.0
    done = b.update(3 * seg + 1e-4)  # just past key 3: euler (20, 180, 0) -> yaw 180 then pitch 20 about x
    assert not done
    assert np.allclose(list(cam.d_camera.origin), [5, 2, 1], atol=1e-2)
    f = _forward(cam)
    # q = qy(180) * qx(20): rotate (0,0,1) about x by 20 deg -> (0,-sin20,cos20), then about y by 180 -> (0,-sin20,-cos20)
    assert np.allclose(f, [0, -math.sin(math.radians(20)), -math.cos(math.radians(20))], atol=2e-3)
    frames = 1
    while not b.update(0.25):
        frames += 1
        assert frames < 400
    r = b.report()
    assert abs(r["max_frame_ms"] - (3 * seg + 1e-4) * 1000) < 1 and abs(r["min_frame_ms"] - 250) < 1e-3
    # after the last key the camera stays at the last interpolated state (index guard, Benchmark.zig:51,59)
    assert np.allclose(list(cam.d_camera.origin), [0, 13, 0], atol=1.0)  # frozen within one 0.25 s step of the last key
