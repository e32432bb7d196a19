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


# ---- committed camera states of the path (tests/golden/benchmark/benchmark_path.npz, made by tests/golden/make_benchmark_golden.py) ----
import os

import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "benchmark", "benchmark_path.npz")
PATH_POINTS = np.array([[0, 0, 0], [2, 5, 0], [3, 5, 5], [5, 2, 1], [10, 0, 10], [20, -20, 20], [10, -25, 15], [10, -22, 20], [10, -30, 25],
                        [5, -10, 10], [0, 13, 0]], dtype=np.float64)                                  # Benchmark.zig:146-158
PATH_EULER = np.array([[0, 0, 0], [0, 45, 0], [10, -20, 0], [20, 180, 0], [50, 90, 0], [60, 0, 0], [80, -10, 0], [75, -40, 0], [80, -10, 0],
                       [80, -90, 0], [0, -145, 0]], dtype=np.float64)                                 # Benchmark.zig:159-171


def test_committed_camera_blobs_are_reproduced_byte_for_byte():
    from tests.golden.make_benchmark_golden import key_blobs, sampled_blobs
    z = np.load(GOLDEN)
    assert z["keys"].shape == (11, 128) and z["sampled"].shape == (13, 128)
    assert np.array_equal(key_blobs(), z["keys"])
    frames, sampled, total, _ = sampled_blobs()
    assert frames.tolist() == z["sampled_frames"].tolist() and total == int(z["total_frames"]) == 1801
    assert np.array_equal(sampled, z["sampled"])


def _q_axis(deg, axis):
    r = math.radians(deg) / 2
    return np.array([math.cos(r), *(np.asarray(axis, dtype=np.float64) * math.sin(r))])


def _q_mul(l, r):
    lw, lx, ly, lz = l
    rw, rx, ry, rz = r
    return np.array([lw * rw - lx * rx - ly * ry - lz * rz, lw * rx + lx * rw + ly * rz - lz * ry,
                     lw * ry - lx * rz + ly * rw + lz * rx, lw * rz + lx * ry - ly * rx + lz * rw])


def _q_euler(e):  # zalgebra Quat.fromEulerAngles: z * (y * x)
    return _q_mul(_q_axis(e[2], (0, 0, 1)), _q_mul(_q_axis(e[1], (0, 1, 0)), _q_axis(e[0], (1, 0, 0))))


def _rotate(q, v):  # standard unit-quaternion rotation, written differently from host_benchmark.cpp's q_rotate
    q = q / np.linalg.norm(q)
    w, u = q[0], q[1:]
    return v + 2 * np.cross(u, np.cross(u, v) + w * v)


def _expected_camera(t, width, height):
    """Float64 restatement of Benchmark.update (Benchmark.zig:47-74) + Camera.propogatePitchChange (Camera.zig:167-180)."""
    seg = 60.0 / 11.0
    k = int(math.floor(t / seg))
    a = math.fmod(t, seg) / seg
    origin = PATH_POINTS[k] + (PATH_POINTS[k + 1] - PATH_POINTS[k]) * a
    q = _q_euler(PATH_EULER[k]) * (1 - a) + _q_euler(PATH_EULER[k + 1]) * a     # component-wise lerp, then normalised
    fwd = _rotate(q, np.array([0.0, 0.0, 1.0]))
    right = np.cross([0.0, 1.0, 0.0], fwd)
    right /= np.linalg.norm(right)
    up = np.cross(fwd, right)
    up /= np.linalg.norm(up)
    vh = 2.0 * math.tan(math.radians(75.0) / 2)
    vw = vh * width / height
    h, v = right * vw, up * vh
    return origin, h, v, origin - h / 2 - v / 2 - fwd


def test_key_camera_blobs_match_an_independent_float64_restatement():
    z = np.load(GOLDEN)
    w, h = int(z["width"]), int(z["height"])
    seg = 60.0 / 11.0
    for k in range(10):   # key 10 is past the last segment: the camera keeps its last state (Benchmark.zig:51,59)
        b = z["keys"][k]
        f = np.frombuffer(b.tobytes(), dtype=np.float32)
        assert np.frombuffer(b.tobytes()[:8], dtype=np.uint32).tolist() == [w, h]
        origin, hor, ver, llc = _expected_camera(k * seg + 1e-3, w, h)
        assert np.allclose(f[16:19], origin, atol=2e-5), k
        assert np.allclose(f[4:7], hor, atol=2e-5) and np.allclose(f[8:11], ver, atol=2e-5) and np.allclose(f[12:15], llc, atol=3e-5), k
    # the sampled frames of the 30 fps run: frame n is at t = n / 30 (within float accumulation of the timer)
    for n, b in zip(z["sampled_frames"].tolist(), z["sampled"]):
        if n / 30.0 >= 10 * seg:   # past the last key: frozen at the last interpolated state
            continue
        f = np.frombuffer(b.tobytes(), dtype=np.float32)
        origin, hor, ver, llc = _expected_camera(n / 30.0, w, h)
        assert np.allclose(f[16:19], origin, atol=2e-3) and np.allclose(f[4:7], hor, atol=2e-3) and np.allclose(f[12:15], llc, atol=3e-3), n


def test_report_min_avg_max_over_a_scripted_dt_sequence():
    """Report (Benchmark.zig:80-136): min / max / mean of the delta times handed to update()."""
    cam = Camera(75.0, 320, 200)
    b = Benchmark(cam)
    dts = [0.016, 0.033, 0.008, 0.120, 0.016, 0.050]
    for dt in dts:
        assert b.update(dt) is False
    r = b.report()
    assert r["min_frame_ms"] == pytest.approx(8.0, abs=1e-4)
    assert r["max_frame_ms"] == pytest.approx(120.0, abs=1e-3)
    assert r["avg_frame_ms"] == pytest.approx(1000.0 * sum(dts) / len(dts), abs=1e-3)


@pytest.mark.gpu
def test_flythrough_frames_equal_the_oracle_bit_for_bit():
    """vrt_benchmark_update stepped at 30 fps over the 60 s path; every 150th frame is rendered through vrt_dispatch on a
    320 x 180 frame of a headline-shaped scene (8^3 bricks, primary + shadow ray) and compared with the oracle bit for bit;
    the camera bytes at those frames are the committed ones."""
    import ctypes as C
    from tests.helpers import O, oracle_scene_from_grid
    from zig_vulkan_amd import workloads as W
    z = np.load(GOLDEN)
    w = W.Workload("fly", int(z["width"]), int(z["height"]), 128, 8, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    scene = oracle_scene_from_grid(grid)
    rt = W.make_renderer(w, grid, want_float_output=True)
    bench = rt.create_benchmark()
    golden = dict(zip(z["sampled_frames"].tolist(), z["sampled"]))
    frame, done, checked, hits = 0, False, 0, 0
    while True:
        if frame in golden:
            pc = O.push_constants(rt.camera.blob(), rt.sun.blob())
            assert np.array_equal(pc[:96], golden[frame][:96]), f"camera bytes of frame {frame} drifted"
            rt.draw()
            f, u = rt.read_rgba32f(), rt.read_rgba8()
            fo, uo, co = O.render(scene, pc)
            assert np.array_equal(f.view(np.uint32), fo.view(np.uint32)) and np.array_equal(u, uo), f"frame {frame}"
            checked += 1
            hits += co["hits"]
        if done:
            break
        done = bench.update(1.0 / 30.0)
        frame += 1
    rt.deinit()
    assert checked == 13 and frame == 1801 and hits > 0
    r = bench.report()
    assert r["min_frame_ms"] == pytest.approx(1000.0 / 30.0, abs=1e-3) and r["max_frame_ms"] == pytest.approx(1000.0 / 30.0, abs=1e-3)
