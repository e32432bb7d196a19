"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py): the oracle built here
must reproduce them bit-for-bit (CPU); the HIP path must too (gpu)."""
import glob
import hashlib
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import oracle_scene_from_grid
from zig_vulkan_amd import workloads as W

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
COUNTER_KEYS = ("rays", "status_loads", "bricks_entered", "voxel_steps", "hits", "grid_steps")


def _load(path):
    z = np.load(path)
    wl = z["workload"].tolist()
    w = W.Workload(os.path.basename(path), wl[0], wl[1], wl[2], wl[3], wl[4], wl[5], bool(wl[6]), float(z["sun_radius"]))
    return z, w


def _check(z, f, u, counters):
    assert np.array_equal(u, z["rgba8"])
    assert hashlib.sha256(np.ascontiguousarray(f).tobytes()).hexdigest() == str(z["float_sha256"])
    y0, x0 = z["crop_origin"].tolist()
    crop = z["float_crop"]
    assert np.array_equal(f[y0:y0 + crop.shape[0], x0:x0 + crop.shape[1]].view(np.uint32), crop.view(np.uint32))
    if counters is not None:
        assert [counters[k] for k in COUNTER_KEYS] == z["counters"].tolist()


def test_fixtures_exist():
    assert len(GOLDEN) >= 5


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    z, w = _load(path)
    grid = W.build_grid(w)
    from tests.golden.make_golden import scene_digest
    assert scene_digest(grid) == str(z["scene_sha256"]), "synthetic scene generator drifted"
    # the camera/sun builders must still produce the committed push-constant bytes
    cam, sun = W.camera_for(w, str(z["view"])), W.sun_for(w)
    assert np.array_equal(O.push_constants(cam.blob(), sun.blob()), z["push_constants"])
    f, u, c = O.render(oracle_scene_from_grid(grid), z["push_constants"].copy())
    _check(z, f, u, c)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_reproduces_golden(path):
    import ctypes as C
    z, w = _load(path)
    grid = W.build_grid(w)
    rt = W.make_renderer(w, grid, want_float_output=True, enable_counters=True)
    pc = z["push_constants"].tobytes()
    C.memmove(C.byref(rt.camera.d_camera), pc[:96], 96)   # drive the kernel with the committed bytes
    C.memmove(C.byref(rt.sun.device_data), pc[96:], 32)
    rt.draw()
    f, u, c = rt.read_rgba32f(), rt.read_rgba8(), rt.counters()
    rt.deinit()
    _check(z, f, u, c)


FULL = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "full", "*.npz")))


@pytest.mark.parametrize("path", FULL, ids=[os.path.basename(p)[:-4] for p in FULL])
def test_oracle_reproduces_fullsize_headline_golden(path):
    """tests/golden/full: the headline workload at its full 1920x1080 (hard sun), whole-frame SHA-256 + crops.  The GPU
    side is tests/test_fullsize_gpu.py::test_headline_settled_frame_is_the_committed_golden_frame."""
    z = np.load(path)
    w = W.WORKLOADS[str(z["workload"])]
    grid = W.build_grid(w)
    from tests.golden.make_golden import scene_digest
    assert scene_digest(grid) == str(z["scene_sha256"]), "synthetic scene generator drifted"
    cam, sun = W.camera_for(w, str(z["view"])), W.sun_for(w, 0.0)
    assert np.array_equal(O.push_constants(cam.blob(), sun.blob()), z["push_constants"])
    f, u, c = O.render(oracle_scene_from_grid(grid), z["push_constants"].copy())
    assert hashlib.sha256(u.tobytes()).hexdigest() == str(z["rgba8_sha256"])
    assert hashlib.sha256(f.tobytes()).hexdigest() == str(z["float_sha256"])
    assert [c[k] for k in COUNTER_KEYS] == z["counters"].tolist()


def test_fullsize_fixtures_exist():
    assert len(FULL) == 4
