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


FULL = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "full", "cfg2_*.npz")))


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


# ---- round 5: whole-frame digests of BASELINE configs[1], [3] and [4] (tests/golden/make_full_golden.py) ----
WHOLE = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "full", "cfg[134]_*.npz")))


def test_whole_frame_fixtures_exist():
    names = {os.path.basename(p)[:-4] for p in WHOLE}
    assert {"cfg1_V0", "cfg1_V1", "cfg1_V2", "cfg3_V1", "cfg3_V2", "cfg4_V1", "cfg4_V1x"} <= names
    for p in WHOLE:
        z = np.load(p)
        assert "oracle/vrt_oracle.c" in str(z["provenance"]) and len(z["band_sha256"]) == (int(z["size"][1]) + 15) // 16


@pytest.mark.parametrize("path", [p for p in WHOLE if "cfg4_" not in p or os.environ.get("VRT_SLOW_TESTS")], ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_reproduces_whole_frame_digests(path):
    """The oracle built here renders the WHOLE frame again (configs[1]: 1 s, configs[3]: 5 s on 8 cores; configs[4] — 0.5 G rays,
    minutes — only with VRT_SLOW_TESTS=1) and must reproduce the committed digests: frame hashes, band hashes, crops, counters."""
    from tests.golden.make_full_golden import digest
    from tests.golden.make_golden import scene_digest
    z = np.load(path)
    w = W.WORKLOADS[str(z["workload"])]
    grid = W.build_grid(w)
    assert scene_digest(grid) == str(z["scene_sha256"]), "synthetic scene generator drifted"
    cam, sun = W.camera_for(w, str(z["view"])), W.sun_for(w)
    assert np.array_equal(O.push_constants(cam.blob(), sun.blob()), z["push_constants"])
    f, u, c = O.render(oracle_scene_from_grid(grid), z["push_constants"].copy())
    d = digest(f, u)
    assert str(d["float_sha256"]) == str(z["float_sha256"]) and str(d["rgba8_sha256"]) == str(z["rgba8_sha256"])
    assert [str(x) for x in d["band_sha256"]] == [str(x) for x in z["band_sha256"]]
    assert np.array_equal(d["float_crops"].view(np.uint32), z["float_crops"].view(np.uint32))
    assert [c[k] for k in COUNTER_KEYS] == z["counters"].tolist()


@pytest.mark.parametrize("view", ["V0", "V1", "V2"])
def test_oracle_and_reference_shader_agree_on_configs1_at_full_size(view):
    """configs[1] at 1920x1080: the oracle's whole frame (tests/golden/full) and the REFERENCE SHADER's under llvmpipe
    (tests/golden/ref_full, make_ref_golden.py) are the same bytes — float frame, RGBA8 frame, every band — for the same 128
    push-constant bytes and the same scene."""
    here = os.path.dirname(__file__)
    a, b = np.load(os.path.join(here, "golden", "full", f"cfg1_{view}.npz")), np.load(os.path.join(here, "golden", "ref_full", f"cfg1_{view}.npz"))
    assert a["push_constants"].tobytes() == b["push_constants"].tobytes() and str(a["scene_sha256"]) == str(b["scene_sha256"])
    assert str(a["float_sha256"]) == str(b["float_sha256"]) and str(a["rgba8_sha256"]) == str(b["rgba8_sha256"])
    assert [str(x) for x in a["band_sha256"]] == [str(x) for x in b["band_sha256"]]
    assert "brick_raytracer.comp" in str(b["provenance"])


# ---- round 6: the 4K fixtures of the reference shader on the 1024^3 sparse scene (tests/golden/make_ref_golden.py big) ----
BIG = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ref_full", "big_*.npz")))


@pytest.mark.parametrize("path", BIG, ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_reproduces_the_reference_shaders_4k_frames(path):
    """The oracle against the REFERENCE SHADER's 4K frames that pin configs[3]'s and [4]'s kernels (GPU side:
    tests/test_reference_parity_gpu.py): the two-sample shadow frames whole (float + RGBA8 hashes, every band), the four-sample
    three-bounce frames on sixteen bands of sixteen rows spread over the frame (the whole frame is 0.1 G rays: VRT_SLOW_TESTS=1)."""
    from tests.golden.make_golden import scene_digest
    z = np.load(path)
    w = W.WORKLOADS[str(z["workload"])]
    grid = W.build_grid(w)
    assert scene_digest(grid) == str(z["scene_sha256"]), "synthetic scene generator drifted"
    assert "brick_raytracer.comp" in str(z["provenance"])
    scene = oracle_scene_from_grid(grid)
    height, band = int(z["size"][1]), int(z["band_rows"])
    whole = int(z["max_bounce"]) == 0 or bool(os.environ.get("VRT_SLOW_TESTS"))
    if whole:
        f, u, _ = O.render(scene, z["push_constants"].copy())
        assert hashlib.sha256(f.tobytes()).hexdigest() == str(z["float_sha256"]) and hashlib.sha256(u.tobytes()).hexdigest() == str(z["rgba8_sha256"])
        return
    nb = (height + band - 1) // band
    for i in range(2, nb, max(1, nb // 16)):
        f, _, _ = O.render(scene, z["push_constants"].copy(), rows=(i * band, min(height, (i + 1) * band)))
        rows = f[i * band:(i + 1) * band] if f.shape[0] == height else f
        assert hashlib.sha256(np.ascontiguousarray(rows).tobytes()).hexdigest() == str(z["band_sha256"][i]), f"band {i}"
