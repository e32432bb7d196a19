"""Pins the oracle against the reference's OWN shader.

tests/golden/ref/*.npz hold frames that /root/reference/assets/shaders/brick_raytracer.comp produced when Mesa
23.2.1 compiled it and llvmpipe ran it (oracle/_ref; made by tests/golden/make_ref_golden.py).  Three layers:

  1. fixtures (always run, CPU): the oracle, oracle/vrt_oracle.c — its GLSL built-ins lowered as llvmpipe lowers them
     (fma unfused, dot from the last channel, two algebraic rewrites in hash12) — reproduces every
     fixture BIT FOR BIT — float colour and RGBA8.  That checks every statement of the restatement: ray
     generation, sample jitter, both DDA levels, shadow rays, soft sun, all scatter functions, the RNG, tone-map.
     Since round 3 this IS the arithmetic contract of the HIP kernels (vrt_math.h): kernel == oracle == reference shader.
  2. the fused build of the oracle ("fused" / "hw" lowering: fused fma, dot as an fma chain — what a GPU driver would emit, and
     libvrt_hip_fused.so's counterpart; sin is gallivm's in both builds) against the same fixtures within north_star's 1e-4 per
     channel.  Two conforming GLSL implementations differ in the last bits of fma / dot; a last-bit difference flips a DDA tie or
     a hit / miss at isolated pixels, and flips the sin-hash RNG wholesale.  So: deterministic fixtures must agree within
     1e-4 on all but a stated handful of pixels; stochastic ones (soft sun, bounces) are compared as images
     (mean colour), never pixel by pixel.
  3. live (only where oracle/_ref can run: this container, or a box holding the program binaries): the runner
     still reproduces the fixtures; random scenes agree bit for bit; the lowering rules assumed in (1) are
     measured with small GLSL probes of our own.
The -m gpu tests replay the fixtures' inputs through libvrt_hip.so.
"""
import glob
import hashlib
import os

import numpy as np
import pytest

from oracle import oracle as O
from oracle import ref_gl

_ALL = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ref", "*.npz")))
FIXTURES = [p for p in _ALL if not os.path.basename(p).startswith("present_")]
PRESENT = [p for p in _ALL if os.path.basename(p).startswith("present_")]
IDS = [os.path.basename(p)[:-4] for p in FIXTURES]
TOL = 1e-4                     # north_star: "within 1e-4 per channel"
# fixtures whose pixels do not depend on sin(): primary rays, hard-sun shadow rays, 1 sample
DETERMINISTIC = ("cfg0_", "shadow_b8_r0_")
MAX_FLIPPED_FRACTION = 2e-4    # deterministic fixtures: pixels allowed to differ by more than TOL (DDA ties, hit/miss)


def _scene(z) -> O.OracleScene:
    mats = np.frombuffer(z["materials"].tobytes(), dtype=np.dtype([("type", "<u4"), ("r", "<f4"), ("g", "<f4"), ("b", "<f4"), ("d", "<f4")]))
    return O.OracleScene(z["grid_state"].tobytes(), mats, z["brick_status"], z["brick_index"], z["brick_occupancy"],
                         z["brick_start_index"], z["material_index"], int(z["brick_dimension"]))


def _flipped(f, ref_rgb):
    d = np.abs(f[:, :, :3] - ref_rgb).max(axis=2)
    return d, int((d > TOL).sum())


def test_every_reference_fixture_names_the_one_implementation_that_rendered_it():
    """the product's arithmetic is pinned to ONE GL implementation's lowering of the shader; the fixtures say which, all of
    them the same, and tests/golden/make_ref_golden.py refuses to regenerate them under another without being told to."""
    import glob
    import re
    names = set()
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref*", "*.npz"))):
        prov = str(np.load(f)["provenance"])
        m = re.search(r"Mesa [0-9][^|]*\| llvmpipe \(LLVM [0-9.]+, \d+ bits\)", prov)
        assert m, (f, prov)
        names.add(m.group(0))
    assert len(names) == 1, names


def test_fixtures_present():
    assert len(FIXTURES) >= 11
    for p in FIXTURES:
        assert "brick_raytracer.comp" in str(np.load(p)["provenance"])


@pytest.mark.parametrize("path", FIXTURES, ids=IDS)
def test_restatement_with_llvmpipe_lowering_equals_reference_shader_bit_for_bit(path):
    z = np.load(path)
    pc = z["push_constants"].copy()
    f, u, _ = O.render(_scene(z), pc, lowering="llvmpipe")
    assert np.array_equal(u, z["rgba8"])
    assert np.array_equal(f[:, :, :3].view(np.uint32), z["rgb32f"].view(np.uint32))
    assert np.all(f[:, :, 3] == 1.0)
    assert hashlib.sha256(np.ascontiguousarray(f).tobytes()).hexdigest() == str(z["float_sha256"])


@pytest.mark.parametrize("path", FIXTURES, ids=IDS)
def test_fused_oracle_within_tolerance_of_reference_shader(path):
    z = np.load(path)
    f, u, _ = O.render(_scene(z), z["push_constants"].copy(), lowering="fused")
    _compare_hw_lowering(os.path.basename(path), f, u, z)


def _compare_hw_lowering(name, f, u, z):
    ref = z["rgb32f"]
    d, flipped = _flipped(f, ref)
    n = d.size
    if name.startswith(DETERMINISTIC):
        assert flipped <= max(2, int(MAX_FLIPPED_FRACTION * n)), f"{flipped} of {n} pixels differ by more than {TOL}"
        ok = d <= TOL
        assert float(d[ok].max()) <= 1e-6          # what is not a flip is a last-bit difference
        assert int((u != z["rgba8"]).any(axis=2).sum()) <= flipped
    else:
        # sin-hash RNG: one ulp in its argument gives another random number, so samples are independent draws.
        # Pixels whose path never reaches the RNG still agree; the image as a whole must be the same picture.
        assert float((d <= TOL).mean()) >= 0.5
        assert np.abs(f[:, :, :3].mean(axis=(0, 1)) - ref.mean(axis=(0, 1))).max() <= 4e-3


# ---------------------------------------------------------------------------------------------- live: oracle/_ref
_UNAVAILABLE = ref_gl.available()
live = pytest.mark.skipif(_UNAVAILABLE is not None, reason=f"oracle/_ref cannot run here: {_UNAVAILABLE}")


@live
@pytest.mark.parametrize("path", FIXTURES, ids=IDS)
def test_live_reference_shader_reproduces_fixture(path):
    z = np.load(path)
    ref = ref_gl.ReferenceShader(int(z["brick_dimension"]))
    f, u = ref.render(_scene(z), z["push_constants"].copy())
    assert np.array_equal(u, z["rgba8"])
    assert np.array_equal(f[:, :, :3].view(np.uint32), z["rgb32f"].view(np.uint32))


@live
@pytest.mark.parametrize("seed", range(6))
def test_live_random_scenes_bit_for_bit(seed):
    """Random small grids (odd dimensions, non-power-of-two scale, every material type incl. unknown ones, random
    cameras inside and outside the box, 1-3 samples, 0-2 bounces, sun off / hard / soft): reference shader under
    llvmpipe == restatement with llvmpipe's lowering, every pixel, every bit."""
    from zig_vulkan_amd import BrickGrid, Camera, CameraConfig, Sun, SunConfig, default_materials
    from tests.helpers import oracle_scene_from_grid
    rng = np.random.default_rng(1000 + seed)
    b = int(rng.choice([4, 8]))
    dim = [int(v) for v in rng.integers(2, 7, 3)]
    scale = float(rng.choice([1.0, 0.5, 0.37, 2.3]))
    grid = BrickGrid(*dim, min_point=tuple(float(v) for v in rng.uniform(-3, 1, 3)), scale=scale, brick_dimension=b,
                     brick_alloc=int(rng.integers(max(1, dim[0] * dim[1] * dim[2] // 2), dim[0] * dim[1] * dim[2] + 1)))
    nvox = int(rng.integers(20, 400))
    xyz = np.stack([rng.integers(0, dim[i] * b, nvox) for i in range(3)], 1)
    try:
        grid.insert_many(xyz, rng.integers(0, 12, nvox))
    except Exception:  # brick_alloc exhausted: keep what fitted
        pass
    mats = default_materials(256).copy()
    mats["type"][:12] = rng.integers(0, 5, 12)
    mats["type_data"][:12] = rng.choice([0.0, 0.3, 1.333, 1.5, 1.0], 12)
    scene = oracle_scene_from_grid(grid, mats)
    ref = ref_gl.ReferenceShader(b)
    w, h = 96, 64
    for _ in range(3):
        cam = Camera(75.0, w, h, CameraConfig(samples_per_pixel=int(rng.integers(1, 4)), max_bounce=int(rng.integers(0, 3))))
        centre = np.array(grid.device_state.min_point_base_t[:3]) + 0.5 * scale * np.array(dim)
        cam.look_at(tuple(float(v) for v in centre + rng.uniform(-1.5, 1.5, 3) * scale * max(dim)), tuple(float(v) for v in centre))
        sun = Sun(SunConfig(enabled=bool(rng.integers(0, 2)), radius=float(rng.choice([0.0, 5.0]))))
        pc = O.push_constants(cam.blob(), sun.blob())
        f, u = ref.render(scene, pc)
        fo, uo, _ = O.render(scene, pc, lowering="llvmpipe")
        assert np.array_equal(f.view(np.uint32), fo.view(np.uint32)), f"{int((f != fo).any(axis=2).sum())} pixels differ"
        assert np.array_equal(u, uo)


def _probe(gl, expr, a, b, c):
    """Our own GLSL (not the reference's): evaluates `expr` per element; A,B,C vec3 and x,y,z float inputs."""
    src = f"""#version 450
layout(local_size_x=32, local_size_y=32) in;
layout(std430, binding=0) buffer Ab {{ vec4 a[]; }};
layout(std430, binding=1) buffer Bb {{ vec4 b[]; }};
layout(std430, binding=2) buffer Cb {{ vec4 c[]; }};
layout(rgba32f, binding=0) uniform writeonly image2D img;
void main() {{
    uint i = gl_GlobalInvocationID.y * 128u + gl_GlobalInvocationID.x;
    vec3 A = a[i].xyz, B = b[i].xyz, C = c[i].xyz; float x = a[i].w, y = b[i].w, z = c[i].w;
    imageStore(img, ivec2(gl_GlobalInvocationID.xy), vec4({expr}));
}}"""
    prog = gl.compile(src)
    out = gl.dispatch(prog, 128, 128, {}, {0: a, 1: b, 2: c}, True)
    gl.delete(prog)
    return out.reshape(-1, 4)


@live
def test_live_llvmpipe_lowering_rules_are_as_assumed():
    """The four rules behind -DORACLE_LOWERING_LLVMPIPE, measured on Mesa itself."""
    gl = ref_gl.GlRef()
    rng = np.random.default_rng(7)
    n = 128 * 128
    a, b, c = ((rng.standard_normal((n, 4)) * 3).astype(np.float32) for _ in range(3))
    bits = lambda v: np.ascontiguousarray(v).view(np.uint32)  # noqa: E731
    o = _probe(gl, "fma(x, y, z), dot(A, B), x / y, sqrt(abs(x))", a, b, c)
    assert np.array_equal(bits(o[:, 0]), bits(a[:, 3] * b[:, 3] + c[:, 3]))                       # fma: two roundings
    A, B = a[:, :3], b[:, :3]
    assert np.array_equal(bits(o[:, 1]), bits((A[:, 2] * B[:, 2] + A[:, 1] * B[:, 1]) + A[:, 0] * B[:, 0]))  # dot: z, y, x
    assert np.array_equal(bits(o[:, 2]), bits(a[:, 3] / b[:, 3]))                                 # IEEE division
    assert np.array_equal(bits(o[:, 3]), bits(np.sqrt(np.abs(a[:, 3]))))                          # IEEE sqrt
    o = _probe(gl, "normalize(A), fract(x)", a, b, c)
    d = (A[:, 2] * A[:, 2] + A[:, 1] * A[:, 1]) + A[:, 0] * A[:, 0]
    assert np.array_equal(bits(o[:, :3]), bits(A * (np.float32(1) / np.sqrt(d))[:, None]))        # normalize: v * (1/sqrt)
    assert np.array_equal(bits(o[:, 3]), bits(a[:, 3] - np.floor(a[:, 3])))                       # fract: x - floor
    L = O.lib("llvmpipe")
    for scale in (1.0, 50.0, 1e4, 1e7):
        s = (a * np.float32(scale)).astype(np.float32)
        o = _probe(gl, "sin(x), 0, 0, 0", s, b, c)
        mine = np.array([L.oracle_sinf(float(v)) for v in s[:, 3]], dtype=np.float32)
        assert np.array_equal(bits(o[:, 0]), bits(mine))                                          # gallivm's sine


# ---------------------------------------------------------------------------------------------- full-size reference frames
# tests/golden/ref_full/*.npz (make_ref_golden.py main_full): the headline workload at 1920x1080 / 512^3 / 8^3 bricks and the
# reference app's own default run, rendered by the reference shader under llvmpipe; whole-frame SHA-256 + band hashes + crops.
FULL = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "ref_full", "*.npz")))
_FULL_GRIDS = {}


def _full_scene(z):
    from tests.golden.make_golden import scene_digest
    from tests.helpers import oracle_scene_from_grid
    from zig_vulkan_amd import workloads as W
    w = W.WORKLOADS[str(z["workload"])]
    if w.name not in _FULL_GRIDS:
        _FULL_GRIDS.clear()
        grid = W.build_grid(w)
        assert scene_digest(grid) == str(z["scene_sha256"]), "the synthetic scene generator no longer produces the fixture's scene"
        _FULL_GRIDS[w.name] = oracle_scene_from_grid(grid)
    return _FULL_GRIDS[w.name]


@pytest.mark.parametrize("path", FULL, ids=[os.path.basename(p)[:-4] for p in FULL])
def test_restatement_with_llvmpipe_lowering_equals_reference_shader_at_full_size(path):
    """Every pixel of the full-size frames: oracle (llvmpipe lowering) == reference shader, by the frame's SHA-256."""
    z = np.load(path)
    f, u, _ = O.render(_full_scene(z), z["push_constants"].copy(), lowering="llvmpipe", want_counters=False)
    crop = z["float_crops"].shape[1]
    for (y, x), want in zip(z["crop_origins"], z["float_crops"]):
        assert np.array_equal(f[y:y + crop, x:x + crop, :3].view(np.uint32), want.view(np.uint32)), (x, y)
    assert hashlib.sha256(f.tobytes()).hexdigest() == str(z["float_sha256"])
    assert hashlib.sha256(u.tobytes()).hexdigest() == str(z["rgba8_sha256"])


@live
def test_live_reference_shader_reproduces_fullsize_headline_fixture():
    z = np.load([p for p in FULL if p.endswith("cfg2_r5_V2.npz")][0])
    f, u = ref_gl.ReferenceShader(int(z["brick_dimension"])).render(_full_scene(z), z["push_constants"].copy())
    assert hashlib.sha256(f.tobytes()).hexdigest() == str(z["float_sha256"])
    assert hashlib.sha256(u.tobytes()).hexdigest() == str(z["rgba8_sha256"])


# ---------------------------------------------------------------------------------------------- GPU: HIP vs fixtures
def _hip_render(z, **config):
    """The fixture's inputs through the C ABI; config: extra VoxelRT.Config fields (library=..., kernel_variant=...)."""
    from zig_vulkan_amd import BrickGrid, Config, VoxelRT
    from zig_vulkan_amd import _lib as L
    import ctypes as C
    gs = np.frombuffer(z["grid_state"].tobytes(), dtype=np.uint32)
    dim = [int(v) for v in gs[3:6]]
    fl = np.frombuffer(z["grid_state"].tobytes(), dtype=np.float32)
    b = int(z["brick_dimension"])
    pc = z["push_constants"].tobytes()
    w, h = (int(v) for v in np.frombuffer(pc[:8], dtype=np.uint32))
    grid = BrickGrid(*dim, min_point=tuple(float(v) for v in fl[8:11]), scale=float(fl[15]), brick_dimension=b,
                     brick_alloc=int(z["brick_start_index"].size))
    cfg = Config(internal_resolution_width=w, internal_resolution_height=h, want_float_output=True, **config)
    rt = VoxelRT(grid, cfg, upload_grid=False)
    # the fixture's seven buffers, byte for byte, through the same entry point the host uses (Pipeline.transfer*)
    for buf, key in ((L.BUF_GRID_STATE, "grid_state"), (L.BUF_MATERIALS, "materials"), (L.BUF_BRICK_STATUS, "brick_status"),
                     (L.BUF_BRICK_INDEX, "brick_index"), (L.BUF_BRICK_OCCUPANCY, "brick_occupancy"),
                     (L.BUF_BRICK_START_INDEX, "brick_start_index"), (L.BUF_MATERIAL_INDEX, "material_index")):
        rt.upload(buf, 0, z[key])
    C.memmove(C.byref(rt.camera.d_camera), pc[:96], 96)
    C.memmove(C.byref(rt.sun.device_data), pc[96:], 32)
    rt.draw()
    f, u = rt.read_rgba32f(), rt.read_rgba8()
    rt.deinit()
    return f, u


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=IDS)
def test_hip_against_reference_shader_fixture(path):
    z = np.load(path)
    f, u = _hip_render(z)
    # the kernel is the oracle, bit for bit, on the fixture's own inputs ...
    fo, uo, _ = O.render(_scene(z), z["push_constants"].copy())
    assert np.array_equal(f.view(np.uint32), fo.view(np.uint32)) and np.array_equal(u, uo)
    # ... and both are what the reference shader itself produced
    assert np.array_equal(u, z["rgba8"]) and np.array_equal(f[:, :, :3].view(np.uint32), z["rgb32f"].view(np.uint32))


# ---------------------------------------------------------------------------------------------- the present / denoise pass
# tests/golden/ref/present_*.npz: image.vert + image.frag of the reference through the same llvmpipe route (one dialect
# edit: push constants -> std140 block).  pow() and the bilinear filter are implementation-defined, so this pass is held to
# north_star's 1e-4, not to bit equality; and a tap whose four texels are black is exactly 0 -> normalize() = NaN only if the
# filter weights are exactly 0 / 1, which llvmpipe's coordinate arithmetic and ours decide differently at a handful of pixels.
PRESENT_IDS = [os.path.basename(p)[:-4] for p in PRESENT]


def _present_args(z):
    p = z["params"]
    return dict(samples=int(p[0]), distribution_bias=float(p[1]), pixel_multiplier=float(p[2]), inverse_hue_tolerance=float(p[3]))


def _compare_present(f, z):
    ref = z["rgb32f"]
    nan_ref, nan_got = np.isnan(ref).any(axis=2), np.isnan(f[..., :3]).any(axis=2)
    black = int((z["image_rgba8"][..., :3] == 0).all(axis=2).sum())
    assert int((nan_ref != nan_got).sum()) <= black          # NaN only ever comes from black texels
    both = ~nan_ref & ~nan_got
    assert both.mean() > 0.99
    assert float(np.abs(f[..., :3][both] - ref[both]).max()) <= TOL


def test_present_fixtures_present():
    assert len(PRESENT) >= 3


@pytest.mark.parametrize("path", PRESENT, ids=PRESENT_IDS)
def test_denoise_oracle_within_tolerance_of_reference_fragment_shader(path):
    z = np.load(path)
    ow, oh = (int(v) for v in z["out_size"])
    f, _ = O.denoise(z["image_rgba8"], ow, oh, **_present_args(z))
    _compare_present(f, z)


@live
@pytest.mark.parametrize("path", PRESENT, ids=PRESENT_IDS)
def test_live_reference_fragment_shader_reproduces_fixture(path):
    z = np.load(path)
    ow, oh = (int(v) for v in z["out_size"])
    f = ref_gl.ReferencePresent().render(z["image_rgba8"], ow, oh, **_present_args(z))
    assert np.array_equal(np.isnan(f[..., :3]), np.isnan(z["rgb32f"]))
    ok = ~np.isnan(z["rgb32f"])
    assert np.array_equal(f[..., :3][ok].view(np.uint32), z["rgb32f"][ok].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("path", PRESENT, ids=PRESENT_IDS)
def test_hip_denoise_within_tolerance_of_reference_fragment_shader(path):
    """vrt_denoise over the fixture's image (set as the context's target through the ABI) against the reference's frame."""
    import ctypes as C
    import torch
    from zig_vulkan_amd import BrickGrid, Config, VoxelRT
    z = np.load(path)
    img = np.ascontiguousarray(z["image_rgba8"])
    h, w = img.shape[:2]
    ow, oh = (int(v) for v in z["out_size"])
    dev = torch.from_numpy(img.reshape(-1)).cuda()
    grid = BrickGrid(1, 1, 1)
    rt = VoxelRT(grid, Config(internal_resolution_width=w, internal_resolution_height=h, external_target_rgba8=dev.data_ptr()))
    a = _present_args(z)
    u, f = rt.denoise(ow, oh, samples=a["samples"], distribution_bias=a["distribution_bias"], pixel_multiplier=a["pixel_multiplier"],
                      inverse_hue_tolerance=a["inverse_hue_tolerance"], want_float=True)
    rt.deinit()
    _compare_present(f, z)
