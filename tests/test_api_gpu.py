"""Boundary behaviour on a real device: error codes instead of aborts, upload semantics (caller keeps
ownership, ranges, device-to-device), buffer sizes as Pipeline.zig:273-283 derives them."""
import ctypes as C

import numpy as np
import pytest

from tests.helpers import O, oracle_scene_from_grid
from zig_vulkan_amd import BrickGrid, Config, VoxelRT, default_materials
from zig_vulkan_amd import _lib as L
from zig_vulkan_amd import workloads as W
from zig_vulkan_amd._lib import VrtError, lib

pytestmark = pytest.mark.gpu


def test_buffer_sizes_follow_the_grid():
    g = BrickGrid(6, 5, 4, brick_dimension=8, brick_alloc=50)
    rt = VoxelRT(g, Config(internal_resolution_width=64, internal_resolution_height=48))
    cells = 6 * 5 * 4
    assert rt.buffer_size(L.BUF_GRID_STATE) == 64
    assert rt.buffer_size(L.BUF_MATERIALS) == 256 * 20
    assert rt.buffer_size(L.BUF_BRICK_STATUS) == ((cells + 31) // 32) * 4
    assert rt.buffer_size(L.BUF_BRICK_INDEX) == cells * 4
    assert rt.buffer_size(L.BUF_BRICK_OCCUPANCY) == 50 * 64
    assert rt.buffer_size(L.BUF_BRICK_START_INDEX) == 50 * 4
    assert rt.buffer_size(L.BUF_MATERIAL_INDEX) == 50 * 512
    rt.deinit()


def test_errors_are_codes_not_aborts():
    g = BrickGrid(4, 4, 4)
    rt = VoxelRT(g, Config(internal_resolution_width=64, internal_resolution_height=48))
    data = np.zeros(64, dtype=np.uint8)
    with pytest.raises(VrtError) as e:  # DestOutOfDeviceMemory, StagingRamp.zig:320-325
        rt.upload(L.BUF_BRICK_STATUS, rt.buffer_size(L.BUF_BRICK_STATUS) - 8, data)
    assert e.value.code == L.VRT_E_OUT_OF_RANGE and "exceeds" in str(e.value)
    with pytest.raises(VrtError) as e:
        rt.upload(99, 0, data)
    assert e.value.code == L.VRT_E_INVALID_ARG
    other = W.Camera(75.0, 32, 32)  # camera for another image size
    assert lib.vrt_dispatch(rt._h, C.byref(other.d_camera), C.byref(rt.sun.device_data)) == L.VRT_E_INVALID_ARG
    assert lib.vrt_dispatch(rt._h, None, None) == L.VRT_E_INVALID_ARG
    assert lib.vrt_read_rgba32f(rt._h, data.ctypes.data, 16) == L.VRT_E_STATE  # want_float_output was 0
    c = L.Counters()
    assert lib.vrt_get_counters(rt._h, C.byref(c)) == L.VRT_E_STATE
    big = np.zeros(64 * 48 * 4 + 4, dtype=np.uint8)
    assert lib.vrt_read_rgba8(rt._h, big.ctypes.data, big.nbytes) == L.VRT_E_OUT_OF_RANGE
    # a grid state whose brick dimensions contradict the context is refused at dispatch
    bad = bytearray(bytes(g.device_state))
    bad[12:16] = (5).to_bytes(4, "little")  # dim_x = 5
    rt.upload(L.BUF_GRID_STATE, 0, np.frombuffer(bytes(bad), dtype=np.uint8))
    with pytest.raises(VrtError) as e:
        rt.draw()
    assert e.value.code == L.VRT_E_INVALID_ARG
    rt.deinit()


def test_upload_copies_before_returning_and_partial_ranges():
    """The caller keeps ownership of the source (Grid.zig:117-126): overwriting it right after the call
    must not change the frame; partial uploads land at their byte offset."""
    w = W.Workload("t", 160, 100, 64, 4, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    from zig_vulkan_amd.voxel_rt import CameraConfig, SunConfig
    rt = VoxelRT(grid, Config(internal_resolution_width=160, internal_resolution_height=100, camera=CameraConfig(samples_per_pixel=1, max_bounce=0),
                              sun=SunConfig(enabled=True, radius=0.0)), upload_grid=False)
    rt.push_materials(default_materials(256))
    rt.upload(L.BUF_GRID_STATE, 0, np.frombuffer(bytes(grid.device_state), dtype=np.uint8))
    for bid in (L.BUF_BRICK_STATUS, L.BUF_BRICK_INDEX, L.BUF_BRICK_OCCUPANCY, L.BUF_BRICK_START_INDEX, L.BUF_MATERIAL_INDEX):
        src = grid.array(bid)
        raw = src.view(np.uint8)
        half = (raw.size // 2) & ~3
        scratch = raw[:half].copy()
        rt.upload(bid, 0, scratch)
        scratch[:] = 0xEE  # scribble over the source immediately
        tail = raw[half:].copy()
        rt.upload(bid, half, tail)
        tail[:] = 0x11
    W.set_view(rt, "V1")
    rt.draw()
    u = rt.read_rgba8()
    pc = O.push_constants(rt.camera.blob(), rt.sun.blob())
    rt.deinit()
    _, uo, _ = O.render(oracle_scene_from_grid(grid), pc)
    assert np.array_equal(u, uo)


def test_device_to_device_upload_and_caller_stream():
    import torch
    w = W.Workload("t", 160, 100, 64, 4, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    stream = torch.cuda.Stream()
    target = torch.zeros(100 * 160 * 4, dtype=torch.uint8, device="cuda")
    from zig_vulkan_amd.voxel_rt import CameraConfig, SunConfig
    rt = VoxelRT(grid, Config(internal_resolution_width=160, internal_resolution_height=100, camera=CameraConfig(samples_per_pixel=1, max_bounce=0),
                              sun=SunConfig(enabled=True, radius=0.0), stream=stream.cuda_stream, external_target_rgba8=target.data_ptr()),
                 upload_grid=False)
    rt.push_materials(default_materials(256))
    rt.upload(L.BUF_GRID_STATE, 0, np.frombuffer(bytes(grid.device_state), dtype=np.uint8))
    keep = []
    for bid in (L.BUF_BRICK_STATUS, L.BUF_BRICK_INDEX, L.BUF_BRICK_OCCUPANCY, L.BUF_BRICK_START_INDEX, L.BUF_MATERIAL_INDEX):
        dev = torch.from_numpy(grid.array(bid).view(np.uint8).copy()).cuda()
        keep.append(dev)
        torch.cuda.synchronize()
        L.check(lib.vrt_upload_device(rt._h, bid, 0, dev.data_ptr(), dev.numel()), rt._h)
    W.set_view(rt, "V2")
    rt.draw()
    rt.wait()
    stream.synchronize()
    u = target.cpu().numpy().reshape(100, 160, 4)  # the caller-owned image was written
    pc = O.push_constants(rt.camera.blob(), rt.sun.blob())
    rt.deinit()
    _, uo, _ = O.render(oracle_scene_from_grid(grid), pc)
    assert np.array_equal(u, uo)


def test_hardware_assumptions_of_the_hand_written_loops():
    """tools/isa_probe (built by __graft_entry__.build): a VOP3 carry-out under a partial EXEC mask writes 0 for the
    inactive lanes, and an `idxen` buffer load beyond num_records returns 0 — what the asm walk loops rely on."""
    import os
    import subprocess
    probe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "isa_probe")
    if not os.path.exists(probe):
        pytest.skip("tools/isa_probe not built (run __graft_entry__.build())")
    out = subprocess.run([probe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "ISA_PROBE_OK" in out.stdout, out.stdout + out.stderr
    # ... and global_load_lds_dwordx4 under a partial EXEC mask puts lane l's 16 bytes at base + 16 l and leaves the inactive
    # lanes' slots alone — the layout vrt_path_kernel's staged bricks are read back from
    glds = os.path.join(os.path.dirname(probe), "ubench", "glds_probe")
    if not os.path.exists(glds):
        pytest.skip("tools/ubench/glds_probe not built (run __graft_entry__.build())")
    out = subprocess.run([glds], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "GLDS_PROBE_OK" in out.stdout, out.stdout + out.stderr


def test_compiled_host_on_the_c_abi_renders_the_same_frame(tmp_path):
    """examples/render_ppm (C++, links libvrt_hip.so only — no Python, no PyTorch in that process) builds the terrain
    scene, renders view V2 and prints a digest of the RGBA8 frame; the Python host must produce the same bytes."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "render_ppm")
    if not os.path.exists(exe):
        pytest.skip("examples/render_ppm not built (run __graft_entry__.build())")
    out = subprocess.run([exe, str(tmp_path / "frame.ppm"), "640", "360", "128", "8"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    digest = out.stdout.strip().splitlines()[-1]
    w = W.Workload("t", 640, 360, 128, 8, 1, 0, True, 5.0)
    grid = W.build_grid(w)
    rt = W.make_renderer(w, grid)
    W.set_view(rt, "V2")
    rt.draw()
    frame = rt.read_rgba8()
    rt.deinit()
    h = 1469598103934665603
    for v in frame.tobytes():
        h = ((h ^ v) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    assert digest == f"{h:016x}"
    ppm = (tmp_path / "frame.ppm").read_bytes()
    assert ppm.startswith(b"P6\n640 360\n255\n") and len(ppm) == len(b"P6\n640 360\n255\n") + 640 * 360 * 3


def test_kernel_name_is_the_symbol_that_ran():
    """vrt_kernel_name(): the template-id of the kernel of the most recent frame, as rocprofv3 prints it (it used
    to format the variant "as asked").  One context, three kinds of frame."""
    import re
    from zig_vulkan_amd import workloads as W
    w = W.Workload("t", 96, 64, 64, 8, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    rt = W.make_renderer(w, grid)
    rt.draw()
    rt.wait()
    assert rt.kernel_name() == "vrt_trace_kernel<8, false, 7, 7, 2, 256>"      # one sample, no bounce, grid <= 64^3 cells: byte status, 7 waves
    rt.camera.d_camera.samples_per_pixel = 3
    rt.draw()
    rt.wait()
    assert rt.kernel_name() == "vrt_trace_kernel<8, false, 7, 7, 1, 256>"      # several samples
    rt.camera.d_camera.max_bounce = 3
    rt.draw()
    rt.wait()
    assert rt.kernel_name() == "vrt_trace_kernel<8, false, 4, 5, 0, 256>"      # bounces on a small scene: the lockstep kernel on the shader's words
    rt.deinit()
    from zig_vulkan_amd import _lib as L
    # persistent lanes forced; power-of-two grid: the half-block walk on a dilated cell index, or (flag) on the linear one
    # (a terrain: its occupied cells do not reach the grid's top, so the walk keeps its steps-left counters — kind 1, not 2)
    for flags, name in ((0, "vrt_path_kernel<8, 5, false, false, false, false, 1>"), (L.TUNE_NO_PATH_DILATED, "vrt_path_kernel<8, 5, false, true, false, false, 0>"),
                        (L.TUNE_NO_PATH_HALFBLOCKS, "vrt_path_kernel<8, 5, false, false, false, false, 0>")):
        rt = W.make_renderer(w, grid, kernel_variant=1 << 23, tuning_flags=flags)
        rt.camera.d_camera.max_bounce = 3
        rt.draw()
        rt.wait()
        assert rt.kernel_name() == name
        rt.deinit()
    assert L.lib.vrt_compiled_kernel_count() == 28   # 26 + vrt_pool_kernel<8, ...> (round 4) + vrt_pool_kernel<4, ...> (round 5)
    assert re.fullmatch(r"vrt_(trace|path)_kernel<[^>]+>", "vrt_trace_kernel<8, false, 7, 7, 2, 256>")


def test_region_and_present_timing_by_device_events():
    """Round 4 (ABI version 2): vrt_region_begin / _end bracket dispatches with HIP events on both of the context's streams;
    vrt_last_denoise_ms times the present pass.  The region covers at least the frames' own kernel times; calls out of order are errors."""
    w = W.Workload("t", 320, 200, 64, 8, 1, 0, True, 5.0)
    grid = W.build_grid(w)
    for fif in (1, 2):
        rt = W.make_renderer(w, grid, frames_in_flight=fif)
        with pytest.raises(VrtError):
            rt.region_end()                      # no region begun
        W.set_view(rt, "V1")
        rt.draw(frames=4)
        rt.wait()
        one = rt.last_kernel_ms()
        rt.region_begin()
        for _ in range(6):
            rt.draw()
        ms = rt.region_end()
        assert ms > 0.0 and ms >= 0.5 * 6 * one / fif and ms < 1000.0, (fif, ms, one)
        assert rt.last_denoise_ms() < 0.0        # no present pass issued yet
        rt.present(400, 260)
        assert 0.0 < rt.last_denoise_ms() < 100.0
        rt.deinit()


@pytest.mark.timeout(600)
def test_bench_line_carries_the_round_4_fields():
    """`python bench.py` at N = 1 (short form, no CPU baseline, no counter passes): ONE JSON line whose `value` is the cold protocol's,
    with the device-event and sustained values beside it, the roofline's validity flag, the present pass and the wall-clock phases."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "12", "--warmup", "3", "--no-cpu-baseline", "--pmc", "off"],
                       cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=550)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 12 and out["warmup"] == 3 and out["precondition_frames"] == 0
    assert out["value"] > 1000.0 and out["value_device_events"] > 1000.0                      # (>= 1 Grays/s, north_star's floor)
    assert 0.5 < out["value_device_events"] / out["value"] < 2.0
    assert out["value_sustained"]["value"] > 1000.0 and out["value_sustained"]["precondition_frames"] > 0
    rf = out["roofline"]
    assert rf["bound"] in ("hbm", "issue") and rf["frac_model_valid"] == (rf["frac"] <= 1.0) and rf["kernel"].startswith("vrt_trace_kernel<8,")
    assert rf["lane_util"] is None and rf["traffic"] is None                                  # --pmc off
    pp = out["present_pass"]
    assert pp["kernel"] == "vrt_denoise_kernel" and 10.0 < pp["us_median"] < 5000.0 and pp["algorithmic_bytes"] == (22 * 16 + 4) * 1920 * 1080
    assert {"startup", "count_rays", "timed_legs", "roofline_leg"} <= set(out["phase_seconds"])
    assert "cpu_baseline" not in out
