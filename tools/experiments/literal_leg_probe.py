#!/usr/bin/env python3
"""Where the literal (one gather per frame) leg's 27 us per frame of a non-root rank go: rank 1 of N over the RCCL stand-in in zero-copy mode
(a send is an event record, nobody receives), launches in flight 1 / 2 / 4 / 8, frames per launch 1 / 2 / 8: us per frame by the host's clock,
the host's own submission time, and the kernel's duration per launch by events on its stream (vrt_dist_profile).
usage: literal_leg_probe.py [world] [workload]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from zig_vulkan_amd import _lib as VL
from zig_vulkan_amd import workloads as W
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
fake = C.CDLL(FAKE)
fake.fake_rccl_set_zero_copy(1)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w = W.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else W.HEADLINE]
grid = W.build_grid(w)
for batch, slots in ((1, 4), (1, 8), (1, 12), (1, 16), (2, 8), (2, 16), (8, 4)):
    rt = W.make_renderer(w, grid, shard_rank=1, shard_count=world)
    rt.dist_init(b"literal-probe" + bytes([batch, slots]) + os.urandom(16) + bytes(128 - 31), 1, world, frames_in_flight=slots, rccl_path=FAKE, frames_per_launch=batch)
    W.set_view(rt, "V1")
    n = 960
    arr = (VL.CameraDevice * n)()
    for i in range(n):
        C.memmove(C.byref(arr[i]), bytes(rt.camera.d_camera), 96)
    rt.dist_frames(arr); rt.dist_wait()
    t0 = time.perf_counter(); rt.dist_frames(arr); t1 = time.perf_counter(); rt.dist_wait(); t2 = time.perf_counter()
    rt.dist_profile(True)
    rt.dist_frames(arr); rt.dist_wait()
    st = rt.dist_stats()
    print(f"rank 1 of {world}, {batch} frame(s) per launch, {slots} launch(es) in flight: {1e6 * (t2 - t0) / n:6.1f} us per frame (host submission {1e6 * (t1 - t0) / n:5.1f}); "
          f"kernel {1e3 * st['kernel_ms_per_launch']:6.1f} us per launch, send {1e3 * st['collective_ms_per_launch']:5.1f} us  [{rt.kernel_name()}, {st['owned_tiles']} tiles]", flush=True)
    rt.deinit()
