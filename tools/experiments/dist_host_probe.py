import os, sys, time
sys.path.insert(0, '.')
from zig_vulkan_amd import workloads as W
FAKE = os.path.join(os.getcwd(), "tests", "fake_rccl", "libfake_rccl.so")
w = W.WORKLOADS[W.HEADLINE]
grid = W.build_grid(w)
for batch, slots in ((1, 8), (1, 4), (8, 4)):
    rt = W.make_renderer(w, grid, shard_rank=1, shard_count=8, shard_root_weight=30)
    rt.dist_init(b"hostprobe" + bytes([batch, slots]) + os.urandom(16) + bytes(128 - 27), 1, 8, frames_in_flight=slots, rccl_path=FAKE, frames_per_launch=batch)
    W.set_view(rt, "V1")
    for _ in range(64): rt.dist_frame()
    rt.dist_wait()
    n = 960
    t0 = time.perf_counter()
    for _ in range(n): rt.dist_frame()
    t1 = time.perf_counter()
    rt.dist_wait()
    t2 = time.perf_counter()
    print(f"batch {batch} slots {slots}: host submit {1e6*(t1-t0)/n:.1f} us per frame, total {1e6*(t2-t0)/n:.1f} us per frame")
    rt.deinit()
