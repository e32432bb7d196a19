#!/usr/bin/env python3
"""What the ROOT rank of an 8-GPU run has to sustain on its own GPU: its eighth of the tiles, the landing of the other seven
ranks' shards and the un-swizzle of every frame.  Rank 0 of 8 over the test-only RCCL stand-in, the seven peers played by a
feeder thread that posts ready-made shards (no tracing), so the only GPU work is rank 0's."""
import ctypes as C, os, sys, threading, time
os.environ["FAKE_RCCL_ZERO_COPY"] = "1"  # the feeder never rewrites its buffer: let the receives copy straight out of it
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from zig_vulkan_amd import workloads as W
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
fake = C.CDLL(FAKE)
class Uid(C.Structure): _fields_ = [("b", C.c_char * 128)]
fake.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
fake.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
w = W.WORKLOADS[W.HEADLINE]
grid = W.build_grid(w)
# usage: root_rank.py [world] [frames per launch] [launches in flight] [root shares in %, comma separated; 0 = equal share]
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 4
for weight in ([int(v) for v in sys.argv[4].split(',')] if len(sys.argv) > 4 else (0, 65, 40)):
    uid = b"root-rank" + os.urandom(16) + bytes(128 - 25)
    rt = W.make_renderer(w, grid, shard_rank=0, shard_count=world, shard_root_weight=weight)
    rt.dist_init(uid, 0, world, frames_in_flight=slots, rccl_path=FAKE, frames_per_launch=batch)
    shard_bytes = rt.shard_info().tiles_per_rank * 256 * 3  # shards travel as RGB
    u = Uid(); C.memmove(C.byref(u), uid, 128)
    comms = []
    for r in range(1, world):
        c = C.c_void_p(); assert fake.ncclCommInitRank(C.byref(c), world, u, r) == 0; comms.append(c)
    dummy = torch.zeros(batch * shard_bytes, dtype=torch.uint8, device="cuda")
    feeder_stream = torch.cuda.Stream()
    def feed(nbatches):
        for _ in range(nbatches):
            for c in comms:
                assert fake.ncclSend(dummy.data_ptr(), batch * shard_bytes, 1, 0, c, feeder_stream.cuda_stream) == 0
    out = {}
    for view in ["V0", "V1", "V2"]:
        W.set_view(rt, view)
        n = 480
        th = threading.Thread(target=feed, args=((32 + n) // batch,)); th.start()
        for _ in range(32): rt.dist_frame()
        rt.dist_wait(); t0 = time.perf_counter()
        for _ in range(n): rt.dist_frame()
        rt.dist_wait(); out[view] = round((time.perf_counter() - t0) / n * 1e6, 1)
        th.join()
    print(f"rank 0 of {world}, root share {weight or 100} % ({batch} frames per launch, {slots} launches in flight), peers fed: us per frame", out, " mean", round(sum(out.values()) / 3, 1))
    rt.deinit()
    for c in comms: fake.ncclCommDestroy(c)
