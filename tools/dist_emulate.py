#!/usr/bin/env python3
"""The multi-GPU pipeline's two sides, emulated on ONE GPU over the test-only RCCL stand-in (tests/fake_rccl), for any workload:

  root   rank 0 of N rendering its share, the other N-1 ranks' shards posted by a feeder thread (ready-made buffers, no tracing):
         what the root has to sustain — its tiles, the landing of every peer's shard, the un-swizzle of every frame;
  peer   rank 1 of N rendering its share, its shards taken in by a consumer thread (one device copy each, no staging).

The job's frame rate is that of the slower side; `projected speed-up` = the single-GPU frame time of the same frames (one context,
two frames in flight) / max(root, peer).  The stand-in's receive is a device copy where RCCL's is an xGMI write into the buffer, so the
root's side is an estimate; everything else (kernels, launches, streams, events, the host's submission) is the product's.

Round 6: the stand-in models RCCL's two costs by default (--model ORDER:WORKGROUPS, default 1:2 — a communicator's groups execute in issue
order; a group is one kernel of 2 workgroups per operation; 0:0 = round 5's stand-in: copies by hipMemcpyAsync, no order), and the
pipeline's launch slots issue on --comms communicators (0: the library's default, one per slot up to 8; 1: round 5's single one).

usage: dist_emulate.py [--workload NAME] [--worlds 2,4,8] [--batches 1,8] [--shares auto|30,60,100] [--submit call|frame] [--frames N]
                       [--model 1:2] [--comms 0]
Prints one line per (world, batch, share) and a JSON summary (--json PATH)."""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from zig_vulkan_amd import _lib as VL  # noqa: E402
from zig_vulkan_amd import workloads as W  # noqa: E402

FAKE = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
VIEWS = ["V0", "V1", "V2"]


class Uid(C.Structure):
    _fields_ = [("b", C.c_char * 128)]


def cam_array(rt, view, n):
    W.set_view(rt, view)
    arr = (VL.CameraDevice * n)()
    for i in range(n):
        C.memmove(C.byref(arr[i]), bytes(rt.camera.d_camera), 96)
    return arr


def submit(rt, arr, n, how):
    if how == "call":
        rt.dist_frames(arr)
    else:
        for _ in range(n):
            rt.dist_frame()


def timed_views(rt, n, warm, how, before=None):
    """us per frame per view; also the host's share of it (time until the submitting call(s) returned)."""
    out, host = {}, {}
    for v in VIEWS:
        aw, an = cam_array(rt, v, warm), cam_array(rt, v, n)
        th = before(warm + n) if before else None
        submit(rt, aw, warm, how)
        rt.dist_wait()
        t0 = time.perf_counter()
        submit(rt, an, n, how)
        t1 = time.perf_counter()
        rt.dist_wait()
        t2 = time.perf_counter()
        out[v], host[v] = (t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6
        if th:
            th.join()
    return out, host


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default=W.HEADLINE)
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--batches", default="1,8")
    ap.add_argument("--shares", default="auto", help="auto: bench.py's three candidates per world; or a comma list in percent (100 = equal)")
    ap.add_argument("--submit", choices=["call", "frame"], default="call")
    ap.add_argument("--frames", type=int, default=0, help="timed frames per view (0: about 0.3 s worth, 16..480)")
    ap.add_argument("--slots", type=int, default=0, help="launches in flight (0: 8 for one frame per launch, 4 otherwise: bench.py's)")
    ap.add_argument("--json", default="")
    ap.add_argument("--model", default="1:2", help="the stand-in's cost model ORDER:WORKGROUPS_PER_OPERATION (0:0: none)")
    ap.add_argument("--comms", type=int, default=0, help="communicators of the pipeline's launch slots (0: one per slot up to 8; 1: one)")
    ap.add_argument("--partner-priority", type=int, default=0,
                    help="stream priority of the OTHER side's operations (the consumer of a peer's shards, the feeder of the root's): they run on the same GPU "
                         "here and share the runtime's hardware queues with the rank under test; -1: high priority (a queue of their own)")
    args = ap.parse_args()
    model = tuple(int(x) for x in args.model.split(":"))
    w = W.WORKLOADS[args.workload]
    grid = W.build_grid(w)
    fake = C.CDLL(FAKE)
    fake.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, Uid, C.c_int]
    fake.ncclSend.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    fake.ncclRecv.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    fake.ncclCommDestroy.argtypes = [C.c_void_p]
    fake.ncclCommSplit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_void_p]
    fake.fake_rccl_set_model.argtypes = [C.c_int, C.c_int]
    fake.fake_rccl_set_model(*model)

    def comm_family(uid_struct, world, rank, n):
        """rank `rank`'s communicators as the pipeline's other side makes them: the first from the id, n - 1 duplicates split from it."""
        base = C.c_void_p()
        assert fake.ncclCommInitRank(C.byref(base), world, uid_struct, rank) == 0
        fam = [base]
        for _ in range(n - 1):
            c = C.c_void_p()
            assert fake.ncclCommSplit(base, 0, rank, C.byref(c), None) == 0
            fam.append(c)
        return fam

    # the single-GPU reference: one context, two frames in flight, the same views
    full = W.make_renderer(w, grid, frames_in_flight=2)
    whole = {}
    W.set_view(full, "V0")
    full.draw()
    full.wait()
    t0 = time.perf_counter()
    full.draw()
    full.wait()
    est_ms = max((time.perf_counter() - t0) * 1e3, 0.02)
    n1 = int(min(400, max(6, 600.0 / est_ms)))
    for v in VIEWS:
        W.set_view(full, v)
        for _ in range(max(2, n1 // 8)):
            full.draw()
        full.wait()
        t0 = time.perf_counter()
        for _ in range(n1):
            full.draw()
        full.wait()
        whole[v] = (time.perf_counter() - t0) / n1 * 1e6
    name1 = full.kernel_name()
    full.deinit()
    single_us = sum(whole.values()) / len(whole)
    print(f"# {w.name}: one GPU, two frames in flight: {single_us:.1f} us per frame {dict((v, round(t, 1)) for v, t in whole.items())}  [{name1}]", flush=True)
    summary = {"workload": w.name, "single_gpu_us_per_frame": single_us, "single_gpu_per_view": whole, "submit": args.submit, "rows": []}

    for world in [int(x) for x in args.worlds.split(",")]:
        shares = bench.root_share_candidates(world) if args.shares == "auto" else [int(x) for x in args.shares.split(",")]
        for batch in [int(x) for x in args.batches.split(",")]:
            slots = args.slots or (8 if batch == 1 else 4)
            per_rank_ms = est_ms / world
            n = args.frames or int(min(480, max(16, 300.0 / per_rank_ms)))
            n = max(batch, (n // batch) * batch)
            warm = max(batch, ((n // 8) // batch) * batch)
            for share in shares:
                weight = share if share < 100 else 0
                row = {"world": world, "frames_per_launch": batch, "launches_in_flight": slots, "root_share": share, "frames_per_view": n, "model": list(model)}
                # ---- peer: rank 1 of N; a consumer thread plays rank 0's receive of its shard (one device copy out of the send buffer:
                # the send's read of the shard, a local write instead of the xGMI one) ----
                fake.fake_rccl_set_zero_copy(1)
                uid = b"emul-peer" + bytes([world, batch, share]) + os.urandom(16) + bytes(128 - 28)
                rt = W.make_renderer(w, grid, shard_rank=1, shard_count=world, shard_root_weight=weight)
                rt.dist_init(uid, 1, world, frames_in_flight=slots, rccl_path=FAKE, frames_per_launch=batch, communicators=args.comms)
                ncomms = rt.dist_comm_info()["communicators"]
                if w.max_bounce > 0:
                    rt.reserve_samples(w.spp)
                shard_bytes = rt.shard_info().tiles_per_rank * 256 * 3   # shards travel as RGB
                u = Uid()
                C.memmove(C.byref(u), uid, 128)
                fam0 = comm_family(u, world, 0, ncomms)
                sink = torch.zeros(batch * shard_bytes, dtype=torch.uint8, device="cuda")
                sink_stream = torch.cuda.Stream(priority=args.partner_priority)
                launch_no = [0]

                def consume(frames):
                    def run():
                        for _ in range(frames // batch):
                            c = fam0[(launch_no[0] % slots) % ncomms]   # (launch L goes to slot L % slots, whose communicator is slot % ncomms)
                            launch_no[0] += 1
                            assert fake.ncclRecv(sink.data_ptr(), batch * shard_bytes, 1, 1, c, sink_stream.cuda_stream) == 0
                    th = threading.Thread(target=run)
                    th.start()
                    return th

                peer, peer_host = timed_views(rt, n, warm, args.submit, before=consume)
                row["peer_kernel"] = rt.kernel_name()
                torch.cuda.synchronize()
                rt.deinit()
                for c in reversed(fam0):
                    fake.ncclCommDestroy(c)
                del sink
                # ---- root: rank 0 of N, its peers fed ----
                uid = b"emul-root" + bytes([world, batch, share]) + os.urandom(16) + bytes(128 - 28)
                rt = W.make_renderer(w, grid, shard_rank=0, shard_count=world, shard_root_weight=weight)
                rt.dist_init(uid, 0, world, frames_in_flight=slots, rccl_path=FAKE, frames_per_launch=batch, communicators=args.comms)
                ncomms = rt.dist_comm_info()["communicators"]
                if w.max_bounce > 0:
                    rt.reserve_samples(w.spp)
                shard_bytes = rt.shard_info().tiles_per_rank * 256 * 3   # shards travel as RGB
                u = Uid()
                C.memmove(C.byref(u), uid, 128)
                fams = [comm_family(u, world, r, ncomms) for r in range(1, world)]
                dummy = torch.zeros(batch * shard_bytes, dtype=torch.uint8, device="cuda")
                feeder_stream = torch.cuda.Stream(priority=args.partner_priority)
                launch_no = [0]

                def feed(frames):
                    def run():
                        for _ in range(frames // batch):
                            k = (launch_no[0] % slots) % ncomms
                            launch_no[0] += 1
                            for fam in fams:
                                assert fake.ncclSend(dummy.data_ptr(), batch * shard_bytes, 1, 0, fam[k], feeder_stream.cuda_stream) == 0
                    th = threading.Thread(target=run)
                    th.start()
                    return th

                root, root_host = timed_views(rt, n, warm, args.submit, before=feed)
                rt.deinit()
                for fam in fams:
                    for c in reversed(fam):
                        fake.ncclCommDestroy(c)
                del dummy
                mean = lambda d: sum(d.values()) / len(d)  # noqa: E731
                slower = {v: max(root[v], peer[v]) for v in VIEWS}
                row.update({"peer_us": mean(peer), "root_us": mean(root), "peer_host_us": mean(peer_host), "root_host_us": mean(root_host),
                            "job_us": mean(slower), "projected_speedup": single_us / mean(slower), "peer_per_view": peer, "root_per_view": root})
                summary["rows"].append(row)
                row["communicators"] = ncomms
                print(f"{w.name} N={world} batch={batch} slots={slots} comms={ncomms} model={args.model} root share {share:3d} %: peer {row['peer_us']:8.1f} us (host {row['peer_host_us']:.1f}), "
                      f"root {row['root_us']:8.1f} us (host {row['root_host_us']:.1f}) -> job {row['job_us']:8.1f} us per frame = {row['projected_speedup']:.2f} x one GPU"
                      f"  [{row['peer_kernel']}]", flush=True)
    if args.json:
        with open(args.json, "w") as fh:
            json.dump(summary, fh, indent=1)


if __name__ == "__main__":
    main()
