"""The multi-rank branch of the native frame pipeline (vrt_dist_frame: kernel -> grouped recv on rank 0 / send
elsewhere -> un-swizzle, several frames in flight) with R ranks as R contexts of this process on the one GPU of the
box, bound to tests/fake_rccl/libfake_rccl.so instead of RCCL (which refuses two ranks on one device).  Everything
of the pipeline except RCCL itself runs: packed tile-major shards of every rank, slot streams and events, frame
ordering across ranks, the gathered layout, the un-swizzle.  Each rank is driven from its own thread, like a process."""
import os
import threading

import numpy as np
import pytest

from zig_vulkan_amd import workloads as W

pytestmark = pytest.mark.gpu

FAKE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl", "libfake_rccl.so")


@pytest.mark.parametrize("world,frames_in_flight,frames_per_launch,root_weight",
                         [(2, 2, 1, 0), (3, 4, 1, 0), (8, 8, 1, 0), (8, 16, 1, 30), (4, 12, 2, 0), (8, 1, 1, 0), (2, 2, 3, 0), (8, 3, 8, 0), (5, 2, 4, 0), (8, 1, 2, 0),
                          (8, 3, 8, 65), (2, 2, 1, 30), (5, 2, 3, 85), (8, 2, 4, 1),
                          # bench.py's root-share candidates (root_share_candidates: the emulated default, 0.6 x, 1.5 x) at 8 / 4 / 2 ranks,
                          # one collective per frame and eight frames per collective
                          (8, 4, 8, 30), (8, 4, 8, 18), (8, 4, 8, 45), (8, 4, 1, 30), (8, 4, 1, 18), (8, 4, 1, 45),
                          (4, 4, 8, 77), (4, 4, 8, 46), (4, 4, 8, 100), (4, 4, 1, 77), (4, 4, 1, 46),
                          (2, 4, 8, 100), (2, 4, 8, 60), (2, 4, 1, 60)])
def test_native_pipeline_with_many_ranks_on_one_gpu(world, frames_in_flight, frames_per_launch, root_weight):
    if not os.path.exists(FAKE):
        pytest.skip("tests/fake_rccl/libfake_rccl.so not built (run __graft_entry__.build())")
    # (width 332: the four-pixels-per-thread un-swizzle; 330: the one-pixel one)
    w = W.Workload("t", 332 if frames_per_launch > 1 else 330, 210, 64, 4, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    views = ["V0", "V1", "V2", "V1", "V2", "V0", "V0", "V2", "V1", "V1", "V0"]
    plain = W.make_renderer(w, grid)
    ref = {}
    for v in set(views):
        W.set_view(plain, v)
        plain.draw()
        ref[v] = plain.read_rgba8().copy()
    plain.deinit()

    uid = b"fake-rccl-test" + bytes([world, frames_in_flight]) + os.urandom(16) + bytes(128 - 32)
    reads = [i for i in range(len(views)) if (i + 1) % frames_per_launch == 0 and i % 3 == 2]  # only where the queue has just been launched
    ranks = [W.make_renderer(w, grid, shard_rank=r, shard_count=world, shard_root_weight=root_weight) for r in range(world)]
    if root_weight:
        owned = [rt.shard_info().owned_tiles for rt in ranks]
        assert sum(owned) == ranks[0].shard_info().tiles_x * ranks[0].shard_info().tiles_y
        assert owned[0] < min(owned[1:]) if root_weight < 100 else max(owned) - min(owned) <= 1   # (100 % = an equal share)
    for r, rt in enumerate(ranks):
        rt.dist_init(uid, r, world, frames_in_flight=frames_in_flight, rccl_path=FAKE, frames_per_launch=frames_per_launch)
    got, errors = [], []

    def drive(r):
        try:
            rt = ranks[r]
            for i, v in enumerate(views):
                W.set_view(rt, v)
                rt.dist_frame()
                if r == 0 and i in reads:  # read some frames mid-stream, the rest stay in flight
                    got.append((v, rt.dist_read_frame().copy()))
            rt.dist_wait()
            if r == 0:
                got.append((views[-1], rt.dist_read_frame().copy()))
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=drive, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a rank hung"
    assert not errors, errors
    for rt in ranks:
        rt.deinit()
    assert len(got) == len(reads) + 1
    for v, frame in got:
        assert np.array_equal(frame, ref[v]), f"assembled frame of view {v} differs from the single-context frame"


def _fake():
    import ctypes as C
    f = C.CDLL(FAKE)
    f.fake_rccl_set_model.argtypes = [C.c_int, C.c_int]
    f.fake_rccl_live_communicators.restype = C.c_int
    return f


@pytest.mark.parametrize("world,frames_in_flight,frames_per_launch,communicators,model",
                         [(8, 8, 1, 0, (1, 2)), (8, 8, 1, 1, (1, 2)), (4, 16, 1, 0, (1, 4)), (4, 6, 2, 3, (1, 1)), (2, 4, 8, 0, (0, 2)), (8, 16, 1, 16, (1, 2)),
                          (3, 5, 1, 2, (1, 0))])
def test_a_communicator_per_launch_slot_and_the_stand_ins_cost_model(world, frames_in_flight, frames_per_launch, communicators, model):
    """Round 6 (VERDICT r05 #2a/b): launch slot i issues its gather on communicator i % n (ncclCommSplit duplicates of the first), n = one
    per slot up to 8 unless the host says otherwise; and the stand-in now models RCCL's two costs — a communicator's groups execute in
    issue order, a group is one kernel of k workgroups per operation.  Frames must be the single context's under every combination
    (the model changes timing, never bytes), and the communicators must all be destroyed with their contexts."""
    if not os.path.exists(FAKE):
        pytest.skip("tests/fake_rccl/libfake_rccl.so not built (run __graft_entry__.build())")
    fake = _fake()
    live_before = fake.fake_rccl_live_communicators()
    fake.fake_rccl_set_model(*model)
    try:
        w = W.Workload("t", 332, 210, 64, 4, 1, 0, True, 0.0)
        grid = W.build_grid(w)
        views = ["V0", "V1", "V2", "V1", "V2", "V0", "V0", "V2", "V1", "V1", "V0", "V2", "V0", "V1", "V2", "V0", "V1", "V1", "V2", "V0"]
        views = views[:max(frames_per_launch, (len(views) // frames_per_launch) * frames_per_launch)]
        plain = W.make_renderer(w, grid)
        ref = {}
        for v in set(views):
            W.set_view(plain, v)
            plain.draw()
            ref[v] = plain.read_rgba8().copy()
        plain.deinit()
        uid = b"fake-rccl-comms" + bytes([world, frames_in_flight, communicators]) + os.urandom(16) + bytes(128 - 34)
        ranks = [W.make_renderer(w, grid, shard_rank=r, shard_count=world) for r in range(world)]
        for r, rt in enumerate(ranks):
            rt.dist_init(uid, r, world, frames_in_flight=frames_in_flight, rccl_path=FAKE, frames_per_launch=frames_per_launch, communicators=communicators)
            info = rt.dist_comm_info()
            want = min(communicators, frames_in_flight) if communicators else min(frames_in_flight, 8)
            assert info == {"communicators": want, "made_by_this_init": want, "library_splits": True, "agreed_by_all_reduce": False}, info
        assert fake.fake_rccl_live_communicators() == live_before + world * want
        got, errors = [], []

        def drive(r):
            try:
                rt = ranks[r]
                for i, v in enumerate(views):
                    W.set_view(rt, v)
                    rt.dist_frame()
                    if r == 0 and (i + 1) % frames_per_launch == 0 and i % 4 == 3:
                        got.append((v, rt.dist_read_frame().copy()))
                rt.dist_wait()
                if r == 0:
                    got.append((views[-1], rt.dist_read_frame().copy()))
            except Exception as e:  # noqa: BLE001
                errors.append((r, repr(e)))

        threads = [threading.Thread(target=drive, args=(r,), daemon=True) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in threads), "a rank hung"
        assert not errors, errors
        for rt in ranks:
            rt.deinit()
        for v, frame in got:
            assert np.array_equal(frame, ref[v]), f"assembled frame of view {v} differs from the single-context frame"
        assert fake.fake_rccl_live_communicators() == live_before
    finally:
        fake.fake_rccl_set_model(0, 0)


def test_communicators_kept_in_the_pool_are_reused_by_the_next_context_with_the_same_id():
    """vrt_dist_keep_communicators(1): a context's communicators outlive it; the next context initialised with the SAME id takes them
    from the pool (made_by_this_init == 0: no collective), two live contexts of one id get disjoint ones, and
    vrt_dist_release_communicators destroys what nobody holds.  (bench.py makes a dozen contexts per run — one per root-share candidate
    and leg — on one id.)"""
    if not os.path.exists(FAKE):
        pytest.skip("tests/fake_rccl/libfake_rccl.so not built (run __graft_entry__.build())")
    from zig_vulkan_amd import VoxelRT
    fake = _fake()
    live_before = fake.fake_rccl_live_communicators()
    w = W.Workload("t", 330, 210, 64, 4, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    plain = W.make_renderer(w, grid)
    W.set_view(plain, "V2")
    plain.draw()
    ref = plain.read_rgba8().copy()
    plain.deinit()
    world = 3
    uid = b"fake-rccl-pool-keep" + os.urandom(16) + bytes(128 - 35)
    was = VoxelRT.dist_keep_communicators(True)
    try:
        def run(slots, expect_made, hold=None):
            ranks = [W.make_renderer(w, grid, shard_rank=r, shard_count=world) for r in range(world)]
            for r, rt in enumerate(ranks):
                rt.dist_init(uid, r, world, frames_in_flight=slots, rccl_path=FAKE)
                info = rt.dist_comm_info()
                assert info["communicators"] == slots and info["made_by_this_init"] == expect_made, info
            errors = []

            def drive(r):
                try:
                    W.set_view(ranks[r], "V2")
                    for _ in range(2 * slots + 1):
                        ranks[r].dist_frame()
                    ranks[r].dist_wait()
                except Exception as e:  # noqa: BLE001
                    errors.append((r, repr(e)))

            threads = [threading.Thread(target=drive, args=(r,), daemon=True) for r in range(world)]
            for t in threads:
                t.start()
            for t in threads:
                t.join(timeout=120)
            assert not any(t.is_alive() for t in threads) and not errors, errors
            assert np.array_equal(ranks[0].dist_read_frame(), ref)
            if hold is not None:
                hold.extend(ranks)
            else:
                for rt in ranks:
                    rt.deinit()

        run(4, 4)                      # makes 4 per rank
        assert fake.fake_rccl_live_communicators() == live_before + 4 * world      # ... which stay
        run(3, 0)                      # takes 3 of them: nothing made
        held = []
        run(4, 0, hold=held)           # holds all 4 ...
        run(2, 2)                      # ... so a second live context of the same id makes 2 more (indices 4, 5 on every rank)
        for rt in held:
            rt.deinit()
        run(6, 0)                      # all six from the pool
        assert fake.fake_rccl_live_communicators() == live_before + 6 * world
        assert VoxelRT.dist_release_communicators() == 6 * world
        assert fake.fake_rccl_live_communicators() == live_before
        run(2, 2)                      # (after a release the same id starts over)
    finally:
        VoxelRT.dist_keep_communicators(was)
        VoxelRT.dist_release_communicators()
    assert fake.fake_rccl_live_communicators() == live_before


@pytest.mark.parametrize("size,world,root_weight,frames_per_launch", [((40, 24), 8, 60, 8), ((16, 16), 4, 50, 2), ((130, 70), 8, 0, 8), ((1, 1), 2, 90, 1)])
def test_more_ranks_than_tiles_and_tiny_frames(size, world, root_weight, frames_per_launch):
    """Frames with fewer tiles than ranks (some ranks, possibly the root, own nothing), with and without the weighted pattern."""
    if not os.path.exists(FAKE):
        pytest.skip("tests/fake_rccl/libfake_rccl.so not built (run __graft_entry__.build())")
    w = W.Workload("t", size[0], size[1], 64, 4, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    plain = W.make_renderer(w, grid)
    W.set_view(plain, "V2")
    plain.draw()
    ref = plain.read_rgba8().copy()
    plain.deinit()
    uid = b"tiny" + bytes([world, root_weight, frames_per_launch]) + os.urandom(16) + bytes(128 - 23)
    ranks = [W.make_renderer(w, grid, shard_rank=r, shard_count=world, shard_root_weight=root_weight) for r in range(world)]
    owned = [rt.shard_info().owned_tiles for rt in ranks]
    assert sum(owned) == ranks[0].shard_info().tiles_x * ranks[0].shard_info().tiles_y
    for r, rt in enumerate(ranks):
        W.set_view(rt, "V2")
        rt.dist_init(uid, r, world, frames_in_flight=2, rccl_path=FAKE, frames_per_launch=frames_per_launch)
    errors = []

    def drive(r):
        try:
            for _ in range(11):
                ranks[r].dist_frame()
            ranks[r].dist_wait()
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=drive, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads) and not errors, errors
    got = ranks[0].dist_read_frame()
    for rt in ranks:
        rt.deinit()
    both_nan_free = np.array_equal(got, ref)
    assert both_nan_free, f"owned tiles per rank {owned}"


@pytest.mark.parametrize("world,no_broadcast", [(2, False), (5, True), (8, False)])
def test_edits_made_on_one_rank_reach_every_replica(world, no_broadcast):
    """SURVEY.md §8(f) #1, the multi-GPU half: only rank 0's host edits its BrickGrid; vrt_dist_broadcast_grid_delta uploads
    rank 0's dirty ranges and broadcasts them (ncclBroadcast, or grouped send / recv where the library lacks it), frames are in
    flight before and after; every rank renders its own tiles from its own replica, so the assembled frame equals the
    single-context frame of the edited scene only if every replica took the edit."""
    if not os.path.exists(FAKE):
        pytest.skip("tests/fake_rccl/libfake_rccl.so not built (run __graft_entry__.build())")
    from zig_vulkan_amd import _lib as L
    flags = L.TUNE_DIST_NO_BROADCAST if no_broadcast else 0   # (every rank alike: a mix would pair ncclBroadcast with send / recv)
    w = W.Workload("t", 330, 210, 64, 4, 1, 0, True, 0.0)
    grids = [W.build_grid(w) for _ in range(world)]          # one host copy per rank, identical so far
    edited = W.build_grid(w)

    def edit(g):
        for y in range(20, 60):
            for dx in range(3):
                for dz in range(3):
                    g.insert(30 + dx, y, 30 + dz, 7)
                    g.insert(5 + dx, y, 50 + dz, 5)

    edit(edited)
    plain = W.make_renderer(w, edited)
    W.set_view(plain, "V1")
    plain.draw()
    want = plain.read_rgba8().copy()
    plain.deinit()

    uid = b"fake-rccl-edit" + bytes([world]) + os.urandom(16) + bytes(128 - 31)
    ranks = [W.make_renderer(w, grids[r], shard_rank=r, shard_count=world, tuning_flags=flags) for r in range(world)]
    for r, rt in enumerate(ranks):
        W.set_view(rt, "V1")
        rt.dist_init(uid, r, world, frames_in_flight=3, rccl_path=FAKE)
    before, errors, shared = [], [], {}
    ready = threading.Barrier(world)

    def drive(r):
        try:
            rt = ranks[r]
            for _ in range(4):
                rt.dist_frame()                                  # frames of the unedited scene, left in flight
            if r == 0:
                before.append(rt.dist_read_frame().copy())
                edit(grids[0])                                   # only this host edits
                shared["ranges"] = rt.grid_delta_ranges()
            ready.wait(timeout=60)                               # (a process would get the ranges over its own channel)
            rt.dist_broadcast_grid_delta(root=0, ranges=shared["ranges"])
            for _ in range(4):
                rt.dist_frame()
            rt.dist_wait()
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=drive, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a rank hung"
    assert not errors, errors
    after = ranks[0].dist_read_frame().copy()
    for rt in ranks:
        rt.deinit()
    assert shared["ranges"] and not np.array_equal(before[0], want)
    assert np.array_equal(after, want)


def test_bench_root_share_candidates_are_the_tested_ones():
    import bench
    assert bench.root_share_candidates(8) == [18, 30, 45] and bench.root_share_candidates(4) == [46, 77, 100] and bench.root_share_candidates(2) == [60, 100]


@pytest.mark.parametrize("world,frames_per_launch", [(2, 1), (4, 8)])
def test_pipeline_stage_profile(world, frames_per_launch):
    """vrt_dist_profile / vrt_dist_stats: per-launch stage times by events on the launch's own stream — what bench.py's per-rank
    breakdown reads.  Every rank samples launches; kernel time is positive everywhere, only rank 0 un-swizzles; frames still equal
    the single-context frame with the extra events in the stream."""
    if not os.path.exists(FAKE):
        pytest.skip("tests/fake_rccl/libfake_rccl.so not built (run __graft_entry__.build())")
    w = W.Workload("t", 332, 210, 64, 4, 1, 0, True, 0.0)
    grid = W.build_grid(w)
    plain = W.make_renderer(w, grid)
    W.set_view(plain, "V1")
    plain.draw()
    ref = plain.read_rgba8().copy()
    plain.deinit()
    uid = b"fake-rccl-prof" + bytes([world, frames_per_launch]) + os.urandom(16) + bytes(128 - 32)
    ranks = [W.make_renderer(w, grid, shard_rank=r, shard_count=world) for r in range(world)]
    for r, rt in enumerate(ranks):
        W.set_view(rt, "V1")
        rt.dist_init(uid, r, world, frames_in_flight=3, rccl_path=FAKE, frames_per_launch=frames_per_launch)
    stats, errors = {}, []

    def drive(r):
        try:
            rt = ranks[r]
            for _ in range(2 * frames_per_launch):
                rt.dist_frame()
            rt.dist_wait()
            assert rt.dist_stats()["launches_sampled"] == 0      # nothing sampled before it is asked for
            rt.dist_profile(True)
            for _ in range(6 * frames_per_launch):
                rt.dist_frame()
            rt.dist_wait()
            stats[r] = rt.dist_stats()
            rt.dist_profile(False)
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=drive, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads) and not errors, errors
    got = ranks[0].dist_read_frame()
    for rt in ranks:
        rt.deinit()
    assert np.array_equal(got, ref)
    for r in range(world):
        st = stats[r]
        assert 3 <= st["launches_sampled"] <= 6 and st["frames_sampled"] == st["launches_sampled"] * frames_per_launch
        assert st["frames_per_launch"] == frames_per_launch and st["kernel_ms_per_launch"] > 0
        assert (st["unswizzle_ms_per_launch"] > 0) == (r == 0)


def _bounce_scene():
    """A cfg4-shaped scene at test size: 8^3 bricks, power-of-two grid, sparse spheres that reach the grid's faces, 2 spp, 2 bounces,
    soft sun — the shape on which the library picks vrt_pool_kernel once the host knows the box of the occupied cells."""
    w = W.Workload("t", 330, 210, 256, 8, 2, 2, True, 5.0, "sparse", 0.08, 30000)
    return w, W.build_grid(w)


def _single_context_frames(w, grid, views, variant):
    plain = W.make_renderer(w, grid, kernel_variant=variant)
    W.set_view(plain, "V0")
    plain.draw()
    plain.wait()          # (the box of the occupied cells has reached the host)
    ref = {}
    for v in set(views):
        W.set_view(plain, v)
        plain.draw()
        ref[v] = plain.read_rgba8().copy()
    name = plain.kernel_name()
    plain.deinit()
    return ref, name


@pytest.mark.parametrize("world,frames_in_flight,frames_per_launch,root_weight,reserve",
                         [(2, 2, 1, 0, False), (4, 3, 1, 77, True), (8, 2, 1, 30, False), (8, 8, 1, 0, True), (2, 2, 3, 0, False), (4, 2, 8, 46, True), (8, 3, 2, 18, False)])
def test_bounce_frames_go_through_the_pipeline_on_the_persistent_kernels(world, frames_in_flight, frames_per_launch, root_weight, reserve):
    """BASELINE configs[4] is an 8-GPU path trace, and until round 4 the pipeline sent bounce frames to the lockstep
    kernel.  Now every launch slot has its own unit counters, path records and sample buffer, and vrt_pool_resolve_kernel writes the
    packed RGB shard: inside the pipeline the frames are traced by vrt_path_kernel<..., DIL 1> until the host knows the box of the
    occupied cells and by vrt_pool_kernel from then on — the kernels a single context uses — and every assembled frame equals the
    single-context frame.  Ranks are threads over tests/fake_rccl; with and without vrt_reserve_samples; batches (the persistent
    kernels then take their frames one per launch into the batch's buffer, ONE collective per batch)."""
    if not os.path.exists(FAKE):
        pytest.skip("tests/fake_rccl/libfake_rccl.so not built (run __graft_entry__.build())")
    path = 1 << 23
    w, grid = _bounce_scene()
    views = ["V0", "V1x", "V0", "V2", "V1x", "V2", "V0", "V1x"]
    ref, single_name = _single_context_frames(w, grid, views, path)
    assert single_name == "vrt_pool_kernel<8, 6, 60, 2>"
    uid = b"fake-rccl-pool" + bytes([world, frames_in_flight, frames_per_launch]) + os.urandom(16) + bytes(128 - 33)
    ranks = [W.make_renderer(w, grid, kernel_variant=path, shard_rank=r, shard_count=world, shard_root_weight=root_weight) for r in range(world)]
    for r, rt in enumerate(ranks):
        rt.dist_init(uid, r, world, frames_in_flight=frames_in_flight, rccl_path=FAKE, frames_per_launch=frames_per_launch)
        if reserve:
            rt.reserve_samples(w.spp)
    got, errors, names = [], [], {r: [] for r in range(world)}
    reads = [i for i in range(len(views)) if (i + 1) % frames_per_launch == 0]   # only where the queue has just been launched

    def drive(r):
        try:
            rt = ranks[r]
            for i, v in enumerate(views):
                W.set_view(rt, v)
                rt.dist_frame()
                names[r].append(rt.kernel_name())
                if i in reads:
                    rt.dist_wait()        # (lets the box of the occupied cells reach the host between frames, as a renderer's frames do)
                    if r == 0:
                        got.append((v, rt.dist_read_frame().copy()))
            rt.dist_wait()
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=drive, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=180)
    assert not any(t.is_alive() for t in threads), "a rank hung"
    assert not errors, errors
    for rt in ranks:
        rt.deinit()
    assert len(got) == len(reads)
    for v, frame in got:
        assert np.array_equal(frame, ref[v]), f"assembled frame of view {v} differs from the single-context frame"
    for r in range(world):
        assert all(n.startswith(("vrt_path_kernel<8, 5,", "vrt_pool_kernel<8, 6,")) for n in names[r]), names[r]
        if reads[0] < len(views) - 1:     # (a wait before the last frame: the box has reached the host by then; without one it is a race)
            assert names[r][-1] == "vrt_pool_kernel<8, 6, 60, 2>", names[r]


def test_bounce_frames_without_a_sample_buffer_keep_the_lockstep_kernel_in_the_pipeline():
    """VRT_TUNE_NO_SAMPLE_UNITS: vrt_path_kernel would store whole RGBA pixels from whichever lane finished them — not a shard the
    gather can carry — so the pipeline substitutes the lockstep kernel, as it did for every bounce frame until round 4."""
    if not os.path.exists(FAKE):
        pytest.skip("tests/fake_rccl/libfake_rccl.so not built (run __graft_entry__.build())")
    from zig_vulkan_amd import _lib as L
    path = 1 << 23
    w, grid = _bounce_scene()
    views = ["V0", "V1x", "V2"]
    ref, _ = _single_context_frames(w, grid, views, path)
    world = 2
    uid = b"fake-rccl-lock" + os.urandom(16) + bytes(128 - 30)
    ranks = [W.make_renderer(w, grid, kernel_variant=path, shard_rank=r, shard_count=world, tuning_flags=L.TUNE_NO_SAMPLE_UNITS) for r in range(world)]
    for r, rt in enumerate(ranks):
        rt.dist_init(uid, r, world, frames_in_flight=2, rccl_path=FAKE, frames_per_launch=1)
    got, errors = [], []

    def drive(r):
        try:
            rt = ranks[r]
            for v in views:
                W.set_view(rt, v)
                rt.dist_frame()
                rt.dist_wait()
                if r == 0:
                    got.append((v, rt.dist_read_frame().copy(), rt.kernel_name()))
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=drive, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a rank hung"
    assert not errors, errors
    for rt in ranks:
        rt.deinit()
    assert len(got) == len(views)
    for v, frame, name in got:
        assert name.startswith("vrt_trace_kernel<8, false,"), name
        assert np.array_equal(frame, ref[v]), f"assembled frame of view {v} differs from the single-context frame"


def test_reserve_samples_on_a_sharded_cache_resident_bounce_context_has_nothing_to_reserve():
    """a cache-resident power-of-two scene with bounces (the reference app's run in shape) has ONE persistent kernel, the
    auto-tune's candidate — which contexts of the pipeline never run, so their launch slots have no lanes.  vrt_reserve_samples used to
    answer VRT_E_OOM there (tools/dist_emulate.py and bench.py's secondary leg call it whenever max_bounce > 0); the header promises
    VRT_OK for 'nothing to reserve'.  The frames still come out as the single context's, on the lockstep kernel."""
    if not os.path.exists(FAKE):
        pytest.skip("tests/fake_rccl/libfake_rccl.so not built (run __graft_entry__.build())")
    w = W.Workload("t", 330, 210, 256, 4, 2, 2, True, 5.0, dims=(64, 32, 64))
    grid = W.build_grid(w)
    plain = W.make_renderer(w, grid)
    W.set_view(plain, "V1")
    plain.draw()
    ref = plain.read_rgba8().copy()
    plain.deinit()
    world = 2
    uid = b"fake-rccl-reserve" + os.urandom(16) + bytes(128 - 33)
    ranks = [W.make_renderer(w, grid, shard_rank=r, shard_count=world) for r in range(world)]
    for r, rt in enumerate(ranks):
        rt.dist_init(uid, r, world, frames_in_flight=2, rccl_path=FAKE, frames_per_launch=1)
        rt.reserve_samples(w.spp)          # (raised VRT_E_OOM before the fix)
    errors, names = [], []

    def drive(r):
        try:
            rt = ranks[r]
            for _ in range(6):             # (more frames than the auto-tune's four trials: none may be one inside the pipeline)
                W.set_view(rt, "V1")
                rt.dist_frame()
                names.append(rt.kernel_name())
            rt.dist_wait()
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=drive, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a rank hung"
    assert not errors, errors
    got = ranks[0].dist_read_frame().copy()
    for rt in ranks:
        rt.deinit()
    assert all(n.startswith("vrt_trace_kernel<4, false,") for n in names), names
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("world,frames_per_launch,bounce", [(2, 1, False), (4, 8, False), (8, 1, False), (2, 1, True)])
def test_frames_submitted_by_one_call(world, frames_per_launch, bounce):
    """vrt_dist_frames (a C-side multi-frame submit for the pipeline): n cameras, one call across the ABI; the frames
    are those of n vrt_dist_frame calls."""
    if not os.path.exists(FAKE):
        pytest.skip("tests/fake_rccl/libfake_rccl.so not built (run __graft_entry__.build())")
    if bounce:
        w, grid = _bounce_scene()
        variant = 1 << 23
    else:
        w, grid, variant = W.Workload("t", 332, 210, 64, 4, 1, 0, True, 0.0), None, 0
        grid = W.build_grid(w)
    views = ["V0", "V1", "V2", "V1", "V2", "V0", "V0", "V2", "V1", "V1", "V0", "V2", "V0", "V1", "V2", "V2"]
    plain = W.make_renderer(w, grid, kernel_variant=variant)
    W.set_view(plain, views[-1])
    plain.draw()
    ref = plain.read_rgba8().copy()
    plain.deinit()
    uid = b"fake-rccl-many" + bytes([world, frames_per_launch]) + os.urandom(16) + bytes(128 - 32)
    ranks = [W.make_renderer(w, grid, kernel_variant=variant, shard_rank=r, shard_count=world) for r in range(world)]
    cams = []
    for v in views:
        W.set_view(ranks[0], v)
        cams.append(bytes(ranks[0].camera.d_camera))
    for r, rt in enumerate(ranks):
        rt.dist_init(uid, r, world, frames_in_flight=4, rccl_path=FAKE, frames_per_launch=frames_per_launch)
    errors = []

    def drive(r):
        try:
            ranks[r].dist_frames(cams[:5])
            ranks[r].dist_frames(cams[5:])
            ranks[r].dist_wait()
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=drive, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads) and not errors, errors
    got = ranks[0].dist_read_frame()
    for rt in ranks:
        rt.deinit()
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("order", [("one", "bounce", "two", "one"), ("two", "one", "one", "bounce"), ("bounce", "bounce", "one", "two")])
def test_a_batch_may_hold_frames_of_several_kernels(order):
    """A launch carries a collective, so WHEN a queue is launched must never depend on anything a rank learns for itself (the box of the
    occupied cells arrives at another moment on every rank) — and it must not depend on the kernels of the queued frames either: frames of
    one sample, of two samples and with bounces (three kernels; the bounce frames by the persistent kernels, one launch each) share
    batches of four, launched run by run, gathered by ONE collective.  The last frame of the sequence is checked for each order."""
    if not os.path.exists(FAKE):
        pytest.skip("tests/fake_rccl/libfake_rccl.so not built (run __graft_entry__.build())")
    w, grid = _bounce_scene()
    kinds = {"one": (1, 0), "two": (2, 0), "bounce": (2, 2)}     # (samples per pixel, Camera.Config.max_bounce: the device value is + 1)
    world = 3

    def set_kind(rt, kind):
        spp, bounce = kinds[kind]
        rt.camera.d_camera.samples_per_pixel = spp
        rt.camera.d_camera.max_bounce = bounce + 1

    plain = W.make_renderer(w, grid, kernel_variant=1 << 23)
    W.set_view(plain, "V2")
    set_kind(plain, order[-1])
    plain.draw()
    plain.wait()
    plain.draw()
    ref = plain.read_rgba8().copy()
    plain.deinit()
    uid = b"fake-rccl-mix" + os.urandom(16) + bytes(128 - 29)
    ranks = [W.make_renderer(w, grid, kernel_variant=1 << 23, shard_rank=r, shard_count=world) for r in range(world)]
    for r, rt in enumerate(ranks):
        W.set_view(rt, "V2")
        rt.dist_init(uid, r, world, frames_in_flight=2, rccl_path=FAKE, frames_per_launch=4)
    errors, names = [], []

    def drive(r):
        try:
            rt = ranks[r]
            for _ in range(3):           # (three batches: by the second the box of the occupied cells is known, each rank in its own time)
                for kind in order:
                    set_kind(rt, kind)
                    rt.dist_frame()
                    if r == 0:
                        names.append(rt.kernel_name().split("<")[0])
            rt.dist_wait()
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=drive, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads) and not errors, errors
    got = ranks[0].dist_read_frame()
    for rt in ranks:
        rt.deinit()
    assert np.array_equal(got, ref)
    assert {"vrt_trace_kernel"} < set(names) <= {"vrt_trace_kernel", "vrt_path_kernel", "vrt_pool_kernel"}, names


@pytest.mark.parametrize("fixture,world,root_weight,frames_per_launch", [("cfg3_V2", 8, 30, 1), ("cfg3_V2", 4, 0, 8), ("cfg4_V1", 8, 18, 1)])
def test_sharded_frames_of_the_baseline_configurations_at_full_size_are_the_oracles(fixture, world, root_weight, frames_per_launch):
    """VERDICT r04 ("the sharded form of configs[3] / [4] has only ever run ... at <= 332 x 210"): BASELINE configs[3] (4K, 1024^3, 4 rays
    per pixel) and configs[4] (4K, 2048^3 sparse, 16 spp path trace) THROUGH the multi-GPU pipeline at full size — 8 (4) ranks as threads
    over tests/fake_rccl, each a context with its own replica of the scene, a weighted root share, one collective per frame (and eight
    frames per collective) — and the frame rank 0 assembles is, byte for byte, the oracle's whole frame (the RGBA8 digest of
    tests/golden/full/, made by tests/golden/make_full_golden.py).  configs[4]: both frames — traced by vrt_path_kernel<..., DIL 1> until
    the ranks know the box of the occupied cells, by vrt_pool_kernel behind that."""
    import hashlib
    from tests.golden.make_golden import scene_digest
    from tests.helpers import O
    if not os.path.exists(FAKE):
        pytest.skip("tests/fake_rccl/libfake_rccl.so not built (run __graft_entry__.build())")
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "full", fixture + ".npz"))
    w = W.WORKLOADS[str(z["workload"])]
    grid = W.build_grid(w)
    assert scene_digest(grid) == str(z["scene_sha256"])
    ranks = [W.make_renderer(w, grid, shard_rank=r, shard_count=world, shard_root_weight=root_weight) for r in range(world)]
    uid = b"fake-rccl-full" + bytes([world, frames_per_launch]) + os.urandom(16) + bytes(128 - 32)
    for r, rt in enumerate(ranks):
        rt.dist_init(uid, r, world, frames_in_flight=2, rccl_path=FAKE, frames_per_launch=frames_per_launch)
        W.set_view(rt, str(z["view"]))
    assert O.push_constants(ranks[0].camera.blob(), ranks[0].sun.blob()).tobytes() == z["push_constants"].tobytes()
    batches = 2 if w.max_bounce > 0 else 1
    got, errors, names = [], [], {r: [] for r in range(world)}

    def drive(r):
        try:
            rt = ranks[r]
            for _ in range(batches):
                for _ in range(frames_per_launch):
                    rt.dist_frame()
                names[r].append(rt.kernel_name())
                rt.dist_wait()            # (a renderer's frames: by the next one the box of the occupied cells has reached the host)
                if r == 0:
                    got.append(hashlib.sha256(np.ascontiguousarray(rt.dist_read_frame()).tobytes()).hexdigest())
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=drive, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), "a rank hung"
    assert not errors, errors
    for rt in ranks:
        rt.deinit()
    assert got == [str(z["rgba8_sha256"])] * batches, (got, str(z["rgba8_sha256"]))
    if str(z["workload"]).startswith("cfg4"):
        for r in range(world):
            # (the first frame: vrt_path_kernel<..., DIL 1> unless the box of the occupied cells has reached the host before it — eight ranks
            # now initialise their communicators one after the other, which can take that long; the second: the pool kernel)
            assert names[r][0] in ("vrt_path_kernel<8, 5, false, false, false, false, 1>", "vrt_pool_kernel<8, 6, 60, 2>") and names[r][1] == "vrt_pool_kernel<8, 6, 60, 2>", names[r]
