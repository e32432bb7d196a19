// vrt_api.hip — implementation of the C ABI in include/vrt_hip.h: context
// (device buffers, stream, events), stream-ordered uploads through a pinned
// staging ring, dispatch of the traversal kernel, read-back, timing.
//
// Replaces src/modules/voxel_rt/ComputePipeline.zig (init / dispatch / deinit)
// and the Pipeline.transfer* family (Pipeline.zig:560-652) with its StagingRamp
// (render/StagingRamp.zig) for this one path.  There is no CPU fallback: without
// a HIP device vrt_create fails with VRT_E_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h> // types and prototypes only: the library is reached through dlopen, not linked
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>
#include <thread>
#include "host_brick_grid.hpp"
#include "vrt_internal.h"
#include "vrt_kernels.h"

namespace vrt {
KernelFn select_trace_kernel(int brick_dimension, bool counters, uint32_t variant, int shade);
KernelFn path_kernel_halfblock_twin(KernelFn fn);
KernelFn path_kernel_ahead_twin(KernelFn fn);
KernelFn path_kernel_dist_twin(KernelFn fn);
KernelFn path_kernel_dilated_twin(KernelFn fn, int kind);
int path_kernel_dilated_kind(KernelFn fn);
bool is_path_halfblock_kernel(KernelFn fn);
const char *kernel_name_of(KernelFn fn);
uint32_t resolve_variant(uint32_t variant);
size_t trace_lds_bytes(const TraceParams &p, uint32_t variant);
hipError_t launch_trace(KernelFn fn, const TraceParams &p, size_t lds_bytes, hipStream_t stream, uint32_t frames = 1);
bool is_path_kernel(KernelFn fn);
hipError_t launch_schedule(const uint32_t *cost, uint32_t *snap, const uint32_t *prev_order, uint32_t *order, uint32_t n, uint32_t extra_max, uint32_t extra, uint32_t wave_slots,
                           hipStream_t stream);
hipError_t launch_assemble_rgb(const void *gathered, void *frame, uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t shard_count,
                               uint32_t tiles_per_rank, const TileOwnership &own, hipStream_t stream, uint32_t frames, uint32_t frame_src_stride_bytes);
hipError_t launch_build_status_blocks(const TraceParams &p, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, hipStream_t stream);
hipError_t launch_build_status_bytes(const TraceParams &p, hipStream_t stream);
hipError_t launch_build_status_halfblocks(const TraceParams &p, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, hipStream_t stream);
hipError_t launch_build_cell_distance(const TraceParams &p, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, hipStream_t stream);
hipError_t launch_build_cell_bounds(const TraceParams &p, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, hipStream_t stream);
hipError_t launch_build_cell_occupancy(const TraceParams &p, uint32_t brick_dimension, uint64_t brick_alloc, uint64_t cell_lo, uint64_t cell_hi, uint64_t slot_lo,
                                       uint64_t slot_hi, hipStream_t stream);
hipError_t launch_check_start_is_slot(const TraceParams &p, uint32_t brick_dimension, uint64_t brick_alloc, hipStream_t stream);
hipError_t launch_check_materials_plain(const TraceParams &p, uint32_t count, hipStream_t stream);
hipError_t launch_denoise(const void *img, int W, int H, int samples, float bias, float mult, float tol, int out_w, int out_h, void *out_u8,
                          void *out_f32, hipStream_t stream);
hipError_t launch_assemble(const void *gathered, void *frame, uint32_t bytes_per_pixel, uint32_t width, uint32_t height, uint32_t tiles_x,
                           uint32_t shard_count, uint32_t tiles_per_rank, const TileOwnership &own, hipStream_t stream, uint32_t frames = 1,
                           uint32_t frame_src_stride_pixels = 0);
} // namespace vrt

namespace {
thread_local std::string g_create_error;

constexpr size_t kStagingSlotBytes = 32u << 20; // pinned staging slot
constexpr int kStagingSlots = 2;
} // namespace

// RCCL entry points resolved with dlsym from the library the host process already uses.
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr; // optional: replica updates fall back to send / recv from the root
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;       // optional (vrt_dist_info)
    decltype(&ncclCommUserRank) CommUserRank = nullptr; // optional
    bool load(const char *path, std::string &err) {
        lib = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
        if (!lib) {
            err = std::string("dlopen(") + (path ? path : "NULL") + "): " + dlerror();
            return false;
        }
#define VRT_RCCL_SYM(field, name)                                   \
        field = reinterpret_cast<decltype(field)>(dlsym(lib, name)); \
        if (!field) {                                                \
            err = std::string("dlsym ") + name + " failed";         \
            return false;                                            \
        }
        VRT_RCCL_SYM(GetUniqueId, "ncclGetUniqueId")
        VRT_RCCL_SYM(CommInitRank, "ncclCommInitRank")
        VRT_RCCL_SYM(CommDestroy, "ncclCommDestroy")
        VRT_RCCL_SYM(GroupStart, "ncclGroupStart")
        VRT_RCCL_SYM(GroupEnd, "ncclGroupEnd")
        VRT_RCCL_SYM(Send, "ncclSend")
        VRT_RCCL_SYM(Recv, "ncclRecv")
        VRT_RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef VRT_RCCL_SYM
        Broadcast = reinterpret_cast<decltype(Broadcast)>(dlsym(lib, "ncclBroadcast"));
        CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(lib, "ncclCommCount"));
        CommUserRank = reinterpret_cast<decltype(CommUserRank)>(dlsym(lib, "ncclCommUserRank"));
        return true;
    }
};

constexpr uint32_t kMaxDistSlots = 8;

// One launch in flight of the multi-GPU pipeline: its stream carries kernel -> gather -> un-swizzle for a batch of up to
// `batch` consecutive frames (see Dist).
struct DistSlot {
    hipStream_t stream = nullptr;
    uint8_t *shard = nullptr;    // this rank's packed tiles, frame-major: batch x shard_bytes (on rank 0: region 0 of `gathered`)
    uint8_t *gathered = nullptr; // rank 0: world x batch x shard_bytes, rank-major then frame-major
    uint8_t *frame = nullptr;    // rank 0: batch row-major RGBA8 frames
    hipEvent_t done = nullptr;
    uint64_t seen_upload = 0;
    uint32_t frames = 0;         // frames of the batch this slot holds
    bool used = false;
    // vrt_dist_profile: events around the three stages of the slot's most recent launch (kernel | collective | un-swizzle)
    hipEvent_t mark[4] = {};
    bool marked = false;         // the most recent launch recorded its marks and they have not been read yet
};

struct Dist {
    RcclApi api;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    uint32_t nslots = 0;
    DistSlot slots[kMaxDistSlots];
    uint64_t frame_no = 0;       // batches launched so far (slot = frame_no % nslots)
    int last_slot = -1;
    size_t shard_bytes = 0;
    // Frames are traced `batch` to a launch (grid.y): a rank owns 1/world of the tiles, too few waves to fill the GPU
    // and no shorter than the frame's longest wave, so single-frame launches leave most of the machine idle
    // (tools/shard_streams.py: 19-31 us per 1/8 frame with eight single-frame launches in flight, against 8-18 us for
    // an eighth of a whole-frame launch).  vrt_dist_frame queues; a full queue, vrt_dist_wait, vrt_dist_read_frame or a
    // scene upload launches what is queued.
    uint32_t batch = 1;
    bool failed = false;         // a collective failed: peers are out of step, every later vrt_dist_* call fails
    uint32_t npend = 0;
    vrt::PushConstants pend[vrt::kMaxBatchFrames];
    vrt::KernelFn pend_fn = nullptr;
    // vrt_dist_profile / vrt_dist_stats: per-launch stage times, summed over the launches sampled
    bool profile = false;
    uint64_t prof_launches = 0, prof_frames = 0;
    double prof_ms[3] = {0.0, 0.0, 0.0}; // kernel, collective, un-swizzle
};

struct vrt_ctx {
    vrt_config cfg{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // second frame slot (frames_in_flight == 2): own stream + own target images
    uint32_t frames_in_flight = 1;
    hipStream_t stream_b = nullptr;
    uint8_t *target8_b = nullptr;
    float *target32f_b = nullptr;
    hipEvent_t ev_b_done = nullptr, ev_upload = nullptr;
    bool b_pending = false;          // stream_b has frames the primary stream has not been ordered after
    uint64_t upload_seq = 0, b_seen_upload = 0;
    uint32_t frame_seq = 0;
    int last_slot = 0;
    void *dbuf[VRT_BUF_COUNT] = {};
    uint64_t dsize[VRT_BUF_COUNT] = {};
    uint8_t *target8 = nullptr;
    float *target32f = nullptr;
    bool own_t8 = false, own_t32 = false;
    uint64_t target_pixels = 0; // pixels in the (possibly sharded, padded) target
    vrt::DeviceCounters *d_counters = nullptr;
    uint32_t *d_tile_cost = nullptr, *d_tile_schedule = nullptr; // cost-feedback tile schedule (two order buffers + snapshot)
    // amortised cost-feedback schedule (tile_order 7): re-sorted every sched_period frames into the other buffer
    uint32_t sched_period = 0, sched_since = 0, sched_cur = 0;
    bool order_auto = false; // kernel_variant left the tile order to the library
    uint32_t bounce_variant = 0; // kernel_variant with the occupancy choice of the bounce kernel filled in
    uint32_t single_variant = 0; // kernel_variant with the library's choice of mode for frames without bounces filled in
    uint32_t tile_order = 0, sched_extra = 0, sched_stride = 0, wave_slots = 0;
    // the cost schedule's two rules (index 1: frames whose split tiles trace their second sample on the idle lanes — two samples per
    // pixel — where a split costs nothing but the second workgroup's fixed part): how many tiles an order may split, and the wave
    // slots the "time the frame needs anyway" is computed for; sched_mode: the rule the current order was sorted under
    uint32_t sched_cap[2] = {0, 0}, sched_slots[2] = {0, 0}, sched_mode = 0;
    uint64_t sched_seq = 0, b_seen_sched = 0;
    hipEvent_t ev_sched = nullptr, ev_b_sched = nullptr;
    bool b_sched_recorded = false;
    void *d_denoised8 = nullptr, *d_denoised32f = nullptr;       // output of the present/denoise pass
    struct Dist *dist = nullptr;                                 // multi-GPU frame pipeline (vrt_dist_*)
    uint32_t denoised_w = 0, denoised_h = 0;
    hipStream_t denoised_stream = nullptr;
    void *d_status_blocks = nullptr; // derived: 4x4x4 block words + block filter (vrt_trace.hip)
    int *d_cell_bounds = nullptr;    // derived: bounding box of the occupied cells (TraceParams::cell_bounds)
    // The host's copy of that box (read back behind every rebuild, never waited for): when it is, or nearly is, the grid, bounce
    // frames of a context whose kernel is the dilated-index path kernel are traced by its twin without steps-left counters.
    int *h_cell_bounds = nullptr;
    hipEvent_t ev_bounds = nullptr;
    bool bounds_pending = false, box_is_grid = false;
    vrt::KernelFn kernel_grid_exit = nullptr, product_grid_exit = nullptr;
    uint8_t *d_status_bytes = nullptr; // derived: one byte per grid cell (TraceParams::status_bytes)
    uint8_t *d_cell_distance = nullptr;      // derived: L1 distance of every cell to the nearest occupied cell (vrt_path_kernel<DIST>)
    uint32_t *d_status_halfblocks = nullptr; // derived: status bits by 4 x 4 x 2 cells per word (vrt_path_kernel on eligible grids)
    bool cell_occupancy_lockstep = false;    // ... read by the lockstep bounce kernel too (scenes that stay in the caches)
    uint8_t *d_cell_occupancy = nullptr;     // derived: occupancy bits by cell (TraceParams::cell_occupancy; vrt_path_kernel, within a memory budget)
    uint32_t *d_start_is_slot = nullptr;     // derived: 1 = binding 6 holds slot * B^3 for every allocated brick (TraceParams::start_is_slot)
    bool occupancy_dirty = true;             // bindings 3-5 changed since the by-cell copy was built ...
    // ... in these ranges (the reference issues a single-brick delta every frame, VoxelRT.zig:107-172; the copy is then
    // refreshed for the cells and brick slots it names, not gathered anew over the whole grid): cells whose status bit / brick index
    // changed and brick slots whose occupancy bytes changed, both [lo, hi); lo >= hi: none
    uint64_t occ_cell_lo = 0, occ_cell_hi = ~0ull, occ_slot_lo = 0, occ_slot_hi = 0;
    uint32_t *d_materials_plain = nullptr;   // derived: 1 = no material record has the type MAT_NONE (TraceParams::materials_plain)
    bool materials_dirty = true;             // binding 0 changed since it was checked
    bool start_dirty = true;                 // binding 6 changed since it was checked
    vrt::TileOwnership own{};        // weighted tile ownership (period 0: tile t belongs to rank t % shard_count)
    bool status_dirty = true;        // brick_status changed since the derived copy was built
    size_t lds_bytes = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool split_ok = false;   // small frames may go to half-tile workgroups (vrt_create's conditions other than the number of waves)
    uint32_t simds = 1024u;
    hipEvent_t ev_region[4] = {}; // vrt_region_begin / _end: {begin, end} on the primary stream, {begin, end} on the second
    hipEvent_t ev_post_start = nullptr, ev_post_stop = nullptr; // around the most recent present / denoise pass (vrt_last_denoise_ms)
    bool post_timed = false;
    bool in_flight = false;
    bool timing_valid = false;
    uint32_t timed_frames = 0;
    double last_ms = -1.0;
    void *staging[kStagingSlots] = {};
    hipEvent_t staging_ev[kStagingSlots] = {};
    bool staging_busy[kStagingSlots] = {};
    int staging_next = 0;
    vrt::TraceParams params{};
    vrt::KernelFn kernel = nullptr;        // frames with bounces: persistent lanes (vrt_path_kernel) unless kernel_variant bit 21
    vrt::KernelFn kernel_lockstep = nullptr; // ... the lockstep bounce loop (always used by the multi-GPU pipeline: RGB shards)
    uint32_t *d_work_counter = nullptr;    // vrt_path_kernel's pixel counters: [2 streams][kMaxBatchFrames]
    uint32_t *d_pool_paths = nullptr;      // vrt_pool_kernel's path records: [2 streams][pool_groups * 4 waves][16 dwords][128 paths]
    size_t pool_stream_dwords = 0;
    float4 *d_pool_samples = nullptr;      // vrt_pool_kernel -> vrt_pool_resolve_kernel: [2 streams][owned pixels][samples] terms of the sample sum, sized by the frames asked for
    size_t pool_samples_stream_elems = 0;
    uint32_t path_lds_bytes = 0;           // LDS block filter of vrt_path_kernel (0: grid not eligible)
    vrt::KernelFn kernel_single = nullptr; // specialisation for max_bounce <= 1
    vrt::KernelFn kernel_single1 = nullptr; // ... and samples_per_pixel == 1
    vrt::KernelFn product[3] = {};         // counting contexts: the product kernel that renders the frame read back, by shade (0 bounces, 1, 2)
    vrt::KernelFn last_fn = nullptr;       // the kernel of the most recent frame (vrt_kernel_name)
    vrt_shard_info shard{};
    std::string err;
    std::string kernel_name, name_note;
};

namespace {

int fail(vrt_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    else g_create_error = msg;
    return code;
}

// remember which kernel rendered the most recent frame (vrt_kernel_name reports what ran, not what was asked for)
void note_kernel(vrt_ctx *c, vrt::KernelFn fn) {
    if (fn == c->last_fn) return;
    c->last_fn = fn;
    c->kernel_name = std::string(vrt::kernel_name_of(fn)) + c->name_note;
}

int hip_fail(vrt_ctx *ctx, hipError_t e, const char *what) {
    return fail(ctx, e == hipErrorOutOfMemory ? VRT_E_OOM : VRT_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define VRT_HIP(ctx, call)                                  \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

void free_ctx(vrt_ctx *c) {
    if (!c) return;
    DeviceGuard dg(c->device); // the caller's current device is restored on return
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < VRT_BUF_COUNT; i++)
        if (c->dbuf[i]) (void)hipFree(c->dbuf[i]);
    if (c->own_t8 && c->target8) (void)hipFree(c->target8);
    if (c->own_t32 && c->target32f) (void)hipFree(c->target32f);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->d_work_counter) (void)hipFree(c->d_work_counter);
    if (c->d_pool_paths) (void)hipFree(c->d_pool_paths);
    if (c->d_pool_samples) (void)hipFree(c->d_pool_samples);
    if (c->d_status_blocks) (void)hipFree(c->d_status_blocks);
    if (c->d_cell_bounds) (void)hipFree(c->d_cell_bounds);
    if (c->h_cell_bounds) (void)hipHostFree(c->h_cell_bounds);
    if (c->ev_bounds) (void)hipEventDestroy(c->ev_bounds);
    if (c->d_status_bytes) (void)hipFree(c->d_status_bytes);
    if (c->d_status_halfblocks) (void)hipFree(c->d_status_halfblocks);
    if (c->d_cell_distance) (void)hipFree(c->d_cell_distance);
    if (c->d_cell_occupancy) (void)hipFree(c->d_cell_occupancy);
    if (c->d_start_is_slot) (void)hipFree(c->d_start_is_slot);
    if (c->d_materials_plain) (void)hipFree(c->d_materials_plain);
    if (c->dist) {
        Dist *d = c->dist;
        for (uint32_t i = 0; i < d->nslots; i++) {
            DistSlot &sl = d->slots[i];
            if (sl.stream) {
                (void)hipStreamSynchronize(sl.stream);
                (void)hipStreamDestroy(sl.stream);
            }
            if (sl.gathered) (void)hipFree(sl.gathered);
            else if (sl.shard) (void)hipFree(sl.shard);
            if (sl.frame) (void)hipFree(sl.frame);
            if (sl.done) (void)hipEventDestroy(sl.done);
            for (hipEvent_t e : sl.mark)
                if (e) (void)hipEventDestroy(e);
        }
        if (d->comm && d->api.CommDestroy) (void)d->api.CommDestroy(d->comm);
        delete d;
        c->dist = nullptr;
    }
    if (c->d_denoised8) (void)hipFree(c->d_denoised8);
    if (c->d_denoised32f) (void)hipFree(c->d_denoised32f);
    if (c->d_tile_cost) (void)hipFree(c->d_tile_cost);
    if (c->d_tile_schedule) (void)hipFree(c->d_tile_schedule);
    if (c->ev_sched) (void)hipEventDestroy(c->ev_sched);
    if (c->ev_b_sched) (void)hipEventDestroy(c->ev_b_sched);
    for (int i = 0; i < kStagingSlots; i++) {
        if (c->staging[i]) (void)hipHostFree(c->staging[i]);
        if (c->staging_ev[i]) (void)hipEventDestroy(c->staging_ev[i]);
    }
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    for (hipEvent_t e : c->ev_region)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_post_start) (void)hipEventDestroy(c->ev_post_start);
    if (c->ev_post_stop) (void)hipEventDestroy(c->ev_post_stop);
    if (c->stream_b) {
        (void)hipStreamSynchronize(c->stream_b);
        (void)hipStreamDestroy(c->stream_b);
    }
    if (c->target8_b) (void)hipFree(c->target8_b);
    if (c->target32f_b) (void)hipFree(c->target32f_b);
    if (c->ev_b_done) (void)hipEventDestroy(c->ev_b_done);
    if (c->ev_upload) (void)hipEventDestroy(c->ev_upload);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// Waiting for a stream / an event: poll for up to a few milliseconds before handing the thread to the runtime's blocking wait.
// The blocking wait sleeps on an interrupt and wakes up tens of microseconds after the GPU has finished — as long as a whole
// frame of the headline workload (tools/short_trace.py: a 20-frame region took 1.45 ms on the GPU and 1.59 ms on the host's clock).
template <typename Query>
hipError_t poll_then(Query query) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        for (int i = 0; i < 64; i++) {
            const hipError_t e = query();
            if (e != hipErrorNotReady) return e;
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) return hipErrorNotReady; // long frame: sleep instead
    }
}
hipError_t wait_stream(hipStream_t s) {
    const hipError_t e = poll_then([&] { return hipStreamQuery(s); });
    return e == hipErrorNotReady ? hipStreamSynchronize(s) : e;
}
hipError_t wait_event(hipEvent_t ev) {
    const hipError_t e = poll_then([&] { return hipEventQuery(ev); });
    return e == hipErrorNotReady ? hipEventSynchronize(ev) : e;
}

// wait for the frame in flight (the fence wait of ComputePipeline.zig:423-434)
int finish_frame(vrt_ctx *c) {
    if (!c->in_flight) return VRT_OK;
    VRT_HIP(c, wait_event(c->ev_stop));
    float ms = 0.0f;
    VRT_HIP(c, hipEventElapsedTime(&ms, c->ev_start, c->ev_stop));
    c->last_ms = (double)ms / (double)(c->timed_frames ? c->timed_frames : 1u);
    c->timing_valid = true;
    c->in_flight = false;
    return VRT_OK;
}

// Scene writes happen on the primary stream.  With two frames in flight they must not overtake a frame
// that is still reading the scene on stream_b, and later frames on stream_b must see them.
int dist_flush(vrt_ctx *ctx);
int begin_scene_write(vrt_ctx *c) {
    if (c->dist && c->dist->npend) { // frames queued before this write must see the scene as it was
        const int rcf = dist_flush(c);
        if (rcf != VRT_OK) return rcf;
    }
    if (c->stream_b && c->b_pending) {
        VRT_HIP(c, hipStreamWaitEvent(c->stream, c->ev_b_done, 0));
        c->b_pending = false;
    }
    if (c->dist) {
        for (uint32_t i = 0; i < c->dist->nslots; i++)
            if (c->dist->slots[i].used) VRT_HIP(c, hipStreamWaitEvent(c->stream, c->dist->slots[i].done, 0));
    }
    return VRT_OK;
}
int end_scene_write(vrt_ctx *c) {
    if (c->stream_b || c->dist) {
        VRT_HIP(c, hipEventRecord(c->ev_upload, c->stream));
        c->upload_seq++;
    }
    return VRT_OK;
}

// Host copy into the pinned slot.  One core moves ~20 GB/s, a third of the PCIe Gen5 x16 link the DMA
// that follows can use, so large pieces are split over a few short-lived threads.
void staging_copy(void *dst, const uint8_t *src, size_t n) {
    constexpr size_t kParallelFrom = 8u << 20;
    constexpr unsigned kThreads = 4;
    if (n < kParallelFrom) {
        std::memcpy(dst, src, n);
        return;
    }
    const size_t piece = ((n / kThreads) + 4095u) & ~(size_t)4095u;
    std::thread workers[kThreads - 1];
    unsigned started = 0;
    for (unsigned t = 1; t < kThreads; t++) {
        const size_t off = piece * t;
        if (off >= n) break;
        const size_t len = (off + piece <= n) ? piece : n - off;
        workers[started++] = std::thread([=] { std::memcpy(static_cast<uint8_t *>(dst) + off, src + off, len); });
    }
    std::memcpy(dst, src, piece < n ? piece : n);
    for (unsigned t = 0; t < started; t++) workers[t].join();
}

int copy_h2d(vrt_ctx *c, void *dst, const void *src, uint64_t nbytes) {
    const uint8_t *s = static_cast<const uint8_t *>(src);
    uint8_t *d = static_cast<uint8_t *>(dst);
    int rc0 = begin_scene_write(c);
    if (rc0 != VRT_OK) return rc0;
    while (nbytes) {
        const int slot = c->staging_next;
        c->staging_next = (slot + 1) % kStagingSlots;
        if (c->staging_busy[slot]) {
            VRT_HIP(c, hipEventSynchronize(c->staging_ev[slot]));
            c->staging_busy[slot] = false;
        }
        const size_t n = nbytes < kStagingSlotBytes ? (size_t)nbytes : kStagingSlotBytes;
        staging_copy(c->staging[slot], s, n);
        VRT_HIP(c, hipMemcpyAsync(d, c->staging[slot], n, hipMemcpyHostToDevice, c->stream));
        VRT_HIP(c, hipEventRecord(c->staging_ev[slot], c->stream));
        c->staging_busy[slot] = true;
        s += n;
        d += n;
        nbytes -= n;
    }
    return end_scene_write(c);
}

} // namespace

extern "C" {

uint32_t vrt_abi_version(void) { return VRT_ABI_VERSION; }

int vrt_device_info(int device, int64_t out[4]) {
    if (!out) return VRT_E_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, VRT_E_NO_DEVICE, "no HIP device");
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return fail(nullptr, VRT_E_HIP, "hipGetDevice failed");
    if (device >= ndev) return fail(nullptr, VRT_E_INVALID_ARG, "device out of range");
    const hipDeviceAttribute_t attrs[4] = {hipDeviceAttributeClockRate, hipDeviceAttributeMultiprocessorCount, hipDeviceAttributeWarpSize,
                                           hipDeviceAttributeL2CacheSize};
    for (int i = 0; i < 4; i++) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, attrs[i], device) != hipSuccess) return fail(nullptr, VRT_E_HIP, "hipDeviceGetAttribute failed");
        out[i] = v;
    }
    return VRT_OK;
}

const char *vrt_last_error(const vrt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

const char *vrt_kernel_name(const vrt_ctx *ctx) { return ctx ? ctx->kernel_name.c_str() : ""; }

int vrt_compiled_kernel_count(void) { return vrt::compiled_kernel_count(); }

int vrt_create(const vrt_config *cfg, vrt_ctx **out) {
    if (!out) return fail(nullptr, VRT_E_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!cfg) return fail(nullptr, VRT_E_INVALID_ARG, "cfg is NULL");
    if (cfg->struct_size != sizeof(vrt_config) || cfg->abi_version != VRT_ABI_VERSION)
        return fail(nullptr, VRT_E_INVALID_ARG, "vrt_config size/ABI version mismatch");
    if (cfg->width == 0 || cfg->height == 0) return fail(nullptr, VRT_E_INVALID_ARG, "zero image size");
    if (cfg->brick_dimension != 4 && cfg->brick_dimension != 8) return fail(nullptr, VRT_E_INVALID_ARG, "brick_dimension must be 4 or 8");
    const uint64_t cells = (uint64_t)cfg->dim_x * cfg->dim_y * cfg->dim_z;
    if (cells == 0) return fail(nullptr, VRT_E_INVALID_ARG, "zero grid dimension");
    if (cells > 0xFFFFFFFFull) return fail(nullptr, VRT_E_OUT_OF_RANGE, "grid has more than 2^32-1 cells (u32 grid index, comp:318)");
    if ((uint64_t)cfg->dim_x * cfg->brick_dimension > 0xFFFFFFFFull || (uint64_t)cfg->dim_y * cfg->brick_dimension > 0xFFFFFFFFull ||
        (uint64_t)cfg->dim_z * cfg->brick_dimension > 0xFFFFFFFFull)
        return fail(nullptr, VRT_E_OUT_OF_RANGE, "dim * brick_dimension exceeds u32 (State.Device.voxel_dim_*, State.zig:61-63)");
    const uint64_t brick_alloc = cfg->brick_alloc ? cfg->brick_alloc : cells;
    const uint64_t bits = (uint64_t)cfg->brick_dimension * cfg->brick_dimension * cfg->brick_dimension;
    if (brick_alloc * bits > 0x80000000ull)
        return fail(nullptr, VRT_E_OUT_OF_RANGE, "brick_alloc * brick_bits exceeds the u31 start index (State.zig:117-120)");
    if ((cfg->tile_w && cfg->tile_w != (uint32_t)vrt::kTileW) || (cfg->tile_h && cfg->tile_h != (uint32_t)vrt::kTileH))
        return fail(nullptr, VRT_E_INVALID_ARG, "tile size must be 16x16 (or 0)");
    const uint32_t shard_count = cfg->shard_count ? cfg->shard_count : 1u;
    if (cfg->shard_rank >= shard_count) return fail(nullptr, VRT_E_INVALID_ARG, "shard_rank >= shard_count");
    if (cfg->tuning_flags & ~VRT_TUNE_ALL) return fail(nullptr, VRT_E_INVALID_ARG, "unknown tuning_flags bit");
#ifndef VRT_DEV_VARIANTS
    if (cfg->tuning_flags & VRT_TUNE_PATH_AHEAD)
        return fail(nullptr, VRT_E_INVALID_ARG, "VRT_TUNE_PATH_AHEAD selects a development kernel: not in the product build of libvrt_hip (make dev)");
    if (cfg->tuning_flags & VRT_TUNE_PATH_DISTANCE)
        return fail(nullptr, VRT_E_INVALID_ARG, "VRT_TUNE_PATH_DISTANCE selects a development kernel: not in the product build of libvrt_hip (make dev)");
    if (cfg->tuning_flags & VRT_TUNE_PATH_TWO_AHEAD)
        return fail(nullptr, VRT_E_INVALID_ARG, "VRT_TUNE_PATH_TWO_AHEAD selects a development kernel: not in the product build of libvrt_hip (make dev)");
    if (cfg->tuning_flags & VRT_TUNE_PATH_BLOCKS64)
        return fail(nullptr, VRT_E_INVALID_ARG, "VRT_TUNE_PATH_BLOCKS64 selects a development kernel: not in the product build of libvrt_hip (make dev)");
#endif
    if ((cfg->kernel_variant & 0xFFu) >= vrt::kVariantCount || ((cfg->kernel_variant >> 28) && ((cfg->kernel_variant >> 16) & 0xFu) != 7u)) return fail(nullptr, VRT_E_INVALID_ARG, "unknown kernel_variant");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, VRT_E_NO_DEVICE, "no HIP device: libvrt_hip has no CPU path");
    int device = cfg->device_id;
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) return fail(nullptr, VRT_E_HIP, "hipGetDevice failed");
    }
    if (device >= ndev) return fail(nullptr, VRT_E_INVALID_ARG, "device_id out of range");

    vrt_ctx *c = new (std::nothrow) vrt_ctx();
    if (!c) return fail(nullptr, VRT_E_OOM, "host allocation failed");
    c->cfg = *cfg;
    c->cfg.shard_count = shard_count;
    c->cfg.brick_alloc = brick_alloc;
    c->cfg.material_capacity = cfg->material_capacity ? cfg->material_capacity : 256u;
    c->device = device;

#define VRT_CREATE_HIP(call)                                                                   \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            const int rc_ = fail(nullptr, e_ == hipErrorOutOfMemory ? VRT_E_OOM : VRT_E_HIP,   \
                                 std::string(#call) + ": " + hipGetErrorString(e_));           \
            free_ctx(c);                                                                       \
            return rc_;                                                                        \
        }                                                                                      \
    } while (0)

    DeviceGuard dg(device); // everything below runs on `device`; the caller's current device is restored on every return path
    if (!dg.ok) {
        free_ctx(c);
        return fail(nullptr, VRT_E_HIP, "hipSetDevice failed");
    }
    if (cfg->stream) {
        c->stream = static_cast<hipStream_t>(cfg->stream);
    } else {
        VRT_CREATE_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    VRT_CREATE_HIP(hipEventCreate(&c->ev_start));
    VRT_CREATE_HIP(hipEventCreate(&c->ev_stop));
    VRT_CREATE_HIP(hipEventCreate(&c->ev_post_start));
    VRT_CREATE_HIP(hipEventCreate(&c->ev_post_stop));

    // buffer sizes as Pipeline.zig:273-283 derives them from the State slices
    c->dsize[VRT_BUF_GRID_STATE] = sizeof(vrt_grid_state);
    c->dsize[VRT_BUF_MATERIALS] = (uint64_t)sizeof(vrt_material) * c->cfg.material_capacity;
    c->dsize[VRT_BUF_BRICK_STATUS] = ((cells + 31u) / 32u) * 4u;
    c->dsize[VRT_BUF_BRICK_INDEX] = cells * 4u;
    c->dsize[VRT_BUF_BRICK_OCCUPANCY] = brick_alloc * (bits / 8u);
    c->dsize[VRT_BUF_BRICK_START_INDEX] = brick_alloc * 4u;
    c->dsize[VRT_BUF_MATERIAL_INDEX] = brick_alloc * bits;
    for (int i = 0; i < VRT_BUF_COUNT; i++) {
        // +16: the kernel reads occupancy as aligned 64-bit words; keep slack at the tail.  The material table is
        // allocated (and zeroed) for all 256 values a u8 material id can take, whatever material_capacity says:
        // a voxel whose id is beyond the uploaded table reads a zero record instead of foreign memory.
        uint64_t alloc = c->dsize[i] + 16u;
        if (i == VRT_BUF_MATERIALS) alloc = std::max<uint64_t>(alloc, 256u * sizeof(vrt_material) + 16u);
        VRT_CREATE_HIP(hipMalloc(&c->dbuf[i], alloc));
        VRT_CREATE_HIP(hipMemsetAsync(c->dbuf[i], 0, alloc, c->stream));
    }

    // target image (Pipeline.zig:103-126), whole frame or this rank's packed tiles
    vrt_shard_info &sh = c->shard;
    sh.tile_w = vrt::kTileW;
    sh.tile_h = vrt::kTileH;
    sh.tiles_x = (cfg->width + vrt::kTileW - 1) / vrt::kTileW;
    sh.tiles_y = (cfg->height + vrt::kTileH - 1) / vrt::kTileH;
    sh.shard_rank = cfg->shard_rank;
    sh.shard_count = shard_count;
    const uint32_t total_tiles = sh.tiles_x * sh.tiles_y;
    sh.owned_tiles = (total_tiles > cfg->shard_rank) ? (total_tiles - cfg->shard_rank + shard_count - 1u) / shard_count : 0u;
    sh.tiles_per_rank = (total_tiles + shard_count - 1u) / shard_count;
    if (cfg->shard_root_weight > 0u && cfg->shard_root_weight < 100u) {
        if (shard_count < 2u || shard_count > 8u) {
            free_ctx(c);
            return fail(nullptr, VRT_E_INVALID_ARG, "shard_root_weight needs 2..8 ranks");
        }
        // Periodic pattern: 8 slots per period for every rank but the root, round(8 * weight) >= 1 for the root, laid out
        // so that each rank's slots are spread evenly over the period (largest deficit first; ties to the lower rank).
        vrt::TileOwnership &o = c->own;
        uint32_t want[8];
        want[0] = (8u * cfg->shard_root_weight + 50u) / 100u;
        if (want[0] < 1u) want[0] = 1u;
        for (uint32_t r = 1; r < shard_count; r++) want[r] = 8u;
        o.ranks = shard_count;
        o.period = want[0] + 8u * (shard_count - 1u);
        uint32_t given[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t j = 0; j < o.period; j++) {
            uint32_t best = 0;
            int64_t best_deficit = INT64_MIN;
            for (uint32_t r = 0; r < shard_count; r++) {
                // deficit of rank r after j+1 slots, scaled by the period
                const int64_t deficit = (int64_t)want[r] * (int64_t)(j + 1u) - (int64_t)given[r] * (int64_t)o.period;
                if (given[r] < want[r] && deficit > best_deficit) {
                    best_deficit = deficit;
                    best = r;
                }
            }
            o.owner[j] = (uint8_t)best;
            o.prefix[j] = (uint8_t)given[best];
            given[best]++;
        }
        for (uint32_t r = 0; r < shard_count; r++) o.count[r] = (uint8_t)want[r];
        auto owned_by = [&](uint32_t r) {
            uint32_t n = (total_tiles / o.period) * want[r];
            for (uint32_t j = 0; j < total_tiles % o.period; j++) n += (o.owner[j] == r) ? 1u : 0u;
            return n;
        };
        sh.owned_tiles = owned_by(cfg->shard_rank);
        sh.tiles_per_rank = 0;
        for (uint32_t r = 0; r < shard_count; r++) sh.tiles_per_rank = std::max(sh.tiles_per_rank, owned_by(r));
    }
    c->target_pixels = (shard_count > 1u) ? (uint64_t)sh.tiles_per_rank * vrt::kTileW * vrt::kTileH : (uint64_t)cfg->width * cfg->height;

    if (cfg->external_target_rgba8) {
        c->target8 = static_cast<uint8_t *>(cfg->external_target_rgba8);
    } else {
        VRT_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&c->target8), c->target_pixels * 4u));
        c->own_t8 = true;
        VRT_CREATE_HIP(hipMemsetAsync(c->target8, 0, c->target_pixels * 4u, c->stream));
    }
    if (cfg->external_target_rgba32f) {
        c->target32f = static_cast<float *>(cfg->external_target_rgba32f);
    } else if (cfg->want_float_output) {
        VRT_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&c->target32f), c->target_pixels * 16u));
        c->own_t32 = true;
        VRT_CREATE_HIP(hipMemsetAsync(c->target32f, 0, c->target_pixels * 16u, c->stream));
    }
    if (cfg->frames_in_flight == 2 && !cfg->stream && !cfg->external_target_rgba8 && !cfg->external_target_rgba32f && !cfg->enable_counters) {
        c->frames_in_flight = 2;
        VRT_CREATE_HIP(hipStreamCreateWithFlags(&c->stream_b, hipStreamNonBlocking));
        VRT_CREATE_HIP(hipEventCreateWithFlags(&c->ev_b_done, hipEventDisableTiming));
        VRT_CREATE_HIP(hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming));
        VRT_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&c->target8_b), c->target_pixels * 4u));
        VRT_CREATE_HIP(hipMemsetAsync(c->target8_b, 0, c->target_pixels * 4u, c->stream));
        if (c->target32f) {
            VRT_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&c->target32f_b), c->target_pixels * 16u));
            VRT_CREATE_HIP(hipMemsetAsync(c->target32f_b, 0, c->target_pixels * 16u, c->stream));
        }
    } else if (cfg->frames_in_flight > 2) {
        free_ctx(c);
        return fail(nullptr, VRT_E_INVALID_ARG, "frames_in_flight must be 0, 1 or 2");
    }
    if (cfg->enable_counters) {
        VRT_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_counters), sizeof(vrt::DeviceCounters)));
        VRT_CREATE_HIP(hipMemsetAsync(c->d_counters, 0, sizeof(vrt::DeviceCounters), c->stream));
    }
    const uint32_t nbx = (cfg->dim_x + 3u) / 4u, nby = (cfg->dim_y + 3u) / 4u, nbz = (cfg->dim_z + 3u) / 4u;
    VRT_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_cell_bounds), 6 * sizeof(int)));
    VRT_CREATE_HIP(hipMemsetAsync(c->d_cell_bounds, 0x80, 6 * sizeof(int), c->stream)); // no cell occupied yet
    // (the other derived copies of the status bits — byte per cell, half-block words, 4^3 block words — are allocated further
    // down, each only when a kernel this context selects reads it)
    {
        // tile schedule starts as reverse raster (bottom rows first); the feedback kernel refines it
        const uint32_t n = sh.owned_tiles ? sh.owned_tiles : 1u;
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || cus <= 0) cus = 256;
        c->wave_slots = 24u * (uint32_t)cus; // 4 SIMDs x 6 waves of the trace kernel per CU
        // default order: frames that run one at a time take the amortised cost-feedback schedule (7) unless all their workgroups
        // are resident at once (6 per CU: no launch order to speak of) or the frame is many times that (4K: 21 rounds of resident
        // workgroups, the tail is a small part of it and the schedule measured +1 %); the schedule may split tiles (spare entries)
        uint32_t order = (cfg->kernel_variant >> 16) & 0xFu;
        c->order_auto = (order == 0u);
        if (order == 0u) order = (n > 6u * (uint32_t)cus && n <= 64u * (uint32_t)cus) ? 7u : 3u;
        c->tile_order = order;
        const bool plain_tiles = (vrt::resolve_variant(cfg->kernel_variant) & 0xFFu) != vrt::kVariantLinearLds512;
        // spare entries: one per tile (the list's layout); a sort may use min(1024, n / 8) of them — or all, with a lower bar, for
        // frames of two samples per pixel (measured on the reference app's run, same box, V0 / V1 / V2: 0.336 / 0.340 / 0.364 ms with
        // n / 8 and a bar of 1.25 x the frame's wave-cycles over 24 slots per CU; 0.294 / 0.306 / 0.330 with every tile eligible
        // and 40 slots per CU — 28 / 32 / 36 / 44: 0.329 / 0.306 / 0.293 / 0.295 on V0; every tile split 0.339 / 0.345 / 0.370; a cap
        // below what the bar asks for makes the order flip between sorts: n / 4 0.395 / 0.367 / 0.396)
        c->sched_extra = (order == 7u && plain_tiles) ? n : 0u;
        c->sched_cap[0] = std::min(std::min(1024u, n / 8u), c->sched_extra);
        c->sched_cap[1] = c->sched_extra;
        c->sched_slots[0] = c->wave_slots;
        // (with the halves in cost classes of their own — vrt_schedule_kernel, round 4 — a lower bar pays: 28 / 40 / 52 / 64 / 96 slots
        // per CU 0.300 / 0.260 / 0.259 / 0.259 / 0.259 ms on V0, 0.323 / 0.285 / 0.262 / 0.255 / 0.257 on V1, 0.350 / 0.312 / 0.286 / 0.284 / 0.282 on V2)
        c->sched_slots[1] = 64u * (uint32_t)cus;
        VRT_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_tile_cost), n * 32u)); // [half][tile][wave]
        const uint32_t ns = 8u * ((n + c->sched_extra + 7u) / 8u); // an order buffer is stored XCD-major: 8 rows of ceil((n + extra) / 8)
        c->sched_stride = ns;
        VRT_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_tile_schedule), (2u * (size_t)ns + 2u * (size_t)n) * 4u)); // order A, order B, snapshot + split state
        VRT_CREATE_HIP(hipMemsetAsync(c->d_tile_cost, 0, n * 32u, c->stream));
        VRT_CREATE_HIP(hipMemsetAsync(c->d_tile_schedule + 2u * (size_t)ns, 0, 2u * (size_t)n * 4u, c->stream));
        uint32_t *init = static_cast<uint32_t *>(std::malloc((size_t)ns * 4u));
        if (!init) {
            free_ctx(c);
            return fail(nullptr, VRT_E_OOM, "host allocation failed");
        }
        std::memset(init, 0xFF, (size_t)ns * 4u); // spare entries: idle workgroups
        for (uint32_t i = 0; i < n; i++) init[(i & 7u) * (ns / 8u) + (i >> 3)] = n - 1u - i;
        const hipError_t e = hipMemcpy(c->d_tile_schedule, init, ns * 4u, hipMemcpyHostToDevice);
        std::free(init);
        VRT_CREATE_HIP(e);
        // first launch of the schedule kernel now (code-object load, about 2 ms, stays out of the frames): with no cost
        // measured yet it copies the initial order into the second buffer
        if (n > 1u) VRT_CREATE_HIP(vrt::launch_schedule(c->d_tile_cost, c->d_tile_schedule + 2u * (size_t)ns, c->d_tile_schedule, c->d_tile_schedule + ns, n, c->sched_extra, c->sched_cap[0], c->sched_slots[0], c->stream));
    }
    for (int i = 0; i < kStagingSlots; i++) {
        VRT_CREATE_HIP(hipHostMalloc(&c->staging[i], kStagingSlotBytes, hipHostMallocDefault));
        VRT_CREATE_HIP(hipEventCreateWithFlags(&c->staging_ev[i], hipEventDisableTiming));
    }

    // the bounce kernel comes in a 4- and an 8-waves-per-SIMD build (vrt_trace.hip, select_trace_kernel): the second one for
    // scenes whose traversal structures (bindings 3-5) exceed what the caches hold
    c->bounce_variant = cfg->kernel_variant;
    if (((cfg->kernel_variant >> 8) & 0xFFu) == 0u &&
        c->dsize[VRT_BUF_BRICK_STATUS] + c->dsize[VRT_BUF_BRICK_INDEX] + c->dsize[VRT_BUF_BRICK_OCCUPANCY] > (192ull << 20))
        c->bounce_variant |= 8u << 8;
    const uint32_t mwv = (cfg->kernel_variant >> 8) & 0xFFu;
    // Frames without bounces, mode left to the library: the hand-written loops on the byte-per-cell copy of the status bits for
    // grids up to 64^3 cells (1080p / 512^3 / 8^3 bricks V1, V2: 0.122 -> 0.118 ms, 1080p / 256^3 / 4^3: 0.098 -> 0.095; a tail-bound
    // frame from outside the grid pays 5 % for the larger footprint), the words beyond (128^3 cells: -4 % inside, +9 ... +22 % outside)
    uint32_t single_variant = ((cfg->kernel_variant & 0xFFu) == vrt::kVariantDefault && cells <= (1ull << 18))
                                  ? (cfg->kernel_variant | (uint32_t)vrt::kVariantBytes) : cfg->kernel_variant;
    {
        // development variants that stage a structure in LDS: a grid whose structure exceeds the budget reads global memory instead
        vrt::TraceParams sizes{};
        sizes.nbx = nbx, sizes.nby = nby, sizes.nbz = nbz;
        sizes.status_words = (uint32_t)((cells + 31u) / 32u);
        c->lds_bytes = vrt::trace_lds_bytes(sizes, cfg->kernel_variant);
        if (c->lds_bytes > 64u * 1024u) {
            const uint32_t mode = vrt::resolve_variant(cfg->kernel_variant) & 0xFFu;
            const uint32_t fallback = (mode == vrt::kVariantLinearLds || mode == vrt::kVariantLinearLds512) ? vrt::kVariantLinearAlways : vrt::kVariantBlocked;
            c->cfg.kernel_variant = (cfg->kernel_variant & ~0xFFu) | fallback;
            single_variant = c->cfg.kernel_variant;
            c->bounce_variant = (c->bounce_variant & ~0xFFu) | fallback;
            c->lds_bytes = 0;
            c->name_note = "[LDS structure > 64 KiB: global-memory variant]";
        }
    }
    uint32_t lockstep_variant = c->bounce_variant | vrt::kVariantLockstepBounce; // (before the path kernel's occupancy is filled in below)
    bool want_halfblocks = false, want_distance = false, want_dilated = false;
    {
        auto pow2 = [](uint32_t v) { return v >= 4u && (v & (v - 1u)) == 0u; };
        // Development build only (kernel_variant bit 22): the block-skipping walk of vrt_path_kernel<FILTER> — lanes in empty
        // 4x4x4 blocks jump to the block's exit face instead of taking a trip per cell.  x and z dimensions powers of two >= 4,
        // y a multiple of 4, filter <= 32 KiB, cell index < 2^31.  Measured on the 2048^3 path trace: 10 % fewer wave-cycles
        // per frame, but the filter's 32 KiB of LDS allow four waves per SIMD instead of five: 200 ms against 176.
        const uint64_t nblocks64 = (uint64_t)nbx * nby * nbz;
        size_t bytes = 16;
        while (bytes < ((nblocks64 + 31u) / 32u) * 4u) bytes <<= 1;
        const bool eligible = pow2(cfg->dim_x) && pow2(cfg->dim_z) && cfg->dim_y % 4u == 0u && bytes <= (32u << 10);
        const bool block_skip = eligible && (cfg->kernel_variant & vrt::kVariantPathFilter) != 0u;
        if (block_skip) {
            c->path_lds_bytes = (uint32_t)bytes;
            c->bounce_variant |= vrt::kVariantPathFilter;
        } else {
            c->bounce_variant &= ~vrt::kVariantPathFilter;
        }
        // Which kernel traces frames with bounces.  Scenes whose traversal structures exceed the caches (the 8-wave criterion
        // above): vrt_path_kernel at 5 waves per SIMD (96 VGPRs) — 4K / 2048^3 sparse / 16 spp / 3 bounces 193.6 ms per frame
        // against 206 for the lockstep kernel at 8 waves, 50.7 against 62 from outside the grid.  Scenes that stay in the caches:
        // the lockstep kernel — the reference app's shape (1024x576, 512^3 terrain, 2 spp, 2 bounces) 0.38 / 0.48 / 0.54 ms against
        // 0.58 / 0.92 / 1.11 for the path kernel, whose waiting for batches of lanes costs more than the coherent rays lose.
        // kernel_variant bit 21 forces the lockstep kernel, bit 23 the path kernel.
        const bool big_scene = ((c->bounce_variant >> 8) & 0xFFu) == 8u && mwv == 0u;
        const bool want_path = (cfg->kernel_variant & vrt::kVariantForcePath) || (big_scene && !(cfg->kernel_variant & vrt::kVariantLockstepBounce));
        if (want_path) {
            c->bounce_variant &= ~vrt::kVariantLockstepBounce;
            if (mwv == 0u) c->bounce_variant = (c->bounce_variant & ~0xFF00u) | (5u << 8);
            // the path kernel's walk loop reads the status bits by half-blocks of 4 x 4 x 2 cells where the grid allows (x, z
            // powers of two >= 4, y even): a third of the L1 requests of the linear words (vrt_trace_kernels.h)
            want_halfblocks = pow2(cfg->dim_x) && pow2(cfg->dim_z) && cfg->dim_y % 2u == 0u && !block_skip &&
                              !(cfg->tuning_flags & VRT_TUNE_NO_PATH_HALFBLOCKS);
            // ... or the L1 distance field of the occupied cells, one byte per cell (any dimensions; cells < 2^31: the byte offset is
            // the walk's 32-bit cell index)
            want_distance = (cfg->tuning_flags & VRT_TUNE_PATH_DISTANCE) != 0u && !block_skip && cells < (1ull << 31);
            if (want_distance) want_halfblocks = false;
            // the half-block words through a dilated cell index (22 instead of 29 vector instructions per trip): all three dimensions
            // powers of two
            want_dilated = want_halfblocks && pow2(cfg->dim_y) && !(cfg->tuning_flags & VRT_TUNE_NO_PATH_DILATED);
        } else {
            c->bounce_variant |= vrt::kVariantLockstepBounce;
        }
    }
    auto select_all = [&]() {
        const bool cnt = cfg->enable_counters != 0;
        c->kernel = vrt::select_trace_kernel((int)cfg->brick_dimension, cnt, c->bounce_variant, 0);
        const bool ahead = (cfg->tuning_flags & VRT_TUNE_PATH_AHEAD) != 0u; // (development build: the product holds no such kernel)
        if (ahead) c->kernel = vrt::path_kernel_ahead_twin(c->kernel);
        else if (want_distance) c->kernel = vrt::path_kernel_dist_twin(c->kernel);
        else if (want_dilated) c->kernel = vrt::path_kernel_dilated_twin(c->kernel, 1);
        else if (want_halfblocks) c->kernel = vrt::path_kernel_halfblock_twin(c->kernel);
        c->kernel_lockstep = vrt::select_trace_kernel((int)cfg->brick_dimension, cnt, lockstep_variant, 0);
        c->kernel_single = vrt::select_trace_kernel((int)cfg->brick_dimension, cnt, single_variant, 1);
        c->kernel_single1 = vrt::select_trace_kernel((int)cfg->brick_dimension, cnt, single_variant, 2);
        // a counting context renders the frame that is read back with the product kernels (do_dispatch)
        for (int shade = 0; shade < 3; shade++) {
            c->product[shade] = nullptr;
            if (!cnt) continue;
            c->product[shade] = vrt::select_trace_kernel((int)cfg->brick_dimension, false, shade == 0 ? c->bounce_variant : single_variant, shade);
            if (shade == 0 && ahead) c->product[shade] = vrt::path_kernel_ahead_twin(c->product[shade]);
            else if (shade == 0 && want_distance) c->product[shade] = vrt::path_kernel_dist_twin(c->product[shade]);
            else if (shade == 0 && want_dilated) c->product[shade] = vrt::path_kernel_dilated_twin(c->product[shade], 1);
            else if (shade == 0 && want_halfblocks) c->product[shade] = vrt::path_kernel_halfblock_twin(c->product[shade]);
        }
    };
    select_all();
    if (!(cfg->tuning_flags & VRT_TUNE_NO_PATH_GRID_EXIT)) {
        // (chosen per dispatch, once the host knows the box of the occupied cells: pre_dispatch)
        const int kind = ((cfg->tuning_flags & VRT_TUNE_PATH_BLOCKS64) && cfg->dim_y % 4u == 0u) ? 3 : ((cfg->tuning_flags & VRT_TUNE_PATH_TWO_AHEAD) ? 4 : 2);
        if (vrt::path_kernel_dilated_kind(c->kernel) == 1) c->kernel_grid_exit = vrt::path_kernel_dilated_twin(c->kernel, kind);
        if (c->product[0] && vrt::path_kernel_dilated_kind(c->product[0]) == 1) c->product_grid_exit = vrt::path_kernel_dilated_twin(c->product[0], kind);
        if (c->kernel_grid_exit == c->kernel) c->kernel_grid_exit = nullptr;
        if (c->product_grid_exit == c->product[0]) c->product_grid_exit = nullptr;
        // round 4: where that kernel would run on 8^3 bricks staged in LDS, a pool of rays per wave runs instead (vrt_pool_kernel.h)
        int pool_mw = 0, pool_slots = 0, pool_stages = 0;
#ifdef VRT_DEV_VARIANTS
        if (const char *e = std::getenv("VRT_DEV_POOL_KERNEL")) (void)std::sscanf(e, "%d:%d:%d", &pool_mw, &pool_slots, &pool_stages); // "<waves per SIMD>:<LDS slots>:<staging areas>"
#endif
        const vrt::KernelEntry *pool = (kind == 2 && cfg->brick_dimension == 8u && !(cfg->tuning_flags & (VRT_TUNE_NO_PATH_BRICK_LDS | VRT_TUNE_NO_PATH_POOL)))
                                           ? vrt::find_pool_kernel((int)cfg->brick_dimension, pool_mw, pool_slots, pool_stages) : nullptr;
        if (pool && c->kernel_grid_exit) c->kernel_grid_exit = pool->fn;
        if (pool && c->product_grid_exit) c->product_grid_exit = pool->fn;
    }
    c->single_variant = single_variant;
    {
        const bool cnt = cfg->enable_counters != 0;
        if (!c->kernel || !c->kernel_lockstep || !c->kernel_single || !c->kernel_single1 ||
            (cnt && (!c->product[0] || !c->product[1] || !c->product[2]))) {
            free_ctx(c);
#ifdef VRT_DEV_VARIANTS
            return fail(nullptr, VRT_E_INVALID_ARG, "no kernel for this configuration");
#else
            return fail(nullptr, VRT_E_INVALID_ARG, "no kernel for this kernel_variant in the product build of libvrt_hip (development variants: make dev)");
#endif
        }
        // derived copies of the status bits, each only if a kernel of this context reads it
        auto any_kernel = [&](auto pred) {
            const vrt::KernelFn fns[9] = {c->kernel, c->kernel_lockstep, c->kernel_single, c->kernel_single1, c->product[0], c->product[1], c->product[2],
                                          c->kernel_grid_exit, c->product_grid_exit};
            for (vrt::KernelFn fn : fns) {
                const vrt::KernelEntry *e = fn ? vrt::kernel_entry_of(fn) : nullptr;
                if (e && pred(*e)) return true;
            }
            return false;
        };
        if (any_kernel([](const vrt::KernelEntry &e) { return !e.path && e.mode == vrt::kStatusBytes; })) {
            const size_t status_bytes_size = (size_t)((cells + 31u) / 32u) * 32u + 64u; // 32 bytes per status word
            VRT_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_status_bytes), status_bytes_size));
            VRT_CREATE_HIP(hipMemsetAsync(c->d_status_bytes, 0, status_bytes_size, c->stream));
        }
        if (any_kernel([](const vrt::KernelEntry &e) { return e.path && (e.half || e.dil); })) {
            const size_t bytes_hb = (size_t)(cells / 32u) * 4u + 64u;
            VRT_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_status_halfblocks), bytes_hb));
            VRT_CREATE_HIP(hipMemsetAsync(c->d_status_halfblocks, 0, bytes_hb, c->stream));
        }
        if (any_kernel([](const vrt::KernelEntry &e) { return e.path && e.dist; })) {
            VRT_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_cell_distance), (size_t)cells + 64u));
            VRT_CREATE_HIP(hipMemsetAsync(c->d_cell_distance, 0xFF, (size_t)cells + 64u, c->stream));
        }
        // the by-cell copy of the occupancy bits, for the persistent-lane kernel (scenes larger than the caches, where a brick entry
        // is a chain of dependent misses): at most 2 GiB, and — walked in global memory instead of LDS — a 32-bit bit index
        const uint64_t by_cell_bytes = cells * (bits / 8u);
        const bool lds_walk = cfg->brick_dimension == 8u && !(cfg->tuning_flags & VRT_TUNE_NO_PATH_BRICK_LDS);
        // (an optional structure — within a quarter of the memory that is free now, and a failed allocation means "no
        // by-cell copy", not a failed vrt_create: the kernels then reach a brick's bits through brick_index as the shader does)
        size_t mem_free = 0, mem_total = 0;
        if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) mem_free = 0;
        // (round 4: also for the lockstep bounce kernel on scenes that stay in the caches — up to 64 MiB of it —, whose brick entries then
        // request the bits without waiting for brick_index[cell]: brick_walk_gfx950<..., BY_CELL>)
        const bool persistent = any_kernel([](const vrt::KernelEntry &e) { return e.path != 0; });
        const bool lockstep_bounce = any_kernel([](const vrt::KernelEntry &e) { return e.path == 0 && e.shade == 0 && !e.count; }) &&
                                     by_cell_bytes <= (64ull << 20) && cells * bits <= (1ull << 32);
        if ((persistent || lockstep_bounce) && !(cfg->tuning_flags & VRT_TUNE_NO_CELL_OCCUPANCY) &&
            by_cell_bytes <= (2ull << 30) && by_cell_bytes + 64u <= mem_free / 4u && (lds_walk || cells * bits <= (1ull << 32))) {
            if (hipMalloc(reinterpret_cast<void **>(&c->d_cell_occupancy), by_cell_bytes + 64u) != hipSuccess) {
                (void)hipGetLastError();
                c->d_cell_occupancy = nullptr;
            } else {
                VRT_CREATE_HIP(hipMemsetAsync(c->d_cell_occupancy, 0, by_cell_bytes + 64u, c->stream));
                c->cell_occupancy_lockstep = lockstep_bounce;
            }
        }
        if (!(cfg->tuning_flags & VRT_TUNE_NO_DEFERRED_MATERIAL)) {
            VRT_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_materials_plain), 64u));
            VRT_CREATE_HIP(hipMemsetAsync(c->d_materials_plain, 0, 64u, c->stream));
        }
        if (!(cfg->tuning_flags & VRT_TUNE_NO_START_SHORTCUT)) {
            VRT_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_start_is_slot), 64u));
            VRT_CREATE_HIP(hipMemsetAsync(c->d_start_is_slot, 0, 64u, c->stream));
        }
        if (any_kernel([](const vrt::KernelEntry &e) { return (e.path && (e.filter || e.dil == 3)) || (!e.path && (e.mode == vrt::kStatusBlocked || e.mode == vrt::kStatusBlockedLds)); })) {
            const size_t nblocks = (size_t)nbx * nby * nbz;
            const size_t status_blocks_bytes = nblocks * 8u + ((nblocks + 31u) / 32u) * 4u + 16u;
            VRT_CREATE_HIP(hipMalloc(&c->d_status_blocks, status_blocks_bytes));
            VRT_CREATE_HIP(hipMemsetAsync(c->d_status_blocks, 0, status_blocks_bytes, c->stream));
        }
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || cus <= 0) cus = 256;
    {
        const vrt::KernelEntry *e1 = c->kernel_grid_exit ? vrt::kernel_entry_of(c->kernel_grid_exit) : nullptr;
        const vrt::KernelEntry *e2 = c->product_grid_exit ? vrt::kernel_entry_of(c->product_grid_exit) : nullptr;
        if ((e1 && e1->path == 2) || (e2 && e2->path == 2)) {
            // (never read before it is written: a path's record is filled by the transition that gives the path its first pixel)
            c->pool_stream_dwords = (size_t)(8 * cus) * 4u * vrt::kPoolPaths * vrt::kPoolPathDwords; // (room for any occupancy)
            VRT_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_pool_paths), 2u * c->pool_stream_dwords * sizeof(uint32_t)));
        }
    }
    VRT_CREATE_HIP(hipMalloc(reinterpret_cast<void **>(&c->d_work_counter), 2u * vrt::kMaxBatchFrames * sizeof(uint32_t)));
    VRT_CREATE_HIP(hipMemsetAsync(c->d_work_counter, 0, 2u * vrt::kMaxBatchFrames * sizeof(uint32_t), c->stream));

    vrt::TraceParams &p = c->params;
    std::memset(&p, 0, sizeof p);
    p.materials = static_cast<const vrt_material *>(c->dbuf[VRT_BUF_MATERIALS]);
    p.brick_status = static_cast<const uint32_t *>(c->dbuf[VRT_BUF_BRICK_STATUS]);
    p.brick_index = static_cast<const uint32_t *>(c->dbuf[VRT_BUF_BRICK_INDEX]);
    p.brick_occupancy = static_cast<const uint8_t *>(c->dbuf[VRT_BUF_BRICK_OCCUPANCY]);
    p.brick_start_index = static_cast<const uint32_t *>(c->dbuf[VRT_BUF_BRICK_START_INDEX]);
    p.material_index = static_cast<const uint8_t *>(c->dbuf[VRT_BUF_MATERIAL_INDEX]);
    p.target_rgba8 = c->target8;
    p.target_rgba32f = c->target32f;
    p.counters = c->d_counters;
    p.count_box = (cfg->enable_counters == 2u) ? 1u : 0u;
    p.skip_to_box = (cfg->tuning_flags & VRT_TUNE_NO_SKIP_TO_BOX) ? 0u : 1u;
    p.work_counter = c->d_work_counter;
    p.path_lds_bytes = c->path_lds_bytes;
    {
        p.pool_paths = c->d_pool_paths;
        p.pool_cus = (uint32_t)cus;
        p.pool_walk_k = 12u;
        p.pool_brick_thr = 48u; // (tools/pool_sweep.py: a plateau from 48 to 56, walk_min 32 to 40, walk_k 16 to 20)
        p.pool_trans_thr = 48u;
        p.pool_walk_min = 32u;
        p.path_groups = 8u * (uint32_t)cus; // twice what 4 waves per SIMD hold: late groups find the counter exhausted and leave
        p.path_fin_batch = 32u;
        p.path_brick_lds = (cfg->brick_dimension == 8u && !(cfg->tuning_flags & VRT_TUNE_NO_PATH_BRICK_LDS)) ? 1u : 0u;
        p.path_skip_rounds = 8u;
        p.path_ready_batch = 32u;
        p.path_eager_start = (cfg->tuning_flags & VRT_TUNE_PATH_EAGER_START) ? 1u : 0u;
#ifdef VRT_DEV_VARIANTS
        // development build only: numeric knobs of vrt_path_kernel for parameter sweeps (tools/); the product reads no environment
        auto dev_knob = [](const char *name, uint32_t &v) {
            if (const char *e = std::getenv(name)) v = (uint32_t)std::max(1, std::atoi(e));
        };
        dev_knob("VRT_DEV_PATH_FIN_BATCH", p.path_fin_batch);
        dev_knob("VRT_DEV_PATH_SKIP_ROUNDS", p.path_skip_rounds);
        dev_knob("VRT_DEV_PATH_READY_BATCH", p.path_ready_batch);
        dev_knob("VRT_DEV_PATH_GROUPS", p.path_groups);
        dev_knob("VRT_DEV_POOL_WALK_K", p.pool_walk_k);
        dev_knob("VRT_DEV_POOL_BRICK_THR", p.pool_brick_thr);
        dev_knob("VRT_DEV_POOL_TRANS_THR", p.pool_trans_thr);
        dev_knob("VRT_DEV_POOL_WALK_MIN", p.pool_walk_min);
#endif
    }
    p.width = cfg->width;
    p.height = cfg->height;
    p.tiles_x = sh.tiles_x;
    p.pool_tiles_x_magic = sh.tiles_x > 1u ? (uint32_t)((1ull << 32) / sh.tiles_x) + 1u : 0u;
    p.tiles_y = sh.tiles_y;
    p.shard_rank = sh.shard_rank;
    p.shard_count = sh.shard_count;
    p.owned_tiles = sh.owned_tiles;
    if (c->own.period) {
        p.own_period = c->own.period;
        p.own_count = c->own.count[sh.shard_rank];
        uint32_t k = 0;
        for (uint32_t j = 0; j < c->own.period; j++)
            if (c->own.owner[j] == sh.shard_rank) p.own_slots[k++] = (uint8_t)j;
    }
    p.status_words = (uint32_t)((cells + 31u) / 32u);
    // the hand-written voxel-level loop addresses brick_occupancy by a 32-bit global bit index: brick_alloc * B^3 <= 2^31
    // (the u31 start-index check above), so it always reaches
    p.occupancy_words = (uint32_t)(c->dsize[VRT_BUF_BRICK_OCCUPANCY] / 4u);
    p.status_blocks = static_cast<const uint2 *>(c->d_status_blocks);
    p.cell_bounds = c->d_cell_bounds;
    p.status_bytes = c->d_status_bytes;
    p.status_halfblocks = c->d_status_halfblocks;
    p.cell_distance = c->d_cell_distance;
    p.cell_occupancy = c->d_cell_occupancy;
    p.cell_occupancy_lockstep = (c->d_cell_occupancy && c->cell_occupancy_lockstep) ? 1u : 0u;
    p.start_is_slot = c->d_start_is_slot;
    p.materials_plain = c->d_materials_plain;
    p.status_cells = (uint32_t)cells;
    // (order_auto: frames that alternate between the two streams of a frames_in_flight = 2 context take reverse raster (3)
    // instead, see do_dispatch and DESIGN.md §4)
    p.tile_order = c->tile_order;
    p.sched_extra = c->sched_extra;
    p.sched_units = c->sched_cap[0];
    if (p.tile_order == 7u) {
        // the cost-feedback schedule, re-sorted every 32 frames instead of every frame: the kernel sees order 5
        p.tile_order = 5u;
        c->sched_period = (cfg->kernel_variant >> 28) ? (1u << (cfg->kernel_variant >> 28)) : 32u; // tuning knob: log2 of the period
        VRT_CREATE_HIP(hipEventCreateWithFlags(&c->ev_sched, hipEventDisableTiming));
        VRT_CREATE_HIP(hipEventCreateWithFlags(&c->ev_b_sched, hipEventDisableTiming));
    }
    p.tile_cost = c->d_tile_cost;
    p.tile_schedule = c->d_tile_schedule;
    p.wave_groups = (cfg->kernel_variant >> 20) & 0x1u;
    p.wave_groups_bounce = (cfg->tuning_flags & VRT_TUNE_NO_BOUNCE_WAVE_GROUPS) ? 0u : 1u;
    {
        // small frames (configs[0], 256 tiles = one wave per SIMD, lasts as long as its slowest wave's chain): when the
        // frame's waves do not fill the SIMDs twice, every tile goes to two workgroups of 32-lane waves
        int cus2 = 0;
        if (hipDeviceGetAttribute(&cus2, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || cus2 <= 0) cus2 = 256;
        const uint32_t waves = sh.owned_tiles * 4u, simds = 4u * (uint32_t)cus2;
        // (a shard of a multi-GPU frame is a small frame as well: a rank of eight owns 1 020 tiles of the headline's 8 160)
        const bool eligible = c->order_auto && c->tile_order == 3u && !p.wave_groups && !(cfg->tuning_flags & VRT_TUNE_NO_SMALL_FRAME_SPLIT) &&
                              (vrt::resolve_variant(cfg->kernel_variant) & 0xFFu) != vrt::kVariantLinearLds512;
        c->split_ok = eligible;
        c->simds = simds;
        // (configs[0], same box, us per frame V0 / V1 / V2 / V1x: whole tiles 16.3 / 24.0 / 27.3 / 28.6, halves 15.7 / 21.8 / 23.8 / 25.8,
        // quarters 16.7 / 21.5 / 24.1 / 23.4, eighths 16.4 / 23.3 / 25.4 / 24.1: halves; the rest of such a frame is its launch and the
        // fixed part of a wave's chain — the status bits in LDS change nothing, tools/small_frame_ab.py)
        p.split_all = (eligible && waves <= 2u * simds) ? 1u : 0u;
#ifdef VRT_EXP_SPLIT_ALL
        if (c->tile_order == 3u) p.split_all = VRT_EXP_SPLIT_ALL;
#endif
    }
    p.brick_batch = (cfg->kernel_variant >> 24) & 0xFu ? ((cfg->kernel_variant >> 24) & 0xFu) * 4u : 8u; // tuning knob: units of 4 lanes
    p.path_brick_batch = (cfg->kernel_variant >> 24) & 0xFu ? ((cfg->kernel_variant >> 24) & 0xFu) * 4u : 32u; // vrt_path_kernel: waiting is cheap there
    p.block_threads = ((vrt::resolve_variant(c->cfg.kernel_variant) & 0xFFu) == vrt::kVariantLinearLds512) ? 512u : 256u;
    {
        // a stride near owned_tiles * 0.618 that is coprime to owned_tiles
        auto gcd = [](uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b; b = t; } return a; };
        uint32_t st = (uint32_t)((double)sh.owned_tiles * 0.6180339887) | 1u;
        while (sh.owned_tiles > 1u && gcd(st, sh.owned_tiles) != 1u) st += 2u;
        p.tile_stride = sh.owned_tiles > 1u ? st % sh.owned_tiles : 0u;
        if (sh.owned_tiles > 1u && p.tile_stride == 0u) p.tile_stride = 1u;
    }
    p.nbx = nbx;
    p.nby = nby;
    p.nbz = nbz;
    // the status bitmap is read 16 bytes at a time by the LDS-staging variant: the +16 slack of dbuf covers the tail
    if (c->stream_b) {
        // everything enqueued on the primary stream so far (clears) precedes the first frame on stream_b
        VRT_CREATE_HIP(hipEventRecord(c->ev_upload, c->stream));
        c->upload_seq = 1;
    }
#undef VRT_CREATE_HIP
    note_kernel(c, c->product[2] ? c->product[2] : c->kernel_single1);
    *out = c;
    return VRT_OK;
}

void vrt_destroy(vrt_ctx *ctx) { free_ctx(ctx); }

uint64_t vrt_buffer_size(const vrt_ctx *ctx, vrt_buffer_id id) {
    if (!ctx || (int)id < 0 || id >= VRT_BUF_COUNT) return 0;
    return ctx->dsize[id];
}

// which derived structures a write to scene buffer `id` invalidates (rebuilt before the next frame, pre_dispatch)
static void mark_dirty(vrt_ctx *ctx, vrt_buffer_id id, uint64_t byte_offset, uint64_t nbytes) {
    if (id == VRT_BUF_BRICK_STATUS) ctx->status_dirty = true;
    if (id == VRT_BUF_BRICK_START_INDEX) ctx->start_dirty = true;
    if (id == VRT_BUF_MATERIALS) ctx->materials_dirty = true;
    if (id != VRT_BUF_BRICK_STATUS && id != VRT_BUF_BRICK_INDEX && id != VRT_BUF_BRICK_OCCUPANCY) return;
    auto widen = [](uint64_t &lo, uint64_t &hi, uint64_t a, uint64_t b) {
        if (lo >= hi) lo = a, hi = b;
        else lo = std::min(lo, a), hi = std::max(hi, b);
    };
    const uint64_t end = byte_offset + nbytes;
    if (id == VRT_BUF_BRICK_STATUS) widen(ctx->occ_cell_lo, ctx->occ_cell_hi, byte_offset * 8u, end * 8u);
    else if (id == VRT_BUF_BRICK_INDEX) widen(ctx->occ_cell_lo, ctx->occ_cell_hi, byte_offset / 4u, (end + 3u) / 4u);
    else {
        const uint64_t brick_bytes = (uint64_t)ctx->cfg.brick_dimension * ctx->cfg.brick_dimension * ctx->cfg.brick_dimension / 8u;
        widen(ctx->occ_slot_lo, ctx->occ_slot_hi, byte_offset / brick_bytes, (end + brick_bytes - 1u) / brick_bytes);
    }
    ctx->occupancy_dirty = true;
}

static int check_range(vrt_ctx *ctx, vrt_buffer_id id, uint64_t byte_offset, const void *src, uint64_t nbytes) {
    if (!ctx) return VRT_E_INVALID_ARG;
    if ((int)id < 0 || id >= VRT_BUF_COUNT) return fail(ctx, VRT_E_INVALID_ARG, "bad buffer id");
    if (nbytes && !src) return fail(ctx, VRT_E_INVALID_ARG, "src is NULL");
    if (byte_offset > ctx->dsize[id] || nbytes > ctx->dsize[id] - byte_offset)
        return fail(ctx, VRT_E_OUT_OF_RANGE, "upload exceeds device buffer (DestOutOfDeviceMemory)");
    return VRT_OK;
}

int vrt_upload(vrt_ctx *ctx, vrt_buffer_id id, uint64_t byte_offset, const void *src, uint64_t nbytes) {
    const int rc = check_range(ctx, id, byte_offset, src, nbytes);
    if (rc != VRT_OK || nbytes == 0) return rc;
    DeviceGuard dg(ctx->device);
    if (id == VRT_BUF_GRID_STATE) {
        // the kernel takes the UBO through its argument block; keep the host mirror current
        std::memcpy(reinterpret_cast<uint8_t *>(&ctx->params.grid) + byte_offset, src, (size_t)nbytes);
    }
    mark_dirty(ctx, id, byte_offset, nbytes);
    return copy_h2d(ctx, static_cast<uint8_t *>(ctx->dbuf[id]) + byte_offset, src, nbytes);
}

int vrt_upload_device(vrt_ctx *ctx, vrt_buffer_id id, uint64_t byte_offset, const void *dev_src, uint64_t nbytes) {
    const int rc = check_range(ctx, id, byte_offset, dev_src, nbytes);
    if (rc != VRT_OK || nbytes == 0) return rc;
    DeviceGuard dg(ctx->device);
    if (id == VRT_BUF_GRID_STATE) {
        VRT_HIP(ctx, hipMemcpyAsync(reinterpret_cast<uint8_t *>(&ctx->params.grid) + byte_offset, dev_src, nbytes, hipMemcpyDeviceToHost,
                                    ctx->stream));
        VRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    mark_dirty(ctx, id, byte_offset, nbytes);
    const int rcb = begin_scene_write(ctx);
    if (rcb != VRT_OK) return rcb;
    VRT_HIP(ctx, hipMemcpyAsync(static_cast<uint8_t *>(ctx->dbuf[id]) + byte_offset, dev_src, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
    return end_scene_write(ctx);
}

// Common front part of a frame: argument checks, push constants, derived-structure refresh.  Leaves the
// kernel to launch in *fn.  Runs on the primary stream.
// vrt_pool_kernel packs a path's sample index into 16 bits and its bounce count into 4, and divides a tile's number by the tiles
// per row with one multiplication (exact while tiles * tiles_x < 2^32); frames beyond that keep vrt_path_kernel
// ... and takes SAMPLES from its counter (32 bits), whose terms of the sample sum it leaves in a buffer of 16 bytes per sample and stream
// for vrt_pool_resolve_kernel: 2 x 2 GiB for a 4K frame of 16 samples.  The buffer grows with the frames asked for (both streams idle
// first); where it cannot be had — more than half of the free memory, or a failed allocation — the frame keeps vrt_path_kernel.
static bool pool_samples_ready(vrt_ctx *ctx, const vrt_camera_device *camera) {
    ctx->params.pool_samples = nullptr; // (until this frame's buffer is known to be there)
    if (ctx->cfg.tuning_flags & VRT_TUNE_NO_SAMPLE_UNITS) return false;
    if (camera->samples_per_pixel < 1) return false; // (a frame of no samples is not a frame of units)
    const uint64_t units = (uint64_t)ctx->shard.owned_tiles * 256u * (uint64_t)camera->samples_per_pixel;
    if (units >= (1ull << 32) - (1ull << 26)) return false; // (the counter keeps counting, a chunk per wave, after it has run out)
    if (ctx->pool_samples_stream_elems >= units) {
        ctx->params.pool_samples = ctx->d_pool_samples;
        return true;
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return false;
    if (ctx->stream_b && hipStreamSynchronize(ctx->stream_b) != hipSuccess) return false;
    if (ctx->d_pool_samples) (void)hipFree(ctx->d_pool_samples);
    ctx->d_pool_samples = nullptr;
    ctx->pool_samples_stream_elems = 0;
    ctx->params.pool_samples = nullptr;
    size_t mem_free = 0, mem_total = 0;
    if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) return false;
    const size_t bytes = 2u * (size_t)units * sizeof(float4);
    if (bytes > mem_free / 2u) return false;
    if (hipMalloc(reinterpret_cast<void **>(&ctx->d_pool_samples), bytes) != hipSuccess) {
        (void)hipGetLastError();
        ctx->d_pool_samples = nullptr;
        return false;
    }
    ctx->pool_samples_stream_elems = (size_t)units;
    ctx->params.pool_samples = ctx->d_pool_samples;
    return true;
}
static bool grid_exit_fits(vrt_ctx *ctx, vrt::KernelFn fn, const vrt_camera_device *camera, bool tiles_fit) {
    const vrt::KernelEntry *e = vrt::kernel_entry_of(fn);
    return !(e && e->path == 2) || (camera->max_bounce <= 15 && tiles_fit && pool_samples_ready(ctx, camera));
}

static bool pool_tiles_fit(const vrt_ctx *ctx) {
    return (unsigned long long)ctx->shard.tiles_x * ctx->shard.tiles_y * ctx->shard.tiles_x < (1ull << 32);
}

static int pre_dispatch(vrt_ctx *ctx, const vrt_camera_device *camera, const vrt_sun_device *sun, vrt::KernelFn *fn) {
    if (!ctx || !camera || !sun) return ctx ? fail(ctx, VRT_E_INVALID_ARG, "NULL camera/sun") : VRT_E_INVALID_ARG;
    if (camera->image_width != ctx->cfg.width || camera->image_height != ctx->cfg.height)
        return fail(ctx, VRT_E_INVALID_ARG, "camera image size differs from the target image");
    // The reference blocks here on the previous frame's fence because it re-records its one
    // command buffer (ComputePipeline.zig:423-436).  Launches are stream-ordered and carry their
    // arguments by value, so frames may queue; vrt_wait / vrt_read_* are the synchronisation points.
    ctx->in_flight = false;
    ctx->params.pcs[0].cam = *camera;
    ctx->params.pcs[0].sun = *sun;
    const vrt_grid_state &g = ctx->params.grid;
    {
        auto pow2_with_normal_reciprocal = [](float v) {
            uint32_t b;
            std::memcpy(&b, &v, 4);
            const uint32_t e = (b >> 23) & 0xFFu;
            return (b & 0x7FFFFFu) == 0u && e >= 2u && e <= 252u;
        };
        const float gs = g.max_point_scale[3];
        const float vs = gs * (1.0f / (float)ctx->cfg.brick_dimension); // as the kernel forms it (Pipeline.zig:313)
        const bool ok = pow2_with_normal_reciprocal(gs) && pow2_with_normal_reciprocal(vs);
        ctx->params.scale_pow2 = ok ? 1u : 0u;
        ctx->params.inv_grid_scale = ok ? 1.0f / gs : 0.0f;
        ctx->params.inv_voxel_scale = ok ? 1.0f / vs : 0.0f;
    }
    if (g.dim_x != 0 && (g.dim_x != ctx->cfg.dim_x || g.dim_y != ctx->cfg.dim_y || g.dim_z != ctx->cfg.dim_z))
        return fail(ctx, VRT_E_INVALID_ARG, "uploaded grid state has other brick dimensions than the context was created with");
    if (ctx->d_counters) VRT_HIP(ctx, hipMemsetAsync(ctx->d_counters, 0, sizeof(vrt::DeviceCounters), ctx->stream));
    if (ctx->status_dirty) {
        // refresh the derived block words / filter from the uploaded status bits (stream-ordered after the uploads)
        int rcw = begin_scene_write(ctx);
        if (rcw != VRT_OK) return rcw;
        VRT_HIP(ctx, vrt::launch_build_status_blocks(ctx->params, ctx->cfg.dim_x, ctx->cfg.dim_y, ctx->cfg.dim_z, ctx->stream));
        VRT_HIP(ctx, vrt::launch_build_cell_bounds(ctx->params, ctx->cfg.dim_x, ctx->cfg.dim_y, ctx->cfg.dim_z, ctx->stream));
        if (ctx->kernel_grid_exit || ctx->product_grid_exit) {
            if (!ctx->h_cell_bounds) VRT_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->h_cell_bounds), 6 * sizeof(int), hipHostMallocDefault));
            if (!ctx->ev_bounds) VRT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_bounds, hipEventDisableTiming));
            if (ctx->bounds_pending) VRT_HIP(ctx, hipEventSynchronize(ctx->ev_bounds)); // (the copy before this one still owns the buffer)
            VRT_HIP(ctx, hipMemcpyAsync(ctx->h_cell_bounds, ctx->d_cell_bounds, 6 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            VRT_HIP(ctx, hipEventRecord(ctx->ev_bounds, ctx->stream));
            ctx->bounds_pending = true;
            ctx->box_is_grid = false; // until the new box is known
        }
        VRT_HIP(ctx, vrt::launch_build_status_bytes(ctx->params, ctx->stream));
        VRT_HIP(ctx, vrt::launch_build_status_halfblocks(ctx->params, ctx->cfg.dim_x, ctx->cfg.dim_y, ctx->cfg.dim_z, ctx->stream));
        VRT_HIP(ctx, vrt::launch_build_cell_distance(ctx->params, ctx->cfg.dim_x, ctx->cfg.dim_y, ctx->cfg.dim_z, ctx->stream));
        rcw = end_scene_write(ctx);
        if (rcw != VRT_OK) return rcw;
        ctx->status_dirty = false;
    }
    if ((ctx->occupancy_dirty && ctx->d_cell_occupancy) || (ctx->start_dirty && ctx->d_start_is_slot) || (ctx->materials_dirty && ctx->d_materials_plain)) {
        int rcw = begin_scene_write(ctx);
        if (rcw != VRT_OK) return rcw;
        if (ctx->occupancy_dirty)
            VRT_HIP(ctx, vrt::launch_build_cell_occupancy(ctx->params, ctx->cfg.brick_dimension, ctx->cfg.brick_alloc, ctx->occ_cell_lo, ctx->occ_cell_hi, ctx->occ_slot_lo,
                                                          ctx->occ_slot_hi, ctx->stream));
        if (ctx->start_dirty) VRT_HIP(ctx, vrt::launch_check_start_is_slot(ctx->params, ctx->cfg.brick_dimension, ctx->cfg.brick_alloc, ctx->stream));
        if (ctx->materials_dirty) VRT_HIP(ctx, vrt::launch_check_materials_plain(ctx->params, std::max<uint32_t>(256u, (uint32_t)(ctx->dsize[VRT_BUF_MATERIALS] / sizeof(vrt_material))), ctx->stream));
        rcw = end_scene_write(ctx);
        if (rcw != VRT_OK) return rcw;
    }
    ctx->occupancy_dirty = ctx->start_dirty = ctx->materials_dirty = false;
    ctx->occ_cell_lo = ctx->occ_cell_hi = ctx->occ_slot_lo = ctx->occ_slot_hi = 0;
    // max_bounce <= 1 ("only primary ray" + its shadow ray): the bounce loop runs at most once
    *fn = (camera->max_bounce <= 1) ? (camera->samples_per_pixel == 1 ? ctx->kernel_single1 : ctx->kernel_single) : ctx->kernel;
    if (ctx->bounds_pending && hipEventQuery(ctx->ev_bounds) == hipSuccess) {
        // the box of the occupied cells {-min, max} per axis: "the grid, or nearly" = at most an eighth of the axis free on either side
        const int *b = ctx->h_cell_bounds;
        const int dim[3] = {(int)ctx->cfg.dim_x, (int)ctx->cfg.dim_y, (int)ctx->cfg.dim_z};
        bool all = b[0] != (int)0x80808080;
        for (int a = 0; a < 3 && all; a++) all = (-b[a]) * 8 <= dim[a] && (dim[a] - 1 - b[3 + a]) * 8 <= dim[a];
        ctx->box_is_grid = all;
        ctx->bounds_pending = false;
    }
    if (camera->max_bounce > 1 && ctx->box_is_grid && !ctx->d_counters && ctx->kernel_grid_exit && grid_exit_fits(ctx, ctx->kernel_grid_exit, camera, pool_tiles_fit(ctx)))
        *fn = ctx->kernel_grid_exit;
    return VRT_OK;
}

static int do_dispatch(vrt_ctx *ctx, const vrt_camera_device *camera, const vrt_sun_device *sun, uint32_t frames, bool primary_only = false,
                       hipEvent_t *marks = nullptr) {
    if (frames == 0) return ctx ? fail(ctx, VRT_E_INVALID_ARG, "zero frames") : VRT_E_INVALID_ARG;
    if (ctx && ctx->dist) return fail(ctx, VRT_E_STATE, "this context runs the multi-GPU pipeline: use vrt_dist_frame");
    DeviceGuard dg(ctx ? ctx->device : 0);
    vrt::KernelFn fn = nullptr;
    const int rcp = pre_dispatch(ctx, camera, sun, &fn);
    if (rcp != VRT_OK) return rcp;

    // With counters enabled the counting build of the kernel (compiler-generated loops, per-lane counters) runs
    // first and fills the counters; the frame that is read back is then rendered by the product kernel itself,
    // so that every parity check made on a counting context checks the shipped code path.
    vrt::KernelFn product_fn = nullptr;
    if (ctx->d_counters) {
        product_fn = ctx->product[(camera->max_bounce <= 1) ? (camera->samples_per_pixel == 1 ? 2 : 1) : 0];
        if (!product_fn) return fail(ctx, VRT_E_STATE, "no product kernel for this configuration");
        if (camera->max_bounce > 1 && ctx->box_is_grid && ctx->product_grid_exit && grid_exit_fits(ctx, ctx->product_grid_exit, camera, pool_tiles_fit(ctx))) product_fn = ctx->product_grid_exit;
    }
    // (vrt_path_kernel takes samples as units of work where the sample buffer can be had, whole pixels otherwise)
    if (const vrt::KernelEntry *pe = vrt::kernel_entry_of(product_fn ? product_fn : fn); pe && pe->path == 1) (void)pool_samples_ready(ctx, camera);
    note_kernel(ctx, product_fn ? product_fn : fn);
    // (the persistent-lane kernel takes its pixels from a counter: it neither reads the tile schedule nor reports tile costs)
    const bool scheduled = !vrt::is_path_kernel(product_fn ? product_fn : fn);

    // (tile_order 5 re-sorts the tile schedule in place before every frame on the primary stream: a frame running on
    // the second stream would read it while it is being rewritten, so that order runs one frame at a time.  The
    // amortised form, tile_order 7, sorts into the other of two buffers and may use both streams.)
    // which of the cost schedule's two rules serves this frame (vrt_trace_kernel's `dual`: two samples per pixel, whole RGBA pixels); a
    // change re-sorts at once, on the primary stream
    const uint32_t sched_mode = (ctx->sched_period && scheduled && camera->samples_per_pixel == 2 && !ctx->params.packed_rgb) ? 1u : 0u;
    const bool sched_changes = ctx->sched_period && scheduled && sched_mode != ctx->sched_mode;
    const bool slot_b = ctx->stream_b && frames == 1 && !primary_only && (ctx->frame_seq & 1u) && (ctx->params.tile_order != 5u || ctx->sched_period) &&
                        !sched_changes;
    if (slot_b) {
        // second frame slot: its own stream and target; ordered after every scene write so far
        if (ctx->b_seen_upload != ctx->upload_seq) {
            VRT_HIP(ctx, hipStreamWaitEvent(ctx->stream_b, ctx->ev_upload, 0));
            ctx->b_seen_upload = ctx->upload_seq;
        }
        if (ctx->b_seen_sched != ctx->sched_seq) {
            // the schedule buffer this frame reads was sorted on the primary stream
            VRT_HIP(ctx, hipStreamWaitEvent(ctx->stream_b, ctx->ev_sched, 0));
        }
        vrt::TraceParams pb = ctx->params;
        // (frames of two samples per pixel keep the cost schedule on both streams: their split tiles trace the second sample on the idle
        // lanes, which is worth more than reverse raster's neighbourhood — the app's run, two frames in flight, V0 / V1 / V2: 0.205 /
        // 0.214 / 0.237 ms per frame against 0.267 / 0.262 / 0.286, tools/fif_order_ab.py)
        if (ctx->order_auto && sched_mode == 0u) pb.tile_order = 3u;
        pb.target_rgba8 = ctx->target8_b;
        pb.target_rgba32f = ctx->target32f_b;
        pb.work_counter = ctx->d_work_counter + vrt::kMaxBatchFrames; // its frames run beside the primary stream's
        if (pb.pool_paths) pb.pool_paths += ctx->pool_stream_dwords;
        if (pb.pool_samples) pb.pool_samples += ctx->pool_samples_stream_elems;
        VRT_HIP(ctx, vrt::launch_trace(fn, pb, ctx->lds_bytes, ctx->stream_b));
        if (product_fn) VRT_HIP(ctx, vrt::launch_trace(product_fn, pb, ctx->lds_bytes, ctx->stream_b));
        VRT_HIP(ctx, hipEventRecord(ctx->ev_b_done, ctx->stream_b));
        if (ctx->b_seen_sched != ctx->sched_seq) {
            // everything this stream read from the OTHER schedule buffer is finished once this event is
            VRT_HIP(ctx, hipEventRecord(ctx->ev_b_sched, ctx->stream_b));
            ctx->b_sched_recorded = true;
            ctx->b_seen_sched = ctx->sched_seq;
        }
        ctx->sched_since++;
        ctx->b_pending = true;
        ctx->frame_seq++;
        ctx->last_slot = 1;
        return VRT_OK;
    }
    if (ctx->stream_b && (frames > 1 || primary_only) && ctx->b_pending) {
        // timed back-to-back launches: do not let a frame on the other stream run underneath them
        VRT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_b_done, 0));
        ctx->b_pending = false;
    }
    VRT_HIP(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    const uint32_t nt = ctx->shard.owned_tiles;
    const uint32_t ns = ctx->sched_stride; // stride of a schedule buffer
    if (ctx->params.tile_order == 5u && nt > 1u && !ctx->sched_period && scheduled) {
        // re-sort the tile list by last frame's measured cost (inside the timed region: it is per-frame work)
        VRT_HIP(ctx, vrt::launch_schedule(ctx->d_tile_cost, ctx->d_tile_schedule + 2u * (size_t)ns, ctx->d_tile_schedule, ctx->d_tile_schedule, nt, ctx->sched_extra, 0u, ctx->wave_slots, ctx->stream));
    }
    if (ctx->order_auto && ctx->stream_b && frames == 1 && !primary_only && sched_mode == 0u && !sched_changes) {
        // the even frames of two frames in flight: the other stream fills this frame's tail, and reverse raster keeps
        // neighbouring tiles together (measured 4 % faster than the cost order in that mode)
        vrt::TraceParams pa = ctx->params;
        pa.tile_order = 3u;
        VRT_HIP(ctx, vrt::launch_trace(fn, pa, ctx->lds_bytes, ctx->stream));
        if (product_fn) VRT_HIP(ctx, vrt::launch_trace(product_fn, pa, ctx->lds_bytes, ctx->stream));
        VRT_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
        ctx->timed_frames = 1;
        ctx->in_flight = true;
        ctx->frame_seq++;
        ctx->last_slot = 0;
        return VRT_OK;
    }
    if (product_fn) {
        // Counting context: the counting build runs ONCE per call (the counters are those of one frame however many
        // frames were asked for), then both targets are overwritten with 0xCD, then the product kernel renders the
        // frame(s): a pixel the product kernel fails to write reads back as 0xCDCDCDCD / -4.3e8, not as the counting
        // build's (correct) colour.
        VRT_HIP(ctx, vrt::launch_trace(fn, ctx->params, ctx->lds_bytes, ctx->stream));
        VRT_HIP(ctx, hipMemsetAsync(ctx->target8, 0xCD, ctx->target_pixels * 4u, ctx->stream));
        if (ctx->target32f) VRT_HIP(ctx, hipMemsetAsync(ctx->target32f, 0xCD, ctx->target_pixels * 16u, ctx->stream));
        fn = product_fn;
    }
    if (sched_changes) {
        ctx->sched_mode = sched_mode;
        ctx->params.sched_units = ctx->sched_cap[sched_mode];
        ctx->sched_since = ctx->sched_period;
    }
    for (uint32_t f = 0; f < frames; f++) {
        if (marks) VRT_HIP(ctx, hipEventRecord(marks[f], ctx->stream)); // per-frame timing (vrt_dispatch_timed)
        if (ctx->sched_period && nt > 1u && ctx->sched_since >= ctx->sched_period && scheduled) {
            // Amortised re-sort, in the frames' own stream (inside the timed region as well): the measured costs (running mean)
            // order the tiles into the buffer no frame reads; frames launched from here on read that one.  A sort costs about
            // 35 us of the stream's time (the kernel plus the two kernel boundaries).  Measured alternatives: on a second
            // stream with event waits 80 us per sort; on a second stream with the host polling for its completion nothing, but
            // then the order lags behind frames that are queued ahead (vrt_dispatch_repeat) by a whole call.
            if (ctx->b_sched_recorded) VRT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_b_sched, 0)); // (signalled a period ago)
            uint32_t *cur = ctx->d_tile_schedule + (size_t)ctx->sched_cur * ns, *alt = ctx->d_tile_schedule + (size_t)(ctx->sched_cur ^ 1u) * ns;
            VRT_HIP(ctx, vrt::launch_schedule(ctx->d_tile_cost, ctx->d_tile_schedule + 2u * (size_t)ns, cur, alt, nt, ctx->sched_extra, ctx->sched_cap[ctx->sched_mode], ctx->sched_slots[ctx->sched_mode], ctx->stream));
            VRT_HIP(ctx, hipEventRecord(ctx->ev_sched, ctx->stream));
            ctx->sched_cur ^= 1u;
            ctx->params.tile_schedule = alt;
            ctx->sched_seq++;
            ctx->sched_since = 0;
        }
        VRT_HIP(ctx, vrt::launch_trace(fn, ctx->params, ctx->lds_bytes, ctx->stream));
        ctx->sched_since++;
    }
    if (marks) VRT_HIP(ctx, hipEventRecord(marks[frames], ctx->stream));
    VRT_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    ctx->timed_frames = frames;
    ctx->in_flight = true;
    ctx->frame_seq++;
    ctx->last_slot = 0;
    return VRT_OK;
}

int vrt_dispatch(vrt_ctx *ctx, const vrt_camera_device *camera, const vrt_sun_device *sun) { return do_dispatch(ctx, camera, sun, 1); }

int vrt_dispatch_repeat(vrt_ctx *ctx, const vrt_camera_device *camera, const vrt_sun_device *sun, uint32_t frames) {
    return do_dispatch(ctx, camera, sun, frames);
}

int vrt_dispatch_timed(vrt_ctx *ctx, const vrt_camera_device *camera, const vrt_sun_device *sun, uint32_t frames, float *ms_per_frame) {
    if (!ctx || !ms_per_frame) return ctx ? fail(ctx, VRT_E_INVALID_ARG, "ms_per_frame is NULL") : VRT_E_INVALID_ARG;
    if (frames == 0 || frames > 4096u) return fail(ctx, VRT_E_INVALID_ARG, "vrt_dispatch_timed: 1..4096 frames");
    DeviceGuard dg(ctx->device);
    std::vector<hipEvent_t> marks(frames + 1u, nullptr);
    int rc = VRT_OK;
    for (uint32_t i = 0; i <= frames && rc == VRT_OK; i++)
        if (hipEventCreate(&marks[i]) != hipSuccess) rc = fail(ctx, VRT_E_HIP, "hipEventCreate failed");
    // frames > 1 or primary_only keeps every frame on the primary stream, one after another
    if (rc == VRT_OK) rc = do_dispatch(ctx, camera, sun, frames, true, marks.data());
    if (rc == VRT_OK && hipEventSynchronize(marks[frames]) != hipSuccess) rc = fail(ctx, VRT_E_HIP, "hipEventSynchronize failed");
    for (uint32_t f = 0; f < frames && rc == VRT_OK; f++)
        if (hipEventElapsedTime(&ms_per_frame[f], marks[f], marks[f + 1u]) != hipSuccess) rc = fail(ctx, VRT_E_HIP, "hipEventElapsedTime failed");
    for (hipEvent_t e : marks)
        if (e) (void)hipEventDestroy(e);
    return rc;
}

int vrt_wait(vrt_ctx *ctx) {
    if (!ctx) return VRT_E_INVALID_ARG;
    DeviceGuard dg(ctx->device);
    const int rc = finish_frame(ctx);
    if (rc != VRT_OK) return rc;
    VRT_HIP(ctx, wait_stream(ctx->stream));
    if (ctx->stream_b) {
        VRT_HIP(ctx, wait_stream(ctx->stream_b));
        ctx->b_pending = false;
    }
    return VRT_OK;
}

double vrt_last_kernel_ms(vrt_ctx *ctx) {
    if (!ctx) return -1.0;
    DeviceGuard dg(ctx->device);
    if (finish_frame(ctx) != VRT_OK) return -1.0;
    return ctx->timing_valid ? ctx->last_ms : -1.0;
}

static int read_back(vrt_ctx *ctx, void *dst, uint64_t nbytes, const void *src, uint64_t avail) {
    if (!ctx || !dst) return ctx ? fail(ctx, VRT_E_INVALID_ARG, "dst is NULL") : VRT_E_INVALID_ARG;
    if (!src) return fail(ctx, VRT_E_STATE, "target not allocated (want_float_output = 0?)");
    if (nbytes > avail) return fail(ctx, VRT_E_OUT_OF_RANGE, "read exceeds the target image");
    DeviceGuard dg(ctx->device);
    const int rc = finish_frame(ctx);
    if (rc != VRT_OK) return rc;
    const hipStream_t s = (ctx->last_slot == 1) ? ctx->stream_b : ctx->stream; // the stream that rendered the most recent frame
    VRT_HIP(ctx, hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToHost, s));
    VRT_HIP(ctx, wait_stream(s));
    return VRT_OK;
}

int vrt_read_rgba8(vrt_ctx *ctx, void *dst, uint64_t nbytes) {
    return read_back(ctx, dst, nbytes, ctx ? (ctx->last_slot == 1 ? ctx->target8_b : ctx->target8) : nullptr, ctx ? ctx->target_pixels * 4u : 0);
}
int vrt_read_rgba32f(vrt_ctx *ctx, void *dst, uint64_t nbytes) {
    return read_back(ctx, dst, nbytes, ctx ? (ctx->last_slot == 1 ? ctx->target32f_b : ctx->target32f) : nullptr,
                     ctx ? ctx->target_pixels * 16u : 0);
}
int vrt_set_target(vrt_ctx *ctx, void *rgba8, void *rgba32f) {
    if (!ctx) return VRT_E_INVALID_ARG;
    if (!rgba8) return fail(ctx, VRT_E_INVALID_ARG, "rgba8 target is NULL");
    if (ctx->stream_b) return fail(ctx, VRT_E_STATE, "vrt_set_target needs frames_in_flight = 1");
    DeviceGuard dg(ctx->device);
    if (ctx->own_t8 && ctx->target8) {
        VRT_HIP(ctx, hipStreamSynchronize(ctx->stream)); // frames in flight still write the owned image
        (void)hipFree(ctx->target8);
        ctx->own_t8 = false;
    }
    if (ctx->own_t32 && ctx->target32f) {
        VRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->target32f);
        ctx->own_t32 = false;
    }
    ctx->target8 = static_cast<uint8_t *>(rgba8);
    ctx->target32f = static_cast<float *>(rgba32f);
    ctx->params.target_rgba8 = ctx->target8;
    ctx->params.target_rgba32f = ctx->target32f;
    return VRT_OK;
}
void *vrt_device_target_rgba8(vrt_ctx *ctx) { return ctx ? (ctx->last_slot == 1 ? ctx->target8_b : ctx->target8) : nullptr; }
void *vrt_device_target_rgba32f(vrt_ctx *ctx) { return ctx ? (ctx->last_slot == 1 ? ctx->target32f_b : ctx->target32f) : nullptr; }
uint64_t vrt_target_bytes_rgba8(const vrt_ctx *ctx) { return ctx ? ctx->target_pixels * 4u : 0; }

int vrt_denoise(vrt_ctx *ctx, const vrt_denoise_config *cfg, uint32_t out_w, uint32_t out_h, uint32_t want_float) {
    if (!ctx || out_w == 0 || out_h == 0) return ctx ? fail(ctx, VRT_E_INVALID_ARG, "zero output size") : VRT_E_INVALID_ARG;
    if (ctx->shard.shard_count > 1u) return fail(ctx, VRT_E_STATE, "vrt_denoise needs the whole frame (unsharded context)");
    vrt_denoise_config c{20, 0.6f, 1.5f, 20.0f}; // GraphicsPipeline.Config, GraphicsPipeline.zig:34-39
    if (cfg) c = *cfg;
    if (c.samples < 0 || c.samples > 4096) return fail(ctx, VRT_E_INVALID_ARG, "samples out of range");
    DeviceGuard dg(ctx->device);
    // runs on the stream that rendered the most recent frame, so it is ordered after that frame
    const hipStream_t s = (ctx->last_slot == 1) ? ctx->stream_b : ctx->stream;
    if (ctx->denoised_w != out_w || ctx->denoised_h != out_h || (want_float && !ctx->d_denoised32f)) {
        VRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->stream_b) VRT_HIP(ctx, hipStreamSynchronize(ctx->stream_b));
        if (ctx->d_denoised8) (void)hipFree(ctx->d_denoised8);
        if (ctx->d_denoised32f) (void)hipFree(ctx->d_denoised32f);
        ctx->d_denoised8 = ctx->d_denoised32f = nullptr;
        VRT_HIP(ctx, hipMalloc(&ctx->d_denoised8, (size_t)out_w * out_h * 4u));
        if (want_float) VRT_HIP(ctx, hipMalloc(&ctx->d_denoised32f, (size_t)out_w * out_h * 16u));
        ctx->denoised_w = out_w;
        ctx->denoised_h = out_h;
    }
    const void *img = (ctx->last_slot == 1) ? ctx->target8_b : ctx->target8;
    VRT_HIP(ctx, hipEventRecord(ctx->ev_post_start, s));
    VRT_HIP(ctx, vrt::launch_denoise(img, (int)ctx->cfg.width, (int)ctx->cfg.height, c.samples, c.distribution_bias, c.pixel_multiplier,
                                     c.inverse_hue_tolerance, (int)out_w, (int)out_h, ctx->d_denoised8, want_float ? ctx->d_denoised32f : nullptr, s));
    VRT_HIP(ctx, hipEventRecord(ctx->ev_post_stop, s));
    ctx->post_timed = true;
    ctx->denoised_stream = s;
    return VRT_OK;
}

int vrt_region_begin(vrt_ctx *ctx) {
    if (!ctx) return VRT_E_INVALID_ARG;
    DeviceGuard dg(ctx->device);
    for (hipEvent_t &e : ctx->ev_region)
        if (!e) VRT_HIP(ctx, hipEventCreate(&e));
    VRT_HIP(ctx, hipEventRecord(ctx->ev_region[0], ctx->stream));
    if (ctx->stream_b) VRT_HIP(ctx, hipEventRecord(ctx->ev_region[2], ctx->stream_b));
    return VRT_OK;
}

int vrt_region_end(vrt_ctx *ctx, double *ms) {
    if (!ctx || !ms) return ctx ? fail(ctx, VRT_E_INVALID_ARG, "ms is NULL") : VRT_E_INVALID_ARG;
    if (!ctx->ev_region[0]) return fail(ctx, VRT_E_STATE, "vrt_region_end without vrt_region_begin");
    DeviceGuard dg(ctx->device);
    VRT_HIP(ctx, hipEventRecord(ctx->ev_region[1], ctx->stream));
    if (ctx->stream_b) VRT_HIP(ctx, hipEventRecord(ctx->ev_region[3], ctx->stream_b));
    VRT_HIP(ctx, wait_event(ctx->ev_region[1]));
    float a1 = 0.0f, b0 = 0.0f, b1 = 0.0f;
    VRT_HIP(ctx, hipEventElapsedTime(&a1, ctx->ev_region[0], ctx->ev_region[1]));
    double begin = 0.0, end = a1; // (times relative to the primary stream's begin event)
    if (ctx->stream_b) {
        VRT_HIP(ctx, wait_event(ctx->ev_region[3]));
        VRT_HIP(ctx, hipEventElapsedTime(&b0, ctx->ev_region[0], ctx->ev_region[2]));
        VRT_HIP(ctx, hipEventElapsedTime(&b1, ctx->ev_region[0], ctx->ev_region[3]));
        begin = std::min(0.0, (double)b0);
        end = std::max((double)a1, (double)b1);
    }
    *ms = end - begin;
    return VRT_OK;
}

double vrt_last_denoise_ms(vrt_ctx *ctx) {
    if (!ctx || !ctx->post_timed) return -1.0;
    DeviceGuard dg(ctx->device);
    if (wait_event(ctx->ev_post_stop) != hipSuccess) return -1.0;
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ctx->ev_post_start, ctx->ev_post_stop) != hipSuccess) return -1.0;
    return (double)ms;
}

static int read_denoised(vrt_ctx *ctx, void *dst, uint64_t nbytes, const void *src, uint64_t avail) {
    if (!ctx || !dst) return VRT_E_INVALID_ARG;
    if (!src) return fail(ctx, VRT_E_STATE, "no denoised image (call vrt_denoise first; want_float for the float image)");
    if (nbytes > avail) return fail(ctx, VRT_E_OUT_OF_RANGE, "read exceeds the denoised image");
    DeviceGuard dg(ctx->device);
    VRT_HIP(ctx, hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToHost, ctx->denoised_stream));
    VRT_HIP(ctx, hipStreamSynchronize(ctx->denoised_stream));
    return VRT_OK;
}
int vrt_read_denoised_rgba8(vrt_ctx *ctx, void *dst, uint64_t nbytes) {
    return read_denoised(ctx, dst, nbytes, ctx ? ctx->d_denoised8 : nullptr, ctx ? (uint64_t)ctx->denoised_w * ctx->denoised_h * 4u : 0);
}
int vrt_read_denoised_rgba32f(vrt_ctx *ctx, void *dst, uint64_t nbytes) {
    return read_denoised(ctx, dst, nbytes, ctx ? ctx->d_denoised32f : nullptr, ctx ? (uint64_t)ctx->denoised_w * ctx->denoised_h * 16u : 0);
}
void *vrt_device_denoised_rgba8(vrt_ctx *ctx) { return ctx ? ctx->d_denoised8 : nullptr; }

int vrt_get_shard_info(const vrt_ctx *ctx, vrt_shard_info *out) {
    if (!ctx || !out) return VRT_E_INVALID_ARG;
    *out = ctx->shard;
    return VRT_OK;
}

int vrt_assemble_frame(vrt_ctx *ctx, const void *gathered, void *dst_frame, uint32_t bytes_per_pixel) {
    if (!ctx || !gathered || !dst_frame) return ctx ? fail(ctx, VRT_E_INVALID_ARG, "NULL buffer") : VRT_E_INVALID_ARG;
    if (bytes_per_pixel != 4 && bytes_per_pixel != 16) return fail(ctx, VRT_E_INVALID_ARG, "bytes_per_pixel must be 4 or 16");
    DeviceGuard dg(ctx->device);
    VRT_HIP(ctx, vrt::launch_assemble(gathered, dst_frame, bytes_per_pixel, ctx->cfg.width, ctx->cfg.height, ctx->shard.tiles_x,
                                      ctx->shard.shard_count, ctx->shard.tiles_per_rank, ctx->own, ctx->stream));
    return VRT_OK;
}

int vrt_trace_wave_timeline(vrt_ctx *ctx, const vrt_camera_device *camera, const vrt_sun_device *sun, uint64_t *out, uint64_t capacity_pairs,
                            uint64_t *n_pairs) {
    if (!ctx || !out || !n_pairs) return VRT_E_INVALID_ARG;
    // (the cost-ordered launch has spare workgroups for the halves of split tiles: their waves are listed too; a workgroup that
    // stayed idle leaves zeros)
    const uint64_t waves = ((uint64_t)ctx->shard.owned_tiles + (ctx->params.tile_order == 5u ? ctx->params.sched_units : 0u)) * 4u;
    if (capacity_pairs < waves) return fail(ctx, VRT_E_OUT_OF_RANGE, "timeline buffer too small");
    DeviceGuard dg(ctx->device);
    const size_t bytes = std::max<size_t>(waves * 16u, 32u * sizeof(unsigned long long)); // (the profile build of vrt_path_kernel writes 20 words)
#ifndef VRT_DEV_PROFILE
    {
        // the persistent-lane kernel has no wave -> tile map to report: refuse instead of returning zeros
        const vrt::KernelFn would = (camera && camera->max_bounce > 1) ? (ctx->d_counters ? ctx->product[0] : ctx->kernel) : nullptr;
        if (would && vrt::is_path_kernel(would)) return fail(ctx, VRT_E_STATE, "vrt_trace_wave_timeline: frames with bounces run vrt_path_kernel on this context (no per-tile waves)");
    }
#endif
    unsigned long long *d = nullptr;
    VRT_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&d), bytes));
    VRT_HIP(ctx, hipMemsetAsync(d, 0, bytes, ctx->stream));
    ctx->params.wave_timeline = d;
    const uint32_t split_all = ctx->params.split_all;
    ctx->params.split_all = 0u; // (one row of the timeline per wave of a whole tile)
    const int rc = do_dispatch(ctx, camera, sun, 1, true);
    ctx->params.split_all = split_all;
    ctx->params.wave_timeline = nullptr;
    if (rc != VRT_OK) {
        (void)hipFree(d);
        return rc;
    }
    VRT_HIP(ctx, hipMemcpyAsync(out, d, waves * 16u, hipMemcpyDeviceToHost, ctx->stream));
    VRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(d);
    *n_pairs = waves;
    return VRT_OK;
}

// ---- multi-GPU frame pipeline ------------------------------------------------------------------------
#define VRT_NCCL(ctx, d, call)                                                                               \
    do {                                                                                                     \
        const ncclResult_t r_ = (call);                                                                      \
        if (r_ != ncclSuccess) return fail(ctx, VRT_E_RCCL, std::string(#call) + ": " + (d)->api.GetErrorString(r_)); \
    } while (0)

int vrt_dist_unique_id(const char *rccl_path, void *out_id128) {
    if (!rccl_path || !out_id128) return VRT_E_INVALID_ARG;
    RcclApi api;
    std::string err;
    if (!api.load(rccl_path, err)) return fail(nullptr, VRT_E_RCCL, err);
    ncclUniqueId id;
    const ncclResult_t r = api.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, VRT_E_RCCL, std::string("ncclGetUniqueId: ") + api.GetErrorString(r));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(out_id128, &id, sizeof id);
    return VRT_OK;
}

int vrt_dist_init_batched(vrt_ctx *ctx, const char *rccl_path, const void *id128, int rank, int world, uint32_t frames_in_flight,
                          uint32_t frames_per_launch) {
    if (!ctx || !rccl_path || !id128) return ctx ? fail(ctx, VRT_E_INVALID_ARG, "NULL argument") : VRT_E_INVALID_ARG;
    if (ctx->dist) return fail(ctx, VRT_E_STATE, "vrt_dist_init called twice");
    if (world < 1 || rank < 0 || rank >= world) return fail(ctx, VRT_E_INVALID_ARG, "bad rank / world");
    if ((uint32_t)world != ctx->shard.shard_count || (uint32_t)rank != ctx->shard.shard_rank)
        return fail(ctx, VRT_E_INVALID_ARG, "context was not created with shard_rank / shard_count = rank / world");
    if (ctx->stream_b || ctx->cfg.stream || ctx->cfg.external_target_rgba8 || ctx->d_counters)
        return fail(ctx, VRT_E_STATE, "the multi-GPU pipeline owns its streams and targets (no frames_in_flight=2, caller stream/target or counters)");
    if (frames_in_flight == 0) frames_in_flight = 4;
    if (frames_in_flight > kMaxDistSlots) return fail(ctx, VRT_E_INVALID_ARG, "at most 8 launches in flight");
    if (frames_per_launch == 0) frames_per_launch = 1;
    if (frames_per_launch > (uint32_t)vrt::kMaxBatchFrames) return fail(ctx, VRT_E_INVALID_ARG, "at most 8 frames per launch");
    DeviceGuard dg(ctx->device);
    Dist *d = new (std::nothrow) Dist();
    if (!d) return fail(ctx, VRT_E_OOM, "host allocation failed");
    std::string err;
    if (!d->api.load(rccl_path, err)) {
        delete d;
        return fail(ctx, VRT_E_RCCL, err);
    }
    if (ctx->cfg.tuning_flags & VRT_TUNE_DIST_NO_BROADCAST) d->api.Broadcast = nullptr; // the send / recv form of vrt_dist_broadcast
    d->rank = rank;
    d->world = world;
    d->nslots = frames_in_flight;
    d->batch = frames_per_launch;
    // shards travel as RGB (the alpha of the RGBA8 target is the constant 255): a quarter less for rank 0's links to take in
    d->shard_bytes = (size_t)ctx->shard.tiles_per_rank * vrt::kTileW * vrt::kTileH * 3u;
    ctx->dist = d; // from here free_ctx cleans up
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof id);
    VRT_NCCL(ctx, d, d->api.CommInitRank(&d->comm, world, id, rank));
    const size_t region = d->shard_bytes * d->batch; // one rank's shards of a batch, frame-major
    for (uint32_t i = 0; i < d->nslots; i++) {
        DistSlot &sl = d->slots[i];
        VRT_HIP(ctx, hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        VRT_HIP(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
        if (rank == 0) {
            VRT_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&sl.gathered), region * (size_t)world));
            VRT_HIP(ctx, hipMemsetAsync(sl.gathered, 0, region * (size_t)world, ctx->stream));
            sl.shard = sl.gathered; // rank 0's own tiles are region 0 of the gathered buffer: no copy
            VRT_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&sl.frame), (size_t)ctx->cfg.width * ctx->cfg.height * 4u * d->batch));
        } else {
            VRT_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&sl.shard), region));
            VRT_HIP(ctx, hipMemsetAsync(sl.shard, 0, region, ctx->stream));
        }
    }
    if (!ctx->ev_upload) VRT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_upload, hipEventDisableTiming));
    // everything enqueued on the primary stream so far (scene uploads, clears) precedes the first frame of every slot
    VRT_HIP(ctx, hipEventRecord(ctx->ev_upload, ctx->stream));
    ctx->upload_seq++;
    return VRT_OK;
}

int vrt_dist_init(vrt_ctx *ctx, const char *rccl_path, const void *id128, int rank, int world, uint32_t frames_in_flight) {
    return vrt_dist_init_batched(ctx, rccl_path, id128, rank, world, frames_in_flight, 1);
}

int vrt_dist_frame(vrt_ctx *ctx, const vrt_camera_device *camera, const vrt_sun_device *sun) {
    if (!ctx || !ctx->dist) return ctx ? fail(ctx, VRT_E_STATE, "vrt_dist_init has not been called") : VRT_E_INVALID_ARG;
    Dist *d = ctx->dist;
    DeviceGuard dg(ctx->device);
    vrt::KernelFn fn = nullptr;
    const int rcp = pre_dispatch(ctx, camera, sun, &fn); // (a scene write it may have to do launches the queued frames first)
    if (rcp != VRT_OK) return rcp;
    // the pipeline's launches overlap on several streams and write RGB shards with a row-of-eight-lanes shuffle: both
    // need the lockstep kernel (vrt_path_kernel has one pixel counter per stream and no fixed lane -> pixel map)
    if (fn == ctx->kernel || (ctx->kernel_grid_exit && fn == ctx->kernel_grid_exit)) fn = ctx->kernel_lockstep;
    if (d->npend > 0 && d->pend_fn != fn) { // another kernel specialisation (bounces / samples changed): not in the same launch
        const int rcf = dist_flush(ctx);
        if (rcf != VRT_OK) return rcf;
    }
    note_kernel(ctx, fn);
    d->pend[d->npend].cam = *camera;
    d->pend[d->npend].sun = *sun;
    d->pend_fn = fn;
    d->npend++;
    return d->npend >= d->batch ? dist_flush(ctx) : VRT_OK;
}

} // extern "C"

namespace {
// vrt_dist_profile: add the stage times of the slot's most recent launch to the sums (wait: the launch is known to have
// finished; otherwise only if it has)
void dist_collect(Dist *d, DistSlot &sl, bool finished) {
    if (!sl.marked) return;
    sl.marked = false;
    if (!finished && hipEventQuery(sl.mark[3]) != hipSuccess) return;
    float ms[3] = {0.0f, 0.0f, 0.0f};
    for (int k = 0; k < 3; k++)
        if (hipEventElapsedTime(&ms[k], sl.mark[k], sl.mark[k + 1]) != hipSuccess) return;
    for (int k = 0; k < 3; k++) d->prof_ms[k] += (double)ms[k];
    d->prof_launches++;
    d->prof_frames += sl.frames;
}

// Launch the queued frames: one kernel over (tiles of this rank) x (frames), ONE collective, one un-swizzle per frame.
int dist_flush(vrt_ctx *ctx) {
    Dist *d = ctx->dist;
    if (d->failed) return fail(ctx, VRT_E_RCCL, "an earlier collective of this context failed: its ranks are out of step, destroy it");
    const uint32_t n = d->npend;
    if (n == 0) return VRT_OK;
    d->npend = 0; // (also on failure: the frames are dropped, not retried)
    const int k = (int)(d->frame_no % d->nslots);
    DistSlot &sl = d->slots[k];
    if (sl.seen_upload != ctx->upload_seq) { // scene writes happen on the primary stream
        VRT_HIP(ctx, hipStreamWaitEvent(sl.stream, ctx->ev_upload, 0));
        sl.seen_upload = ctx->upload_seq;
    }
    // 1. this rank's tiles of the n frames, packed tile-major, frame after frame, straight into the buffer RCCL sends
    //    (rank 0: into region 0 of `gathered`)
    vrt::TraceParams pk = ctx->params;
    if (ctx->order_auto) pk.tile_order = 3u; // (no cost feedback across the slots of the pipeline yet)
    for (uint32_t f = 0; f < n; f++) pk.pcs[f] = d->pend[f];
    pk.target_rgba8 = sl.shard;
    pk.target_rgba32f = nullptr;
    pk.packed_tiles = 1u;
    pk.packed_rgb = 1u;
    // (a launch of n frames of this rank's tiles: half-tile workgroups while its waves do not fill the SIMDs twice)
    pk.split_all = (ctx->split_ok && pk.tile_order == 3u && (uint64_t)ctx->shard.owned_tiles * 4u * n <= 2ull * ctx->simds) ? 1u : 0u;
    pk.batch_target_stride = (uint32_t)d->shard_bytes;
    if (sl.marked) dist_collect(d, sl, false); // (profile: the slot's previous launch, if it has finished; else that sample is dropped)
    const bool mark = d->profile;
    if (mark) {
        for (hipEvent_t &e : sl.mark)
            if (!e) VRT_HIP(ctx, hipEventCreate(&e));
        VRT_HIP(ctx, hipEventRecord(sl.mark[0], sl.stream));
    }
    VRT_HIP(ctx, vrt::launch_trace(d->pend_fn, pk, ctx->lds_bytes, sl.stream, n));
    if (mark) VRT_HIP(ctx, hipEventRecord(sl.mark[1], sl.stream));
    // 2. the one collective of the batch: every rank's shards -> rank 0 (grouped point-to-point = gather)
    const size_t region = d->shard_bytes * d->batch;
    if (d->world > 1) {
        VRT_NCCL(ctx, d, d->api.GroupStart());
        // a failing Send / Recv must not leave the group open on this rank: close it, then report, and the context
        // stays failed (every later vrt_dist_* call returns VRT_E_RCCL) because its peers are now out of step
        ncclResult_t first_bad = ncclSuccess;
        if (d->rank == 0) {
            for (int r = 1; r < d->world && first_bad == ncclSuccess; r++)
                first_bad = d->api.Recv(sl.gathered + (size_t)r * region, d->shard_bytes * n, ncclUint8, r, d->comm, sl.stream);
        } else {
            first_bad = d->api.Send(sl.shard, d->shard_bytes * n, ncclUint8, 0, d->comm, sl.stream);
        }
        const ncclResult_t end = d->api.GroupEnd();
        if (first_bad != ncclSuccess || end != ncclSuccess) {
            d->failed = true;
            return fail(ctx, VRT_E_RCCL, std::string("RCCL gather failed: ") + d->api.GetErrorString(first_bad != ncclSuccess ? first_bad : end));
        }
    }
    if (mark) VRT_HIP(ctx, hipEventRecord(sl.mark[2], sl.stream));
    // 3. rank 0: tile-major shards -> row-major frames
    if (d->rank == 0) {
        // (one launch for the n frames of the batch: grid.z)
        VRT_HIP(ctx, vrt::launch_assemble_rgb(sl.gathered, sl.frame, ctx->cfg.width, ctx->cfg.height, ctx->shard.tiles_x, (uint32_t)d->world,
                                              ctx->shard.tiles_per_rank * d->batch, ctx->own, sl.stream, n, (uint32_t)d->shard_bytes));
    }
    if (mark) VRT_HIP(ctx, hipEventRecord(sl.mark[3], sl.stream));
    sl.marked = mark;
    VRT_HIP(ctx, hipEventRecord(sl.done, sl.stream));
    sl.used = true;
    sl.frames = n;
    d->last_slot = k;
    d->frame_no++;
    return VRT_OK;
}
} // namespace

extern "C" {

int vrt_dist_wait(vrt_ctx *ctx) {
    if (!ctx || !ctx->dist) return ctx ? fail(ctx, VRT_E_STATE, "vrt_dist_init has not been called") : VRT_E_INVALID_ARG;
    DeviceGuard dg(ctx->device);
    const int rcf = dist_flush(ctx);
    if (rcf != VRT_OK) return rcf;
    for (uint32_t i = 0; i < ctx->dist->nslots; i++) VRT_HIP(ctx, wait_stream(ctx->dist->slots[i].stream));
    VRT_HIP(ctx, wait_stream(ctx->stream));
    for (uint32_t i = 0; i < ctx->dist->nslots; i++) dist_collect(ctx->dist, ctx->dist->slots[i], true);
    return VRT_OK;
}

int vrt_dist_profile(vrt_ctx *ctx, uint32_t enable) {
    if (!ctx || !ctx->dist) return ctx ? fail(ctx, VRT_E_STATE, "vrt_dist_init has not been called") : VRT_E_INVALID_ARG;
    Dist *d = ctx->dist;
    d->profile = enable != 0u;
    d->prof_launches = d->prof_frames = 0;
    d->prof_ms[0] = d->prof_ms[1] = d->prof_ms[2] = 0.0;
    return VRT_OK;
}

int vrt_dist_stats(vrt_ctx *ctx, double out[8]) {
    if (!ctx || !ctx->dist || !out) return ctx ? fail(ctx, VRT_E_STATE, "vrt_dist_init has not been called / out NULL") : VRT_E_INVALID_ARG;
    const Dist *d = ctx->dist;
    const double n = d->prof_launches ? (double)d->prof_launches : 1.0;
    out[0] = (double)d->prof_launches;
    out[1] = (double)d->prof_frames;
    out[2] = d->prof_ms[0] / n;
    out[3] = d->prof_ms[1] / n;
    out[4] = d->rank == 0 ? d->prof_ms[2] / n : 0.0; // (only rank 0 un-swizzles; elsewhere the interval holds two event records)
    out[5] = (double)ctx->shard.owned_tiles;
    out[6] = (double)d->shard_bytes;
    out[7] = (double)d->batch;
    return VRT_OK;
}

int vrt_dist_info(vrt_ctx *ctx, int32_t out[4]) {
    if (!ctx || !ctx->dist || !out) return ctx ? fail(ctx, VRT_E_STATE, "vrt_dist_init has not been called / out NULL") : VRT_E_INVALID_ARG;
    Dist *d = ctx->dist;
    int rank = d->rank, world = d->world;
    // what the communicator itself says, not what vrt_dist_init was told
    if (d->comm && d->api.CommCount && d->api.CommUserRank) {
        VRT_NCCL(ctx, d, d->api.CommCount(d->comm, &world));
        VRT_NCCL(ctx, d, d->api.CommUserRank(d->comm, &rank));
    }
    out[0] = rank;
    out[1] = world;
    out[2] = (int32_t)d->batch;
    out[3] = (int32_t)d->nslots;
    return VRT_OK;
}

int vrt_dist_read_frame(vrt_ctx *ctx, void *dst, uint64_t nbytes) {
    if (!ctx || !ctx->dist || !dst) return ctx ? fail(ctx, VRT_E_STATE, "vrt_dist_init has not been called / dst NULL") : VRT_E_INVALID_ARG;
    Dist *d = ctx->dist;
    if (d->rank != 0) return fail(ctx, VRT_E_STATE, "only rank 0 holds the assembled frame");
    if (nbytes > (uint64_t)ctx->cfg.width * ctx->cfg.height * 4u) return fail(ctx, VRT_E_OUT_OF_RANGE, "read exceeds the frame");
    DeviceGuard dg(ctx->device);
    // A launch carries a collective, so every rank must launch the same frames together: rank 0 cannot launch a partial
    // queue on its own.  Queues empty themselves when full and in vrt_dist_wait, which every rank calls.
    if (d->npend) return fail(ctx, VRT_E_STATE, "frames are still queued for the next launch: call vrt_dist_wait on every rank first");
    if (d->last_slot < 0) return fail(ctx, VRT_E_STATE, "no frame submitted yet");
    DistSlot &sl = d->slots[d->last_slot];
    const size_t frame_bytes = (size_t)ctx->cfg.width * ctx->cfg.height * 4u;
    VRT_HIP(ctx, hipMemcpyAsync(dst, sl.frame + (size_t)(sl.frames - 1u) * frame_bytes, nbytes, hipMemcpyDeviceToHost, sl.stream));
    VRT_HIP(ctx, hipStreamSynchronize(sl.stream));
    return VRT_OK;
}

int vrt_dist_selftest(vrt_ctx *ctx) {
    if (!ctx || !ctx->dist) return ctx ? fail(ctx, VRT_E_STATE, "vrt_dist_init has not been called") : VRT_E_INVALID_ARG;
    Dist *d = ctx->dist;
    DeviceGuard dg(ctx->device);
    const size_t n = d->shard_bytes;
    uint8_t *a = nullptr, *b = nullptr;
    VRT_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&a), n));
    VRT_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&b), n));
    std::string host(n, '\0'), back(n, '\0');
    for (size_t i = 0; i < n; i++) host[i] = (char)((i * 131u + 7u) & 0xFFu);
    hipStream_t s = d->slots[0].stream;
    int rc = VRT_OK;
    do {
        if (hipMemcpyAsync(a, host.data(), n, hipMemcpyHostToDevice, s) != hipSuccess || hipMemsetAsync(b, 0, n, s) != hipSuccess) {
            rc = fail(ctx, VRT_E_HIP, "selftest copy failed");
            break;
        }
        ncclResult_t r = d->api.GroupStart();
        if (r == ncclSuccess) r = d->api.Send(a, n, ncclUint8, d->rank, d->comm, s);
        if (r == ncclSuccess) r = d->api.Recv(b, n, ncclUint8, d->rank, d->comm, s);
        const ncclResult_t r2 = d->api.GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) {
            rc = fail(ctx, VRT_E_RCCL, std::string("self send/recv: ") + d->api.GetErrorString(r));
            break;
        }
        if (hipMemcpyAsync(&back[0], b, n, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            rc = fail(ctx, VRT_E_HIP, "selftest read-back failed");
            break;
        }
        if (back != host) rc = fail(ctx, VRT_E_RCCL, "self send/recv returned different bytes");
    } while (0);
    (void)hipFree(a);
    (void)hipFree(b);
    return rc;
}

// Replica update (delta upload "+ replica broadcast"): one collective per dirty range.
int vrt_dist_broadcast(vrt_ctx *ctx, vrt_buffer_id id, uint64_t byte_offset, uint64_t nbytes, int root) {
    if (!ctx || !ctx->dist) return ctx ? fail(ctx, VRT_E_STATE, "vrt_dist_init has not been called") : VRT_E_INVALID_ARG;
    Dist *d = ctx->dist;
    if (d->failed) return fail(ctx, VRT_E_RCCL, "an earlier collective failed: the ranks are out of step, destroy the context");
    if ((int)id < 0 || id >= VRT_BUF_COUNT) return fail(ctx, VRT_E_INVALID_ARG, "bad buffer id");
    if (root < 0 || root >= d->world) return fail(ctx, VRT_E_INVALID_ARG, "root is not a rank of the communicator");
    if (byte_offset > ctx->dsize[id] || nbytes > ctx->dsize[id] - byte_offset)
        return fail(ctx, VRT_E_OUT_OF_RANGE, "range exceeds device buffer (DestOutOfDeviceMemory)");
    if (nbytes == 0) return VRT_OK;
    DeviceGuard dg(ctx->device);
    // a scene write: frames queued or in flight see the scene as it was, later ones as it becomes
    const int rcb = begin_scene_write(ctx);
    if (rcb != VRT_OK) return rcb;
    uint8_t *range = static_cast<uint8_t *>(ctx->dbuf[id]) + byte_offset;
    ncclResult_t r = ncclSuccess;
    {
        if (d->api.Broadcast) { // (also with a single rank: the call is then RCCL's own no-op, and the binding is exercised)
            r = d->api.Broadcast(range, range, (size_t)nbytes, ncclUint8, root, d->comm, ctx->stream);
        } else {
            r = d->api.GroupStart();
            if (d->rank == root) {
                for (int peer = 0; peer < d->world && r == ncclSuccess; peer++)
                    if (peer != root) r = d->api.Send(range, (size_t)nbytes, ncclUint8, peer, d->comm, ctx->stream);
            } else if (r == ncclSuccess) {
                r = d->api.Recv(range, (size_t)nbytes, ncclUint8, root, d->comm, ctx->stream);
            }
            const ncclResult_t r2 = d->api.GroupEnd(); // (always: an open group would swallow every later call)
            if (r == ncclSuccess) r = r2;
        }
    }
    if (r != ncclSuccess) {
        d->failed = true;
        return fail(ctx, VRT_E_RCCL, std::string("replica broadcast: ") + d->api.GetErrorString(r));
    }
    if (id == VRT_BUF_GRID_STATE && d->rank != root) {
        // the kernel takes the UBO through its argument block: bring the host mirror up to date
        VRT_HIP(ctx, hipMemcpyAsync(reinterpret_cast<uint8_t *>(&ctx->params.grid) + byte_offset, range, nbytes, hipMemcpyDeviceToHost, ctx->stream));
        VRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    mark_dirty(ctx, id, byte_offset, nbytes);
    return end_scene_write(ctx);
}

int vrt_get_counters(vrt_ctx *ctx, vrt_counters *out) {
    if (!ctx || !out) return VRT_E_INVALID_ARG;
    if (!ctx->d_counters) return fail(ctx, VRT_E_STATE, "context created with enable_counters = 0");
    DeviceGuard dg(ctx->device);
    const int rc = finish_frame(ctx);
    if (rc != VRT_OK) return rc;
    vrt::DeviceCounters h;
    VRT_HIP(ctx, hipMemcpyAsync(&h, ctx->d_counters, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    VRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out->rays = h.rays;
    out->status_loads = h.status_loads;
    out->bricks_entered = h.bricks_entered;
    out->voxel_steps = h.voxel_steps;
    out->hits = h.hits;
    out->grid_steps = h.grid_steps;
    return VRT_OK;
}

int vrt_get_wave_counters(vrt_ctx *ctx, uint64_t out[3]) {
    if (!ctx || !out) return VRT_E_INVALID_ARG;
    if (!ctx->d_counters) return fail(ctx, VRT_E_STATE, "context created with enable_counters = 0");
    DeviceGuard dg(ctx->device);
    const int rc = finish_frame(ctx);
    if (rc != VRT_OK) return rc;
    vrt::DeviceCounters h;
    VRT_HIP(ctx, hipMemcpyAsync(&h, ctx->d_counters, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    VRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out[0] = h.wave_grid_iters;
    out[1] = h.wave_brick_walks;
    out[2] = h.wave_voxel_iters;
    return VRT_OK;
}

// VoxelRT.init's transferGridState (VoxelRT.zig:62) plus the five arrays in full.
int vrt_upload_grid(vrt_ctx *ctx, vrt_grid *gh) {
    if (!ctx || !gh) return VRT_E_INVALID_ARG;
    vrt::BrickGrid *g = reinterpret_cast<vrt::BrickGrid *>(gh);
    if (g->brickDimension() != ctx->cfg.brick_dimension) return fail(ctx, VRT_E_INVALID_ARG, "grid brick_dimension differs from the context");
    static const vrt_buffer_id ids[6] = {VRT_BUF_GRID_STATE,      VRT_BUF_BRICK_STATUS,      VRT_BUF_BRICK_INDEX,
                                         VRT_BUF_BRICK_OCCUPANCY, VRT_BUF_BRICK_START_INDEX, VRT_BUF_MATERIAL_INDEX};
    for (vrt_buffer_id id : ids) {
        uint64_t n = 0;
        const void *ptr = g->dataFor(id, &n);
        if (n != ctx->dsize[id]) return fail(ctx, VRT_E_INVALID_ARG, "grid array size differs from the context's buffer");
        const int rc = vrt_upload(ctx, id, 0, ptr, n);
        if (rc != VRT_OK) return rc;
        if (vrt::DeviceDataDelta *d = g->deltaFor(id)) {
            std::lock_guard<std::mutex> lk(d->mutex);
            d->resetDelta();
        }
    }
    return VRT_OK;
}

// VoxelRT.updateGridDelta, VoxelRT.zig:107-172
int vrt_update_grid_delta(vrt_ctx *ctx, vrt_grid *gh) {
    if (!ctx || !gh) return VRT_E_INVALID_ARG;
    vrt::BrickGrid *g = reinterpret_cast<vrt::BrickGrid *>(gh);
    static const vrt_buffer_id ids[5] = {VRT_BUF_BRICK_STATUS, VRT_BUF_BRICK_INDEX, VRT_BUF_BRICK_OCCUPANCY, VRT_BUF_BRICK_START_INDEX,
                                         VRT_BUF_MATERIAL_INDEX};
    for (vrt_buffer_id id : ids) {
        vrt::DeviceDataDelta *d = g->deltaFor(id);
        std::lock_guard<std::mutex> lk(d->mutex);
        if (d->state != vrt::DeviceDataDelta::DeltaState::active) continue;
        uint64_t n = 0;
        const uint8_t *base = static_cast<const uint8_t *>(g->dataFor(id, &n));
        const size_t es = g->elementSize(id);
        const int rc = vrt_upload(ctx, id, (uint64_t)d->from * es, base + d->from * es, (uint64_t)(d->to - d->from) * es);
        if (rc != VRT_OK) return rc;
        d->resetDelta();
    }
    return VRT_OK;
}

} // extern "C"
