// vrt_ctx.h — the context behind the C ABI of include/vrt_hip.h, shared by the units that implement it:
//   vrt_api.hip    creation / destruction, the seven uploads (pinned staging ring), read-back, counters
//   vrt_frame.hip  one frame: refresh of the derived structures, kernel choice, tile schedule, launch, timing
//   vrt_dist.hip   the multi-GPU frame pipeline (RCCL through dlopen)
//   vrt_post.hip   the present / denoise pass
// Replaces src/modules/voxel_rt/ComputePipeline.zig (init / dispatch / deinit) and the Pipeline.transfer* family
// (Pipeline.zig:560-652) with its StagingRamp (render/StagingRamp.zig) for this one path.
#pragma once
#include <hip/hip_runtime.h>
#include <chrono>
#include <string>
#include <vector>
#include "vrt_internal.h"
#include "vrt_kernels.h"

namespace vrt {
// ---- launchers and kernel selection (vrt_trace.hip, vrt_inst_*.hip, vrt_post.hip) ----
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
hipError_t launch_build_cell_material(const TraceParams &p, uint32_t brick_dimension, uint64_t brick_alloc, uint64_t cell_lo, uint64_t cell_hi, uint64_t slot_lo,
                                      uint64_t slot_hi, uint64_t mat_lo, uint64_t mat_hi, hipStream_t stream);
hipError_t launch_check_start_is_slot(const TraceParams &p, uint32_t brick_dimension, uint64_t brick_alloc, hipStream_t stream);
hipError_t launch_check_materials_plain(const TraceParams &p, uint32_t count, hipStream_t stream);
hipError_t launch_denoise(const void *img, int W, int H, int samples, float bias, float mult, float tol, int out_w, int out_h, void *out_u8,
                          void *out_f32, hipStream_t stream);
hipError_t launch_assemble(const void *gathered, void *frame, uint32_t bytes_per_pixel, uint32_t width, uint32_t height, uint32_t tiles_x,
                           uint32_t shard_count, uint32_t tiles_per_rank, const TileOwnership &own, hipStream_t stream, uint32_t frames = 1,
                           uint32_t frame_src_stride_pixels = 0);

// Everything a context owns on the runtime's side — device memory, pinned host memory, events, streams — in ONE container: made
// through it, released by it in reverse order of creation (streams drained first).  The context keeps plain pointers for use; none
// of them is freed anywhere else.  `drop` releases one item early (a buffer that is replaced by a larger one).
class Resources {
public:
    template <class T> hipError_t device(T **out, size_t bytes) {
        void *p = nullptr;
        const hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) {
            items_.push_back({Kind::Device, p});
            *out = static_cast<T *>(p);
        }
        return e;
    }
    template <class T> hipError_t pinned(T **out, size_t bytes) {
        void *p = nullptr;
        const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e == hipSuccess) {
            items_.push_back({Kind::Pinned, p});
            *out = static_cast<T *>(p);
        }
        return e;
    }
    hipError_t event(hipEvent_t *out, unsigned flags = hipEventDefault) {
        const hipError_t e = hipEventCreateWithFlags(out, flags);
        if (e == hipSuccess) items_.push_back({Kind::Event, *out});
        return e;
    }
    hipError_t stream(hipStream_t *out) {
        const hipError_t e = hipStreamCreateWithFlags(out, hipStreamNonBlocking);
        if (e == hipSuccess) items_.push_back({Kind::Stream, *out});
        return e;
    }
    template <class T> void drop(T *&p) { // (nullptr and pointers made elsewhere — a caller's target image — are left alone)
        for (size_t i = items_.size(); i-- > 0;)
            if (items_[i].handle == static_cast<void *>(p) && p) {
                release(items_[i]);
                items_.erase(items_.begin() + (ptrdiff_t)i);
                break;
            }
        p = nullptr;
    }
    void release_all() {
        for (const Item &it : items_)
            if (it.kind == Kind::Stream) (void)hipStreamSynchronize(static_cast<hipStream_t>(it.handle));
        for (size_t i = items_.size(); i-- > 0;) release(items_[i]);
        items_.clear();
    }
    size_t count() const { return items_.size(); }

private:
    enum class Kind : uint8_t { Device, Pinned, Event, Stream };
    struct Item {
        Kind kind;
        void *handle;
    };
    static void release(const Item &it) {
        switch (it.kind) {
        case Kind::Device: (void)hipFree(it.handle); break;
        case Kind::Pinned: (void)hipHostFree(it.handle); break;
        case Kind::Event: (void)hipEventDestroy(static_cast<hipEvent_t>(it.handle)); break;
        case Kind::Stream: (void)hipStreamDestroy(static_cast<hipStream_t>(it.handle)); break;
        }
    }
    std::vector<Item> items_;
};

// What ONE stream of frames traced by the persistent kernels (vrt_path_kernel, vrt_pool_kernel) needs for itself, because its frames
// run beside those of the context's other streams: the unit counters, the pool kernel's path records and the sample buffer
// (vrt_pool_kernel / vrt_path_kernel -> vrt_pool_resolve_kernel).  A context has one per stream it dispatches on (two with two frames
// in flight), the multi-GPU pipeline one per launch slot.
struct PersistentLane {
    uint32_t *work_counter = nullptr; // [kMaxBatchFrames]
    uint32_t *pool_paths = nullptr;   // [pool_groups * 4 waves][16 dwords][128 paths] (nullptr: this context selects no vrt_pool_kernel)
    float4 *samples = nullptr;        // [owned pixels][samples] terms of the sample loop's sum, sized by the frames asked for
    size_t sample_elems = 0;
};
} // namespace vrt

constexpr size_t kStagingSlotBytes = 32u << 20; // pinned staging slot
constexpr int kStagingSlots = 2;

struct Dist; // vrt_dist.hip

struct vrt_ctx {
    vrt_config cfg{};
    int device = 0;
    vrt::Resources res; // owns every allocation, event and stream named below (except a caller's stream / target images)
    hipStream_t stream = nullptr;
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
    uint8_t *target8 = nullptr;      // (ours unless cfg.external_target_rgba8 / vrt_set_target named a caller's)
    float *target32f = nullptr;
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
    Dist *dist = nullptr;                                        // multi-GPU frame pipeline (vrt_dist_*)
    uint32_t denoised_w = 0, denoised_h = 0;
    hipStream_t denoised_stream = nullptr;
    void *d_status_blocks = nullptr; // derived: 4x4x4 block words + block filter (vrt_trace.hip)
    int *d_cell_bounds = nullptr;    // derived: bounding box of the occupied cells (TraceParams::cell_bounds)
    // The host's copy of that box (read back behind every rebuild, never waited for): when it is, or nearly is, the grid, bounce
    // frames of a context whose kernel is the dilated-index path kernel are traced by its twin without steps-left counters.
    int *h_cell_bounds = nullptr;
    hipEvent_t ev_bounds = nullptr;
    bool bounds_pending = false, box_is_grid = false;
    // frames with bounces once the box is known to be the grid: vrt_pool_kernel where it applies, else vrt_path_kernel<..., DIL 2>;
    // *_path: that DIL-2 twin kept beside a pool kernel, for the frames the pool kernel cannot take
    vrt::KernelFn kernel_grid_exit = nullptr, product_grid_exit = nullptr, kernel_grid_exit_path = nullptr, product_grid_exit_path = nullptr;
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
    // derived (round 5): the material every solid voxel of a cell's brick shares, a byte per cell (TraceParams::cell_material; contexts that
    // select vrt_pool_kernel).  Refreshed like the by-cell occupancy, for what was written: cells [occ_cell_lo, hi) (status / index),
    // slots [cm_slot_lo, hi) (occupancy / start index), material entries [cm_mat_lo, hi) (bytes of binding 7); lo >= hi: none
    uint8_t *d_cell_material = nullptr;
    bool cell_material_dirty = true;
    uint64_t cm_cell_lo = 0, cm_cell_hi = ~0ull, cm_slot_lo = 0, cm_slot_hi = 0, cm_mat_lo = 0, cm_mat_hi = 0;
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
    // Round 6, VRT_TUNE_PRESENT_OWN_STREAM: contexts with two frames in flight run the present pass on a stream of its own, behind an
    // event of the frame it reads — the reference's graphics queue behind the compute queue's semaphore (Pipeline.zig:494-517); the frame
    // after next waits for the pass to have read its target (ev_post_done[slot]).  Not the default: see vrt_denoise.
    hipStream_t stream_post = nullptr;
    hipEvent_t ev_post_src[2] = {nullptr, nullptr}, ev_post_done[2] = {nullptr, nullptr};
    bool post_pending[2] = {false, false};
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
    vrt::KernelFn kernel_lockstep = nullptr; // ... the lockstep bounce loop (the multi-GPU pipeline's fallback where a launch slot has no sample buffer)
    // Round 5, bounce frames of scenes that stay in the caches: the size rule gives them to the lockstep kernel, and for a terrain that is
    // right (coherent rays: 1.5-1.8 x faster than vrt_pool_kernel on the reference app's run) — for a sparse field it is wrong by 1.6-2.2 x
    // (profiles/r05_pool_generalised_ab.txt).  Where both kernels can trace the frame (bounce_auto: the pool kernel of this configuration)
    // the library TIMES them — four frames on the primary stream, lockstep / pool / lockstep / pool, HIP events around each — and keeps
    // the faster (pool only if it wins by 15 %); a status upload starts the trials again.  Frames are the same bytes either way.
    vrt::KernelFn bounce_auto = nullptr;
    uint32_t auto_next = 0;                // trial frames launched so far (0 .. 4)
    int32_t auto_spp = 0, auto_bounce = 0; // what trial 0 traced: the other three must trace the same, or the trials start again
    bool auto_decided = false, auto_use_pool = false;
    hipEvent_t auto_ev[4][2] = {};
    float auto_ms[2] = {0.0f, 0.0f};       // what the trials measured: lockstep, pool (vrt_bounce_autotune_ms)
    vrt::PersistentLane lane[2];           // what the persistent kernels need per stream: [0] the primary stream, [1] stream_b
    size_t pool_stream_dwords = 0;         // size of one lane's pool_paths (0: this context selects no vrt_pool_kernel)
    uint32_t path_lds_bytes = 0;           // LDS block filter of vrt_path_kernel (0: grid not eligible)
    vrt::KernelFn kernel_single = nullptr; // specialisation for max_bounce <= 1
    vrt::KernelFn kernel_single1 = nullptr; // ... and samples_per_pixel == 1
    vrt::KernelFn product[3] = {};         // counting contexts: the product kernel that renders the frame read back, by shade (0 bounces, 1, 2)
    vrt::KernelFn last_fn = nullptr;       // the kernel of the most recent frame (vrt_kernel_name)
    vrt_shard_info shard{};
    std::string err;
    std::string kernel_name, name_note;
};

namespace vrt_impl {
// ---- errors ----
int fail(vrt_ctx *ctx, int code, const std::string &msg); // (ctx == nullptr: the thread's create error, vrt_last_error(NULL))
inline int hip_fail(vrt_ctx *ctx, hipError_t e, const char *what) {
    return fail(ctx, e == hipErrorOutOfMemory ? VRT_E_OOM : VRT_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define VRT_HIP(ctx, call)                                              \
    do {                                                                \
        hipError_t e_ = (call);                                         \
        if (e_ != hipSuccess) return vrt_impl::hip_fail(ctx, e_, #call); \
    } while (0)

struct DeviceGuard { // everything inside runs on `dev`; the caller's current device is restored on every return path
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// Waiting for a stream / an event: poll for up to a few milliseconds before handing the thread to the runtime's blocking wait.
// The blocking wait sleeps on an interrupt and wakes up tens of microseconds after the GPU has finished — as long as a whole
// frame of the headline workload (tools/experiments/short_trace.py: a 20-frame region took 1.45 ms on the GPU and 1.59 ms on the host's clock).
template <typename Query> hipError_t poll_then(Query query) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        for (int i = 0; i < 64; i++) {
            const hipError_t e = query();
            if (e != hipErrorNotReady) return e;
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) return hipErrorNotReady; // long frame: sleep instead
    }
}
inline hipError_t wait_stream(hipStream_t s) {
    const hipError_t e = poll_then([&] { return hipStreamQuery(s); });
    return e == hipErrorNotReady ? hipStreamSynchronize(s) : e;
}
inline hipError_t wait_event(hipEvent_t ev) {
    const hipError_t e = poll_then([&] { return hipEventQuery(ev); });
    return e == hipErrorNotReady ? hipEventSynchronize(ev) : e;
}

// ---- vrt_api.hip ----
void free_ctx(vrt_ctx *c);
// Scene writes happen on the primary stream.  With several streams of frames they must not overtake a frame that is still reading
// the scene on another stream, and later frames on those streams must see them.
int begin_scene_write(vrt_ctx *c);
int end_scene_write(vrt_ctx *c);
// which derived structures a write to scene buffer `id` invalidates (rebuilt before the next frame, pre_dispatch)
void mark_dirty(vrt_ctx *ctx, vrt_buffer_id id, uint64_t byte_offset, uint64_t nbytes);
// the unit counters and (contexts that select vrt_pool_kernel) the path records of one stream of persistent-kernel frames
int lane_init(vrt_ctx *c, vrt::PersistentLane &lane);

// ---- vrt_frame.hip ----
void note_kernel(vrt_ctx *c, vrt::KernelFn fn); // remember which kernel rendered the most recent frame (vrt_kernel_name reports what ran)
int finish_frame(vrt_ctx *c);                   // wait for the frame in flight (the fence wait of ComputePipeline.zig:423-434)
// Common front part of a frame: argument checks, push constants, derived-structure refresh, kernel choice.  Leaves the kernel to
// launch in *fn — for a counting context in *product_fn too: the product kernel that renders the frame read back.  Runs on the primary
// stream.  `lane` (may be nullptr) belongs to the stream `lane_stream` the frame will run on: a persistent kernel that needs or can use
// a sample buffer finds it there (grown if it has to be: lane_samples_ready); where it cannot be had the frame keeps a kernel that
// does without.  *with_samples: the frame's kernel is a persistent one and takes samples as its units of work from the lane's buffer.
// trial: >= 0 when this frame is trial `trial` of the bounce kernel's auto-tune (even: lockstep, odd: pool); -1 otherwise.
int pre_dispatch(vrt_ctx *ctx, const vrt_camera_device *camera, const vrt_sun_device *sun, vrt::PersistentLane *lane, hipStream_t lane_stream, vrt::KernelFn *fn,
                 vrt::KernelFn *product_fn, bool *with_samples, int trial = -1);
bool lane_samples_ready(vrt_ctx *ctx, vrt::PersistentLane &lane, uint64_t units, hipStream_t lane_stream);
uint64_t sample_units(const vrt_ctx *ctx, int samples_per_pixel); // units of a frame of this context (0: not a frame of units)
void lane_into_params(const vrt::PersistentLane &lane, bool with_samples, vrt::TraceParams &p);

// ---- vrt_dist.hip ----
int dist_flush(vrt_ctx *ctx);
bool dist_has_pending(const vrt_ctx *ctx);
int dist_order_primary_after_slots(vrt_ctx *ctx); // the primary stream waits for every launch in flight (a scene write follows)
bool dist_reserve_samples(vrt_ctx *ctx, uint64_t units); // every launch slot's sample buffer (vrt_reserve_samples)
void dist_destroy(vrt_ctx *ctx);                  // drains the slots' streams and closes the communicator (the memory is Resources')
} // namespace vrt_impl
