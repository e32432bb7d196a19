// vrt_frame.hip — one frame behind the C ABI: refresh of the structures derived from the scene buffers, the choice of the
// traversal kernel, the cost-feedback tile schedule, the launch on one of the context's streams, its timing.
// Replaces ComputePipeline.dispatch (src/modules/voxel_rt/ComputePipeline.zig:417-463).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>
#include "vrt_ctx.h"

using namespace vrt_impl;

namespace vrt_impl {

void note_kernel(vrt_ctx *c, vrt::KernelFn fn) {
    if (fn == c->last_fn) return;
    c->last_fn = fn;
    c->kernel_name = std::string(vrt::kernel_name_of(fn)) + c->name_note;
}

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

// The persistent kernels' units of work.  vrt_pool_kernel packs a path's sample index into 16 bits and its bounce count into 4, and
// divides a tile's number by the tiles per row with one multiplication (exact while tiles * tiles_x < 2^32); it takes SAMPLES from its
// counter (32 bits), whose terms of the sample sum it leaves in a buffer of 16 bytes per sample for vrt_pool_resolve_kernel — 2 GiB for
// a 4K frame of 16 samples, one buffer per stream of frames (PersistentLane).
uint64_t sample_units(const vrt_ctx *ctx, int samples_per_pixel) {
    if (ctx->cfg.tuning_flags & VRT_TUNE_NO_SAMPLE_UNITS) return 0;
    if (samples_per_pixel < 1) return 0; // (a frame of no samples is not a frame of units)
    const uint64_t units = (uint64_t)ctx->shard.owned_tiles * 256u * (uint64_t)samples_per_pixel;
    return units >= (1ull << 32) - (1ull << 26) ? 0 : units; // (the counter keeps counting, a chunk per wave, after it has run out)
}

// The lane's sample buffer for frames of `units` samples.  It is sized once by vrt_reserve_samples, or grows here with the frames
// asked for: the larger buffer is made FIRST (within half of the free memory), then the lane's stream is drained — the frames in
// flight on it still use the old one — and the old one freed; where the larger one cannot be had the lane keeps what it has and the
// frame a kernel that does without (a failed growth used to take the old buffer with it).
bool lane_samples_ready(vrt_ctx *ctx, vrt::PersistentLane &lane, uint64_t units, hipStream_t lane_stream) {
    if (units == 0) return false;
    if (lane.sample_elems >= units) return true;
    size_t mem_free = 0, mem_total = 0;
    if (hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) return false;
    const size_t bytes = (size_t)units * sizeof(float4);
    if (bytes > mem_free / 2u) return false;
    float4 *grown = nullptr;
    if (ctx->res.device(&grown, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    if (lane.samples) {
        if (hipStreamSynchronize(lane_stream) != hipSuccess) {
            ctx->res.drop(grown);
            return false;
        }
        ctx->res.drop(lane.samples);
    }
    lane.samples = grown;
    lane.sample_elems = (size_t)units;
    return true;
}

void lane_into_params(const vrt::PersistentLane &lane, bool with_samples, vrt::TraceParams &p) {
    p.work_counter = lane.work_counter;
    p.pool_paths = lane.pool_paths;
    p.pool_samples = with_samples ? lane.samples : nullptr;
}

static bool pool_tiles_fit(const vrt_ctx *ctx) {
    return (unsigned long long)ctx->shard.tiles_x * ctx->shard.tiles_y * ctx->shard.tiles_x < (1ull << 32);
}

// Bounce frames once the box of the occupied cells is known to be the grid: `exit_fn` — vrt_pool_kernel where the frame fits it and the
// lane has the sample buffer, its DIL-2 twin `exit_path` otherwise (nullptr: keep `keep`); a DIL-2 path kernel as it is.
static vrt::KernelFn grid_exit_choice(vrt_ctx *ctx, vrt::KernelFn exit_fn, vrt::KernelFn exit_path, vrt::KernelFn keep, const vrt_camera_device *camera,
                                      vrt::PersistentLane *lane, hipStream_t lane_stream) {
    const vrt::KernelEntry *e = vrt::kernel_entry_of(exit_fn);
    if (!(e && e->path == 2)) return exit_fn;
    if (camera->max_bounce <= 15 && pool_tiles_fit(ctx) && lane && lane_samples_ready(ctx, *lane, sample_units(ctx, camera->samples_per_pixel), lane_stream)) return exit_fn;
    return exit_path ? exit_path : keep;
}

// Common front part of a frame: argument checks, push constants, derived-structure refresh.  Leaves the kernel to launch in *fn —
// for a counting context in *product_fn too: the product kernel that renders the frame read back.  Runs on the primary stream.
// with_samples: the kernel of the frame (a persistent one) takes samples as its units of work from `lane`'s buffer.
int pre_dispatch(vrt_ctx *ctx, const vrt_camera_device *camera, const vrt_sun_device *sun, vrt::PersistentLane *lane, hipStream_t lane_stream, vrt::KernelFn *fn,
                 vrt::KernelFn *product_fn, bool *with_samples, int trial) {
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
        if (ctx->kernel_grid_exit || ctx->product_grid_exit || ctx->bounce_auto) {
            if (!ctx->h_cell_bounds) VRT_HIP(ctx, ctx->res.pinned(&ctx->h_cell_bounds, 6 * sizeof(int)));
            if (!ctx->ev_bounds) VRT_HIP(ctx, ctx->res.event(&ctx->ev_bounds, hipEventDisableTiming));
            if (ctx->bounds_pending) VRT_HIP(ctx, hipEventSynchronize(ctx->ev_bounds)); // (the copy before this one still owns the buffer)
            VRT_HIP(ctx, hipMemcpyAsync(ctx->h_cell_bounds, ctx->d_cell_bounds, 6 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            VRT_HIP(ctx, hipEventRecord(ctx->ev_bounds, ctx->stream));
            ctx->bounds_pending = true;
            ctx->box_is_grid = false; // until the new box is known
            ctx->auto_next = 0;       // (another scene: the bounce kernel's auto-tune starts again)
            ctx->auto_decided = ctx->auto_use_pool = false;
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
    if (ctx->cell_material_dirty && ctx->d_cell_material) {
        int rcw = begin_scene_write(ctx);
        if (rcw != VRT_OK) return rcw;
        VRT_HIP(ctx, vrt::launch_build_cell_material(ctx->params, ctx->cfg.brick_dimension, ctx->cfg.brick_alloc, ctx->cm_cell_lo, ctx->cm_cell_hi, ctx->cm_slot_lo, ctx->cm_slot_hi,
                                                     ctx->cm_mat_lo, ctx->cm_mat_hi, ctx->stream));
        rcw = end_scene_write(ctx);
        if (rcw != VRT_OK) return rcw;
    }
    ctx->cell_material_dirty = false;
    ctx->cm_cell_lo = ctx->cm_cell_hi = ctx->cm_slot_lo = ctx->cm_slot_hi = ctx->cm_mat_lo = ctx->cm_mat_hi = 0;
    // max_bounce <= 1 ("only primary ray" + its shadow ray): the bounce loop runs at most once
    *fn = (camera->max_bounce <= 1) ? (camera->samples_per_pixel == 1 ? ctx->kernel_single1 : ctx->kernel_single) : ctx->kernel;
    if (ctx->bounds_pending && hipEventQuery(ctx->ev_bounds) == hipSuccess) {
        // the box of the occupied cells {-min, max} per axis: "the grid, or nearly" = at most an eighth of the axis free on either side
        const int *b = ctx->h_cell_bounds;
        const int dim[3] = {(int)ctx->cfg.dim_x, (int)ctx->cfg.dim_y, (int)ctx->cfg.dim_z};
        bool all = b[0] != (int)0x80808080;
        // (VRT_TUNE_GRID_EXIT_ANY_BOX, round 5's experiment: whatever the box — rays that leave it walk on to the grid's face)
        for (int a = 0; a < 3 && all && !(ctx->cfg.tuning_flags & VRT_TUNE_GRID_EXIT_ANY_BOX); a++) all = (-b[a]) * 8 <= dim[a] && (dim[a] - 1 - b[3 + a]) * 8 <= dim[a];
        ctx->box_is_grid = all;
        ctx->bounds_pending = false;
    }
    if (camera->max_bounce > 1 && ctx->box_is_grid && !ctx->d_counters && ctx->kernel_grid_exit)
        *fn = grid_exit_choice(ctx, ctx->kernel_grid_exit, ctx->kernel_grid_exit_path, *fn, camera, lane, lane_stream);
    // the auto-tuned bounce kernel (bounce_auto): the pool kernel on its trial frames and once the trials have said so
    if (camera->max_bounce > 1 && ctx->box_is_grid && ctx->bounce_auto && !ctx->dist && ((trial >= 0 && (trial & 1)) || (trial < 0 && ctx->auto_decided && ctx->auto_use_pool))) {
        const vrt::KernelFn pool = grid_exit_choice(ctx, ctx->bounce_auto, nullptr, nullptr, camera, lane, lane_stream);
        if (pool) *fn = pool;
        else if (trial >= 0) ctx->auto_decided = true, ctx->auto_use_pool = false; // (this context's frames do not fit the pool kernel: no contest)
    }
    // With counters enabled the counting build of the kernel (compiler-generated loops, per-lane counters) runs
    // first and fills the counters; the frame that is read back is then rendered by the product kernel itself,
    // so that every parity check made on a counting context checks the shipped code path.
    *product_fn = nullptr;
    if (ctx->d_counters) {
        *product_fn = ctx->product[(camera->max_bounce <= 1) ? (camera->samples_per_pixel == 1 ? 2 : 1) : 0];
        if (!*product_fn) return fail(ctx, VRT_E_STATE, "no product kernel for this configuration");
        if (camera->max_bounce > 1 && ctx->box_is_grid && ctx->product_grid_exit)
            *product_fn = grid_exit_choice(ctx, ctx->product_grid_exit, ctx->product_grid_exit_path, *product_fn, camera, lane, lane_stream);
    }
    // (vrt_pool_kernel has its sample buffer by now; vrt_path_kernel takes samples as units of work where the buffer can be had, whole
    // pixels otherwise)
    const vrt::KernelEntry *pe = vrt::kernel_entry_of(*product_fn ? *product_fn : *fn);
    *with_samples = pe && pe->path != 0 && lane && lane_samples_ready(ctx, *lane, sample_units(ctx, camera->samples_per_pixel), lane_stream);
    return VRT_OK;
}

} // namespace vrt_impl

extern "C" {

static int do_dispatch(vrt_ctx *ctx, const vrt_camera_device *camera, const vrt_sun_device *sun, uint32_t frames, bool primary_only = false,
                       hipEvent_t *marks = nullptr) {
    if (frames == 0) return ctx ? fail(ctx, VRT_E_INVALID_ARG, "zero frames") : VRT_E_INVALID_ARG;
    if (ctx && ctx->dist) return fail(ctx, VRT_E_STATE, "this context runs the multi-GPU pipeline: use vrt_dist_frame");
    DeviceGuard dg(ctx ? ctx->device : 0);
    // (the stream a frame of the persistent kernels would run on — they never wait for a re-sort of the tile schedule — and with it
    // the lane whose buffers the frame uses)
    // the bounce kernel's auto-tune (vrt_ctx::bounce_auto): read the trials' events once all four have been launched; is this frame a trial?
    int trial = -1;
    if (ctx && ctx->bounce_auto && !ctx->auto_decided) {
        if (ctx->auto_next == 4u) {
            bool done = true;
            for (int k = 0; k < 4 && done; k++) done = hipEventQuery(ctx->auto_ev[k][1]) == hipSuccess;
            if (done) {
                float ms[4] = {0, 0, 0, 0};
                for (int k = 0; k < 4 && done; k++) done = hipEventElapsedTime(&ms[k], ctx->auto_ev[k][0], ctx->auto_ev[k][1]) == hipSuccess;
                ctx->auto_ms[0] = std::min(ms[0], ms[2]), ctx->auto_ms[1] = std::min(ms[1], ms[3]);
                ctx->auto_use_pool = done && ctx->auto_ms[1] < 0.85f * ctx->auto_ms[0];
                ctx->auto_decided = true;
                if (!ctx->auto_use_pool && !vrt::is_path_kernel(ctx->kernel)) {
                    // the lockstep kernel stays: the sample buffers the pool trials made serve no frame of this context
                    for (int l = 0; l < 2; l++) {
                        if (!ctx->lane[l].samples) continue;
                        ctx->res.drop(ctx->lane[l].samples); // (the trials' events have completed: nothing in flight reads it)
                        ctx->lane[l].samples = nullptr;
                        ctx->lane[l].sample_elems = 0;
                    }
                }
            }
        } else if (frames == 1 && !marks && camera && camera->max_bounce > 1 && ctx->box_is_grid && !ctx->status_dirty && !ctx->params.wave_timeline) {
            // (the four trials must trace the same work: a host that changes the samples per pixel or the bounce count between them
            // starts the trials again with the new pair — ADVICE r05)
            if (ctx->auto_next != 0u && (camera->samples_per_pixel != ctx->auto_spp || camera->max_bounce != ctx->auto_bounce)) ctx->auto_next = 0u;
            if (ctx->auto_next == 0u) ctx->auto_spp = camera->samples_per_pixel, ctx->auto_bounce = camera->max_bounce;
            trial = (int)ctx->auto_next;
            primary_only = true; // (a trial runs alone on the primary stream: its time is the kernel's, not the overlap's)
        }
    }
    const bool b_turn = ctx && ctx->stream_b && frames == 1 && !primary_only && (ctx->frame_seq & 1u) && (ctx->params.tile_order != 5u || ctx->sched_period);
    vrt::PersistentLane *lane = ctx ? &ctx->lane[b_turn ? 1 : 0] : nullptr;
    vrt::KernelFn fn = nullptr, product_fn = nullptr;
    bool with_samples = false;
    const int rcp = pre_dispatch(ctx, camera, sun, lane, ctx ? (b_turn ? ctx->stream_b : ctx->stream) : nullptr, &fn, &product_fn, &with_samples, trial);
    if (rcp != VRT_OK) return rcp;
    if (trial >= 0 && ctx->auto_decided) trial = -1; // (the pool kernel cannot take this context's frames: decided without a contest)
    lane_into_params(ctx->lane[0], with_samples && !b_turn, ctx->params);
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
    const bool slot_b = b_turn && !sched_changes;
    if (slot_b) {
        // second frame slot: its own stream and target; ordered after every scene write so far
        if (ctx->post_pending[1]) { // (... and after the present pass that still reads this slot's target, round 6)
            VRT_HIP(ctx, hipStreamWaitEvent(ctx->stream_b, ctx->ev_post_done[1], 0));
            ctx->post_pending[1] = false;
        }
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
        // 0.214 / 0.237 ms per frame against 0.267 / 0.262 / 0.286, tools/experiments/fif_order_ab.py)
        if (ctx->order_auto && sched_mode == 0u) pb.tile_order = 3u;
        pb.target_rgba8 = ctx->target8_b;
        pb.target_rgba32f = ctx->target32f_b;
        lane_into_params(ctx->lane[1], with_samples, pb); // its frames run beside the primary stream's
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
    if (ctx->post_pending[0]) { // (the present pass on its own stream may still be reading the primary target, round 6)
        VRT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_post_done[0], 0));
        ctx->post_pending[0] = false;
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
        // The re-sort below writes the schedule buffer no frame is SUPPOSED to read any more — ev_b_sched covers the second stream's
        // reads of it only from the first stream-b frame behind the previous sort on.  A change of rule may follow a sort at once
        // (samples per pixel 2 -> 1 -> 2 on consecutive frames): then a frame still running on the second stream may read the very
        // buffer this sort rewrites.  Let the primary stream wait for the second stream's last frame.
        if (ctx->stream_b && ctx->b_pending) {
            VRT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_b_done, 0));
            ctx->b_pending = false;
        }
        ctx->sched_mode = sched_mode;
        ctx->params.sched_units = ctx->sched_cap[sched_mode];
        ctx->sched_since = ctx->sched_period;
    }
    if (trial >= 0) {
        for (int e = 0; e < 2; e++)
            if (!ctx->auto_ev[trial][e]) VRT_HIP(ctx, ctx->res.event(&ctx->auto_ev[trial][e]));
        VRT_HIP(ctx, hipEventRecord(ctx->auto_ev[trial][0], ctx->stream));
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
    if (trial >= 0) {
        VRT_HIP(ctx, hipEventRecord(ctx->auto_ev[trial][1], ctx->stream));
        ctx->auto_next = (uint32_t)trial + 1u;
    }
    VRT_HIP(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    ctx->timed_frames = frames;
    ctx->in_flight = true;
    ctx->frame_seq++;
    ctx->last_slot = 0;
    return VRT_OK;
}

int vrt_bounce_autotune_info(vrt_ctx *ctx, double out[4]) {
    if (!ctx || !out) return VRT_E_INVALID_ARG;
    out[0] = !ctx->bounce_auto || ctx->dist ? 0.0 : (!ctx->auto_decided ? 1.0 : (ctx->auto_use_pool ? 3.0 : 2.0));
    out[1] = (double)ctx->auto_next;
    out[2] = (double)ctx->auto_ms[0];
    out[3] = (double)ctx->auto_ms[1];
    return VRT_OK;
}

int vrt_reserve_samples(vrt_ctx *ctx, uint32_t max_samples_per_pixel) {
    if (!ctx) return VRT_E_INVALID_ARG;
    if (max_samples_per_pixel == 0u || max_samples_per_pixel > 65535u) return fail(ctx, VRT_E_INVALID_ARG, "max_samples_per_pixel: 1..65535");
    // (only contexts whose bounce frames a persistent kernel may trace have anything to reserve)
    const vrt::KernelFn fns[6] = {ctx->kernel, ctx->kernel_grid_exit, ctx->kernel_grid_exit_path, ctx->product[0], ctx->product_grid_exit, ctx->bounce_auto};
    bool persistent = false;
    for (vrt::KernelFn fn : fns) persistent = persistent || (fn && vrt::is_path_kernel(fn));
    const uint64_t units = sample_units(ctx, (int)max_samples_per_pixel);
    if (!persistent || ctx->shard.owned_tiles == 0u) return VRT_OK;
    if (units == 0) return fail(ctx, VRT_E_OUT_OF_RANGE, "frames of that many samples are not traced by units of samples (2^32 units, or VRT_TUNE_NO_SAMPLE_UNITS)");
    DeviceGuard dg(ctx->device);
    bool ok = true;
    if (ctx->dist) ok = dist_reserve_samples(ctx, units);
    else
        for (int l = 0; l < (ctx->stream_b ? 2 : 1); l++) ok = lane_samples_ready(ctx, ctx->lane[l], units, l ? ctx->stream_b : ctx->stream) && ok;
    return ok ? VRT_OK : fail(ctx, VRT_E_OOM, "a sample buffer of that size cannot be had (more than half of the free memory): frames keep a kernel that does without");
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
    if (ctx->stream_post) { // (the present passes issued so far)
        VRT_HIP(ctx, wait_stream(ctx->stream_post));
        ctx->post_pending[0] = ctx->post_pending[1] = false;
    }
    return VRT_OK;
}

double vrt_last_kernel_ms(vrt_ctx *ctx) {
    if (!ctx) return -1.0;
    DeviceGuard dg(ctx->device);
    if (finish_frame(ctx) != VRT_OK) return -1.0;
    return ctx->timing_valid ? ctx->last_ms : -1.0;
}

int vrt_region_begin(vrt_ctx *ctx) {
    if (!ctx) return VRT_E_INVALID_ARG;
    DeviceGuard dg(ctx->device);
    for (hipEvent_t &e : ctx->ev_region)
        if (!e) VRT_HIP(ctx, ctx->res.event(&e)); // (owned by the context's container: released with it)
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

int vrt_trace_wave_timeline(vrt_ctx *ctx, const vrt_camera_device *camera, const vrt_sun_device *sun, uint64_t *out, uint64_t capacity_pairs,
                            uint64_t *n_pairs) {
    if (!ctx || !out || !n_pairs) return VRT_E_INVALID_ARG;
    // (the cost-ordered launch has spare workgroups for the halves of split tiles: their waves are listed too; a workgroup that
    // stayed idle leaves zeros).  The dispatch below may change the schedule's rule for this camera (two samples per pixel) and with
    // it the number of spare workgroups it launches — any number up to the list's spare entries: the device buffer holds a row for
    // every wave ANY rule can launch, and the rows reported are counted after the dispatch (sized before it, a rule
    // change made the kernel write past the buffer).
    const uint64_t waves_max = ((uint64_t)ctx->shard.owned_tiles + (ctx->params.tile_order == 5u ? ctx->sched_extra : 0u)) * 4u;
    if (capacity_pairs < waves_max) return fail(ctx, VRT_E_OUT_OF_RANGE, "timeline buffer too small: 4 x (owned tiles + the schedule's spare entries) pairs");
    DeviceGuard dg(ctx->device);
    const size_t bytes = std::max<size_t>(waves_max * 16u, 32u * sizeof(unsigned long long)); // (the profile build of vrt_path_kernel writes 20 words)
#ifndef VRT_DEV_PROFILE
    {
        // the persistent-lane kernel has no wave -> tile map to report: refuse instead of returning zeros
        const vrt::KernelFn would = (camera && camera->max_bounce > 1) ? (ctx->d_counters ? ctx->product[0] : ctx->kernel) : nullptr;
        if (would && vrt::is_path_kernel(would)) return fail(ctx, VRT_E_STATE, "vrt_trace_wave_timeline: frames with bounces run vrt_path_kernel on this context (no per-tile waves)");
        // (... and so does a context whose auto-tune chose vrt_pool_kernel, or whose next frame would be a pool trial)
        if (camera && camera->max_bounce > 1 && ctx->bounce_auto && !ctx->dist && ((ctx->auto_decided && ctx->auto_use_pool) || (!ctx->auto_decided && (ctx->auto_next & 1u))))
            return fail(ctx, VRT_E_STATE, "vrt_trace_wave_timeline: this context's bounce frames run (or are about to try) vrt_pool_kernel (no per-tile waves); VRT_TUNE_NO_BOUNCE_AUTOTUNE keeps the lockstep kernel");
    }
#endif
    unsigned long long *d = nullptr;
    VRT_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&d), bytes));
    int rc = VRT_OK;
    if (hipMemsetAsync(d, 0, bytes, ctx->stream) != hipSuccess) rc = fail(ctx, VRT_E_HIP, "hipMemsetAsync failed");
    ctx->params.wave_timeline = d;
    const uint32_t split_all = ctx->params.split_all;
    ctx->params.split_all = 0u; // (one row of the timeline per wave of a whole tile)
    if (rc == VRT_OK) rc = do_dispatch(ctx, camera, sun, 1, true);
    ctx->params.split_all = split_all;
    ctx->params.wave_timeline = nullptr;
    const uint64_t waves = ((uint64_t)ctx->shard.owned_tiles + (ctx->params.tile_order == 5u ? ctx->params.sched_units : 0u)) * 4u; // (what that launch held)
    if (rc == VRT_OK && (hipMemcpyAsync(out, d, waves * 16u, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess))
        rc = fail(ctx, VRT_E_HIP, "timeline read-back failed");
    (void)hipFree(d);
    if (rc == VRT_OK) *n_pairs = waves;
    return rc;
}

} // extern "C"
