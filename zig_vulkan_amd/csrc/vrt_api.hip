// vrt_api.hip — the context behind the C ABI of include/vrt_hip.h: creation (device buffers, streams, events, kernel
// selection), destruction, the seven stream-ordered uploads through a pinned staging ring, read-back, counters.
// The frame itself is vrt_frame.hip, the multi-GPU pipeline vrt_dist.hip, the present pass vrt_post.hip.
//
// Replaces src/modules/voxel_rt/ComputePipeline.zig (init / deinit) and the Pipeline.transfer* family
// (Pipeline.zig:560-652) with its StagingRamp (render/StagingRamp.zig) for this one path.  There is no CPU fallback:
// without a HIP device vrt_create fails with VRT_E_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include "host_brick_grid.hpp"
#include "vrt_ctx.h"

using namespace vrt_impl;

namespace {
thread_local std::string g_create_error;
} // namespace

namespace vrt_impl {

int fail(vrt_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    else g_create_error = msg;
    return code;
}

void free_ctx(vrt_ctx *c) {
    if (!c) return;
    DeviceGuard dg(c->device); // the caller's current device is restored on return
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    dist_destroy(c);      // (its streams are drained and its communicator closed before the memory they use goes)
    c->res.release_all(); // every allocation, event and stream the context made, in reverse order
    delete c;
}

int begin_scene_write(vrt_ctx *c) {
    if (dist_has_pending(c)) { // frames queued before this write must see the scene as it was
        const int rcf = dist_flush(c);
        if (rcf != VRT_OK) return rcf;
    }
    if (c->stream_b && c->b_pending) {
        VRT_HIP(c, hipStreamWaitEvent(c->stream, c->ev_b_done, 0));
        c->b_pending = false;
    }
    return dist_order_primary_after_slots(c);
}
int end_scene_write(vrt_ctx *c) {
    if (c->stream_b || c->dist) {
        VRT_HIP(c, hipEventRecord(c->ev_upload, c->stream));
        c->upload_seq++;
    }
    return VRT_OK;
}

// which derived structures a write to scene buffer `id` invalidates (rebuilt before the next frame, pre_dispatch)
void mark_dirty(vrt_ctx *ctx, vrt_buffer_id id, uint64_t byte_offset, uint64_t nbytes) {
    if (id == VRT_BUF_BRICK_STATUS) ctx->status_dirty = true;
    if (id == VRT_BUF_BRICK_START_INDEX) ctx->start_dirty = true;
    if (id == VRT_BUF_MATERIALS) ctx->materials_dirty = true;
    if (id == VRT_BUF_GRID_STATE || id == VRT_BUF_MATERIALS) return;
    auto widen = [](uint64_t &lo, uint64_t &hi, uint64_t a, uint64_t b) {
        if (lo >= hi) lo = a, hi = b;
        else lo = std::min(lo, a), hi = std::max(hi, b);
    };
    const uint64_t end = byte_offset + nbytes;
    const uint64_t brick_bytes = (uint64_t)ctx->cfg.brick_dimension * ctx->cfg.brick_dimension * ctx->cfg.brick_dimension / 8u;
    // (both by-cell structures follow the cells and the occupancy slots; the material byte also the start indices and the entries of binding 7)
    if (id == VRT_BUF_BRICK_STATUS) {
        widen(ctx->occ_cell_lo, ctx->occ_cell_hi, byte_offset * 8u, end * 8u);
        widen(ctx->cm_cell_lo, ctx->cm_cell_hi, byte_offset * 8u, end * 8u);
    } else if (id == VRT_BUF_BRICK_INDEX) {
        widen(ctx->occ_cell_lo, ctx->occ_cell_hi, byte_offset / 4u, (end + 3u) / 4u);
        widen(ctx->cm_cell_lo, ctx->cm_cell_hi, byte_offset / 4u, (end + 3u) / 4u);
    } else if (id == VRT_BUF_BRICK_OCCUPANCY) {
        widen(ctx->occ_slot_lo, ctx->occ_slot_hi, byte_offset / brick_bytes, (end + brick_bytes - 1u) / brick_bytes);
        widen(ctx->cm_slot_lo, ctx->cm_slot_hi, byte_offset / brick_bytes, (end + brick_bytes - 1u) / brick_bytes);
    } else if (id == VRT_BUF_BRICK_START_INDEX) {
        widen(ctx->cm_slot_lo, ctx->cm_slot_hi, byte_offset / 4u, (end + 3u) / 4u);
    } else {
        widen(ctx->cm_mat_lo, ctx->cm_mat_hi, byte_offset, end);
    }
    if (id == VRT_BUF_BRICK_STATUS || id == VRT_BUF_BRICK_INDEX || id == VRT_BUF_BRICK_OCCUPANCY) ctx->occupancy_dirty = true;
    ctx->cell_material_dirty = true;
}

// the unit counters and (contexts that select vrt_pool_kernel) the path records of one stream of persistent-kernel frames
int lane_init(vrt_ctx *c, vrt::PersistentLane &lane) {
    if (lane.work_counter) return VRT_OK;
    VRT_HIP(c, c->res.device(&lane.work_counter, vrt::kMaxBatchFrames * sizeof(uint32_t)));
    VRT_HIP(c, hipMemsetAsync(lane.work_counter, 0, vrt::kMaxBatchFrames * sizeof(uint32_t), c->stream));
    // (never read before it is written: a path's record is filled by the transition that gives the path its first pixel)
    if (c->pool_stream_dwords) VRT_HIP(c, c->res.device(&lane.pool_paths, c->pool_stream_dwords * sizeof(uint32_t)));
    return VRT_OK;
}
} // namespace vrt_impl

namespace {
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
        VRT_CREATE_HIP(c->res.stream(&c->stream));
    }
    VRT_CREATE_HIP(c->res.event(&c->ev_start));
    VRT_CREATE_HIP(c->res.event(&c->ev_stop));
    VRT_CREATE_HIP(c->res.event(&c->ev_post_start));
    VRT_CREATE_HIP(c->res.event(&c->ev_post_stop));

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
        VRT_CREATE_HIP(c->res.device(&c->dbuf[i], alloc));
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
        VRT_CREATE_HIP(c->res.device(&c->target8, c->target_pixels * 4u));
        VRT_CREATE_HIP(hipMemsetAsync(c->target8, 0, c->target_pixels * 4u, c->stream));
    }
    if (cfg->external_target_rgba32f) {
        c->target32f = static_cast<float *>(cfg->external_target_rgba32f);
    } else if (cfg->want_float_output) {
        VRT_CREATE_HIP(c->res.device(&c->target32f, c->target_pixels * 16u));
        VRT_CREATE_HIP(hipMemsetAsync(c->target32f, 0, c->target_pixels * 16u, c->stream));
    }
    if (cfg->frames_in_flight == 2 && !cfg->stream && !cfg->external_target_rgba8 && !cfg->external_target_rgba32f && !cfg->enable_counters) {
        c->frames_in_flight = 2;
        VRT_CREATE_HIP(c->res.stream(&c->stream_b));
        VRT_CREATE_HIP(c->res.event(&c->ev_b_done, hipEventDisableTiming));
        VRT_CREATE_HIP(c->res.event(&c->ev_upload, hipEventDisableTiming));
        VRT_CREATE_HIP(c->res.device(&c->target8_b, c->target_pixels * 4u));
        VRT_CREATE_HIP(hipMemsetAsync(c->target8_b, 0, c->target_pixels * 4u, c->stream));
        if (c->target32f) {
            VRT_CREATE_HIP(c->res.device(&c->target32f_b, c->target_pixels * 16u));
            VRT_CREATE_HIP(hipMemsetAsync(c->target32f_b, 0, c->target_pixels * 16u, c->stream));
        }
    } else if (cfg->frames_in_flight > 2) {
        free_ctx(c);
        return fail(nullptr, VRT_E_INVALID_ARG, "frames_in_flight must be 0, 1 or 2");
    }
    if (cfg->enable_counters) {
        VRT_CREATE_HIP(c->res.device(&c->d_counters, sizeof(vrt::DeviceCounters)));
        VRT_CREATE_HIP(hipMemsetAsync(c->d_counters, 0, sizeof(vrt::DeviceCounters), c->stream));
    }
    const uint32_t nbx = (cfg->dim_x + 3u) / 4u, nby = (cfg->dim_y + 3u) / 4u, nbz = (cfg->dim_z + 3u) / 4u;
    VRT_CREATE_HIP(c->res.device(&c->d_cell_bounds, 6 * sizeof(int)));
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
        VRT_CREATE_HIP(c->res.device(&c->d_tile_cost, n * 32u)); // [half][tile][wave]
        const uint32_t ns = 8u * ((n + c->sched_extra + 7u) / 8u); // an order buffer is stored XCD-major: 8 rows of ceil((n + extra) / 8)
        c->sched_stride = ns;
        VRT_CREATE_HIP(c->res.device(&c->d_tile_schedule, (2u * (size_t)ns + 2u * (size_t)n) * 4u)); // order A, order B, snapshot + split state
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
        VRT_CREATE_HIP(c->res.pinned(&c->staging[i], kStagingSlotBytes));
        VRT_CREATE_HIP(c->res.event(&c->staging_ev[i], hipEventDisableTiming));
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
    bool want_halfblocks = false, want_distance = false, want_dilated = false, auto_candidate = false;
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
            // (the lockstep kernel by the size rule, not by the caller's word: a candidate for the auto-tune below where the pool kernel's
            // walk applies — three power-of-two dimensions)
            auto_candidate = !(cfg->kernel_variant & vrt::kVariantLockstepBounce) && mwv == 0u && pow2(cfg->dim_x) && pow2(cfg->dim_y) && pow2(cfg->dim_z) &&
                             !cfg->enable_counters && !block_skip;
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
        // (8^3 bricks: staged in LDS, so not with VRT_TUNE_NO_PATH_BRICK_LDS; round 5: 4^3 bricks too, read from global memory as they are walked)
        const bool pool_ok = kind == 2 && !(cfg->tuning_flags & VRT_TUNE_NO_PATH_POOL) &&
                             (cfg->brick_dimension == 4u || !(cfg->tuning_flags & VRT_TUNE_NO_PATH_BRICK_LDS));
        const vrt::KernelEntry *pool = pool_ok ? vrt::find_pool_kernel((int)cfg->brick_dimension, pool_mw, pool_slots, pool_stages) : nullptr;
        // (the DIL-2 twin stays beside a pool kernel: frames the pool kernel cannot take — more than 15 bounces, no sample buffer — keep it
        // instead of falling back to the DIL-1 kernel with its steps-left counters)
        if (pool && c->kernel_grid_exit) c->kernel_grid_exit_path = c->kernel_grid_exit, c->kernel_grid_exit = pool->fn;
        if (pool && c->product_grid_exit) c->product_grid_exit_path = c->product_grid_exit, c->product_grid_exit = pool->fn;
    }
    if (auto_candidate && !(cfg->tuning_flags & (VRT_TUNE_NO_BOUNCE_AUTOTUNE | VRT_TUNE_NO_PATH_POOL | VRT_TUNE_NO_PATH_GRID_EXIT | VRT_TUNE_NO_PATH_DILATED |
                                                  VRT_TUNE_NO_PATH_HALFBLOCKS | VRT_TUNE_NO_SAMPLE_UNITS)) &&
        (cfg->brick_dimension == 4u || !(cfg->tuning_flags & VRT_TUNE_NO_PATH_BRICK_LDS))) {
        if (const vrt::KernelEntry *pool = vrt::find_pool_kernel((int)cfg->brick_dimension)) c->bounce_auto = pool->fn;
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
            const vrt::KernelFn fns[12] = {c->kernel, c->kernel_lockstep, c->kernel_single, c->kernel_single1, c->product[0], c->product[1], c->product[2],
                                           c->kernel_grid_exit, c->product_grid_exit, c->kernel_grid_exit_path, c->product_grid_exit_path, c->bounce_auto};
            for (vrt::KernelFn fn : fns) {
                const vrt::KernelEntry *e = fn ? vrt::kernel_entry_of(fn) : nullptr;
                if (e && pred(*e)) return true;
            }
            return false;
        };
        if (any_kernel([](const vrt::KernelEntry &e) { return !e.path && e.mode == vrt::kStatusBytes; })) {
            const size_t status_bytes_size = (size_t)((cells + 31u) / 32u) * 32u + 64u; // 32 bytes per status word
            VRT_CREATE_HIP(c->res.device(&c->d_status_bytes, status_bytes_size));
            VRT_CREATE_HIP(hipMemsetAsync(c->d_status_bytes, 0, status_bytes_size, c->stream));
        }
        if (any_kernel([](const vrt::KernelEntry &e) { return e.path && (e.half || e.dil); })) {
            const size_t bytes_hb = (size_t)(cells / 32u) * 4u + 64u;
            VRT_CREATE_HIP(c->res.device(&c->d_status_halfblocks, bytes_hb));
            VRT_CREATE_HIP(hipMemsetAsync(c->d_status_halfblocks, 0, bytes_hb, c->stream));
        }
        if (any_kernel([](const vrt::KernelEntry &e) { return e.path && e.dist; })) {
            VRT_CREATE_HIP(c->res.device(&c->d_cell_distance, (size_t)cells + 64u));
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
            if (c->res.device(&c->d_cell_occupancy, by_cell_bytes + 64u) != hipSuccess) {
                (void)hipGetLastError();
                c->d_cell_occupancy = nullptr;
            } else {
                VRT_CREATE_HIP(hipMemsetAsync(c->d_cell_occupancy, 0, by_cell_bytes + 64u, c->stream));
                c->cell_occupancy_lockstep = lockstep_bounce;
            }
        }
        // (round 5) the material a cell's brick is made of, for the hits vrt_pool_kernel shades in its rounds of transitions
        if (!(cfg->tuning_flags & (VRT_TUNE_NO_CELL_MATERIAL | VRT_TUNE_NO_DEFERRED_MATERIAL)) && any_kernel([](const vrt::KernelEntry &e) { return e.path == 2; })) {
            VRT_CREATE_HIP(c->res.device(&c->d_cell_material, (size_t)cells + 64u));
            VRT_CREATE_HIP(hipMemsetAsync(c->d_cell_material, 0xFF, (size_t)cells + 64u, c->stream));
        }
        if (!(cfg->tuning_flags & VRT_TUNE_NO_DEFERRED_MATERIAL)) {
            VRT_CREATE_HIP(c->res.device(&c->d_materials_plain, 64u));
            VRT_CREATE_HIP(hipMemsetAsync(c->d_materials_plain, 0, 64u, c->stream));
        }
        if (!(cfg->tuning_flags & VRT_TUNE_NO_START_SHORTCUT)) {
            VRT_CREATE_HIP(c->res.device(&c->d_start_is_slot, 64u));
            VRT_CREATE_HIP(hipMemsetAsync(c->d_start_is_slot, 0, 64u, c->stream));
        }
        if (any_kernel([](const vrt::KernelEntry &e) { return (e.path && (e.filter || e.dil == 3)) || (!e.path && (e.mode == vrt::kStatusBlocked || e.mode == vrt::kStatusBlockedLds)); })) {
            const size_t nblocks = (size_t)nbx * nby * nbz;
            const size_t status_blocks_bytes = nblocks * 8u + ((nblocks + 31u) / 32u) * 4u + 16u;
            VRT_CREATE_HIP(c->res.device(&c->d_status_blocks, status_blocks_bytes));
            VRT_CREATE_HIP(hipMemsetAsync(c->d_status_blocks, 0, status_blocks_bytes, c->stream));
        }
    }
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || cus <= 0) cus = 256;
    {
        const vrt::KernelEntry *e1 = c->kernel_grid_exit ? vrt::kernel_entry_of(c->kernel_grid_exit) : nullptr;
        const vrt::KernelEntry *e2 = c->product_grid_exit ? vrt::kernel_entry_of(c->product_grid_exit) : nullptr;
        // (vrt_pool_kernel's path records, per stream of frames: room for eight workgroups per CU, any occupancy)
        if ((e1 && e1->path == 2) || (e2 && e2->path == 2) || c->bounce_auto) c->pool_stream_dwords = (size_t)(8 * cus) * 4u * vrt::kPoolPaths * vrt::kPoolPathDwords;
    }
    for (int l = 0; l < (c->stream_b ? 2 : 1); l++) {
        const int rcl = lane_init(c, c->lane[l]);
        if (rcl != VRT_OK) {
            const std::string why = c->err;
            free_ctx(c);
            return fail(nullptr, rcl, why);
        }
    }

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
    p.work_counter = c->lane[0].work_counter;
    p.path_lds_bytes = c->path_lds_bytes;
    {
        p.pool_paths = c->lane[0].pool_paths;
        p.pool_cus = (uint32_t)cus;
        p.pool_walk_k = 14u;    // (tools/experiments/pool_sweep.py; round 5, 124 rays per wave: a plateau — walk_k 12 to 16, thresholds 52 to 56, walk_min 36 to 40)
        p.pool_brick_thr = 52u;
        p.pool_trans_thr = 52u;
        p.pool_walk_min = 36u;
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
    p.cell_material = c->d_cell_material;
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
        VRT_CREATE_HIP(c->res.event(&c->ev_sched, hipEventDisableTiming));
        VRT_CREATE_HIP(c->res.event(&c->ev_b_sched, hipEventDisableTiming));
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
        // fixed part of a wave's chain — the status bits in LDS change nothing, tools/experiments/small_frame_ab.py)
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
    // (frames in flight still write the images the context made for itself; a caller's earlier images are left alone)
    VRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->res.drop(ctx->target8);
    ctx->res.drop(ctx->target32f);
    ctx->target8 = static_cast<uint8_t *>(rgba8);
    ctx->target32f = static_cast<float *>(rgba32f);
    ctx->params.target_rgba8 = ctx->target8;
    ctx->params.target_rgba32f = ctx->target32f;
    return VRT_OK;
}
void *vrt_device_target_rgba8(vrt_ctx *ctx) { return ctx ? (ctx->last_slot == 1 ? ctx->target8_b : ctx->target8) : nullptr; }
void *vrt_device_target_rgba32f(vrt_ctx *ctx) { return ctx ? (ctx->last_slot == 1 ? ctx->target32f_b : ctx->target32f) : nullptr; }
uint64_t vrt_target_bytes_rgba8(const vrt_ctx *ctx) { return ctx ? ctx->target_pixels * 4u : 0; }

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