// vrt_internal.h — types shared by the C-ABI implementation and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vrt_hip.h"

namespace vrt {

// push constants as one 128-byte block (brick_raytracer.comp:58-75)
struct PushConstants {
    vrt_camera_device cam; // bytes 0..95
    vrt_sun_device sun;    // bytes 96..127
};
static_assert(sizeof(vrt_camera_device) == 96, "Camera.Device is 96 bytes (Camera.zig:183-193)");
static_assert(sizeof(vrt_sun_device) == 32, "Sun.Device is 32 bytes (Sun.zig:13-18)");
static_assert(sizeof(PushConstants) == 128, "push constant range is 128 bytes (ComputePipeline.zig:258-272)");
static_assert(sizeof(vrt_grid_state) == 64, "State.Device is 64 bytes (State.zig:60-79)");
static_assert(sizeof(vrt_material) == 20, "Material is 20 bytes (gpu_types.zig:16-32)");

struct DeviceCounters {
    unsigned long long rays, status_loads, bricks_entered, voxel_steps, hits, grid_steps;
    // wave-level executions (one count per wave per trip, whatever the number of active lanes)
    unsigned long long wave_grid_iters, wave_brick_walks, wave_voxel_iters;
};

// Which rank owns which 16x16 tile.  Default: tile t belongs to rank t % R (its (t / R)-th tile).  With a root weight
// (vrt_config.shard_root_weight) the ownership is a periodic pattern instead: tile t belongs to owner[t % period], rank 0
// holding fewer slots of the period than the others — at 8 GPUs the root also takes in seven ranks' shards and un-swizzles
// every frame, so it should trace less.  A rank's packed shard holds its tiles in increasing tile order either way.
struct TileOwnership {
    uint32_t period;      // 0: the t % R rule
    uint32_t ranks;
    uint8_t owner[64];    // slot j of the period -> rank
    uint8_t prefix[64];   // number of earlier slots of the period held by the same rank
    uint8_t count[8];     // slots per period of each rank
};

// Kernel argument block.  Passed by value: lives in the kernarg segment and is
// read through the scalar cache, like the reference's UBO + push constants.
struct TraceParams {
    vrt_grid_state grid;                 // binding 1
    // push constants of the frames of this launch.  A launch renders blockIdx.y = 0 .. frames-1 frames of the same
    // scene, frame f with pcs[f] into target + f * batch_target_stride: one rank of a multi-GPU run owns only 1/R
    // of the tiles, too few waves to fill the GPU and no shorter than the frame's longest wave, so its frames
    // are traced several to a launch (vrt_dist_*).  A plain dispatch is a batch of one.
    PushConstants pcs[8];
    const vrt_material *materials;       // binding 2
    const uint32_t *brick_status;        // binding 3
    const uint32_t *brick_index;         // binding 4
    const uint8_t *brick_occupancy;      // binding 5
    const uint32_t *brick_start_index;   // binding 6
    const uint8_t *material_index;       // binding 7
    uint8_t *target_rgba8;               // binding 0 (Rgba8 storage image)
    float *target_rgba32f;               // optional float twin of the target
    DeviceCounters *counters;            // optional
    uint32_t width, height;              // imageSize(img_output)
    uint32_t batch_target_stride;        // bytes between the RGBA8 targets of consecutive frames of a launch
    // tile geometry / sharding
    uint32_t tiles_x, tiles_y;           // 16x16-pixel workgroup tiles in the frame
    uint32_t shard_rank, shard_count;    // this ctx renders tiles t % count == rank
    uint32_t owned_tiles;                // number of tiles this context renders
    uint32_t own_period, own_count;      // weighted ownership (TileOwnership): period 0 = the t % shard_count rule
    uint8_t own_slots[16];               // the slots of the period this rank holds, ascending
    uint32_t status_words;               // length of brick_status in u32 words
    // x / grid scale and x / voxel scale as multiplications when both scales are powers of two (set per dispatch from
    // binding 1): the exact reciprocal gives the same correctly rounded quotient in one instruction instead of the
    // ~10 of an IEEE division (six divisions per ray, three per brick entered)
    uint32_t scale_pow2;
    float inv_grid_scale, inv_voxel_scale;
    uint32_t occupancy_words;            // length of brick_occupancy in u32 words
    // derived from binding 3: the bounding box of the occupied grid cells as {-min_x, -min_y, -min_z, max_x, max_y, max_z}
    // (0x80808080 in all six while no cell is occupied).  A ray that has left this box on the far side of an axis cannot meet an
    // occupied cell any more, so the brick-level walk of the product kernels ends there instead of at the grid's face.
    const int *cell_bounds;
    // derived from binding 3: the status bits ordered by 4 x 4 x 2 cells per word (vrt_path_kernel's walk loop; nullptr: not used)
    const uint32_t *status_halfblocks;
    // derived from binding 3: one byte per grid cell, 1 = occupied (kVariantBytes: the walk loop reads the byte of the next cell)
    const uint8_t *status_bytes;
    uint32_t status_cells;               // grid cells = bytes of status_bytes
    // derived from binding 3 (vrt_path_kernel<DIST>; nullptr: not built): one byte per grid cell, its L1 (Manhattan) distance in
    // cells to the nearest occupied cell, 0 = occupied, capped at 255 — a walk of n trips cannot reach an occupied cell from a cell
    // whose byte is > n, so the walk loop asks only where it has to (grid_walk_park_dist_gfx950)
    const uint8_t *cell_distance;
    // derived from bindings 3-5 (vrt_path_kernel; nullptr: not built): the occupancy bits of every OCCUPIED cell's brick, stored by
    // cell (cell * B^3 / 8 bytes) — a brick entry then needs no brick_index look-up before it can ask for the brick's bits
    // (comp:337 -> comp:415 is one dependent miss less; the index is fetched only when a solid voxel was found)
    const uint8_t *cell_occupancy;
    uint32_t cell_occupancy_lockstep; // 1: the lockstep bounce kernel reads it too (round 4: the scene stays in the caches, cells * B^3 < 2^32)
    // derived from binding 6: *start_is_slot == 1 when every allocated brick's start index is slot * B^3 — the pattern the
    // reference's allocator produces (MaterialAllocator.zig:39 hands out B^3 entries per brick in slot order) — so that comp:422's
    // look-up is replaced by a multiplication; nullptr or 0: look it up
    const uint32_t *start_is_slot;
    // derived from binding 0 (round 4): *materials_plain == 1 when no material record has the type MAT_NONE (3).  comp:427 skips a solid
    // voxel whose material's type is the ray's ignore type (and whose type_data is the ray's refraction index); camera rays, shadow rays
    // and rays scattered by anything but a dielectric ignore MAT_NONE, which then no record can match — vrt_pool_kernel decides the test
    // without the record and leaves the hit's look-ups (brick_index -> material_index -> material: three dependent misses on a scene
    // larger than the caches) to the round of transitions that shades it; nullptr or 0: always look the record up in the brick round
    const uint32_t *materials_plain;
    // derived from bindings 3-7 (round 5; vrt_pool_kernel; nullptr: not built): one byte per grid cell — the material id that ALL solid
    // voxels of the cell's brick share, or 0xFF: look it up.  comp:337 -> :422 -> :425 is a chain of three dependent misses on a scene
    // larger than the caches (brick_index 4 B of a 128-byte line, then one byte of material_index — 2 GiB on the 2048^3 scene — per hit:
    // a third of the path trace's fabric traffic); where a brick is of one material — every brick of a sphere, most bricks of a
    // terrain's height band — the hit needs this one byte instead, from an array 128 times smaller than material_index.
    const uint8_t *cell_material;
    // derived, device-built copy of brick_status: one 64-bit word per 4x4x4 block of grid cells,
    // block index bx + nbx*(bz + nbz*by), bit (x&3) + 4*(z&3) + 16*(y&3)  (x, z, y order as comp:318)
    const uint2 *status_blocks;
    uint32_t nbx, nby, nbz;
    // cost-feedback schedule (tile_order 5): tile_cost[4*i + w] = cycles/64 that wave w of the i-th owned tile took in the
    // most recent frame; tile_schedule (stored XCD-major: entry of workgroup k at (k % 8) * ceil(n / 8) + k / 8) is the
    // owned tile the k-th workgroup renders
    uint32_t *tile_cost;
    unsigned long long *wave_timeline;   // optional (measurement): [begin,end] wall-clock ticks per wave, 2 u64 each
    const uint32_t *tile_schedule;
    uint32_t tile_stride;                // tile_order 4: multiplier coprime to owned_tiles
    uint32_t sched_extra;                // tile_order 5: spare schedule entries for the second halves of split tiles (the list's layout)
    uint32_t sched_units;                // ... and how many of them the current order may use (= extra workgroups launched)
    uint32_t packed_tiles;               // 1: write the packed tile-major shard layout even when shard_count == 1
    uint32_t packed_rgb;                 // 1 (with packed_tiles): 3 bytes per pixel in the shard (alpha is the constant 255): a quarter
                                         // less to gather over xGMI; the un-swizzle on rank 0 puts the alpha back
    uint32_t brick_batch;                // lanes that must be waiting before a batched voxel-level walk runs (bounce frames)
    uint32_t path_brick_batch;           // the same for vrt_path_kernel
    uint32_t block_threads;              // 256, or 512 for kVariantLinearLds512
    uint32_t wave_groups;                // 1: launch one 64-thread workgroup per 8x8 block instead of 256 per 16x16 tile
    uint32_t *work_counter;              // vrt_path_kernel: one pixel counter per frame of the launch (zeroed before every launch)
    uint32_t path_lds_bytes;             // vrt_path_kernel<FILTER>: power-of-two LDS allocation holding the block filter (0: grid not eligible)
    uint32_t path_groups;                // vrt_path_kernel: workgroups to launch (a few times what the GPU holds)
    uint32_t path_fin_batch;             // vrt_path_kernel: lanes that must be waiting before the wave leaves the walk loop for them
    uint32_t path_skip_rounds;           // vrt_path_kernel<FILTER>: block look-ups (and jumps over empty blocks) per lane between two calls of the trip loop
    uint32_t path_ready_batch;           // ... unless this many lanes already stand in blocks that hold occupied cells
    uint32_t path_brick_lds;             // vrt_path_kernel, 8^3 bricks: 1 = a lane's brick is staged in LDS for the voxel-level walk (16 KiB per workgroup)
    uint32_t path_eager_start;           // ... and its brick_start_index entry is requested together with the brick (one dependent miss less per hit)
    // vrt_pool_kernel (round 4: a pool of 128 rays per wave, vrt_pool_kernel.h): the paths' records in global memory, 16 dwords per
    // path by field, one block of 128 paths per wave of the launch; and the phase rule's four numbers
    uint32_t *pool_paths;
    uint32_t pool_cus;                   // compute units: min_waves workgroups are launched for each
    float4 *pool_samples;                // vrt_pool_kernel -> vrt_pool_resolve_kernel: the terms of the sample loop's sum, [owned pixel][sample]
    uint32_t pool_walk_k;                // a call of the walk loop returns once this many of its lanes have parked or left
    uint32_t pool_brick_thr;             // a brick round runs once this many of the wave's 128 rays wait for one
    uint32_t pool_trans_thr;             // a round of transitions once this many wait for one
    uint32_t pool_walk_min;              // below this many rays to walk the fuller of the two other queues is served first
    uint32_t pool_tiles_x_magic;         // 2^32 / tiles_x + 1 (0: tiles_x == 1): tile / tiles_x as one multiplication (exact while tiles * tiles_x < 2^32)
    uint32_t wave_groups_bounce;         // 1: the lockstep bounce kernel is launched as one-wave workgroups (launch_trace; round 4)
    uint32_t split_all;                  // vrt_trace_kernel, tile_order 3: log2 of the workgroups per tile (0: one; small frames: 1 or 2)
    uint32_t count_box;                  // counting build only: 1 = walk to the occupied-cell box like the product kernel (issued loads)
    uint32_t skip_to_box;                // 1: rays that enter the grid in front of the occupied-cell box jump to its near face (skip_to_box())
    uint32_t tile_order;                 // workgroup -> tile mapping: 1 row bands per XCD, 2 column bands per XCD, 3 reverse raster, 4 strided,
                                         // 5 cost-feedback schedule, 6 raster.  (kernel_variant: 0 = the library chooses between 3 and the
                                         // schedule re-sorted every 32 frames, which is 7 there; 5 there re-sorts before every frame)
};

constexpr int kMaxBatchFrames = 8;
constexpr int kTileW = 16;
constexpr int kTileH = 16;

// kernel variants (vrt_config.kernel_variant)
enum : uint32_t {
    kVariantDefault = 0,    // best known (see vrt_trace.hip select_trace_kernel)
    kVariantLiteral = 1,    // the shader's memory behaviour: linear status words, byte occupancy loads
    kVariantBlocked = 2,    // blocked 4^3 status words + 64-bit occupancy words, read from global memory
    kVariantBlockedLds = 3, // same, behind an LDS-resident block filter
    kVariantLinearWide = 4, // linear status words (as the shader) + 64-bit occupancy words
    kVariantLinearAlways = 5, // linear status word loaded on every step (no per-lane cache / branch)
    kVariantLinearLds = 6,    // linear status bitmap staged in LDS per workgroup (grids whose bitmap fits)
    kVariantLinearLds512 = 7, // the same with 512-thread workgroups (two tiles share one LDS copy)
    kVariantLinearAhead = 8,  // uncached status word, software-pipelined one cell ahead
    kVariantBytes = 9,        // the hand-written loops on a byte-per-cell copy of the status bits
    kVariantCount
};
// frames with bounces: bit 21 forces the lockstep kernel (vrt_trace_kernel<SHADE 0>), bit 23 vrt_path_kernel (persistent lanes);
// neither: the library chooses by the size of the scene (vrt_create)
constexpr uint32_t kVariantLockstepBounce = 1u << 21;
constexpr uint32_t kVariantForcePath = 1u << 23;
// bit 22: vrt_path_kernel with the block-skipping walk (LDS block filter; grids whose x and z dimensions are powers of two, y a
// multiple of 4; fewer wave-cycles but one wave per SIMD less: measured slower, opt-in)
constexpr uint32_t kVariantPathFilter = 1u << 22;

} // namespace vrt
