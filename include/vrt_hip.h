/*
 * vrt_hip.h — C ABI of the MI355X-native brickmap ray tracer (libvrt_hip.so).
 *
 * This is the drop-in boundary for ONE path of Avokadoen/zig_vulkan: the
 * vkCmdDispatch of assets/shaders/brick_raytracer.comp
 * (src/modules/voxel_rt/ComputePipeline.zig:550) together with the buffer
 * creation and the seven uploads that feed it.  Every entry point cites the
 * reference interface it replaces.  Plain pointers and sizes only; no C++ or
 * torch types cross this boundary; nothing here throws or aborts.
 *
 * Threading: one vrt_ctx is used from one thread at a time (as the reference
 * uses its ComputePipeline from the main thread, src/main.zig:156-195).
 * Ownership: every pointer argument is borrowed for the duration of the call.
 */
#ifndef VRT_HIP_H
#define VRT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRT_ABI_VERSION 4u /* 4 (round 6): + vrt_dist_init_ex (a communicator per launch slot), vrt_dist_keep_communicators / _release_communicators, vrt_dist_comm_info, vrt_dist_selftest_slots.  3 (round 5): + vrt_dist_frames, vrt_reserve_samples, vrt_bounce_autotune_info; vrt_trace_wave_timeline's capacity rule.  2 (round 4): + vrt_region_begin / _end,
                             vrt_last_denoise_ms, tuning flags 13-17; the product build refuses development kernel_variants */

/* ---- status codes (replace Zig error unions, e.g. StagingRamp.zig:320-325) */
enum {
    VRT_OK = 0,
    VRT_E_INVALID_ARG = -1,
    VRT_E_OOM = -2,
    VRT_E_OUT_OF_RANGE = -3,
    VRT_E_HIP = -4,
    VRT_E_NO_DEVICE = -5,
    VRT_E_STATE = -6,
    VRT_E_RCCL = -7
};

/* ---- buffer ids: same order as shader bindings 1..7
 * (assets/shaders/brick_raytracer.comp:79,105,112,117,124,128,132) and as
 * Pipeline.transfer{GridState,Materials,BrickStatuses,BrickIndices,
 * BrickOccupancy,BrickStartIndex,MaterialIndices} (Pipeline.zig:560-652). */
typedef enum vrt_buffer_id {
    VRT_BUF_GRID_STATE = 0,        /* State.Device, 64 B (State.zig:60-79)        */
    VRT_BUF_MATERIALS = 1,         /* gpu_types.Material[], 20 B each             */
    VRT_BUF_BRICK_STATUS = 2,      /* BrickStatusMask[] u32, 1 bit / grid cell    */
    VRT_BUF_BRICK_INDEX = 3,       /* IndexToBrick[] u32                          */
    VRT_BUF_BRICK_OCCUPANCY = 4,   /* u8[], b^3/8 bytes per brick                 */
    VRT_BUF_BRICK_START_INDEX = 5, /* Brick.StartIndex[] u32 (u31 value + u1 type) */
    VRT_BUF_MATERIAL_INDEX = 6,    /* u8[], b^3 per brick                         */
    VRT_BUF_COUNT = 7
} vrt_buffer_id;

/* ---- data contract structs (byte-for-byte the reference's extern structs) */

/* State.Device (State.zig:60-79) == BrickGridState UBO (comp:79-95). 64 bytes. */
typedef struct vrt_grid_state {
    uint32_t voxel_dim_x, voxel_dim_y, voxel_dim_z;
    uint32_t dim_x, dim_y, dim_z;
    uint32_t padding1, padding2;
    float min_point_base_t[4];
    float max_point_scale[4];
} vrt_grid_state;

/* gpu_types.Material (gpu_types.zig:16-32) == comp:97-104. 20 bytes. */
typedef struct vrt_material {
    uint32_t type; /* 0 lambertian, 1 metal, 2 dielectric */
    float albedo_r, albedo_g, albedo_b;
    float type_data;
} vrt_material;

/* Camera.Device (Camera.zig:183-193), 96 bytes; Zig @Vector(3,f32) is 16-byte
 * sized and aligned, hence the explicit pads.  Push-constant bytes 0..95. */
typedef struct vrt_camera_device {
    uint32_t image_width, image_height;
    uint32_t _pad0[2];
    float horizontal[3];
    float _pad1;
    float vertical[3];
    float _pad2;
    float lower_left_corner[3];
    float _pad3;
    float origin[3];
    float _pad4;
    int32_t samples_per_pixel;
    int32_t max_bounce; /* device value = Config.max_bounce + 1 (Camera.zig:74) */
    uint32_t _pad5[2];
} vrt_camera_device;

/* Sun.Device (Sun.zig:13-18), 32 bytes.  Push-constant bytes 96..127. */
typedef struct vrt_sun_device {
    float position[3];
    uint32_t enabled;
    float color[3];
    float radius;
} vrt_sun_device;

/* ---- creation ------------------------------------------------------------
 * Replaces ComputePipeline.init(allocator, ctx, target_image_info,
 * StateConfigs{uniform_sizes, storage_sizes}, specialization_constants)
 * (ComputePipeline.zig:67-73) as called from Pipeline.init
 * (Pipeline.zig:272-316): image size, buffer sizes (derived here from the grid
 * dimensions exactly as Pipeline.zig:273-283 derives them from the State slice
 * lengths), and the specialization constants (brick_dimension -> brick_bits,
 * brick_bytes, brick_voxel_scale; Pipeline.zig:293-315). */
typedef struct vrt_config {
    uint32_t struct_size;       /* = sizeof(vrt_config)                          */
    uint32_t abi_version;       /* = VRT_ABI_VERSION                             */
    uint32_t width, height;     /* target image (Pipeline.zig:103-126)           */
    uint32_t brick_dimension;   /* 4 (reference, State.zig:5) or 8               */
    uint32_t dim_x, dim_y, dim_z; /* bricks per axis (Grid.zig:36)               */
    uint64_t brick_alloc;       /* 0 => dim_x*dim_y*dim_z (Grid.zig:51)          */
    uint32_t material_capacity; /* 0 => 256 (Pipeline.zig:30)                    */
    int32_t device_id;          /* HIP device; -1 => current device              */
    uint32_t want_float_output; /* also keep an RGBA32F target (parity checks)   */
    uint32_t enable_counters;   /* traversal counters (S,K,V,H,rays): a counting build of the kernel runs once per
                                   call, both targets are then overwritten with 0xCD, and the product kernel renders
                                   the frame that is read back.  1: the counting build walks to the grid's face like
                                   the shader (the reference algorithm's counts).  2: it ends its brick-level walk at
                                   the occupied-cell box like the product kernel (the loads the product issues) */
    /* image-tile sharding across processes (one process per GPU).  The frame is
     * cut into tile_w x tile_h tiles, numbered row-major; this context renders
     * tiles t with t % shard_count == shard_rank into a packed tile-major
     * buffer (see vrt_shard_info).  shard_count <= 1 => whole frame, row-major. */
    uint32_t shard_rank, shard_count;
    uint32_t tile_w, tile_h;    /* 0 => 16 x 16; multiples of 8                  */
    /* optional caller-owned device memory / stream (the reference pipeline does
     * not own its target image either, ComputePipeline.zig:64-66).  0 => owned. */
    void *external_target_rgba8;
    void *external_target_rgba32f;
    void *stream;               /* hipStream_t; 0 => a stream owned by the ctx   */
    uint32_t kernel_variant;    /* 0 => default; see DESIGN.md "kernel variants" */
    /* 1 (or 0): frames execute one after another on the context's stream.  2: vrt_dispatch alternates
     * between two internal streams, each with its own target image, so the tail of one frame overlaps
     * the start of the next (two frames in flight, as a swapchain would have).  Ignored (1) when a
     * caller stream / external target or counters are in use. */
    uint32_t frames_in_flight;
    /* Multi-GPU, 2..8 ranks: rank 0's share of the tiles in percent of an equal share (0 or 100: equal; 1..99: rank 0
     * owns fewer tiles — it also receives every other rank's shards and un-swizzles every frame).  Every rank must pass
     * the same value.  Tile t then belongs to owner[t % period] of a fixed periodic pattern instead of rank t % count;
     * vrt_shard_info.owned_tiles / tiles_per_rank describe the result. */
    uint32_t shard_root_weight;
    /* VRT_TUNE_* bits, 0 = the library's defaults.  Every setting renders the same frame bit for bit: the flags exist for A/B
     * measurements and for the equivalence tests (tests/test_fullsize_gpu.py).  The library reads no environment variable. */
    uint32_t tuning_flags;
    uint32_t _reserved[4];
} vrt_config;

#define VRT_TUNE_NO_SKIP_TO_BOX     (1u << 0) /* walk every cell between the grid's face and the box of the occupied cells */
#define VRT_TUNE_NO_PATH_BRICK_LDS  (1u << 1) /* vrt_path_kernel: walk 8^3 bricks in global memory instead of staging them in LDS */
#define VRT_TUNE_NO_PATH_HALFBLOCKS (1u << 2) /* vrt_path_kernel: the shader's linear status words instead of the half-block words */
#define VRT_TUNE_PATH_EAGER_START   (1u << 3) /* vrt_path_kernel: request brick_start_index together with the staged brick */
#define VRT_TUNE_DIST_NO_BROADCAST  (1u << 4) /* vrt_dist_broadcast as grouped send / recv from the root (every rank alike) */
#define VRT_TUNE_NO_CELL_OCCUPANCY  (1u << 5) /* vrt_path_kernel: reach a brick's bits through brick_index instead of the by-cell copy */
#define VRT_TUNE_NO_START_SHORTCUT  (1u << 6) /* always look brick_start_index up, even when it is slot * B^3 for every brick */
#define VRT_TUNE_PATH_AHEAD          (1u << 7) /* development build only: vrt_path_kernel's walk loop pipelined two trips ahead (measured slower) */
#define VRT_TUNE_PATH_DISTANCE       (1u << 8) /* development build only: vrt_path_kernel's walk loop on the L1 distance field of the occupied cells, a byte per cell (measured slower) */
#define VRT_TUNE_NO_PATH_DILATED     (1u << 9) /* vrt_path_kernel: the half-block walk loop on the linear cell index instead of the dilated one */
#define VRT_TUNE_NO_PATH_GRID_EXIT   (1u << 10) /* vrt_path_kernel, dilated index: keep the steps-left counters in the walk loop (the walk ends at the box of the occupied cells) even when that box is, or nearly is, the grid */
#define VRT_TUNE_PATH_BLOCKS64        (1u << 11) /* development build only: vrt_path_kernel's counter-free dilated-index walk on 4 x 4 x 4-cell words (64 bits) instead of half-block words (measured slower) */
#define VRT_TUNE_PATH_TWO_AHEAD       (1u << 12) /* development build only: vrt_path_kernel's counter-free dilated-index walk with the DDA two cells ahead of the test (two requests in flight per lane; measured +-1 %) */
#define VRT_TUNE_NO_PATH_POOL         (1u << 13) /* frames with bounces on scenes larger than the caches: vrt_path_kernel (a ray per lane) instead of vrt_pool_kernel (a pool of 128 rays per wave; round 4) */
#define VRT_TUNE_NO_SMALL_FRAME_SPLIT (1u << 14) /* frames with fewer waves than twice the SIMDs: one 256-thread workgroup per 16x16 tile as for large frames (instead of two with 32-lane waves) */
#define VRT_TUNE_NO_BOUNCE_WAVE_GROUPS (1u << 15) /* the lockstep bounce kernel as 256-thread workgroups (a tile each) instead of one-wave workgroups */
#define VRT_TUNE_NO_SAMPLE_UNITS     (1u << 16) /* frames with bounces on scenes larger than the caches: a path takes whole pixels from the counter and sums their samples itself (vrt_path_kernel; vrt_pool_kernel is not chosen), instead of single samples whose terms vrt_pool_resolve_kernel adds */
#define VRT_TUNE_NO_DEFERRED_MATERIAL (1u << 17) /* vrt_pool_kernel: look a solid voxel's material up in the brick round (comp:422-427 where the shader has them), not in the round of transitions that shades the hit */
#define VRT_TUNE_NO_CELL_MATERIAL     (1u << 18) /* vrt_pool_kernel: always reach a hit's material through brick_index and material_index (comp:337, :422-425), also where all solid voxels of the brick share one material (round 5: a byte per cell says which) */
#define VRT_TUNE_GRID_EXIT_ANY_BOX    (1u << 19) /* bounce frames of the persistent kernels: the counter-free walk to the grid's face (vrt_pool_kernel, vrt_path_kernel<..., DIL 2>) whatever the box of the occupied cells — by default only where that box is, or nearly is, the grid (a ray that leaves a smaller box walks the empty cells beyond it) */
#define VRT_TUNE_NO_BOUNCE_AUTOTUNE   (1u << 20) /* bounce frames of scenes that stay in the caches: always the lockstep kernel; by default the library times it against vrt_pool_kernel where both apply (four trial frames) and keeps the faster (round 5) */
#define VRT_TUNE_PRESENT_OWN_STREAM   (1u << 21) /* contexts with two frames in flight: vrt_denoise on a stream of its own behind an event of the frame it reads (the reference's arrangement: graphics queue behind the compute queue's semaphore, Pipeline.zig:494-517).  Measured and NOT the default (round 6): on the stream of the frame it reads the pass already overlaps the next frame's trace, which runs on the other stream; the extra stream costs 3-12 % (profiles/r06_present_overlap.txt) */
#define VRT_TUNE_ALL                0x3FFFFFu

typedef struct vrt_ctx vrt_ctx;

int vrt_create(const vrt_config *cfg, vrt_ctx **out);

/* ComputePipeline.deinit (ComputePipeline.zig:385-415): waits, then frees. */
void vrt_destroy(vrt_ctx *ctx);

/* ---- uploads --------------------------------------------------------------
 * Replaces the seven Pipeline.transfer* (Pipeline.zig:560-652).  byte_offset =
 * element offset * element size of the reference call.  The bytes are copied
 * out of `src` before the call returns (the caller keeps ownership, as
 * BrickGrid keeps its slices, Grid.zig:117-126) and the device copy is ordered
 * before the next vrt_dispatch.  Out-of-range => VRT_E_OUT_OF_RANGE (the
 * reference's DestOutOfDeviceMemory, StagingRamp.zig:320-325). */
int vrt_upload(vrt_ctx *ctx, vrt_buffer_id id, uint64_t byte_offset, const void *src, uint64_t nbytes);

/* Size in bytes of device buffer `id` (Pipeline.zig:273-283). */
uint64_t vrt_buffer_size(const vrt_ctx *ctx, vrt_buffer_id id);

/* Device-to-device variant of vrt_upload for sources already in HBM. */
int vrt_upload_device(vrt_ctx *ctx, vrt_buffer_id id, uint64_t byte_offset, const void *dev_src, uint64_t nbytes);

/* ---- dispatch -------------------------------------------------------------
 * Replaces ComputePipeline.dispatch(ctx, workgroup_size, camera, sun)
 * (ComputePipeline.zig:417-463): waits for the previous frame like the fence
 * wait at :423-434, passes the 96+32 bytes the reference pushes as push
 * constants (:488-505) and launches ceil(w/wg) x ceil(h/wg) workgroups
 * (:547-550).  Asynchronous; vrt_wait() plays the role of the returned
 * semaphore/fence. */
int vrt_dispatch(vrt_ctx *ctx, const vrt_camera_device *camera, const vrt_sun_device *sun);
int vrt_wait(vrt_ctx *ctx);

/* Frames with bounces on scenes larger than the caches are traced by persistent kernels whose unit of work is one SAMPLE of a pixel;
 * the samples' terms go through a buffer of 16 bytes per sample — owned pixels x samples_per_pixel, one buffer per stream of frames
 * (two with frames_in_flight = 2, one per launch in flight of the multi-GPU pipeline; 2 GiB for a 4K frame of 16 samples).  By default
 * that buffer is made by the first frame that needs it and grows when a later frame has more samples per pixel: such a vrt_dispatch
 * allocates, and — when it replaces a smaller buffer — waits for the frames in flight on that stream.  vrt_reserve_samples makes the
 * buffers now, for frames of up to max_samples_per_pixel, so that no dispatch does.  VRT_OK also where the context has no such kernel
 * (nothing to reserve); VRT_E_OOM where a buffer cannot be had (more than half of the free memory): the context stays usable, frames
 * keep a kernel that does without, as they do when a growth at dispatch time fails (the buffer it has is kept). */
int vrt_reserve_samples(vrt_ctx *ctx, uint32_t max_samples_per_pixel);

/* Same launch repeated `frames` times back-to-back on the ctx stream without
 * host round trips (benchmarking; every frame is a full render). */
int vrt_dispatch_repeat(vrt_ctx *ctx, const vrt_camera_device *camera, const vrt_sun_device *sun, uint32_t frames);

/* The same, with the hipEvent time of every frame: ms_per_frame[f] = event before frame f's launch(es) to the event
 * after them, f = 0 .. frames-1 (<= 4096), all on the context's primary stream, one frame after another (the periodic
 * re-sort of the tile schedule falls into the frame it precedes).  Blocks until the frames are done.  The measurement
 * SURVEY.md §8(d) asks for: median and p10/p90 of per-frame kernel times. */
int vrt_dispatch_timed(vrt_ctx *ctx, const vrt_camera_device *camera, const vrt_sun_device *sun, uint32_t frames, float *ms_per_frame);

/* ---- results --------------------------------------------------------------
 * Stand in for handing the storage image to the graphics pass
 * (Pipeline.zig:494-517); layout is what Texture.copyToHost (Texture.zig:185-237)
 * yields: rows top to bottom, tightly packed, 4 bytes (or 4 floats) per pixel.
 * For a sharded ctx these return the packed tile-major shard instead. */
int vrt_read_rgba8(vrt_ctx *ctx, void *dst, uint64_t nbytes);
int vrt_read_rgba32f(vrt_ctx *ctx, void *dst, uint64_t nbytes);
/* Re-point the context at other caller-owned target images (device memory, same size as
 * vrt_target_bytes_rgba8 / x4 for the float twin; rgba32f may be NULL).  Takes effect for the next
 * dispatch; lets a caller double-buffer the image a gather or a present pass is still reading. */
int vrt_set_target(vrt_ctx *ctx, void *rgba8, void *rgba32f);
void *vrt_device_target_rgba8(vrt_ctx *ctx);   /* device pointer of the target */
void *vrt_device_target_rgba32f(vrt_ctx *ctx);
uint64_t vrt_target_bytes_rgba8(const vrt_ctx *ctx);

/* Shard geometry.  tiles_x*tiles_y tiles in the frame; this ctx owns
 * owned_tiles of them; every shard buffer is padded to tiles_per_rank tiles of
 * tile_w*tile_h pixels so that a gather has equal counts. */
typedef struct vrt_shard_info {
    uint32_t tiles_x, tiles_y, tile_w, tile_h;
    uint32_t shard_rank, shard_count;
    uint32_t owned_tiles, tiles_per_rank;
} vrt_shard_info;
int vrt_get_shard_info(const vrt_ctx *ctx, vrt_shard_info *out);

/* Root-side un-swizzle after the per-frame gather: `gathered` holds
 * shard_count consecutive packed shards (rank-major) of bytes_per_pixel-sized
 * pixels in device memory; writes the row-major frame to `dst_frame` (device).
 * Runs on the ctx stream. */
int vrt_assemble_frame(vrt_ctx *ctx, const void *gathered, void *dst_frame, uint32_t bytes_per_pixel);

/* ---- multi-GPU frame pipeline (RCCL over xGMI, one process per GPU) -----------------------------
 * The reference is single-GPU; this is the north-star's image-tile sharding.  A context created with
 * shard_rank / shard_count = this process's rank / world size renders its interleaved 16x16 tiles and
 * ONE gather per launch (grouped ncclSend / ncclRecv) brings the packed shards (RGB: the alpha of the target is the
 * constant 255) to rank 0, which un-swizzles them into row-major RGBA8 frames.  Up to 16 launches are in flight, each on
 * its own stream (kernel -> gather -> un-swizzle), so one launch's collective overlaps the next launches' kernels.  (A launch of a
 * rank's 1/N of the tiles lasts as long as its longest wave — about as long as the whole frame's — so a rank's frame RATE is the number of
 * launches the GPU runs side by side: the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues, 4 by default; a host that
 * wants more than 4 launches to overlap sets that variable of the RUNTIME before it initialises HIP — the library itself reads none.
 * Rank 1 of 8 on the headline workload, one frame per launch: 30.9 us per frame with 8 launches on 4 queues, 16.2 with 16 on 24.)
 * RCCL is reached through dlopen(rccl_path) — pass the library the process already uses (PyTorch's
 * bundled librccl.so) so that there is one RCCL in the address space; libvrt_hip.so does not link it.
 * Rank 0 makes the 128-byte id with vrt_dist_unique_id and the host distributes it to every rank.
 * COLLECTIVE CALLS.  On a context with vrt_dist_init done, every call that launches queued frames carries a
 * collective and must be made by EVERY rank at the same point of its frame sequence with the same queue length:
 * vrt_dist_frame (when it fills the queue), vrt_dist_wait, vrt_dist_read_frame — and every scene write
 * (vrt_upload, vrt_upload_device, vrt_upload_grid, vrt_update_grid_delta, vrt_dist_broadcast), because a scene write first launches
 * what is queued.  A rank that uploads at another frame than its peers deadlocks the gather.  If a collective fails
 * (VRT_E_RCCL) the ranks are out of step: the context refuses further vrt_dist_* calls and must be destroyed. */
int vrt_dist_unique_id(const char *rccl_path, void *out_id128);
int vrt_dist_init(vrt_ctx *ctx, const char *rccl_path, const void *id128, int rank, int world, uint32_t frames_in_flight);
/* The same with frames_per_launch (1..8) consecutive frames traced by ONE kernel launch and gathered by ONE collective:
 * a rank owns only 1/world of the tiles — too few waves to fill a GPU, and a launch is never shorter than its
 * longest wave — so single-frame launches leave most of the machine idle.  vrt_dist_frame then
 * queues; a full queue, vrt_dist_wait or a scene upload launches what is queued — events that every rank sees at the same
 * frame, as every launch carries a collective.
 * frames_in_flight counts launches. */
int vrt_dist_init_batched(vrt_ctx *ctx, const char *rccl_path, const void *id128, int rank, int world, uint32_t frames_in_flight,
                          uint32_t frames_per_launch);
/* The general form (round 6).  COMMUNICATORS: RCCL executes the operations of one communicator in the order they were issued, whichever
 * streams they were issued on, so launch slots that share a communicator have their gathers run one behind the other.  Every launch slot
 * therefore issues its gather on a communicator of its own — duplicates of the first (ncclCommInitRank from the id) made by
 * ncclCommSplit; `communicators` = how many (0: one per launch slot, at most 8; 1: the round-5 behaviour, every slot on the first; slot i
 * uses communicator i % communicators; fewer than asked for where the library has no ncclCommSplit or a split fails — the ranks agree on
 * the number by one all-reduce, vrt_dist_comm_info reports it).  All ranks issue their launches in the same order and a gather waits only
 * for operations issued before it, so communicators side by side cannot deadlock.  vrt_dist_init and vrt_dist_init_batched are this call
 * with communicators = 0. */
typedef struct vrt_dist_options {
    uint32_t struct_size;       /* sizeof(vrt_dist_options) */
    uint32_t frames_in_flight;  /* launch slots, 1..16 (0: 4) */
    uint32_t frames_per_launch; /* 1..8 (0: 1) */
    uint32_t communicators;     /* 0..16, see above */
    uint32_t reserved[4];       /* zero */
} vrt_dist_options;
int vrt_dist_init_ex(vrt_ctx *ctx, const char *rccl_path, const void *id128, int rank, int world, const vrt_dist_options *options);
/* Communicators are expensive to make (a collective; of the order of a second for eight GPUs) and a host may make contexts often (one
 * per candidate setting, per workload).  The library keeps them in a per-process pool keyed by (id, rank, world, library): a context
 * takes the free communicators of its id's set, makes the missing ones, and gives them back at vrt_destroy.  By default a set whose
 * last context is destroyed is destroyed with it (as before round 6).  vrt_dist_keep_communicators(1): idle sets stay, and a later
 * vrt_dist_init* with the SAME id (every rank must pass the same id again, and make the same calls in the same order) reuses them
 * without a collective; returns the previous setting.  vrt_dist_release_communicators destroys every set no context holds and returns
 * the number of communicators destroyed (call it on every rank at the same point, before the process ends).  A set on which a
 * collective failed is neither reused nor destroyed. */
int vrt_dist_keep_communicators(int keep);
int vrt_dist_release_communicators(void);
/* out = {communicators this context's launch slots use, how many of them this vrt_dist_init* had to make (the rest came from the pool),
 * 1 if the library has ncclCommSplit, 1 if the ranks agreed on the number by an all-reduce} */
int vrt_dist_comm_info(vrt_ctx *ctx, int32_t out[4]);
int vrt_dist_frame(vrt_ctx *ctx, const vrt_camera_device *camera, const vrt_sun_device *sun);
/* n consecutive vrt_dist_frame calls in one: frame i takes cameras[i] and suns[i * sun_stride] (sun_stride 0: every frame the same
 * sun).  A host that knows its next frames — a scripted fly-through (Benchmark.zig:141-172), a benchmark — submits them without a
 * round trip through its own language per frame; what the call costs per frame is the library's own submission (kernel launch, grouped
 * send / recv, event records).  Stops at the first error.  A collective call like vrt_dist_frame: every rank, same n. */
int vrt_dist_frames(vrt_ctx *ctx, const vrt_camera_device *cameras, const vrt_sun_device *suns, uint32_t n, uint32_t sun_stride);
int vrt_dist_wait(vrt_ctx *ctx);
/* rank 0: the most recently submitted frame, row-major RGBA8 (waits for it).  With frames_per_launch > 1 the queue must
 * be empty (full batch just launched, or after vrt_dist_wait): a launch carries a collective, every rank launches together. */
int vrt_dist_read_frame(vrt_ctx *ctx, void *dst, uint64_t nbytes);
/* Replica update (the reference's incremental edits, VoxelRT.zig:107-172, on a scene replicated per GPU): makes bytes
 * [byte_offset, byte_offset + nbytes) of scene buffer `id` on every rank equal to rank `root`'s — what root holds after its own
 * vrt_upload / vrt_update_grid_delta of that range.  For hosts where only one process edits the grid; hosts that apply the same
 * edits on every rank do not need it.  A collective and a scene write: every rank calls it with the same arguments at the same
 * point of its frame sequence (ncclBroadcast where the library has it, send / recv from the root otherwise).  Range errors
 * are the upload's (VRT_E_OUT_OF_RANGE). */
int vrt_dist_broadcast(vrt_ctx *ctx, vrt_buffer_id id, uint64_t byte_offset, uint64_t nbytes, int root);
/* out = {rank, world size — both as the RCCL communicator reports them (ncclCommUserRank / ncclCommCount) —,
 * frames per launch, launches in flight}: lets a launcher prove how many ranks the gather really spans. */
int vrt_dist_info(vrt_ctx *ctx, int32_t out[4]);
/* Per-launch breakdown of the pipeline, by events on each launch's own stream (measurement; off by default because the extra
 * event records sit between the stages).  vrt_dist_profile(ctx, 1) clears the sums and starts sampling; every launch whose
 * events can be read when its slot is reused, and every slot's last launch at vrt_dist_wait, adds its stage times.
 * vrt_dist_stats: out = {launches sampled, frames in them, kernel ms, collective ms, un-swizzle ms (the three are averages
 * per launch; collective = this rank's part of the grouped send / recv: the receives on rank 0, the send elsewhere, waiting
 * for the peer included; un-swizzle: rank 0 only), tiles this rank owns, bytes of one frame's shard, frames per launch}. */
int vrt_dist_profile(vrt_ctx *ctx, uint32_t enable);
int vrt_dist_stats(vrt_ctx *ctx, double out[8]);
/* ncclSend + ncclRecv of one shard to this rank itself: checks the RCCL binding on a single GPU */
int vrt_dist_selftest(vrt_ctx *ctx);
/* Every launch slot at once against the bound library, on one GPU: `rounds` times, per slot, a kernel that keeps one wave busy for
 * busy_us microseconds on the slot's stream (the frame's trace kernel) followed by the grouped self send + recv of one shard on the
 * slot's communicator and stream (the gather).  Fails if an operation fails, a stream faults or bytes differ; a deadlock shows as a call
 * that does not return (run it under a time limit).  out = {wall ms of the timed rounds (one untimed round precedes them), launches in
 * them, mean ms from a slot's kernel start to its gather's end in the last round, communicators used}: with every slot on one
 * communicator the gathers run in issue order, one behind the other; with one per slot they overlap. */
int vrt_dist_selftest_slots(vrt_ctx *ctx, uint32_t busy_us, uint32_t rounds, double out[4]);

/* ---- measurement ---------------------------------------------------------- */
/* hipEvent time of the most recent vrt_dispatch / average per frame of the
 * most recent vrt_dispatch_repeat, in milliseconds; <0 if none completed. */
double vrt_last_kernel_ms(vrt_ctx *ctx);
/* A region of dispatches by the device's own clock (hipEventElapsedTime around the kernels): _begin records an event on
 * each of the context's streams, _end another pair, waits for them and returns the time from the earlier begin to the later end in
 * milliseconds — what the frames between the two calls took on the GPU, without the host's launch and notification latency. */
int vrt_region_begin(vrt_ctx *ctx);
int vrt_region_end(vrt_ctx *ctx, double *ms);

/* Traversal counters of ONE frame of the last dispatch (enable_counters != 0; the counting build runs once per call):
 * rays = GridHit invocations, S = status-word loads (comp:323-326),
 * K = occupied bricks entered (comp:337), V = voxel steps (comp:415),
 * H = hits (comp:422-427), grid_steps = brick-level DDA iterations. */
typedef struct vrt_counters {
    uint64_t rays, status_loads, bricks_entered, voxel_steps, hits, grid_steps;
} vrt_counters;
int vrt_get_counters(vrt_ctx *ctx, vrt_counters *out);
/* Wave-level execution counts of the last dispatch (enable_counters=1), a tuning aid: out[0] = trips
 * of the brick-level loop, out[1] = executions of the voxel-level walk, out[2] = trips of the
 * voxel-level loop, each counted once per wave however many lanes took part. */
int vrt_get_wave_counters(vrt_ctx *ctx, uint64_t out[3]);

/* Measurement aid: render one frame with per-wave begin/end timestamps (100 MHz wall clock) and copy
 * them to `out` as pairs.  `capacity_pairs` >= 4 x (owned tiles + the cost schedule's spare entries, at most one per owned
 * tile: the second halves of split tiles) — the most waves any launch of this context can hold.  Returns the number of
 * pairs of THIS launch through *n_pairs.  Not part of the reference's interface. */
int vrt_trace_wave_timeline(vrt_ctx *ctx, const vrt_camera_device *camera, const vrt_sun_device *sun, uint64_t *out,
                            uint64_t capacity_pairs, uint64_t *n_pairs);

/* out = {engine clock in kHz, compute units, wavefront size, L2 bytes} of HIP device `device` (-1: current):
 * what bench.py prices an instruction-issue rate against. */
int vrt_device_info(int device, int64_t out[4]);

const char *vrt_last_error(const vrt_ctx *ctx); /* ctx may be NULL: create errors */
uint32_t vrt_abi_version(void);
/* The traversal kernel that rendered the most recent frame (before the first frame: the one a frame without bounces and with
 * one sample per pixel would take), as its template-id — the kernel name rocprofv3 reports minus the `void vrt::` prefix and
 * the argument list: "vrt_trace_kernel<8, false, 7, 7, 2, 256>" (B, COUNT, MODE, MIN_WAVES, SHADE, BLOCK),
 * "vrt_path_kernel<8, 5, false, false, false, false, 1>" (B, MIN_WAVES, FILTER, HALF, AHEAD, DIST, DIL) or
 * "vrt_pool_kernel<8, 6, 60, 2>" (B, MIN_WAVES, SLOTS, STAGES).  Inside the multi-GPU pipeline: the kernel of the frame queued last.  On a counting context: the product kernel, not the counting build that
 * ran before it.  The name is that of the LAST frame and may change between frames of one context: a context whose bounce frames
 * the persistent kernels trace starts on vrt_path_kernel<..., DIL 1> and moves to vrt_pool_kernel (8^3 bricks) or <..., DIL 2> once the
 * host copy of the occupied cells' box has arrived and says that the box is the grid; frames of other sample / bounce counts take
 * other kernels.  The product build (make) holds the kernels the library chooses itself — kernel_variant modes 0, 5 and 9 with the
 * occupancy and order fields — and answers every other kernel_variant with VRT_E_INVALID_ARG; the development build (make dev) holds
 * them all (ABI version 2 recorded that change). */
const char *vrt_kernel_name(const vrt_ctx *ctx);
/* The bounce kernel's auto-tune (round 5).  Bounce frames of a scene that stays in the caches are the lockstep kernel's by the library's
 * size rule; where vrt_pool_kernel can trace them too (three power-of-two grid dimensions, occupied cells reaching the grid's faces, a
 * sample buffer to be had) the library times both — four single-frame vrt_dispatch calls run alone on the primary stream, lockstep /
 * pool / lockstep / pool — and keeps the faster (the pool kernel only if it wins by 15 %): a terrain keeps the lockstep kernel, a sparse
 * field gets the pool kernel (1.6-2.2 x, profiles/r05_pool_generalised_ab.txt).  Same bytes either way; a status upload starts it again;
 * VRT_TUNE_NO_BOUNCE_AUTOTUNE turns it off; contexts of the multi-GPU pipeline, counting contexts and contexts whose kernel_variant names
 * the bounce kernel do not tune.  out = {state: 0 not applicable, 1 trials to come or in flight, 2 decided: lockstep, 3 decided: pool;
 * trial frames launched; lockstep ms; pool ms (the minima of the two trials each, 0 until decided)}. */
int vrt_bounce_autotune_info(vrt_ctx *ctx, double out[4]);
/* number of traversal kernels compiled into this build of the library (tests/test_kernel_resources.py) */
int vrt_compiled_kernel_count(void);

/* =========================================================================
 * Host-side scene objects (CPU only; usable without a GPU).  C view of the
 * C++ mirror of the reference's "side" modules so that other hosts (the Zig
 * app, Python tests) can build the exact byte contract.
 * ========================================================================= */

/* ---- BrickGrid: Grid.zig:36-211 + State.zig + MaterialAllocator.zig ------- */
typedef struct vrt_grid vrt_grid;

typedef struct vrt_grid_config {       /* Grid.zig:13-20 */
    uint64_t brick_alloc;              /* 0 => all bricks                        */
    float base_t;                      /* default 0.01                           */
    float min_point[3];
    float scale;                       /* default 1.0                            */
    uint32_t brick_dimension;          /* 0 => 4 (State.zig:5); 4 or 8           */
} vrt_grid_config;

int vrt_grid_create(uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, const vrt_grid_config *cfg, vrt_grid **out);
void vrt_grid_destroy(vrt_grid *g);
/* BrickGrid.insert (Grid.zig:129-194), including the Y flip and delta tracking.
 * Returns VRT_E_OUT_OF_RANGE where the reference would trip its asserts. */
int vrt_grid_insert(vrt_grid *g, uint64_t x, uint64_t y, uint64_t z, uint8_t material_index);
/* Bulk form: n records of {x,y,z} u32 triples + material byte. */
int vrt_grid_insert_many(vrt_grid *g, const uint32_t *xyz, const uint8_t *materials, uint64_t n);
const vrt_grid_state *vrt_grid_device_state(const vrt_grid *g);
/* Borrowed pointer + byte size of host array `id` (GRID_STATE..MATERIAL_INDEX;
 * MATERIALS is not part of the grid => NULL). */
const void *vrt_grid_data(const vrt_grid *g, vrt_buffer_id id, uint64_t *nbytes);
uint32_t vrt_grid_active_bricks(const vrt_grid *g);
uint32_t vrt_grid_brick_dimension(const vrt_grid *g);
/* DeviceDataDelta (State.zig:14-57): dirty element range [from,to) of array
 * `id`; returns 1 if active, 0 if inactive. */
int vrt_grid_delta(const vrt_grid *g, vrt_buffer_id id, uint64_t *from, uint64_t *to);
void vrt_grid_reset_delta(vrt_grid *g, vrt_buffer_id id);

/* VoxelRT.init's transferGridState (VoxelRT.zig:62) + a full upload of the five
 * arrays (what the first updateGridDelta amounts to after scene build). */
int vrt_upload_grid(vrt_ctx *ctx, vrt_grid *g);
/* VoxelRT.updateGridDelta (VoxelRT.zig:107-172): uploads only the dirty
 * [from,to) of each of the 5 arrays, then resets the deltas. */
int vrt_update_grid_delta(vrt_ctx *ctx, vrt_grid *g);

/* ---- Camera (Camera.zig:36-77,162-180) and Sun (Sun.zig:35-63) ----------- */
typedef struct vrt_camera_config {     /* Camera.zig:5-14 (render-relevant part) */
    float viewport_height;             /* default 2                              */
    float origin[3];
    int32_t samples_per_pixel;         /* default 2                              */
    int32_t max_bounce;                /* default 2 (device gets +1)             */
} vrt_camera_config;
/* Camera.init(vertical_fov, w, h, config): identity orientation, forward (0,0,1). */
int vrt_camera_init(float vertical_fov_deg, uint32_t image_width, uint32_t image_height,
                    const vrt_camera_config *cfg, vrt_camera_device *out);
/* propogatePitchChange for a given unit forward vector (Camera.zig:167-180):
 * recomputes horizontal / vertical / lower_left_corner in place. */
int vrt_camera_set_forward(vrt_camera_device *cam, float vertical_fov_deg, float viewport_height,
                           const float forward[3]);

typedef struct vrt_sun_config {        /* Sun.zig:4-11 (render-relevant part)    */
    uint32_t enabled;                  /* default 1                              */
    float color[3];                    /* default 1, 1.1, 1                      */
    float radius;                      /* default 5                              */
    float sun_distance;                /* default 1000                           */
} vrt_sun_config;
int vrt_sun_init(const vrt_sun_config *cfg, vrt_sun_device *out);

/* The reference's default material table (terrain.zig:130-196), 8 entries. */
uint32_t vrt_default_materials(vrt_material *out, uint32_t capacity);

/* ---- deterministic synthetic scenes (bench/test input) --- */
/* Value-noise terrain shell + water inserted through vrt_grid_insert. */
int vrt_synth_terrain(vrt_grid *g, uint64_t seed);
/* Sparse field of solid spheres; fraction ~p of 32-voxel blocks occupied. */
int vrt_synth_sparse(vrt_grid *g, uint64_t seed, float p);

/* ---- present / denoise pass ------------------------------------------------
 * The step after the path: the reference samples the traced image in a fullscreen-quad fragment pass
 * with the "sirBird" denoiser (assets/shaders/image.frag:18-78; parameters GraphicsPipeline.Config,
 * GraphicsPipeline.zig:34-39: samples 20, bias 0.6, multiplier 1.5, inverse hue tolerance 20) at the
 * window resolution.  vrt_denoise runs that pass as a HIP kernel over the most recent frame of the
 * context into a context-owned out_w x out_h image; unsharded contexts only. */
typedef struct vrt_denoise_config {
    int32_t samples;
    float distribution_bias, pixel_multiplier, inverse_hue_tolerance;
} vrt_denoise_config;
int vrt_denoise(vrt_ctx *ctx, const vrt_denoise_config *cfg /* NULL => reference defaults */, uint32_t out_w, uint32_t out_h,
                uint32_t want_float);
int vrt_read_denoised_rgba8(vrt_ctx *ctx, void *dst, uint64_t nbytes);
int vrt_read_denoised_rgba32f(vrt_ctx *ctx, void *dst, uint64_t nbytes);
void *vrt_device_denoised_rgba8(vrt_ctx *ctx);
/* hipEvent time of the most recent vrt_denoise launch in milliseconds (waits for it); <0 if none was issued.  With
 * vrt_last_kernel_ms this is the app's whole frame: trace (ComputePipeline.zig:417-463) + present (GraphicsPipeline.zig:27-39). */
double vrt_last_denoise_ms(vrt_ctx *ctx);

/* ---- MagicaVoxel .vox input --------------------------------------------
 * Host-side parser with the reference's semantics (src/modules/voxel_rt/vox/loader.zig:41-229,
 * types.zig): MAIN, optional PACK, per model SIZE + XYZI, then an optional RGBA chunk; unknown
 * chunks are skipped 4 bytes at a time; without RGBA the default palette applies.  Unlike the
 * reference (loader.zig:90 "TODO: pos will cause out of bounds easily") every read is bounds
 * checked and a truncated file is VRT_VOX_E_INVALID_FILE_CONTENT. */
enum { /* loader.zig:33-41 ParseError */
    VRT_VOX_E_INVALID_ID = -100,
    VRT_VOX_E_EXPECTED_SIZE_HEADER = -101,
    VRT_VOX_E_EXPECTED_XYZI_HEADER = -102,
    VRT_VOX_E_EXPECTED_RGBA_HEADER = -103,
    VRT_VOX_E_UNEXPECTED_VERSION = -104,
    VRT_VOX_E_INVALID_FILE_CONTENT = -105,
    VRT_VOX_E_MULTIPLE_PACK_CHUNKS = -106
};
typedef struct vrt_vox vrt_vox;
typedef struct vrt_vox_xyzi { uint8_t x, y, z, color_index; } vrt_vox_xyzi; /* types.zig Chunk.XyziElement */
typedef struct vrt_vox_rgba { uint8_t r, g, b, a; } vrt_vox_rgba;           /* types.zig Chunk.RgbaElement */
/* validateHeader (loader.zig:231-245): "VOX ", version byte 150, "MAIN" at offset 8. */
int vrt_vox_validate_header(const void *buffer, uint64_t nbytes);
/* parseBuffer(strict, allocator, buffer) (loader.zig:41). */
int vrt_vox_parse(const void *buffer, uint64_t nbytes, int strict, vrt_vox **out);
void vrt_vox_destroy(vrt_vox *v);
uint32_t vrt_vox_num_models(const vrt_vox *v);
int vrt_vox_model_size(const vrt_vox *v, uint32_t model, int32_t size_xyz[3]);
const vrt_vox_xyzi *vrt_vox_model_voxels(const vrt_vox *v, uint32_t model, uint64_t *count);
const vrt_vox_rgba *vrt_vox_palette(const vrt_vox *v); /* 256 entries */
/* Palette -> materials as the reference app maps them (src/main.zig:93-106): alpha/255 < 0.8 =>
 * dielectric with index 1.52, else lambertian; albedo = rgb/255.  Writes palette entries
 * [0, count) to out[0..count). */
int vrt_vox_materials(const vrt_vox *v, vrt_material *out, uint32_t count);
/* Insert one model into a grid as the reference app does (src/main.zig:109-117): voxel (x,y,z) goes to
 * grid (x + off_x, z + off_y, y + off_z) — .vox is z-up — with material color_index + material_offset. */
int vrt_vox_insert(vrt_grid *g, const vrt_vox *v, uint32_t model, uint32_t off_x, uint32_t off_y, uint32_t off_z,
                   uint32_t material_offset);

/* ---- scripted benchmark fly-through ----------------------------------------
 * Benchmark.init / update / Report of src/modules/voxel_rt/Benchmark.zig:22-136 with its path
 * (Benchmark.zig:141-172): 60 s, 11 positions and 11 orientations, linearly interpolated.  The caller
 * advances it by the frame time it measured (the reference passes the app's delta time) and renders
 * with the updated camera. */
typedef struct vrt_benchmark vrt_benchmark;
int vrt_benchmark_create(vrt_camera_device *cam, float vertical_fov_deg, float viewport_height, vrt_benchmark **out);
void vrt_benchmark_destroy(vrt_benchmark *b);
int vrt_benchmark_update(vrt_benchmark *b, float dt_seconds, vrt_camera_device *cam); /* 1 = path complete */
int vrt_benchmark_report(const vrt_benchmark *b, float *min_ms, float *max_ms, float *avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* VRT_HIP_H */
