//! Zig binding of libvrt_hip.so (include/vrt_hip.h) — what a maintainer of Avokadoen/zig_vulkan adds to route the
//! brick ray tracer's compute dispatch to the MI355X HIP kernels instead of vkCmdDispatch.
//!
//! zig is not available in the build image of this repository, so this file is not compiled here.  It is kept in step
//! with the header mechanically instead: the `extern fn` block is generated from include/vrt_hip.h by
//! tools/gen_zig_binding.py, and tests/test_abi.py checks (1) that the block is up to date, (2) that `Status` lists
//! every status code the header defines, (3) that the field lists of the extern structs equal the header's.
//!
//! Layout notes: GridState, Material, CameraDevice and SunDevice ARE the reference's own device structs
//! (State.Device, gpu_types.Material, Camera.Device, Sun.Device): `&camera.d_camera` etc. can be passed after a
//! @ptrCast, nothing is converted.

const std = @import("std");

pub const Ctx = opaque {};
pub const Grid = opaque {};
pub const Vox = opaque {};
pub const Benchmark = opaque {};

/// Every status code of include/vrt_hip.h.  NON-exhaustive (`_`): a code added by a later library version arrives as an
/// unnamed value and takes the `else` branch of check() instead of being illegal behaviour.
pub const Status = enum(c_int) {
    ok = 0,
    invalid_arg = -1,
    oom = -2,
    out_of_range = -3,
    hip = -4,
    no_device = -5,
    state = -6,
    rccl = -7,
    // vox/loader.zig:33-41 ParseError
    vox_invalid_id = -100,
    vox_expected_size_header = -101,
    vox_expected_xyzi_header = -102,
    vox_expected_rgba_header = -103,
    vox_unexpected_version = -104,
    vox_invalid_file_content = -105,
    vox_multiple_pack_chunks = -106,
    _,
};

/// The reference's error names where it has them (StagingRamp.zig:320-325, vox/loader.zig:33-41).
pub const Error = error{
    InvalidArgument,
    OutOfDeviceMemory,
    DestOutOfDeviceMemory,
    HipFailure,
    NoHipDevice,
    InvalidState,
    RcclFailure,
    InvalidId,
    ExpectedSizeHeader,
    ExpectedXyziHeader,
    ExpectedRgbaHeader,
    UnexpectedVersion,
    InvalidFileContent,
    MultiplePackChunks,
    VrtFailure,
};

pub fn check(rc: c_int) Error!void {
    return switch (@as(Status, @enumFromInt(rc))) {
        .ok => {},
        .invalid_arg => error.InvalidArgument,
        .oom => error.OutOfDeviceMemory,
        .out_of_range => error.DestOutOfDeviceMemory, // StagingRamp.zig:320-325
        .hip => error.HipFailure,
        .no_device => error.NoHipDevice,
        .state => error.InvalidState,
        .rccl => error.RcclFailure,
        .vox_invalid_id => error.InvalidId,
        .vox_expected_size_header => error.ExpectedSizeHeader,
        .vox_expected_xyzi_header => error.ExpectedXyziHeader,
        .vox_expected_rgba_header => error.ExpectedRgbaHeader,
        .vox_unexpected_version => error.UnexpectedVersion,
        .vox_invalid_file_content => error.InvalidFileContent,
        .vox_multiple_pack_chunks => error.MultiplePackChunks,
        else => error.VrtFailure,
    };
}

/// same order as shader bindings 1..7 (brick_raytracer.comp:79..132)
pub const BufferId = enum(c_int) {
    grid_state = 0,
    materials = 1,
    brick_status = 2,
    brick_index = 3,
    brick_occupancy = 4,
    brick_start_index = 5,
    material_index = 6,
};

// ---- data contract structs: field for field include/vrt_hip.h (checked by tests/test_abi.py) ----
pub const GridState = extern struct { // State.Device, State.zig:60-79
    voxel_dim_x: u32,
    voxel_dim_y: u32,
    voxel_dim_z: u32,
    dim_x: u32,
    dim_y: u32,
    dim_z: u32,
    padding1: u32 = 0,
    padding2: u32 = 0,
    min_point_base_t: [4]f32,
    max_point_scale: [4]f32,
};

pub const Material = extern struct { // gpu_types.Material, gpu_types.zig:16-32
    type: u32,
    albedo_r: f32,
    albedo_g: f32,
    albedo_b: f32,
    type_data: f32,
};

pub const CameraDevice = extern struct { // Camera.Device, Camera.zig:183-193 (96 bytes)
    image_width: u32,
    image_height: u32,
    _pad0: [2]u32 = .{ 0, 0 },
    horizontal: [3]f32,
    _pad1: f32 = 0,
    vertical: [3]f32,
    _pad2: f32 = 0,
    lower_left_corner: [3]f32,
    _pad3: f32 = 0,
    origin: [3]f32,
    _pad4: f32 = 0,
    samples_per_pixel: i32,
    max_bounce: i32,
    _pad5: [2]u32 = .{ 0, 0 },
};

pub const SunDevice = extern struct { // Sun.Device, Sun.zig:13-18 (32 bytes)
    position: [3]f32,
    enabled: u32,
    color: [3]f32,
    radius: f32,
};

pub const Config = extern struct {
    struct_size: u32 = @sizeOf(Config),
    abi_version: u32 = 4, // VRT_ABI_VERSION
    width: u32,
    height: u32,
    brick_dimension: u32 = 4, // State.brick_dimension
    dim_x: u32,
    dim_y: u32,
    dim_z: u32,
    brick_alloc: u64 = 0,
    material_capacity: u32 = 256, // Pipeline.Config.material_buffer
    device_id: i32 = -1,
    want_float_output: u32 = 0,
    enable_counters: u32 = 0,
    shard_rank: u32 = 0,
    shard_count: u32 = 1,
    tile_w: u32 = 0,
    tile_h: u32 = 0,
    external_target_rgba8: ?*anyopaque = null,
    external_target_rgba32f: ?*anyopaque = null,
    stream: ?*anyopaque = null,
    kernel_variant: u32 = 0,
    frames_in_flight: u32 = 1,
    shard_root_weight: u32 = 0,
    /// TUNE_* bits, 0 = the library's defaults; every setting renders the same frame (A/B measurements, equivalence tests)
    tuning_flags: u32 = 0,
    _reserved: [4]u32 = [_]u32{0} ** 4,
};

pub const DistOptions = extern struct { // vrt_dist_options (vrt_dist_init_ex)
    struct_size: u32 = @sizeOf(DistOptions),
    frames_in_flight: u32 = 0, // launch slots, 1..16 (0: 4)
    frames_per_launch: u32 = 0, // 1..8 (0: 1)
    communicators: u32 = 0, // 0: one per launch slot, at most 8; 1: every slot on the first
    reserved: [4]u32 = .{ 0, 0, 0, 0 },
};

pub const ShardInfo = extern struct {
    tiles_x: u32,
    tiles_y: u32,
    tile_w: u32,
    tile_h: u32,
    shard_rank: u32,
    shard_count: u32,
    owned_tiles: u32,
    tiles_per_rank: u32,
};

pub const Counters = extern struct {
    rays: u64,
    status_loads: u64,
    bricks_entered: u64,
    voxel_steps: u64,
    hits: u64,
    grid_steps: u64,
};

pub const TUNE_NO_SKIP_TO_BOX: u32 = 1 << 0;
pub const TUNE_NO_PATH_BRICK_LDS: u32 = 1 << 1;
pub const TUNE_NO_PATH_HALFBLOCKS: u32 = 1 << 2;
pub const TUNE_PATH_EAGER_START: u32 = 1 << 3;
pub const TUNE_DIST_NO_BROADCAST: u32 = 1 << 4;
pub const TUNE_NO_CELL_OCCUPANCY: u32 = 1 << 5;
pub const TUNE_NO_START_SHORTCUT: u32 = 1 << 6;
pub const TUNE_PATH_AHEAD: u32 = 1 << 7;
pub const TUNE_PATH_DISTANCE: u32 = 1 << 8;
pub const TUNE_NO_PATH_DILATED: u32 = 1 << 9;
pub const TUNE_NO_PATH_GRID_EXIT: u32 = 1 << 10;
pub const TUNE_PATH_BLOCKS64: u32 = 1 << 11;
pub const TUNE_PATH_TWO_AHEAD: u32 = 1 << 12;
pub const TUNE_NO_PATH_POOL: u32 = 1 << 13;
pub const TUNE_NO_SMALL_FRAME_SPLIT: u32 = 1 << 14;
pub const TUNE_NO_BOUNCE_WAVE_GROUPS: u32 = 1 << 15;
pub const TUNE_NO_SAMPLE_UNITS: u32 = 1 << 16;
pub const TUNE_NO_DEFERRED_MATERIAL: u32 = 1 << 17;
pub const TUNE_NO_CELL_MATERIAL: u32 = 1 << 18;
pub const TUNE_GRID_EXIT_ANY_BOX: u32 = 1 << 19;
pub const TUNE_NO_BOUNCE_AUTOTUNE: u32 = 1 << 20;
pub const TUNE_PRESENT_OWN_STREAM: u32 = 1 << 21;

pub const GridConfig = extern struct { // Grid.zig:13-20
    brick_alloc: u64 = 0,
    base_t: f32 = 0.01,
    min_point: [3]f32 = .{ 0, 0, 0 },
    scale: f32 = 1.0,
    brick_dimension: u32 = 4,
};

pub const CameraConfig = extern struct { // Camera.zig:5-14
    viewport_height: f32 = 2,
    origin: [3]f32 = .{ 0, 0, 0 },
    samples_per_pixel: i32 = 2,
    max_bounce: i32 = 2,
};

pub const SunConfig = extern struct { // Sun.zig:4-11
    enabled: u32 = 1,
    color: [3]f32 = .{ 1, 1.1, 1 },
    radius: f32 = 5,
    sun_distance: f32 = 1000,
};

pub const DenoiseConfig = extern struct { // GraphicsPipeline.Config, GraphicsPipeline.zig:34-39
    samples: i32 = 20,
    distribution_bias: f32 = 0.6,
    pixel_multiplier: f32 = 1.5,
    inverse_hue_tolerance: f32 = 20,
};

pub const VoxXyzi = extern struct { x: u8, y: u8, z: u8, color_index: u8 };
pub const VoxRgba = extern struct { r: u8, g: u8, b: u8, a: u8 };

// BEGIN GENERATED extern declarations (tools/gen_zig_binding.py from include/vrt_hip.h) — do not edit by hand
pub extern fn vrt_create(cfg: [*c]const Config, out: *?*Ctx) c_int;
pub extern fn vrt_destroy(ctx: ?*Ctx) void;
pub extern fn vrt_upload(ctx: ?*Ctx, id: BufferId, byte_offset: u64, src: ?*const anyopaque, nbytes: u64) c_int;
pub extern fn vrt_buffer_size(ctx: ?*const Ctx, id: BufferId) u64;
pub extern fn vrt_upload_device(ctx: ?*Ctx, id: BufferId, byte_offset: u64, dev_src: ?*const anyopaque, nbytes: u64) c_int;
pub extern fn vrt_dispatch(ctx: ?*Ctx, camera: [*c]const CameraDevice, sun: [*c]const SunDevice) c_int;
pub extern fn vrt_wait(ctx: ?*Ctx) c_int;
pub extern fn vrt_reserve_samples(ctx: ?*Ctx, max_samples_per_pixel: u32) c_int;
pub extern fn vrt_dispatch_repeat(ctx: ?*Ctx, camera: [*c]const CameraDevice, sun: [*c]const SunDevice, frames: u32) c_int;
pub extern fn vrt_dispatch_timed(ctx: ?*Ctx, camera: [*c]const CameraDevice, sun: [*c]const SunDevice, frames: u32, ms_per_frame: [*c]f32) c_int;
pub extern fn vrt_read_rgba8(ctx: ?*Ctx, dst: ?*anyopaque, nbytes: u64) c_int;
pub extern fn vrt_read_rgba32f(ctx: ?*Ctx, dst: ?*anyopaque, nbytes: u64) c_int;
pub extern fn vrt_set_target(ctx: ?*Ctx, rgba8: ?*anyopaque, rgba32f: ?*anyopaque) c_int;
pub extern fn vrt_device_target_rgba8(ctx: ?*Ctx) ?*anyopaque;
pub extern fn vrt_device_target_rgba32f(ctx: ?*Ctx) ?*anyopaque;
pub extern fn vrt_target_bytes_rgba8(ctx: ?*const Ctx) u64;
pub extern fn vrt_get_shard_info(ctx: ?*const Ctx, out: [*c]ShardInfo) c_int;
pub extern fn vrt_assemble_frame(ctx: ?*Ctx, gathered: ?*const anyopaque, dst_frame: ?*anyopaque, bytes_per_pixel: u32) c_int;
pub extern fn vrt_dist_unique_id(rccl_path: ?[*:0]const u8, out_id128: ?*anyopaque) c_int;
pub extern fn vrt_dist_init(ctx: ?*Ctx, rccl_path: ?[*:0]const u8, id128: ?*const anyopaque, rank: c_int, world: c_int, frames_in_flight: u32) c_int;
pub extern fn vrt_dist_init_batched(ctx: ?*Ctx, rccl_path: ?[*:0]const u8, id128: ?*const anyopaque, rank: c_int, world: c_int, frames_in_flight: u32, frames_per_launch: u32) c_int;
pub extern fn vrt_dist_init_ex(ctx: ?*Ctx, rccl_path: ?[*:0]const u8, id128: ?*const anyopaque, rank: c_int, world: c_int, options: [*c]const DistOptions) c_int;
pub extern fn vrt_dist_keep_communicators(keep: c_int) c_int;
pub extern fn vrt_dist_release_communicators() c_int;
pub extern fn vrt_dist_comm_info(ctx: ?*Ctx, out: *[4]i32) c_int;
pub extern fn vrt_dist_frame(ctx: ?*Ctx, camera: [*c]const CameraDevice, sun: [*c]const SunDevice) c_int;
pub extern fn vrt_dist_frames(ctx: ?*Ctx, cameras: [*c]const CameraDevice, suns: [*c]const SunDevice, n: u32, sun_stride: u32) c_int;
pub extern fn vrt_dist_wait(ctx: ?*Ctx) c_int;
pub extern fn vrt_dist_read_frame(ctx: ?*Ctx, dst: ?*anyopaque, nbytes: u64) c_int;
pub extern fn vrt_dist_broadcast(ctx: ?*Ctx, id: BufferId, byte_offset: u64, nbytes: u64, root: c_int) c_int;
pub extern fn vrt_dist_info(ctx: ?*Ctx, out: *[4]i32) c_int;
pub extern fn vrt_dist_profile(ctx: ?*Ctx, enable: u32) c_int;
pub extern fn vrt_dist_stats(ctx: ?*Ctx, out: *[8]f64) c_int;
pub extern fn vrt_dist_selftest(ctx: ?*Ctx) c_int;
pub extern fn vrt_dist_selftest_slots(ctx: ?*Ctx, busy_us: u32, rounds: u32, out: *[4]f64) c_int;
pub extern fn vrt_last_kernel_ms(ctx: ?*Ctx) f64;
pub extern fn vrt_region_begin(ctx: ?*Ctx) c_int;
pub extern fn vrt_region_end(ctx: ?*Ctx, ms: [*c]f64) c_int;
pub extern fn vrt_get_counters(ctx: ?*Ctx, out: [*c]Counters) c_int;
pub extern fn vrt_get_wave_counters(ctx: ?*Ctx, out: *[3]u64) c_int;
pub extern fn vrt_trace_wave_timeline(ctx: ?*Ctx, camera: [*c]const CameraDevice, sun: [*c]const SunDevice, out: [*c]u64, capacity_pairs: u64, n_pairs: [*c]u64) c_int;
pub extern fn vrt_device_info(device: c_int, out: *[4]i64) c_int;
pub extern fn vrt_last_error(ctx: ?*const Ctx) [*:0]const u8;
pub extern fn vrt_abi_version() u32;
pub extern fn vrt_kernel_name(ctx: ?*const Ctx) [*:0]const u8;
pub extern fn vrt_bounce_autotune_info(ctx: ?*Ctx, out: *[4]f64) c_int;
pub extern fn vrt_compiled_kernel_count() c_int;
pub extern fn vrt_grid_create(dim_x: u32, dim_y: u32, dim_z: u32, cfg: [*c]const GridConfig, out: *?*Grid) c_int;
pub extern fn vrt_grid_destroy(g: ?*Grid) void;
pub extern fn vrt_grid_insert(g: ?*Grid, x: u64, y: u64, z: u64, material_index: u8) c_int;
pub extern fn vrt_grid_insert_many(g: ?*Grid, xyz: [*c]const u32, materials: [*c]const u8, n: u64) c_int;
pub extern fn vrt_grid_device_state(g: ?*const Grid) [*c]const GridState;
pub extern fn vrt_grid_data(g: ?*const Grid, id: BufferId, nbytes: [*c]u64) ?*const anyopaque;
pub extern fn vrt_grid_active_bricks(g: ?*const Grid) u32;
pub extern fn vrt_grid_brick_dimension(g: ?*const Grid) u32;
pub extern fn vrt_grid_delta(g: ?*const Grid, id: BufferId, from: [*c]u64, to: [*c]u64) c_int;
pub extern fn vrt_grid_reset_delta(g: ?*Grid, id: BufferId) void;
pub extern fn vrt_upload_grid(ctx: ?*Ctx, g: ?*Grid) c_int;
pub extern fn vrt_update_grid_delta(ctx: ?*Ctx, g: ?*Grid) c_int;
pub extern fn vrt_camera_init(vertical_fov_deg: f32, image_width: u32, image_height: u32, cfg: [*c]const CameraConfig, out: [*c]CameraDevice) c_int;
pub extern fn vrt_camera_set_forward(cam: [*c]CameraDevice, vertical_fov_deg: f32, viewport_height: f32, forward: *const [3]f32) c_int;
pub extern fn vrt_sun_init(cfg: [*c]const SunConfig, out: [*c]SunDevice) c_int;
pub extern fn vrt_default_materials(out: [*c]Material, capacity: u32) u32;
pub extern fn vrt_synth_terrain(g: ?*Grid, seed: u64) c_int;
pub extern fn vrt_synth_sparse(g: ?*Grid, seed: u64, p: f32) c_int;
pub extern fn vrt_denoise(ctx: ?*Ctx, cfg: [*c]const DenoiseConfig, out_w: u32, out_h: u32, want_float: u32) c_int;
pub extern fn vrt_read_denoised_rgba8(ctx: ?*Ctx, dst: ?*anyopaque, nbytes: u64) c_int;
pub extern fn vrt_read_denoised_rgba32f(ctx: ?*Ctx, dst: ?*anyopaque, nbytes: u64) c_int;
pub extern fn vrt_device_denoised_rgba8(ctx: ?*Ctx) ?*anyopaque;
pub extern fn vrt_last_denoise_ms(ctx: ?*Ctx) f64;
pub extern fn vrt_vox_validate_header(buffer: ?*const anyopaque, nbytes: u64) c_int;
pub extern fn vrt_vox_parse(buffer: ?*const anyopaque, nbytes: u64, strict: c_int, out: *?*Vox) c_int;
pub extern fn vrt_vox_destroy(v: ?*Vox) void;
pub extern fn vrt_vox_num_models(v: ?*const Vox) u32;
pub extern fn vrt_vox_model_size(v: ?*const Vox, model: u32, size_xyz: *[3]i32) c_int;
pub extern fn vrt_vox_model_voxels(v: ?*const Vox, model: u32, count: [*c]u64) [*c]const VoxXyzi;
pub extern fn vrt_vox_palette(v: ?*const Vox) [*c]const VoxRgba;
pub extern fn vrt_vox_materials(v: ?*const Vox, out: [*c]Material, count: u32) c_int;
pub extern fn vrt_vox_insert(g: ?*Grid, v: ?*const Vox, model: u32, off_x: u32, off_y: u32, off_z: u32, material_offset: u32) c_int;
pub extern fn vrt_benchmark_create(cam: [*c]CameraDevice, vertical_fov_deg: f32, viewport_height: f32, out: *?*Benchmark) c_int;
pub extern fn vrt_benchmark_destroy(b: ?*Benchmark) void;
pub extern fn vrt_benchmark_update(b: ?*Benchmark, dt_seconds: f32, cam: [*c]CameraDevice) c_int;
pub extern fn vrt_benchmark_report(b: ?*const Benchmark, min_ms: [*c]f32, max_ms: [*c]f32, avg_ms: [*c]f32) c_int;
// END GENERATED

/// Drop-in for the compute side of voxel_rt/Pipeline.zig: same call shapes as
/// Pipeline.transfer*(ctx, offset, slice) and compute_pipeline.dispatch(ctx, wg, camera, sun).
/// A Zig host needs exactly this subset: vrt_create / vrt_destroy / vrt_upload / vrt_dispatch / vrt_wait /
/// vrt_read_rgba8 (or vrt_device_target_rgba8 for interop) / vrt_last_error; it keeps its own BrickGrid, Camera, Sun,
/// vox loader and Benchmark, so the vrt_grid_* / vrt_camera_* / vrt_sun_* / vrt_vox_* / vrt_benchmark_* families are
/// for hosts without them (the C++ example, the Python tests).
pub const HipComputePipeline = struct {
    ctx: *Ctx,
    /// ComputePipeline.dispatch waits for the previous frame's fence before it records the next one
    /// (ComputePipeline.zig:423-434).  true keeps that; false lets frames queue in stream order (vrt_dispatch itself
    /// never blocks), which is what a host that double-buffers its target wants.
    wait_for_previous_frame: bool = true,

    pub fn init(width: u32, height: u32, grid_state: anytype) Error!HipComputePipeline {
        const d = grid_state.device_state;
        var out: ?*Ctx = null;
        try check(vrt_create(&Config{
            .width = width,
            .height = height,
            .dim_x = d.dim_x,
            .dim_y = d.dim_y,
            .dim_z = d.dim_z,
            .brick_alloc = grid_state.brick_start_indices.len,
        }, &out));
        return .{ .ctx = out.? };
    }

    pub fn deinit(self: HipComputePipeline) void {
        vrt_destroy(self.ctx);
    }

    /// Pipeline.transferBrickStatuses / Indices / Occupancy / StartIndex / MaterialIndices / Materials
    pub fn transfer(self: HipComputePipeline, id: BufferId, comptime T: type, offset: usize, slice: []const T) Error!void {
        try check(vrt_upload(self.ctx, id, offset * @sizeOf(T), slice.ptr, slice.len * @sizeOf(T)));
    }

    /// ComputePipeline.dispatch(ctx, workgroup_size, camera, sun)
    pub fn dispatch(self: HipComputePipeline, camera: anytype, sun: anytype) Error!void {
        if (self.wait_for_previous_frame) try check(vrt_wait(self.ctx));
        try check(vrt_dispatch(self.ctx, @ptrCast(&camera.d_camera), @ptrCast(&sun.device_data)));
    }

    pub fn lastError(self: HipComputePipeline) [*:0]const u8 {
        return vrt_last_error(self.ctx);
    }
};
