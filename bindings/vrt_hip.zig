//! Zig binding of libvrt_hip.so (include/vrt_hip.h) — the declarations a maintainer of
//! Avokadoen/zig_vulkan adds to route the brick ray tracer's compute dispatch to the MI355X
//! HIP kernels instead of vkCmdDispatch.  Declarative only: zig is not available in the build
//! image of this repository, so this file is not compiled here.
//!
//! Layout notes: the extern structs below are the reference's own device structs
//! (State.Device, gpu_types.Material, Camera.Device, Sun.Device); they can be passed as-is.

const std = @import("std");

pub const Ctx = opaque {};
pub const Grid = opaque {};

pub const Status = enum(c_int) {
    ok = 0,
    invalid_arg = -1,
    oom = -2,
    out_of_range = -3,
    hip = -4,
    no_device = -5,
    state = -6,
};

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

pub const Config = extern struct {
    struct_size: u32 = @sizeOf(Config),
    abi_version: u32 = 1,
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
    _reserved: [5]u32 = [_]u32{0} ** 5,
};

pub extern fn vrt_create(cfg: *const Config, out: *?*Ctx) c_int;
pub extern fn vrt_destroy(ctx: ?*Ctx) void;
pub extern fn vrt_upload(ctx: *Ctx, id: BufferId, byte_offset: u64, src: ?*const anyopaque, nbytes: u64) c_int;
pub extern fn vrt_buffer_size(ctx: *const Ctx, id: BufferId) u64;
/// camera: *const Camera.Device (96 bytes), sun: *const Sun.Device (32 bytes)
pub extern fn vrt_dispatch(ctx: *Ctx, camera: *const anyopaque, sun: *const anyopaque) c_int;
pub extern fn vrt_wait(ctx: *Ctx) c_int;
pub extern fn vrt_read_rgba8(ctx: *Ctx, dst: *anyopaque, nbytes: u64) c_int;
pub extern fn vrt_read_rgba32f(ctx: *Ctx, dst: *anyopaque, nbytes: u64) c_int;
pub extern fn vrt_device_target_rgba8(ctx: *Ctx) ?*anyopaque;
pub extern fn vrt_last_kernel_ms(ctx: *Ctx) f64;
pub extern fn vrt_last_error(ctx: ?*const Ctx) [*:0]const u8;

// the step after the path: image.frag's denoiser as a HIP kernel (GraphicsPipeline.Config defaults when cfg == null)
pub const DenoiseConfig = extern struct { samples: i32 = 20, distribution_bias: f32 = 0.6, pixel_multiplier: f32 = 1.5, inverse_hue_tolerance: f32 = 20 };
pub extern fn vrt_denoise(ctx: *Ctx, cfg: ?*const DenoiseConfig, out_w: u32, out_h: u32, want_float: u32) c_int;
pub extern fn vrt_read_denoised_rgba8(ctx: *Ctx, dst: *anyopaque, nbytes: u64) c_int;
pub extern fn vrt_device_denoised_rgba8(ctx: *Ctx) ?*anyopaque;

// multi-GPU frame pipeline (one process per GPU; rank 0 owns the assembled frame)
pub extern fn vrt_dist_unique_id(rccl_path: [*:0]const u8, out_id128: *[128]u8) c_int;
pub extern fn vrt_dist_init(ctx: *Ctx, rccl_path: [*:0]const u8, id128: *const [128]u8, rank: c_int, world: c_int, frames_in_flight: u32) c_int;
pub extern fn vrt_dist_init_batched(ctx: *Ctx, rccl_path: [*:0]const u8, id128: *const [128]u8, rank: c_int, world: c_int, frames_in_flight: u32, frames_per_launch: u32) c_int;
pub extern fn vrt_dist_frame(ctx: *Ctx, camera: *const anyopaque, sun: *const anyopaque) c_int;
pub extern fn vrt_dist_wait(ctx: *Ctx) c_int;
pub extern fn vrt_dist_read_frame(ctx: *Ctx, dst: *anyopaque, nbytes: u64) c_int;

fn check(rc: c_int) !void {
    return switch (@as(Status, @enumFromInt(rc))) {
        .ok => {},
        .oom => error.OutOfDeviceMemory,
        .out_of_range => error.DestOutOfDeviceMemory, // StagingRamp.zig:320-325
        .no_device => error.NoHipDevice,
        else => error.VrtFailure,
    };
}

/// Drop-in for the compute side of voxel_rt/Pipeline.zig: same call shapes as
/// Pipeline.transfer*(ctx, offset, slice) and compute_pipeline.dispatch(ctx, wg, camera, sun).
pub const HipComputePipeline = struct {
    ctx: *Ctx,

    pub fn init(width: u32, height: u32, grid_state: anytype) !HipComputePipeline {
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
    pub fn transfer(self: HipComputePipeline, id: BufferId, comptime T: type, offset: usize, slice: []const T) !void {
        try check(vrt_upload(self.ctx, id, offset * @sizeOf(T), slice.ptr, slice.len * @sizeOf(T)));
    }

    /// ComputePipeline.dispatch(ctx, workgroup_size, camera, sun)
    pub fn dispatch(self: HipComputePipeline, camera: anytype, sun: anytype) !void {
        try check(vrt_dispatch(self.ctx, &camera.d_camera, &sun.device_data));
    }
};
