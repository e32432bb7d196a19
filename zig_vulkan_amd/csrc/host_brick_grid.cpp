// host_brick_grid.cpp — see host_brick_grid.hpp.
#include "host_brick_grid.hpp"
#include <limits>
#include <new>

namespace vrt {

void DeviceDataDelta::resetDelta() { // State.zig:33-37
    state = DeltaState::inactive;
    from = std::numeric_limits<size_t>::max();
    to = std::numeric_limits<size_t>::min();
}

void DeviceDataDelta::registerDelta(size_t delta_index) { // State.zig:39-46
    std::lock_guard<std::mutex> lk(mutex);
    registerDeltaUnlocked(delta_index);
}

void DeviceDataDelta::registerDeltaRange(size_t from_, size_t to_) { // State.zig:49-56 (to is inclusive there)
    std::lock_guard<std::mutex> lk(mutex);
    state = DeltaState::active;
    if (from_ < from) from = from_;
    if (to_ + 1 > to) to = to_ + 1;
}

int BrickGrid::create(uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, const GridConfig &cfg, BrickGrid **out) {
    if (!out) return VRT_E_INVALID_ARG;
    *out = nullptr;
    const uint32_t b = cfg.brick_dimension ? cfg.brick_dimension : 4u;
    if (b != 4u && b != 8u) return VRT_E_INVALID_ARG;
    const uint64_t brick_count64 = (uint64_t)dim_x * dim_y * dim_z;
    if (brick_count64 == 0) return VRT_E_INVALID_ARG; // Grid.zig:38
    // grid indices are u32 in the shader (comp:318)
    if (brick_count64 > 0xFFFFFFFFull) return VRT_E_OUT_OF_RANGE;
    // State.Device.voxel_dim_* are u32 (State.zig:61-63): dim * brick_dimension must not wrap
    if ((uint64_t)dim_x * b > 0xFFFFFFFFull || (uint64_t)dim_y * b > 0xFFFFFFFFull || (uint64_t)dim_z * b > 0xFFFFFFFFull) return VRT_E_OUT_OF_RANGE;
    const uint64_t brick_alloc = cfg.brick_alloc ? cfg.brick_alloc : brick_count64; // Grid.zig:51
    const uint32_t brick_bits = b * b * b;
    // Brick.StartIndex.value is a u31 (State.zig:117-120)
    if (brick_alloc * brick_bits > 0x80000000ull) return VRT_E_OUT_OF_RANGE;

    BrickGrid *g = new (std::nothrow) BrickGrid();
    if (!g) return VRT_E_OOM;
    try {
        g->brick_dimension_ = b;
        g->brick_bits_ = brick_bits;
        g->brick_bytes_ = brick_bits / 8u;
        g->brick_alloc_ = brick_alloc;
        g->brick_statuses.assign((size_t)((brick_count64 + 31u) / 32u), 0u);        // Grid.zig:44-46
        g->brick_indices.assign((size_t)brick_count64, 0u);                         // Grid.zig:48-49
        g->brick_occupancy.assign((size_t)(brick_alloc * g->brick_bytes_), 0u);     // Grid.zig:53-55
        g->brick_start_indices.assign((size_t)brick_alloc, 0xFFFFFFFFu);            // Grid.zig:57-59 (unset_index)
        g->material_indices.assign((size_t)(brick_alloc * brick_bits), 0u);         // Grid.zig:61-64
    } catch (const std::bad_alloc &) {
        delete g;
        return VRT_E_OOM;
    }
    g->material_capacity_ = g->material_indices.size(); // Grid.zig:107

    vrt_grid_state &d = g->device_state_; // Grid.zig:66-102
    d.voxel_dim_x = dim_x * b;
    d.voxel_dim_y = dim_y * b;
    d.voxel_dim_z = dim_z * b;
    d.dim_x = dim_x;
    d.dim_y = dim_y;
    d.dim_z = dim_z;
    d.padding1 = 0;
    d.padding2 = 0;
    d.min_point_base_t[0] = cfg.min_point[0];
    d.min_point_base_t[1] = cfg.min_point[1];
    d.min_point_base_t[2] = cfg.min_point[2];
    d.min_point_base_t[3] = cfg.base_t;
    d.max_point_scale[0] = d.min_point_base_t[0] + (float)dim_x * cfg.scale;
    d.max_point_scale[1] = d.min_point_base_t[1] + (float)dim_y * cfg.scale;
    d.max_point_scale[2] = d.min_point_base_t[2] + (float)dim_z * cfg.scale;
    d.max_point_scale[3] = cfg.scale;
    *out = g;
    return VRT_OK;
}

template <bool Locked>
int BrickGrid::insertImpl(uint64_t x, uint64_t y, uint64_t z, uint8_t material_index) {
    const vrt_grid_state &d = device_state_;
    // Grid.zig:130-132 (asserts in the reference)
    if (x >= d.voxel_dim_x || y >= d.voxel_dim_y || z >= d.voxel_dim_z) return VRT_E_OUT_OF_RANGE;
    const uint32_t b = brick_dimension_;

    const uint64_t flipped_y = d.voxel_dim_y - 1 - y; // Grid.zig:135

    // gridAt, Grid.zig:206-211
    const size_t grid_index = (size_t)(x / b) + (size_t)d.dim_x * ((size_t)(z / b) + (size_t)d.dim_z * (size_t)(flipped_y / b));
    const size_t brick_status_index = grid_index / 32;
    const uint32_t brick_status_offset = (uint32_t)(grid_index % 32);
    const bool loaded = (brick_statuses[brick_status_index] >> brick_status_offset) & 1u; // BrickStatusMask.read
    uint32_t brick_index;
    if (loaded) {
        brick_index = brick_indices[grid_index];
    } else {
        brick_index = active_bricks_.fetch_add(1, std::memory_order_relaxed); // Grid.zig:147
        if (brick_index >= brick_alloc_) {
            active_bricks_.fetch_sub(1, std::memory_order_relaxed);
            return VRT_E_OOM; // the reference would index past brick_occupancy here
        }
    }

    const size_t occupancy_from = (size_t)brick_index * brick_bytes_;
    uint32_t &brick_material_index = brick_start_indices[brick_index];

    // voxelAt, Grid.zig:198-203
    const uint32_t nth_bit = (uint32_t)(x % b) + b * ((uint32_t)(z % b) + b * (uint32_t)(flipped_y % b));

    auto reg = [](DeviceDataDelta &dd, size_t i) {
        if (Locked) dd.registerDelta(i);
        else dd.registerDeltaUnlocked(i);
    };

    if (brick_material_index == 0xFFFFFFFFu) { // Grid.zig:160-168
        const uint32_t material_entry = material_cursor_.fetch_add(brick_bits_, std::memory_order_relaxed); // MaterialAllocator.zig:39
        if ((size_t)material_entry >= material_capacity_) return VRT_E_OOM;                                  // MaterialAllocator.zig:40
        brick_material_index = material_entry & 0x7FFFFFFFu; // value:u31, type = voxel_start_index (0)
        reg(bricks_start_indices_delta, brick_index);
    }
    const size_t new_voxel_material_index = (size_t)(brick_material_index & 0x7FFFFFFFu) + nth_bit; // Grid.zig:173
    material_indices[new_voxel_material_index] = material_index;
    reg(material_indices_delta, new_voxel_material_index);

    // Grid.zig:180-185
    const size_t mask_index = nth_bit / 8;
    const uint32_t mask_bit = nth_bit % 8;
    brick_occupancy[occupancy_from + mask_index] |= (uint8_t)(1u << mask_bit);
    reg(bricks_occupancy_delta, occupancy_from + mask_index);

    // Grid.zig:188-193
    brick_statuses[brick_status_index] |= (1u << brick_status_offset);
    reg(brick_statuses_delta, brick_status_index);
    brick_indices[grid_index] = brick_index;
    reg(brick_indices_delta, grid_index);
    return VRT_OK;
}

template int BrickGrid::insertImpl<true>(uint64_t, uint64_t, uint64_t, uint8_t);
template int BrickGrid::insertImpl<false>(uint64_t, uint64_t, uint64_t, uint8_t);

DeviceDataDelta *BrickGrid::deltaFor(vrt_buffer_id id) {
    switch (id) {
        case VRT_BUF_BRICK_STATUS: return &brick_statuses_delta;
        case VRT_BUF_BRICK_INDEX: return &brick_indices_delta;
        case VRT_BUF_BRICK_OCCUPANCY: return &bricks_occupancy_delta;
        case VRT_BUF_BRICK_START_INDEX: return &bricks_start_indices_delta;
        case VRT_BUF_MATERIAL_INDEX: return &material_indices_delta;
        default: return nullptr;
    }
}

size_t BrickGrid::elementSize(vrt_buffer_id id) const {
    switch (id) {
        case VRT_BUF_GRID_STATE: return sizeof(vrt_grid_state);
        case VRT_BUF_BRICK_STATUS:
        case VRT_BUF_BRICK_INDEX:
        case VRT_BUF_BRICK_START_INDEX: return 4;
        case VRT_BUF_BRICK_OCCUPANCY:
        case VRT_BUF_MATERIAL_INDEX: return 1;
        default: return 0;
    }
}

const void *BrickGrid::dataFor(vrt_buffer_id id, uint64_t *nbytes) const {
    const void *ptr = nullptr;
    uint64_t n = 0;
    switch (id) {
        case VRT_BUF_GRID_STATE: ptr = &device_state_; n = sizeof(device_state_); break;
        case VRT_BUF_BRICK_STATUS: ptr = brick_statuses.data(); n = brick_statuses.size() * 4ull; break;
        case VRT_BUF_BRICK_INDEX: ptr = brick_indices.data(); n = brick_indices.size() * 4ull; break;
        case VRT_BUF_BRICK_OCCUPANCY: ptr = brick_occupancy.data(); n = brick_occupancy.size(); break;
        case VRT_BUF_BRICK_START_INDEX: ptr = brick_start_indices.data(); n = brick_start_indices.size() * 4ull; break;
        case VRT_BUF_MATERIAL_INDEX: ptr = material_indices.data(); n = material_indices.size(); break;
        default: break;
    }
    if (nbytes) *nbytes = n;
    return ptr;
}

} // namespace vrt
