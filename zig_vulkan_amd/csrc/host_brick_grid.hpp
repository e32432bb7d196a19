// host_brick_grid.hpp — C++ host mirror of the reference's BrickGrid
// (src/modules/voxel_rt/brick/Grid.zig, State.zig, MaterialAllocator.zig).
//
// Produces, byte for byte, the five arrays + State.Device the traversal kernel
// consumes.  brick_dimension is a run-time property of the grid (the reference
// fixes it to 4 at compile time, State.zig:5; every derived constant below
// follows State.zig:6-11 so that 8 is a legitimate instantiation).
#pragma once
#include <atomic>
#include <cstddef>
#include <cstdint>
#include <mutex>
#include <vector>
#include "../../include/vrt_hip.h"

namespace vrt {

// State.zig:14-57
struct DeviceDataDelta {
    enum class DeltaState { invalid, inactive, active };
    std::mutex mutex;
    DeltaState state = DeltaState::inactive;
    // `.empty` starts at from = 0, to = 0 (State.zig:15-20): the first delta
    // after construction therefore always starts at element 0.
    size_t from = 0;
    size_t to = 0;

    void resetDelta();
    void registerDelta(size_t delta_index);
    void registerDeltaRange(size_t from_, size_t to_);
    // same bookkeeping without taking the mutex (bulk inserts from one thread)
    void registerDeltaUnlocked(size_t delta_index) {
        state = DeltaState::active;
        if (delta_index < from) from = delta_index;
        if (delta_index + 1 > to) to = delta_index + 1;
    }
};

struct GridConfig { // Grid.zig:13-20
    uint64_t brick_alloc = 0; // 0 => all bricks
    float base_t = 0.01f;
    float min_point[3] = {0.0f, 0.0f, 0.0f};
    float scale = 1.0f;
    uint32_t brick_dimension = 4;
};

class BrickGrid {
public:
    // BrickGrid.init, Grid.zig:36-115
    static int create(uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, const GridConfig &cfg, BrickGrid **out);

    // BrickGrid.insert, Grid.zig:129-194
    int insert(uint64_t x, uint64_t y, uint64_t z, uint8_t material_index) { return insertImpl<true>(x, y, z, material_index); }
    // single-threaded bulk path: identical results, no per-delta locking
    int insertUnlocked(uint64_t x, uint64_t y, uint64_t z, uint8_t material_index) { return insertImpl<false>(x, y, z, material_index); }

    // State.zig:5-11
    uint32_t brickDimension() const { return brick_dimension_; }
    uint32_t brickBits() const { return brick_bits_; }
    uint32_t brickBytes() const { return brick_bytes_; }

    const vrt_grid_state &deviceState() const { return device_state_; }
    uint32_t activeBricks() const { return active_bricks_.load(std::memory_order_relaxed); }

    std::vector<uint32_t> brick_statuses;      // BrickStatusMask[], State.zig:86-107
    std::vector<uint32_t> brick_indices;       // IndexToBrick[], State.zig:109
    std::vector<uint8_t> brick_occupancy;      // State.zig:125-126
    std::vector<uint32_t> brick_start_indices; // Brick.StartIndex[], State.zig:117-120
    std::vector<uint8_t> material_indices;     // State.zig:129

    DeviceDataDelta brick_statuses_delta, brick_indices_delta, bricks_occupancy_delta, bricks_start_indices_delta,
        material_indices_delta;

    DeviceDataDelta *deltaFor(vrt_buffer_id id);
    const void *dataFor(vrt_buffer_id id, uint64_t *nbytes) const;
    size_t elementSize(vrt_buffer_id id) const;

private:
    BrickGrid() = default;
    template <bool Locked>
    int insertImpl(uint64_t x, uint64_t y, uint64_t z, uint8_t material_index);

    uint32_t brick_dimension_ = 4, brick_bits_ = 64, brick_bytes_ = 8;
    uint64_t brick_alloc_ = 0;
    vrt_grid_state device_state_{};
    std::atomic<uint32_t> active_bricks_{0};   // State.zig:132
    std::atomic<uint32_t> material_cursor_{0}; // MaterialAllocator.next_index
    size_t material_capacity_ = 0;             // MaterialAllocator.capacity
};

} // namespace vrt
