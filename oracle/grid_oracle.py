"""Pure-Python restatement of the reference's BrickGrid host data structure.

TEST INFRASTRUCTURE ONLY (parity oracle for the grid builder; small cases).
Follows /root/reference/src/modules/voxel_rt/brick/Grid.zig:36-211,
State.zig:5-129 and MaterialAllocator.zig:34-43 line by line, with
brick_dimension as a parameter (the reference fixes it to 4, State.zig:5).
PARITY UNPINNED by reference tests (Grid.zig:196 says "TODO: test").
"""
from __future__ import annotations

import numpy as np

UNSET_INDEX = 0xFFFFFFFF  # Brick.unset_index, State.zig:122-123
USIZE_MAX = (1 << 64) - 1


class Delta:  # DeviceDataDelta, State.zig:14-57
    def __init__(self):
        # `.empty`, State.zig:15-20
        self.state = "inactive"
        self.from_ = 0
        self.to = 0

    def reset(self):  # State.zig:33-37
        self.state = "inactive"
        self.from_ = USIZE_MAX
        self.to = 0

    def register(self, i: int):  # State.zig:39-46
        self.state = "active"
        self.from_ = min(self.from_, i)
        self.to = max(self.to, i + 1)


class GridOracle:
    def __init__(self, dim_x: int, dim_y: int, dim_z: int, *, brick_alloc=None, base_t=0.01, min_point=(0.0, 0.0, 0.0),
                 scale=1.0, brick_dimension=4):
        b = brick_dimension
        self.b = b
        self.brick_bits = b * b * b          # State.zig:6
        self.brick_bytes = self.brick_bits // 8  # State.zig:7
        brick_count = dim_x * dim_y * dim_z  # Grid.zig:40
        assert brick_count > 0
        self.brick_statuses = np.zeros((brick_count + 31) // 32, dtype=np.uint32)  # Grid.zig:43-46
        self.brick_indices = np.zeros(brick_count, dtype=np.uint32)                # Grid.zig:48-50
        brick_alloc = brick_alloc or brick_count                                   # Grid.zig:51
        self.brick_alloc = brick_alloc
        self.brick_occupancy = np.zeros(brick_alloc * self.brick_bytes, dtype=np.uint8)  # Grid.zig:53-55
        self.brick_start_indices = np.full(brick_alloc, UNSET_INDEX, dtype=np.uint32)     # Grid.zig:57-59
        self.material_indices = np.zeros(brick_alloc * self.brick_bits, dtype=np.uint8)  # Grid.zig:61-64
        f32 = np.float32
        mn = [f32(v) for v in min_point]
        sc = f32(scale)
        # Grid.zig:66-79, 93-102
        self.device_state = {
            "voxel_dim": (dim_x * b, dim_y * b, dim_z * b),
            "dim": (dim_x, dim_y, dim_z),
            "min_point_base_t": (mn[0], mn[1], mn[2], f32(base_t)),
            "max_point_scale": (f32(mn[0] + f32(dim_x) * sc), f32(mn[1] + f32(dim_y) * sc), f32(mn[2] + f32(dim_z) * sc), sc),
        }
        self.active_bricks = 0
        self.material_next = 0  # MaterialAllocator.next_index
        self.material_capacity = self.material_indices.size
        self.deltas = {k: Delta() for k in ("statuses", "indices", "occupancy", "start_indices", "material_indices")}

    def device_state_bytes(self) -> bytes:
        d = self.device_state
        return (np.array(list(d["voxel_dim"]) + list(d["dim"]) + [0, 0], dtype=np.uint32).tobytes()
                + np.array(d["min_point_base_t"], dtype=np.float32).tobytes()
                + np.array(d["max_point_scale"], dtype=np.float32).tobytes())

    def voxel_at(self, x, y, z):  # Grid.zig:198-203
        b = self.b
        return (x % b) + b * ((z % b) + b * (y % b))

    def grid_at(self, x, y, z):  # Grid.zig:206-211
        b = self.b
        dx, _, dz = self.device_state["dim"]
        return (x // b) + dx * ((z // b) + dz * (y // b))

    def insert(self, x: int, y: int, z: int, material_index: int):  # Grid.zig:129-194
        vx, vy, vz = self.device_state["voxel_dim"]
        assert x < vx and y < vy and z < vz
        flipped_y = vy - 1 - y  # Grid.zig:135
        grid_index = self.grid_at(x, flipped_y, z)
        status_index, status_offset = divmod(grid_index, 32)
        loaded = (int(self.brick_statuses[status_index]) >> status_offset) & 1
        if loaded:
            brick_index = int(self.brick_indices[grid_index])
        else:
            brick_index = self.active_bricks  # fetchAdd, Grid.zig:147
            self.active_bricks += 1
        occupancy_from = brick_index * self.brick_bytes
        nth_bit = self.voxel_at(x, flipped_y, z)
        if int(self.brick_start_indices[brick_index]) == UNSET_INDEX:  # Grid.zig:160-168
            entry = self.material_next  # MaterialAllocator.nextEntry
            self.material_next += self.brick_bits
            assert entry < self.material_capacity
            self.brick_start_indices[brick_index] = entry & 0x7FFFFFFF  # type = voxel_start_index (0)
            self.deltas["start_indices"].register(brick_index)
        new_voxel_material_index = (int(self.brick_start_indices[brick_index]) & 0x7FFFFFFF) + nth_bit
        self.material_indices[new_voxel_material_index] = material_index
        self.deltas["material_indices"].register(new_voxel_material_index)
        mask_index, mask_bit = divmod(nth_bit, 8)
        self.brick_occupancy[occupancy_from + mask_index] |= (1 << mask_bit)
        self.deltas["occupancy"].register(occupancy_from + mask_index)
        self.brick_statuses[status_index] |= np.uint32(1 << status_offset)
        self.deltas["statuses"].register(status_index)
        self.brick_indices[grid_index] = brick_index
        self.deltas["indices"].register(grid_index)
