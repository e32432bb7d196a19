"""Named workloads (BASELINE.json `configs`): grid placement,
synthetic scene, camera views.  Used by bench.py and the parity tests so both
measure and check the same inputs."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Sequence, Tuple

from .voxel_rt import BrickGrid, Camera, CameraConfig, Config, Sun, SunConfig, VoxelRT, default_materials

SEED = 420  # terrain seed of the reference app (src/main.zig:120)


@dataclass(frozen=True)
class Workload:
    name: str
    width: int
    height: int
    voxels: int            # voxels per axis (cubic grid)
    brick_dimension: int
    spp: int
    max_bounce: int        # Camera.Config.max_bounce (device value is +1)
    sun_enabled: bool
    sun_radius: float
    scene: str = "terrain"  # "terrain" | "sparse"
    sparse_p: float = 0.0
    brick_alloc: int = 0   # 0 => dense allocation (every brick slot, Grid.zig:51)
    dims: Tuple[int, int, int] | None = None   # bricks per axis when the grid is not cubic (None: voxels / brick_dimension each)


# BASELINE.json configs[0..4]
WORKLOADS: Dict[str, Workload] = {
    # configs[0]: 256x256, 64^3 dense, 1 primary ray/pixel (the reference's CPU-runnable case)
    "cfg0_256x256_64c_b4": Workload("cfg0_256x256_64c_b4", 256, 256, 64, 4, 1, 0, False, 0.0),
    # configs[1]: 1920x1080, 256^3 dense, 1 primary ray/pixel
    "cfg1_1080p_256c_b4": Workload("cfg1_1080p_256c_b4", 1920, 1080, 256, 4, 1, 0, False, 0.0),
    # configs[2]: 1920x1080, 512^3 brickmap (8^3 bricks), primary + shadow  <- the headline metric
    "cfg2_1080p_512c_b8": Workload("cfg2_1080p_512c_b8", 1920, 1080, 512, 8, 1, 0, True, 5.0),
    # same, reference-native 4^3 bricks
    "cfg2_1080p_512c_b4": Workload("cfg2_1080p_512c_b4", 1920, 1080, 512, 4, 1, 0, True, 5.0),
    # configs[3]: 3840x2160, 1024^3 brickmap, 2 spp x (primary + shadow) = 4 rays/pixel
    "cfg3_4k_1024c_b8": Workload("cfg3_4k_1024c_b8", 3840, 2160, 1024, 8, 2, 0, True, 5.0),
    # configs[4]: 3840x2160, 2048^3 sparse brickmap, 16 spp diffuse path trace
    "cfg4_4k_2048c_b8_sparse": Workload("cfg4_4k_2048c_b8_sparse", 3840, 2160, 2048, 8, 16, 2, True, 5.0, "sparse", 0.08,
                                        4_000_000),
    # not a BASELINE config: the shape of the reference app's own default run (src/main.zig:23,77-81,122-135: 1024x576 internal
    # resolution, 2 samples, max_bounce 2, sun on, 4^3 bricks), on the cubic synthetic terrain
    "refapp_1024x576_512c_b4": Workload("refapp_1024x576_512c_b4", 1024, 576, 512, 4, 2, 2, True, 5.0),
    # the reference app's own default run, grid shape included (src/main.zig:77-81: 128 x 64 x 128 bricks of 4^3, min point
    # (-32, -16, -32), scale 0.5; :23,122-135: 1024x576, 2 samples, max_bounce 2, sun on): the workload a user of the reference sees
    "refapp_1024x576_128x64x128_b4": Workload("refapp_1024x576_128x64x128_b4", 1024, 576, 512, 4, 2, 2, True, 5.0, dims=(128, 64, 128)),
    # not a BASELINE config (round 6): configs[3] / [4]'s frame size on the largest scene of configs[4]'s kind whose buffers fit GL's 128 MiB
    # storage-block limit — the scene of the reference-shader fixtures that pin configs[3]'s and [4]'s kernels directly
    # (tests/golden/make_ref_golden.py big): 128^3 cells -> the word-mode kernels; occupied cells reach the grid's faces -> vrt_pool_kernel
    "refbig_4k_1024c_b8_sparse": Workload("refbig_4k_1024c_b8_sparse", 3840, 2160, 1024, 8, 4, 2, True, 5.0, "sparse", 0.08, 80_000),
    # not a BASELINE config (round 5): a path trace on 4^3 bricks — the reference's own brick size — at a size the persistent kernels
    # are chosen for themselves... only where bindings 3-5 exceed the caches, which 4^3 bricks reach at 2048^3; used with kernel_variant
    # bit 23 by the A/B of vrt_pool_kernel<4, ...> against vrt_path_kernel<4, ...> (tools/lib_ab.py)
    "sparse_4k_1024c_b4": Workload("sparse_4k_1024c_b4", 3840, 2160, 1024, 4, 4, 2, True, 5.0, "sparse", 0.08, 2_000_000),
}

HEADLINE = "cfg2_1080p_512c_b8"

# camera views: (origin, look-at target or None for the reference's start orientation).  V0, V1, V2 are SURVEY.md §8(d)'s
# views, cycled by bench.py's timed region; the others are reported per view only.
VIEWS: Dict[str, Tuple[Sequence[float], Sequence[float] | None]] = {
    "V0": ((0.0, 0.0, 0.0), None),                      # reference start: origin, looking -Z (Camera.zig:7,47)
    "V1": ((0.0, -20.0, 30.0), (0.0, 0.0, 0.0)),        # above the terrain, looking at the grid centre (world is Y-down)
    "V2": ((20.0, -20.0, 20.0), (0.0, 0.0, 0.0)),       # Benchmark.zig:152's corner position, looking at the centre
    # round 1's V1: truly outside the grid box (V1 of §8(d) is above the terrain but inside the box); 10 % of the rays hit
    "V1x": ((0.0, -44.0, 70.0), (0.0, 12.0, 0.0)),
    # all-ground: just above the terrain looking straight down; 98 % of the primary rays hit (no sky to inflate Mrays/s)
    "VG": ((0.0, 2.0, 0.0), (0.5, 32.0, 0.5)),
}


def build_grid(w: Workload) -> BrickGrid:
    n = w.voxels // w.brick_dimension
    # world box 64 units wide like the reference default scene (src/main.zig:77-81)
    scale = 64.0 / n
    dx, dy, dz = w.dims or (n, n, n)
    grid = BrickGrid(dx, dy, dz, min_point=(-32.0, -32.0 * dy / dx, -32.0 * dz / dx), scale=scale, brick_dimension=w.brick_dimension,
                     brick_alloc=w.brick_alloc or None)
    if w.scene == "terrain":
        grid.synth_terrain(SEED)
    else:
        grid.synth_sparse(SEED, w.sparse_p)
    return grid


def make_renderer(w: Workload, grid: BrickGrid, **overrides) -> VoxelRT:
    cfg = Config(
        internal_resolution_width=overrides.pop("width", w.width),
        internal_resolution_height=overrides.pop("height", w.height),
        camera=CameraConfig(samples_per_pixel=w.spp, max_bounce=w.max_bounce),
        sun=SunConfig(enabled=w.sun_enabled, radius=overrides.pop("sun_radius", w.sun_radius)),
        **overrides,
    )
    rt = VoxelRT(grid, cfg)
    rt.push_materials(default_materials(256))
    return rt


def apply_view(camera: Camera, view: str) -> None:
    origin, target = VIEWS[view]
    if target is None:
        camera.set_forward((0.0, 0.0, 1.0))
        camera.set_origin(origin)
    else:
        camera.look_at(origin, target)


def set_view(rt: VoxelRT, view: str) -> None:
    apply_view(rt.camera, view)


def camera_for(w: Workload, view: str, width: int | None = None, height: int | None = None) -> Camera:
    """Camera exactly as VoxelRT.init builds it (fov 75, VoxelRT.zig:42), without a device context."""
    cam = Camera(75.0, width or w.width, height or w.height, CameraConfig(samples_per_pixel=w.spp, max_bounce=w.max_bounce))
    apply_view(cam, view)
    return cam


def sun_for(w: Workload, sun_radius: float | None = None) -> Sun:
    return Sun(SunConfig(enabled=w.sun_enabled, radius=w.sun_radius if sun_radius is None else sun_radius))
