"""zig_vulkan_amd — MI355X-native brickmap voxel ray tracing path.

The product is libvrt_hip.so (hand-written HIP kernels for gfx950 behind the C
ABI of include/vrt_hip.h).  This package is the thin host-side mirror of the
reference's scene / camera / renderer API used by tests and bench.py.
Importing it loads the shared library and fails loudly if it is absent.
"""
from . import _lib
from .voxel_rt import (BrickGrid, Camera, CameraConfig, Config, MATERIAL_DTYPE, Sun, SunConfig, VoxelRT,
                       default_materials)

__all__ = ["BrickGrid", "Camera", "CameraConfig", "Config", "MATERIAL_DTYPE", "Sun", "SunConfig", "VoxelRT",
           "default_materials", "_lib"]
