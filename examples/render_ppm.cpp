// A compiled host on the C ABI alone (no Python, no PyTorch): what VoxelRT.init + pushMaterials +
// draw amount to for the reference app (VoxelRT.zig:40-87, Pipeline.zig:441), written against
// include/vrt_hip.h.  Builds the deterministic terrain scene, renders one frame and writes a PPM.
//
//   g++ -O2 -std=c++17 -I../include render_ppm.cpp -L../zig_vulkan_amd -lvrt_hip -Wl,-rpath,'$ORIGIN/../zig_vulkan_amd' -o render_ppm
//   ./render_ppm out.ppm [width height voxels_per_axis brick_dimension]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vrt_hip.h"

#define CHECK(call)                                                                                      \
    do {                                                                                                 \
        const int rc_ = (call);                                                                          \
        if (rc_ != VRT_OK) {                                                                             \
            std::fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, ctx ? vrt_last_error(ctx) : "no context"); \
            return 1;                                                                                    \
        }                                                                                                \
    } while (0)

int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "frame.ppm";
    const uint32_t width = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 640, height = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 360;
    const uint32_t voxels = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 128, b = argc > 5 ? (uint32_t)std::atoi(argv[5]) : 8;
    const uint32_t n = voxels / b;
    vrt_ctx *ctx = nullptr;

    // scene: BrickGrid (Grid.zig:36) 64 world units wide like src/main.zig:77-81, terrain of SURVEY.md §8(d)
    vrt_grid_config gc;
    std::memset(&gc, 0, sizeof gc);
    gc.base_t = 0.01f;
    gc.min_point[0] = gc.min_point[1] = gc.min_point[2] = -32.0f;
    gc.scale = 64.0f / (float)n;
    gc.brick_dimension = b;
    vrt_grid *grid = nullptr;
    CHECK(vrt_grid_create(n, n, n, &gc, &grid));
    CHECK(vrt_synth_terrain(grid, 420));

    // pipeline (ComputePipeline.init, Pipeline.zig:272-316)
    vrt_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.abi_version = VRT_ABI_VERSION;
    cfg.width = width;
    cfg.height = height;
    cfg.brick_dimension = b;
    cfg.dim_x = cfg.dim_y = cfg.dim_z = n;
    cfg.device_id = -1;
    CHECK(vrt_create(&cfg, &ctx));
    CHECK(vrt_upload_grid(ctx, grid)); // transferGridState + the five arrays (VoxelRT.zig:62)

    std::vector<vrt_material> materials(256);
    std::memset(materials.data(), 0, materials.size() * sizeof(vrt_material));
    vrt_default_materials(materials.data(), 256); // terrain.zig:130-196
    CHECK(vrt_upload(ctx, VRT_BUF_MATERIALS, 0, materials.data(), materials.size() * sizeof(vrt_material)));

    // camera and sun (Camera.zig:36-77, Sun.zig:35-63): fov 75, the corner view of Benchmark.zig:152
    vrt_camera_config cc;
    std::memset(&cc, 0, sizeof cc);
    cc.viewport_height = 2.0f;
    cc.origin[0] = 20.0f; cc.origin[1] = -20.0f; cc.origin[2] = 20.0f;
    cc.samples_per_pixel = 1;
    cc.max_bounce = 0;
    vrt_camera_device cam;
    CHECK(vrt_camera_init(75.0f, width, height, &cc, &cam));
    // view V2: looking at the grid centre (0, 0, 0); the reference's forward is the direction rays leave AGAINST
    // (lower_left_corner = origin - h/2 - v/2 - forward, Camera.zig:177-180); normalised by the callee
    const float fwd[3] = {cc.origin[0] - 0.0f, cc.origin[1] - 0.0f, cc.origin[2] - 0.0f};
    CHECK(vrt_camera_set_forward(&cam, 75.0f, 2.0f, fwd));
    vrt_sun_config sc;
    std::memset(&sc, 0, sizeof sc);
    sc.enabled = 1;
    sc.color[0] = 1.0f; sc.color[1] = 1.1f; sc.color[2] = 1.0f;
    sc.radius = 5.0f;
    sc.sun_distance = 1000.0f;
    vrt_sun_device sun;
    CHECK(vrt_sun_init(&sc, &sun));

    CHECK(vrt_dispatch(ctx, &cam, &sun)); // Pipeline.draw -> compute dispatch (Pipeline.zig:441)
    CHECK(vrt_wait(ctx));
    std::vector<uint8_t> rgba((size_t)width * height * 4);
    CHECK(vrt_read_rgba8(ctx, rgba.data(), rgba.size()));
    std::fprintf(stderr, "%s: %ux%u, %u^3 voxels in %u^3 bricks, kernel %.3f ms\n", vrt_kernel_name(ctx), width, height, voxels, b, vrt_last_kernel_ms(ctx));

    std::FILE *f = std::fopen(path, "wb");
    if (!f) { std::perror(path); return 1; }
    std::fprintf(f, "P6\n%u %u\n255\n", width, height);
    for (size_t i = 0; i < (size_t)width * height; i++) std::fwrite(&rgba[4 * i], 1, 3, f);
    std::fclose(f);
    uint64_t sum = 1469598103934665603ull; // FNV-1a of the RGBA8 frame, for the test that compares with the Python host
    for (uint8_t v : rgba) sum = (sum ^ v) * 1099511628211ull;
    std::printf("%016llx\n", (unsigned long long)sum);
    vrt_destroy(ctx);
    vrt_grid_destroy(grid);
    return 0;
}
