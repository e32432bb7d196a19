// host_scene.cpp — host mirror of the reference's Camera / Sun device structs,
// its default material table, and the deterministic synthetic scenes used as
// bench/test input.  CPU only.
#include <cmath>
#include <cstring>
#include "host_brick_grid.hpp"

namespace {

struct V3 {
    float x, y, z;
};
inline V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 scale(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 norm(V3 a) {
    const float l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    if (l == 0.0f) return a;
    return V3{a.x / l, a.y / l, a.z / l};
}

// Camera.propogatePitchChange + lowerLeftCorner, Camera.zig:167-180
void apply_forward(vrt_camera_device *cam, float viewport_width, float viewport_height, V3 forward) {
    const V3 up0 = V3{0.0f, 1.0f, 0.0f}; // za.Vec3.up()
    const V3 right = norm(cross(up0, forward));
    const V3 up = norm(cross(forward, right));
    const V3 horizontal = scale(right, viewport_width);
    const V3 vertical = scale(up, viewport_height);
    cam->horizontal[0] = horizontal.x; cam->horizontal[1] = horizontal.y; cam->horizontal[2] = horizontal.z;
    cam->vertical[0] = vertical.x; cam->vertical[1] = vertical.y; cam->vertical[2] = vertical.z;
    cam->lower_left_corner[0] = cam->origin[0] - horizontal.x * 0.5f - vertical.x * 0.5f - forward.x;
    cam->lower_left_corner[1] = cam->origin[1] - horizontal.y * 0.5f - vertical.y * 0.5f - forward.y;
    cam->lower_left_corner[2] = cam->origin[2] - horizontal.z * 0.5f - vertical.z * 0.5f - forward.z;
}

void viewport(float vertical_fov_deg, float cfg_viewport_height, uint32_t w, uint32_t h, float *vw, float *vh) {
    // Camera.zig:37-45
    const float aspect_ratio = (float)w / (float)h;
    const float a = (float)(3.14159265358979323846 * (1.0 / 180.0));
    const float theta = vertical_fov_deg * a;
    *vh = cfg_viewport_height * std::tan(theta * 0.5f);
    *vw = aspect_ratio * *vh;
}

// ---- synthetic scene noise: integer hash -> [0,1), no libm ----------------
inline uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352dU;
    h ^= h >> 15; h *= 0x846ca68bU;
    h ^= h >> 16;
    return h;
}
inline uint32_t hash3(uint32_t a, uint32_t b, uint32_t c, uint32_t seed) {
    return mix32(a * 0x9E3779B1U ^ mix32(b * 0x85EBCA77U ^ mix32(c * 0xC2B2AE3DU ^ seed)));
}
inline double unit(uint32_t h) { return (double)(h >> 8) * (1.0 / 16777216.0); }
inline double smooth(double t) { return t * t * (3.0 - 2.0 * t); }

double value_noise2(double x, double z, uint32_t seed) {
    const double fx = std::floor(x), fz = std::floor(z);
    const uint32_t ix = (uint32_t)(int64_t)fx, iz = (uint32_t)(int64_t)fz;
    const double tx = smooth(x - fx), tz = smooth(z - fz);
    const double v00 = unit(hash3(ix, iz, 0, seed)), v10 = unit(hash3(ix + 1, iz, 0, seed));
    const double v01 = unit(hash3(ix, iz + 1, 0, seed)), v11 = unit(hash3(ix + 1, iz + 1, 0, seed));
    const double a = v00 + (v10 - v00) * tx;
    const double b = v01 + (v11 - v01) * tx;
    return a + (b - a) * tz;
}

} // namespace

extern "C" {

int vrt_camera_init(float vertical_fov_deg, uint32_t image_width, uint32_t image_height, const vrt_camera_config *cfg,
                    vrt_camera_device *out) {
    if (!out || image_width == 0 || image_height == 0) return VRT_E_INVALID_ARG;
    vrt_camera_config c;
    if (cfg) c = *cfg;
    else {
        c.viewport_height = 2.0f;
        c.origin[0] = c.origin[1] = c.origin[2] = 0.0f;
        c.samples_per_pixel = 2;
        c.max_bounce = 2;
    }
    std::memset(out, 0, sizeof *out);
    float vw, vh;
    viewport(vertical_fov_deg, c.viewport_height, image_width, image_height, &vw, &vh);
    out->image_width = image_width;
    out->image_height = image_height;
    out->origin[0] = c.origin[0]; out->origin[1] = c.origin[1]; out->origin[2] = c.origin[2];
    out->samples_per_pixel = c.samples_per_pixel;
    out->max_bounce = c.max_bounce + 1; // Camera.zig:74
    apply_forward(out, vw, vh, V3{0.0f, 0.0f, 1.0f}); // za.Vec3.forward(), Camera.zig:47
    return VRT_OK;
}

int vrt_camera_set_forward(vrt_camera_device *cam, float vertical_fov_deg, float viewport_height, const float forward[3]) {
    if (!cam || !forward || cam->image_width == 0 || cam->image_height == 0) return VRT_E_INVALID_ARG;
    float vw, vh;
    viewport(vertical_fov_deg, viewport_height, cam->image_width, cam->image_height, &vw, &vh);
    const V3 f = norm(V3{forward[0], forward[1], forward[2]});
    if (!(f.x == f.x) || (f.x == 0.0f && f.y == 0.0f && f.z == 0.0f)) return VRT_E_INVALID_ARG;
    apply_forward(cam, vw, vh, f);
    return VRT_OK;
}

int vrt_sun_init(const vrt_sun_config *cfg, vrt_sun_device *out) { // Sun.zig:35-63
    if (!out) return VRT_E_INVALID_ARG;
    vrt_sun_config c;
    if (cfg) c = *cfg;
    else {
        c.enabled = 1;
        c.color[0] = 1.0f; c.color[1] = 1.1f; c.color[2] = 1.0f;
        c.radius = 5.0f;
        c.sun_distance = 1000.0f;
    }
    out->position[0] = 0.0f;
    out->position[1] = -c.sun_distance; // static_pos_vec, Sun.zig:41
    out->position[2] = 0.0f;
    out->enabled = c.enabled ? 1u : 0u;
    out->color[0] = c.color[0]; out->color[1] = c.color[1]; out->color[2] = c.color[2];
    out->radius = c.radius;
    return VRT_OK;
}

uint32_t vrt_default_materials(vrt_material *out, uint32_t capacity) { // terrain.zig:130-196
    static const vrt_material table[8] = {
        {2u, 0.117f, 0.45f, 0.85f, 1.333f},  // water (dielectric)
        {0u, 0.0f, 0.6f, 0.0f, 0.0f},        // grass 1
        {0u, 0.0f, 0.5019f, 0.0f, 0.0f},     // grass 2
        {0u, 0.301f, 0.149f, 0.0f, 0.0f},    // dirt 1
        {0u, 0.4f, 0.2f, 0.0f, 0.0f},        // dirt 2
        {0u, 0.275f, 0.275f, 0.275f, 0.0f},  // rock 1
        {0u, 0.225f, 0.225f, 0.225f, 0.0f},  // rock 2
        {1u, 0.6f, 0.337f, 0.282f, 0.45f},   // iron (metal)
    };
    const uint32_t n = capacity < 8u ? capacity : 8u;
    if (out) std::memcpy(out, table, n * sizeof(vrt_material));
    return 8u;
}

// Synthetic terrain: a value-noise height field h(x,z); voxel
// (x,y,z) is solid for h/2 <= y < h (a shell, in the spirit of
// terrain.zig:98-104) or y < ocean (water, material 0, terrain.zig:105-107).
// Columns are visited x-major, z, then y ascending, so brick slots are handed
// out in a reproducible order.
int vrt_synth_terrain(vrt_grid *gh, uint64_t seed) {
    if (!gh) return VRT_E_INVALID_ARG;
    vrt::BrickGrid *g = reinterpret_cast<vrt::BrickGrid *>(gh);
    const vrt_grid_state &d = g->deviceState();
    const uint32_t nx = d.voxel_dim_x, ny = d.voxel_dim_y, nz = d.voxel_dim_z;
    const uint32_t s = (uint32_t)(seed ^ (seed >> 32));
    const uint32_t ocean = (uint32_t)(((uint64_t)ny * 20u) / 256u);
    const double half = (double)ny * 0.5;
    for (uint32_t x = 0; x < nx; x++) {
        for (uint32_t z = 0; z < nz; z++) {
            const double u = (double)x * 4.0 / (double)nx, w = (double)z * 4.0 / (double)nz;
            const double n = 0.7 * value_noise2(u, w, s) + 0.3 * value_noise2(u * 4.0, w * 4.0, s ^ 0x5bd1e995U);
            double hd = (double)ny / 32.0 + (15.0 * (double)ny / 32.0) * (n * n); // low basins fall below the ocean level
            if (hd > half) hd = half;
            const uint32_t height = (uint32_t)hd;
            uint32_t y = height / 2u;
            for (; y < height; y++) {
                const uint32_t hsh = hash3(x, y, z, s ^ 0x27d4eb2fU);
                const double lerp = 1.0 + (3.4 - 1.0) * ((double)y / half);
                int band = (int)std::floor(lerp + unit(hsh) * 0.5);
                if (band < 1) band = 1;
                if (band > 3) band = 3;
                uint8_t m = (uint8_t)(1 + 2 * (band - 1) + (int)((hsh >> 3) & 1u));
                if (band == 3 && (hsh % 61u) == 0u) m = 7; // a little iron in the rock
                const int rc = g->insertUnlocked(x, y, z, m);
                if (rc != VRT_OK) return rc;
            }
            for (; y < ocean; y++) {
                const int rc = g->insertUnlocked(x, y, z, 0);
                if (rc != VRT_OK) return rc;
            }
        }
    }
    return VRT_OK;
}

// Sparse scene: the volume is cut into 32^3-voxel blocks; a block holds a solid
// sphere (radius 6..14 voxels, hashed) with probability p.  Only touched bricks
// get slots, so brick_alloc can be far below the brick count.
int vrt_synth_sparse(vrt_grid *gh, uint64_t seed, float p) {
    if (!gh || !(p >= 0.0f) || p > 1.0f) return VRT_E_INVALID_ARG;
    vrt::BrickGrid *g = reinterpret_cast<vrt::BrickGrid *>(gh);
    const vrt_grid_state &d = g->deviceState();
    const uint32_t nx = d.voxel_dim_x, ny = d.voxel_dim_y, nz = d.voxel_dim_z;
    const uint32_t s = (uint32_t)(seed ^ (seed >> 32));
    const uint32_t B = 32;
    const uint32_t thresh = (uint32_t)((double)p * 16777216.0);
    for (uint32_t bx = 0; bx * B < nx; bx++)
        for (uint32_t bz = 0; bz * B < nz; bz++)
            for (uint32_t by = 0; by * B < ny; by++) {
                const uint32_t h = hash3(bx, by, bz, s);
                if ((h >> 8) >= thresh) continue;
                const uint32_t h2 = mix32(h ^ 0x68bc21ebU);
                const int rad = 6 + (int)(h2 % 9u);
                const int cx = (int)(bx * B) + 16, cy = (int)(by * B) + 16, cz = (int)(bz * B) + 16;
                const uint8_t m = (uint8_t)(1u + (h2 >> 8) % 7u);
                for (int x = cx - rad; x <= cx + rad; x++)
                    for (int z = cz - rad; z <= cz + rad; z++)
                        for (int y = cy - rad; y <= cy + rad; y++) {
                            if (x < 0 || y < 0 || z < 0 || x >= (int)nx || y >= (int)ny || z >= (int)nz) continue;
                            const int ddx = x - cx, ddy = y - cy, ddz = z - cz;
                            if (ddx * ddx + ddy * ddy + ddz * ddz > rad * rad) continue;
                            const int rc = g->insertUnlocked((uint64_t)x, (uint64_t)y, (uint64_t)z, m);
                            if (rc != VRT_OK) return rc;
                        }
            }
    return VRT_OK;
}

// ---- BrickGrid C view -------------------------------------------------------
int vrt_grid_create(uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, const vrt_grid_config *cfg, vrt_grid **out) {
    if (!out) return VRT_E_INVALID_ARG;
    vrt::GridConfig c;
    if (cfg) {
        c.brick_alloc = cfg->brick_alloc;
        c.base_t = cfg->base_t;
        c.min_point[0] = cfg->min_point[0]; c.min_point[1] = cfg->min_point[1]; c.min_point[2] = cfg->min_point[2];
        c.scale = cfg->scale;
        c.brick_dimension = cfg->brick_dimension ? cfg->brick_dimension : 4u;
    }
    vrt::BrickGrid *g = nullptr;
    const int rc = vrt::BrickGrid::create(dim_x, dim_y, dim_z, c, &g);
    *out = reinterpret_cast<vrt_grid *>(g);
    return rc;
}

void vrt_grid_destroy(vrt_grid *g) { delete reinterpret_cast<vrt::BrickGrid *>(g); }

int vrt_grid_insert(vrt_grid *g, uint64_t x, uint64_t y, uint64_t z, uint8_t material_index) {
    if (!g) return VRT_E_INVALID_ARG;
    return reinterpret_cast<vrt::BrickGrid *>(g)->insert(x, y, z, material_index);
}

int vrt_grid_insert_many(vrt_grid *gh, const uint32_t *xyz, const uint8_t *materials, uint64_t n) {
    if (!gh || (n && (!xyz || !materials))) return VRT_E_INVALID_ARG;
    vrt::BrickGrid *g = reinterpret_cast<vrt::BrickGrid *>(gh);
    for (uint64_t i = 0; i < n; i++) {
        const int rc = g->insertUnlocked(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], materials[i]);
        if (rc != VRT_OK) return rc;
    }
    return VRT_OK;
}

const vrt_grid_state *vrt_grid_device_state(const vrt_grid *g) {
    return g ? &reinterpret_cast<const vrt::BrickGrid *>(g)->deviceState() : nullptr;
}

const void *vrt_grid_data(const vrt_grid *g, vrt_buffer_id id, uint64_t *nbytes) {
    if (!g) {
        if (nbytes) *nbytes = 0;
        return nullptr;
    }
    return reinterpret_cast<const vrt::BrickGrid *>(g)->dataFor(id, nbytes);
}

uint32_t vrt_grid_active_bricks(const vrt_grid *g) { return g ? reinterpret_cast<const vrt::BrickGrid *>(g)->activeBricks() : 0u; }
uint32_t vrt_grid_brick_dimension(const vrt_grid *g) { return g ? reinterpret_cast<const vrt::BrickGrid *>(g)->brickDimension() : 0u; }

int vrt_grid_delta(const vrt_grid *gh, vrt_buffer_id id, uint64_t *from, uint64_t *to) {
    if (!gh) return 0;
    vrt::BrickGrid *g = const_cast<vrt::BrickGrid *>(reinterpret_cast<const vrt::BrickGrid *>(gh));
    vrt::DeviceDataDelta *d = g->deltaFor(id);
    if (!d) return 0;
    std::lock_guard<std::mutex> lk(d->mutex);
    if (from) *from = d->from;
    if (to) *to = d->to;
    return d->state == vrt::DeviceDataDelta::DeltaState::active ? 1 : 0;
}

void vrt_grid_reset_delta(vrt_grid *gh, vrt_buffer_id id) {
    if (!gh) return;
    vrt::DeviceDataDelta *d = reinterpret_cast<vrt::BrickGrid *>(gh)->deltaFor(id);
    if (!d) return;
    std::lock_guard<std::mutex> lk(d->mutex);
    d->resetDelta();
}

} // extern "C"
