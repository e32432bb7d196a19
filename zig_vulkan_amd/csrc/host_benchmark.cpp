// host_benchmark.cpp — the reference's scripted benchmark fly-through.
// Mirror of src/modules/voxel_rt/Benchmark.zig: 60 s camera path through 11 points / 11 orientations,
// linear interpolation of position and of the (yaw) quaternion, min / max / avg frame time report.
//
// The quaternion and vector helpers come from zalgebra (build.zig.zon pins
// 7cf3b90edc28a138d666deab5dfde9dce89dff56; not vendored in the reference tree, no network here), so
// their published algorithms are restated below: Quat.fromAxis / fromEulerAngles / mul / lerp / norm /
// rotateVec and Vec3.lerp, with the call sites Benchmark.zig:22-74,146-172 and Camera.zig:154-180.
#include <cmath>
#include <cstring>
#include <new>
#include "../../include/vrt_hip.h"

namespace {

struct V3 {
    float x, y, z;
};
struct Quat {
    float w, x, y, z;
};

inline V3 vscale(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
inline V3 vadd(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 vcross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 vnorm(V3 a) {
    const float l = std::sqrt(vdot(a, a));
    return l == 0.0f ? a : V3{a.x / l, a.y / l, a.z / l};
}
inline float lerp1(float from, float to, float alpha) { return (to - from) * alpha + from; } // zalgebra root.lerp
inline V3 vlerp(V3 l, V3 r, float t) { return V3{lerp1(l.x, r.x, t), lerp1(l.y, r.y, t), lerp1(l.z, r.z, t)}; }

inline Quat q_identity() { return Quat{1.0f, 0.0f, 0.0f, 0.0f}; }
// Quat.fromAxis(degrees, axis)
inline Quat q_from_axis(float degrees, V3 axis) {
    const float radians = degrees * (3.14159265358979323846f / 180.0f);
    const float rot_sin = std::sin(radians / 2.0f);
    const V3 a = vscale(vnorm(axis), rot_sin);
    return Quat{std::cos(radians / 2.0f), a.x, a.y, a.z};
}
// Quat.mul(left, right)
inline Quat q_mul(Quat l, Quat r) {
    return Quat{(-l.x * r.x) - (l.y * r.y) - (l.z * r.z) + (l.w * r.w), (l.x * r.w) + (l.y * r.z) - (l.z * r.y) + (l.w * r.x),
                (-l.x * r.z) + (l.y * r.w) + (l.z * r.x) + (l.w * r.y), (l.x * r.y) - (l.y * r.x) + (l.z * r.w) + (l.w * r.z)};
}
// Quat.fromEulerAngles(degrees xyz): z * (y * x) with axes right, up, forward
inline Quat q_from_euler(float ex, float ey, float ez) {
    const Quat x = q_from_axis(ex, V3{1, 0, 0}), y = q_from_axis(ey, V3{0, 1, 0}), z = q_from_axis(ez, V3{0, 0, 1});
    return q_mul(z, q_mul(y, x));
}
inline Quat q_lerp(Quat l, Quat r, float t) { return Quat{lerp1(l.w, r.w, t), lerp1(l.x, r.x, t), lerp1(l.y, r.y, t), lerp1(l.z, r.z, t)}; }
inline Quat q_norm(Quat q) {
    const float l = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    return l == 0.0f ? q : Quat{q.w / l, q.x / l, q.y / l, q.z / l};
}
// Quat.rotateVec(self, v)
inline V3 q_rotate(Quat self, V3 v) {
    const Quat q = q_norm(self);
    const V3 b{q.x, q.y, q.z};
    const float b2 = vdot(b, b);
    return vadd(vadd(vscale(v, q.w * q.w - b2), vscale(b, vdot(v, b) * 2.0f)), vscale(vcross(b, v), q.w * 2.0f));
}

// Benchmark.Configuration, Benchmark.zig:141-172
constexpr float kDuration = 60.0f;
constexpr int kPoints = 11;
const V3 kPathPoints[kPoints] = {{0, 0, 0},      {2, 5, 0},      {3, 5, 5},      {5, 2, 1},     {10, 0, 10}, {20, -20, 20},
                                 {10, -25, 15}, {10, -22, 20}, {10, -30, 25}, {5, -10, 10}, {0, 13, 0}};
const float kPathEuler[kPoints][3] = {{0, 0, 0},   {0, 45, 0},  {10, -20, 0}, {20, 180, 0}, {50, 90, 0}, {60, 0, 0},
                                      {80, -10, 0}, {75, -40, 0}, {80, -10, 0}, {80, -90, 0}, {0, -145, 0}};

} // namespace

struct vrt_benchmark {
    float timer = 0.0f;
    float path_point_fraction, path_orientation_fraction;
    float viewport_width, viewport_height;
    Quat yaw = q_identity(), pitch = q_identity();
    // Report, Benchmark.zig:80-107
    float min_delta_time = 3.4028235e38f, max_delta_time = 0.0f, delta_time_sum = 0.0f;
    uint32_t delta_time_sum_samples = 0;
};

namespace {
// Camera.propogatePitchChange + lowerLeftCorner with orientation = (yaw * pitch).norm(), Camera.zig:154-180
void propagate(const vrt_benchmark *b, vrt_camera_device *cam) {
    const V3 forward = q_rotate(q_norm(q_mul(b->yaw, b->pitch)), V3{0, 0, 1});
    const V3 right = vnorm(vcross(V3{0, 1, 0}, forward));
    const V3 up = vnorm(vcross(forward, right));
    const V3 h = vscale(right, b->viewport_width), v = vscale(up, b->viewport_height);
    cam->horizontal[0] = h.x; cam->horizontal[1] = h.y; cam->horizontal[2] = h.z;
    cam->vertical[0] = v.x; cam->vertical[1] = v.y; cam->vertical[2] = v.z;
    cam->lower_left_corner[0] = cam->origin[0] - h.x * 0.5f - v.x * 0.5f - forward.x;
    cam->lower_left_corner[1] = cam->origin[1] - h.y * 0.5f - v.y * 0.5f - forward.y;
    cam->lower_left_corner[2] = cam->origin[2] - h.z * 0.5f - v.z * 0.5f - forward.z;
}
} // namespace

extern "C" {

// Benchmark.init, Benchmark.zig:22-44.  `cam` must come from vrt_camera_init (image size, spp, bounces).
int vrt_benchmark_create(vrt_camera_device *cam, float vertical_fov_deg, float viewport_height_cfg, vrt_benchmark **out) {
    if (!cam || !out || cam->image_width == 0 || cam->image_height == 0) return VRT_E_INVALID_ARG;
    vrt_benchmark *b = new (std::nothrow) vrt_benchmark();
    if (!b) return VRT_E_OOM;
    b->path_point_fraction = kDuration / (float)kPoints;
    b->path_orientation_fraction = kDuration / (float)kPoints;
    const float aspect = (float)cam->image_width / (float)cam->image_height; // Camera.zig:37-45
    b->viewport_height = viewport_height_cfg * std::tan(vertical_fov_deg * (3.14159265358979323846f / 180.0f) * 0.5f);
    b->viewport_width = aspect * b->viewport_height;
    cam->origin[0] = kPathPoints[0].x; cam->origin[1] = kPathPoints[0].y; cam->origin[2] = kPathPoints[0].z;
    b->yaw = q_from_euler(kPathEuler[0][0], kPathEuler[0][1], kPathEuler[0][2]); // "use yaw quat as orientation and ignore pitch"
    b->pitch = q_identity();
    propagate(b, cam);
    *out = b;
    return VRT_OK;
}

void vrt_benchmark_destroy(vrt_benchmark *b) { delete b; }

// Benchmark.update, Benchmark.zig:47-74; returns 1 once the 60 s path is complete, 0 before, <0 on error
int vrt_benchmark_update(vrt_benchmark *b, float dt, vrt_camera_device *cam) {
    if (!b || !cam) return VRT_E_INVALID_ARG;
    b->timer += dt;
    const int pi = (int)std::floor(b->timer / b->path_point_fraction);
    if (pi >= 0 && pi < kPoints - 1) {
        const float t = std::fmod(b->timer, b->path_point_fraction) / b->path_point_fraction;
        const V3 o = vlerp(kPathPoints[pi], kPathPoints[pi + 1], t);
        cam->origin[0] = o.x; cam->origin[1] = o.y; cam->origin[2] = o.z;
    }
    const int oi = (int)std::floor(b->timer / b->path_orientation_fraction);
    if (oi >= 0 && oi < kPoints - 1) {
        const float t = std::fmod(b->timer, b->path_orientation_fraction) / b->path_orientation_fraction;
        const Quat l = q_from_euler(kPathEuler[oi][0], kPathEuler[oi][1], kPathEuler[oi][2]);
        const Quat r = q_from_euler(kPathEuler[oi + 1][0], kPathEuler[oi + 1][1], kPathEuler[oi + 1][2]);
        b->yaw = q_lerp(l, r, t);
        b->pitch = q_identity();
    }
    propagate(b, cam);
    if (dt < b->min_delta_time) b->min_delta_time = dt;
    if (dt > b->max_delta_time) b->max_delta_time = dt;
    b->delta_time_sum += dt;
    b->delta_time_sum_samples += 1;
    return b->timer >= kDuration ? 1 : 0;
}

// Report.print's three numbers (Benchmark.zig:109-136), in milliseconds
int vrt_benchmark_report(const vrt_benchmark *b, float *min_ms, float *max_ms, float *avg_ms) {
    if (!b || !min_ms || !max_ms || !avg_ms) return VRT_E_INVALID_ARG;
    *min_ms = b->min_delta_time * 1000.0f;
    *max_ms = b->max_delta_time * 1000.0f;
    *avg_ms = b->delta_time_sum_samples ? b->delta_time_sum / (float)b->delta_time_sum_samples * 1000.0f : 0.0f;
    return VRT_OK;
}

} // extern "C"
