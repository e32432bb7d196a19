// vrt_math.h — arithmetic vocabulary of the gfx950 traversal kernels.
//
// The kernels must produce the same binary32 results as the reference shader's
// operations evaluated one IEEE operation at a time (fma only where the shader
// writes fma), so every helper here spells out its operation order and this
// directory is compiled with -ffp-contract=off and without fast-math.
// GLSL built-ins are lowered as follows (DESIGN.md "Arithmetic contract"):
//   dot(a,b)      = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
//   normalize(v)  = v * (1 / sqrt(dot(v,v)))         (IEEE divide and sqrt)
//   fract(x)      = x - floor(x)
//   reflect(I,N)  = I - (2*dot(N,I))*N
//   sin(x)        = vrt_sin (f64 Cody-Waite reduction + fixed f64 polynomial)
//   int(x)        = (int)clamp(x, -2^31, 2147483520)  (f2i_clamp)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vrt {

struct f3 {
    float x, y, z;
};

#define VRT_DI __device__ __forceinline__

VRT_DI f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
VRT_DI f3 splat3(float s) { return f3{s, s, s}; }
VRT_DI f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
VRT_DI f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
VRT_DI f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
VRT_DI f3 operator/(f3 a, f3 b) { return f3{a.x / b.x, a.y / b.y, a.z / b.z}; }
VRT_DI f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
VRT_DI f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
VRT_DI f3 fma3(f3 a, f3 b, f3 c) { return f3{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y), __builtin_fmaf(a.z, b.z, c.z)}; }
VRT_DI f3 floor3(f3 a) { return f3{__builtin_floorf(a.x), __builtin_floorf(a.y), __builtin_floorf(a.z)}; }
VRT_DI f3 abs3(f3 a) { return f3{__builtin_fabsf(a.x), __builtin_fabsf(a.y), __builtin_fabsf(a.z)}; }
VRT_DI float dot3(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
VRT_DI f3 normalize3(f3 a) {
    const float inv = 1.0f / __builtin_sqrtf(dot3(a, a));
    return a * inv;
}
VRT_DI float sign1(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
VRT_DI float fract1(float x) { return x - __builtin_floorf(x); }
VRT_DI float gl_min(float x, float y) { return (y < x) ? y : x; }
VRT_DI float gl_max(float x, float y) { return (x < y) ? y : x; }
VRT_DI f3 reflect3(f3 I, f3 N) {
    const float k = 2.0f * dot3(N, I);
    return I - N * k;
}

// float -> int32 by clamping first (GLSL leaves out-of-range int(x) undefined): two VALU ops, no
// branches.  NaN takes the lower bound (maxNum semantics of v_max_f32 / fmaxf).
VRT_DI int f2i_clamp(float x) {
    return (int)__builtin_fminf(__builtin_fmaxf(x, -2147483648.0f), 2147483520.0f);
}

// sin by specification (same sequence as the parity oracle's restatement).
VRT_DI float vrt_sin(float xf) {
    const double x = (double)xf;
    const double kd = __builtin_rint(x * 0.63661977236758134308);
    double r = __builtin_fma(-kd, 1.57079632673412561417e+00, x);
    r = __builtin_fma(-kd, 6.07710050650619224932e-11, r);
    const long long k = (__builtin_fabs(kd) < 4611686018427387904.0) ? (long long)kd : 0ll;
    const double r2 = r * r;
    double ps = -1.0 / 6227020800.0;
    ps = __builtin_fma(ps, r2, 1.0 / 39916800.0);
    ps = __builtin_fma(ps, r2, -1.0 / 362880.0);
    ps = __builtin_fma(ps, r2, 1.0 / 5040.0);
    ps = __builtin_fma(ps, r2, -1.0 / 120.0);
    ps = __builtin_fma(ps, r2, 1.0 / 6.0);
    ps = ps * r2;
    const double s = __builtin_fma(-ps, r, r);
    double pc = 1.0 / 479001600.0;
    pc = __builtin_fma(pc, r2, -1.0 / 3628800.0);
    pc = __builtin_fma(pc, r2, 1.0 / 40320.0);
    pc = __builtin_fma(pc, r2, -1.0 / 720.0);
    pc = __builtin_fma(pc, r2, 1.0 / 24.0);
    pc = __builtin_fma(pc, r2, -0.5);
    const double c = __builtin_fma(pc, r2, 1.0);
    const int q = (int)(k & 3);
    const double res = (q == 0) ? s : ((q == 1) ? c : ((q == 2) ? -s : -c));
    return (float)res;
}

// ---- rand.comp (assets/shaders/rand.comp:3-26) -----------------------------
VRT_DI float rand_1(float co) { return fract1(vrt_sin(co * 91.3458f) * 47453.5453f); }
VRT_DI float rand_2(float cx, float cy) {
    const float d = __builtin_fmaf(cy, 78.233f, cx * 12.9898f);
    return fract1(vrt_sin(d) * 43758.5453f);
}
VRT_DI float rand_3(f3 co) {
    const float r = rand_1(co.z);
    return rand_2(co.x + r, co.y + r);
}
VRT_DI float rand_2_range(float cx, float cy, float mn, float mx) { return mn + (mx - mn) * rand_2(cx, cy); }
VRT_DI f3 rand_vec3_range(float cx, float cy, float mn, float mx) {
    const float x = rand_2_range(cx, cy, mn, mx);
    const float y = rand_2_range(cx + x, cy + x, mn, mx);
    const float z = rand_2_range(cx + y, cy + y, mn, mx);
    return f3{x, y, z};
}
VRT_DI float hash_12(float px, float py) {
    f3 p3 = f3{fract1(px * .1031f), fract1(py * .1031f), fract1(px * .1031f)};
    const f3 q = f3{p3.y + 33.33f, p3.z + 33.33f, p3.x + 33.33f};
    const float d = dot3(p3, q);
    p3 = f3{p3.x + d, p3.y + d, p3.z + d};
    return fract1((p3.x + p3.y) * p3.z);
}

} // namespace vrt
