// vrt_math.h — arithmetic vocabulary of the gfx950 traversal kernels.
//
// ARITHMETIC CONTRACT (round 3): the kernels compute what the reference's own shader computes when it runs — under Mesa's
// gallivm / llvmpipe, the back end of lavapipe, the only implementation of the reference that can be executed next to this code.
// One IEEE-754 binary32 operation per source operation (this directory is compiled with -ffp-contract=off, without fast-math,
// IEEE divide / sqrt, denormals on), and the GLSL built-ins whose lowering the language leaves open lowered as Mesa 23.2.1 lowers
// them (measured on Mesa itself, tests/test_ref_gl.py; the parity oracle applies the same rules):
//   fma(a,b,c)    = a*b + c                          (two roundings: nir lower_ffma32)
//   dot(a,b)      = (a.z*b.z + a.y*b.y) + a.x*b.x    (nir lower_fdot, reduction from the last channel)
//   hash12        = the dot with its first two terms factored (nir_opt_algebraic), the jitter's constant folded
//   normalize(v)  = v * (1 / sqrt(dot(v,v)))         (IEEE divide and sqrt)
//   fract(x)      = x - floor(x)
//   reflect(I,N)  = I - (2*dot(N,I))*N
//   sin(x)        = vrt_sin (the Cephes single-precision kernel as gallivm lowers it, its multiply-adds fused as on an FMA host)
//   int(x)        = (int)clamp(x, -2^31, 2147483520)  (f2i_clamp)
// With these rules the frames are the reference shader's BIT FOR BIT — bounces, soft sun and every scatter function included
// (tests/test_reference_parity_gpu.py, tests/golden/ref/, tests/golden/ref_full/, bench.py's parity_vs_reference) — at no measurable cost
// (same-box A/B of the two lowerings: 0.059 / 0.087 / 0.088 ms either way on the headline).
//
// -DVRT_LOWERING_FUSED (make fused -> libvrt_hip_fused.so, test infrastructure): fma fused, dot as an fma chain — what a GPU
// driver's compiler would typically emit for the same GLSL, and this repo's contract until round 3.  Against the reference's frames
// it differs in the last bits everywhere and, where a last bit flips a DDA tie or feeds the sin-hash RNG, in whole pixels.
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
#ifndef VRT_LOWERING_FUSED
VRT_DI float gl_fma(float a, float b, float c) { return a * b + c; } // (-ffp-contract=off: never re-fused)
#else
VRT_DI float gl_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
#endif
VRT_DI f3 fma3(f3 a, f3 b, f3 c) { return f3{gl_fma(a.x, b.x, c.x), gl_fma(a.y, b.y, c.y), gl_fma(a.z, b.z, c.z)}; }
VRT_DI f3 floor3(f3 a) { return f3{__builtin_floorf(a.x), __builtin_floorf(a.y), __builtin_floorf(a.z)}; }
VRT_DI f3 abs3(f3 a) { return f3{__builtin_fabsf(a.x), __builtin_fabsf(a.y), __builtin_fabsf(a.z)}; }
#ifndef VRT_LOWERING_FUSED
VRT_DI float dot3(f3 a, f3 b) { return (a.z * b.z + a.y * b.y) + a.x * b.x; }
VRT_DI float dot2(float ax, float ay, float bx, float by) { return ay * by + ax * bx; }
#else
VRT_DI float dot3(f3 a, f3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
VRT_DI float dot2(float ax, float ay, float bx, float by) { return __builtin_fmaf(ay, by, ax * bx); }
#endif
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

// sin by specification (same operation sequence as the parity oracle's restatement): the Cephes /
// sse_mathfun single-precision kernel as Mesa's gallivm lowers GLSL sin — what the reference shader computes under
// llvmpipe — with every multiply-add fused.  ~30 binary32 operations (round 1's binary64 evaluation was ~75 at half rate:
// 12-15 % of the cycles of a wave that shades hits).
VRT_DI float vrt_sin(float a) {
    const uint32_t ai = __builtin_bit_cast(uint32_t, a);
    float x = __builtin_bit_cast(float, ai & 0x7fffffffu);
    const bool finite = x < __builtin_inff();
    const float scale_y = x * 1.27323954473516f;
    // float -> int32 by truncation; 0x80000000 beyond the int range (the x86 conversion the reference run uses)
    const int emm2_i = (scale_y < 2147483648.0f) ? (int)scale_y : (int)0x80000000;
    const uint32_t emm2_add = (uint32_t)emm2_i + 1u;
    const uint32_t emm2_and = emm2_add & ~1u;
    const float y = (float)(int)emm2_and;
    const uint32_t sign_bit = (ai ^ (emm2_add << 29)) & 0x80000000u;
    const bool use_sin_poly = (emm2_and & 2u) == 0u;
    x = __builtin_fmaf(y, -0.78515625f, x);
    x = __builtin_fmaf(y, -2.4187564849853515625e-4f, x);
    x = __builtin_fmaf(y, -3.77489497744594108e-8f, x);
    const float z = x * x;
    float yc = __builtin_fmaf(z, 2.443315711809948E-005f, -1.388731625493765E-003f);
    yc = __builtin_fmaf(yc, z, 4.166664568298827E-002f);
    yc = yc * z;
    yc = yc * z;
    yc = yc - z * 0.5f;
    yc = yc + 1.0f;
    float ys = __builtin_fmaf(z, -1.9515295891E-4f, 8.3321608736E-3f);
    ys = __builtin_fmaf(ys, z, -1.6666654611E-1f);
    ys = ys * z;
    ys = __builtin_fmaf(ys, x, x);
    float r = use_sin_poly ? ys : yc;
    r = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, r) ^ sign_bit);
    r = (r < -1.0f) ? -1.0f : r;
    r = (r > 1.0f) ? 1.0f : r;
    return finite ? r : __builtin_nanf("");
}

// ---- rand.comp (assets/shaders/rand.comp:3-26) -----------------------------
VRT_DI float rand_1(float co) { return fract1(vrt_sin(co * 91.3458f) * 47453.5453f); }
VRT_DI float rand_2(float cx, float cy) {
    const float d = dot2(cx, cy, 12.9898f, 78.233f);
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
#ifndef VRT_LOWERING_FUSED
    // p3.z == p3.x (p.xyx), so the dot is A*(B+k) + B*(A+k) + A*(A+k); Mesa factors a*b + a*c -> a*(b+c) out of the first two
    // terms of its reduction (the parity oracle states the same rule)
    const float d = (p3.x + p3.y) * (p3.x + 33.33f) + p3.x * (p3.y + 33.33f);
#else
    const f3 q = f3{p3.y + 33.33f, p3.z + 33.33f, p3.x + 33.33f};
    const float d = dot3(p3, q);
#endif
    p3 = f3{p3.x + d, p3.y + d, p3.z + d};
    return fract1((p3.x + p3.y) * p3.z);
}
// comp:167,169: hash12(vec2(ax, ay) * 0.2 * float(sample_i > 0))
VRT_DI float hash_12_jitter(float ax, float ay, float flag) {
#ifndef VRT_LOWERING_FUSED
    // Mesa folds ((a * 0.2) * flag) * .1031 of the inlined hash12 into a * (0.2 * .1031) for flag == 1 (the product is 0 for
    // flag == 0) and factors the dot as above (the parity oracle states the same rule)
    if (flag == 0.0f) return hash_12(0.0f, 0.0f);
    const float k = 0.2f * .1031f;
    const float A = fract1(ax * k), B = fract1(ay * k);
    const float d = (A + B) * (A + 33.33f) + A * (B + 33.33f);
    return fract1(((A + d) + (B + d)) * (A + d));
#else
    return hash_12((ax * 0.2f) * flag, (ay * 0.2f) * flag);
#endif
}

} // namespace vrt
