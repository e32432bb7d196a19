/*
 * vrt_oracle.c — CPU restatement of the reference's brickmap ray tracer.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: a scalar, plain-C
 * restatement of /root/reference/assets/shaders/brick_raytracer.comp:153-596
 * and assets/shaders/rand.comp:3-26.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product (libvrt_hip.so) never
 * links, imports or calls anything in oracle/.
 *
 * PARITY PINNED against the reference's own shader: the image holds Mesa 23.2.1 (llvmpipe), so
 * brick_raytracer.comp itself is compiled by Mesa's GLSL compiler and run on the host cores
 * (oracle/_ref, recipe oracle/ref_gl/recipe.py; the reference's own build — zig + a network-fetched
 * glslang + Vulkan — cannot run here).  This file, built as it stands, reproduces the shader's frames BIT FOR BIT:
 * the committed vectors tests/golden/ref/ and tests/golden/ref_full/ and random scenes (tests/test_ref_gl.py) — primary
 * and shadow rays, soft sun, several samples, bounces, every scatter function, 4^3 and 8^3 bricks, up to the headline
 * workload's full 1920x1080 frame.  The reference itself holds no golden vectors for this path (its only tests are three
 * .vox header checks, src/modules/voxel_rt/vox/loader.zig:265-281).
 *
 * Arithmetic rules — the ones Mesa's gallivm applies to the shader (measured on Mesa itself, tests/test_ref_gl.py), and the
 * ones the HIP kernels follow (zig_vulkan_amd/csrc/vrt_math.h), so that kernel == oracle == reference shader bit for bit:
 *   - GLSL fma(a,b,c)      -> a*b + c            (two roundings: nir lower_ffma32)
 *   - every other * + - /  -> separate IEEE-754 binary32 operations in source
 *                             order; compiled with -ffp-contract=off
 *   - dot(a,b)             -> (a.z*b.z + a.y*b.y) + a.x*b.x   (nir lower_fdot, from the last channel)
 *   - hash12               -> Mesa's two algebraic rewrites, see hash12() / hash12_jitter()
 *   - normalize(v)         -> v * (1.0f / sqrtf(dot(v,v)))
 *   - fract(x)             -> x - floorf(x)
 *   - reflect(I,N)         -> I - (2*dot(N,I))*N
 *   - sin(x)               -> vrt_sinf(x): the Cephes single-precision kernel as Mesa gallivm
 *                             lowers it, specified below (GLSL.std.450 Sin precision is
 *                             implementation defined; libm/ocml sinf differ from each other)
 *   - int(x)               -> (int)clamp(x, -2^31, 2147483520) (vrt_f2i; GLSL leaves the
 *                             out-of-range conversion undefined, C makes it UB)
 *   - Rgba8 imageStore     -> rintf(clamp(c,0,1)*255)
 * -DORACLE_LOWERING_FUSED builds libvrt_oracle_fused.so: fma fused (one rounding), dot as
 * fmaf(a.z,b.z, fmaf(a.y,b.y, a.x*b.x)) — what a GPU driver's compiler typically emits for the same GLSL, this repo's
 * contract until round 3, and the counterpart of libvrt_hip_fused.so.  It agrees with the shader's frames within 1e-4 per
 * channel except at isolated pixels where a last-bit difference flips a DDA tie, and as images only where the sin-hash RNG
 * is reached (one ulp in its argument is another random number).
 * Hang guard: a DDA step along an axis whose ray_step is 0 does not move the
 * position, so a degenerate ray (NaN side distances with a zero step) would spin
 * forever in the reference's loops.  Both loops here allow such a zero-step
 * axis to be selected at most G times (G = dx+dy+dz+8 for the brick-level walk,
 * 3*b+8 for the voxel-level walk) and end the loop on the next selection.  A
 * well-formed ray never selects a zero-step axis (its side distance is 5e11),
 * so the guard is unobservable on valid input.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define MAT_LAMBERTIAN 0u
#define MAT_METAL 1u
#define MAT_DIELECTRIC 2u
#define MAT_NONE 3u

typedef struct { float x, y, z; } v3;

/* State.Device, /root/reference/src/modules/voxel_rt/brick/State.zig:60-79 */
typedef struct {
    uint32_t voxel_dim_x, voxel_dim_y, voxel_dim_z;
    uint32_t dim_x, dim_y, dim_z;
    uint32_t padding1, padding2;
    float min_point_base_t[4];
    float max_point_scale[4];
} oracle_grid_state;

/* gpu_types.Material, gpu_types.zig:16-32 */
typedef struct {
    uint32_t type;
    float albedo_r, albedo_g, albedo_b;
    float type_data;
} oracle_material;

/* push constants, brick_raytracer.comp:58-75 (Camera.zig:183-193 + Sun.zig:13-18) */
typedef struct {
    uint32_t image_width, image_height;
    uint32_t _pad0[2];
    float horizontal[3], _pad1;
    float vertical[3], _pad2;
    float lower_left_corner[3], _pad3;
    float origin[3], paddin;
    int32_t samples_per_pixel, max_bounce;
    uint32_t _pad5[2];
    float sun_position[3];
    uint32_t sun_enabled;
    float sun_color[3];
    float sun_radius;
} oracle_push;

typedef struct {
    const oracle_grid_state *grid;          /* binding 1 */
    const oracle_material *materials;       /* binding 2 */
    const uint32_t *brick_type_bits;        /* binding 3 */
    const uint32_t *brick_indices;          /* binding 4 */
    const uint8_t *brick_solid_mask;        /* binding 5 */
    const uint32_t *brick_type_and_index;   /* binding 6 */
    const uint8_t *material_indices;        /* binding 7 */
    /* specialization constants, Pipeline.zig:293-315 / comp:53-56 */
    uint32_t brick_bytes;
    int32_t brick_dimensions;
    float brick_voxel_scale;
} oracle_scene;

typedef struct {
    uint64_t rays, status_loads, bricks_entered, voxel_steps, hits, grid_steps;
} oracle_counters;

typedef struct {
    v3 origin, direction;
    float internal_reflection;
    uint32_t ignore_type_material;
} Ray;

typedef struct {
    v3 point, normal;
    float t;
    uint32_t index;
} HitRecord;

typedef struct {
    const oracle_scene *s;
    const oracle_push *pc;
    oracle_counters *c;
} Env;

/* ------------------------------------------------------------------ lowering of the GLSL built-ins
 * Default: exactly as Mesa 23.2.1 llvmpipe lowers them (measured through oracle/_ref, tests/test_ref_gl.py):
 *   fma(a,b,c) -> a*b + c with two roundings (nir lower_ffma32), dot(a,b) -> (a.z*b.z + a.y*b.y) + a.x*b.x
 *   (nir lower_fdot, reduction from the last channel); sin is gallivm's in both builds.
 * -DORACLE_LOWERING_FUSED: fused fma, dot as an fma chain (libvrt_oracle_fused.so). */
#ifndef ORACLE_LOWERING_FUSED
#define FMA(a, b, c) ((a) * (b) + (c))
#define DOT3(a, b) (((a).z * (b).z + (a).y * (b).y) + (a).x * (b).x)
#define DOT2(ax, ay, bx, by) ((ay) * (by) + (ax) * (bx))
#else
#define FMA(a, b, c) fmaf((a), (b), (c))
#define DOT3(a, b) fmaf((a).z, (b).z, fmaf((a).y, (b).y, (a).x * (b).x))
#define DOT2(ax, ay, bx, by) fmaf((ay), (by), (ax) * (bx))
#endif

/* ------------------------------------------------------------------ helpers */
static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3s(float s) { return V3(s, s, s); }
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vdiv(v3 a, v3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 vscale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 vneg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline v3 vfma(v3 a, v3 b, v3 c) { return V3(FMA(a.x, b.x, c.x), FMA(a.y, b.y, c.y), FMA(a.z, b.z, c.z)); }
static inline v3 vfloor(v3 a) { return V3(floorf(a.x), floorf(a.y), floorf(a.z)); }
static inline v3 vabs(v3 a) { return V3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
static inline float vdot(v3 a, v3 b) { return DOT3(a, b); }
static inline v3 vnormalize(v3 a) { float inv = 1.0f / sqrtf(vdot(a, a)); return vscale(a, inv); }
static inline float fsign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
static inline float fract(float x) { return x - floorf(x); }
static inline float vidx(v3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
/* GLSL min/max(x,y): "y < x ? y : x" / "x < y ? y : x" */
static inline float gmin(float x, float y) { return (y < x) ? y : x; }
static inline float gmax(float x, float y) { return (x < y) ? y : x; }

/* float -> int32 by clamping first (GLSL leaves out-of-range int(x) undefined); NaN takes the
 * lower bound (fmaxf returns the non-NaN operand). */
static inline int32_t vrt_f2i(float x) {
    return (int32_t)fminf(fmaxf(x, -2147483648.0f), 2147483520.0f);
}

/* sin(x) by specification — GLSL.std.450 Sin has implementation-defined precision, and libm, ocml and every GPU differ.
 * The specified algorithm is the Cephes / sse_mathfun single-precision kernel exactly as Mesa's gallivm lowers GLSL sin
 * (src/gallium/auxiliary/gallivm/lp_bld_arit.c, lp_build_sin_or_cos), i.e. what the reference shader computes when it
 * runs under llvmpipe: |x| scaled by 4/pi, j = (int(y) + 1) & ~1, three-constant Cody-Waite reduction, one of two
 * polynomials by bit 1 of j, sign from x's sign xor bit 2 of j, clamp to [-1, 1], NaN for non-finite input; every
 * multiply-add of it fused (how llvm.fmuladd comes out on an FMA host).  About 30 binary32 operations; absolute error
 * <= 1.2e-7 for |x| <= 8192 (tests/test_oracle_kat.py).  Round 1 specified a binary64 evaluation rounded once: three times
 * the instructions at half rate on the GPU, 12-15 % of the cycles of a wave that shades hits.
 * Identical operation sequence in zig_vulkan_amd/csrc/vrt_math.h; bit-identical to llvmpipe's sin (tests/test_ref_gl.py). */
static inline float vrt_sinf(float a) {
    uint32_t ai; memcpy(&ai, &a, 4);
    uint32_t absi = ai & 0x7fffffffu;
    float x; memcpy(&x, &absi, 4);
    if (!(x < INFINITY)) return NAN;
    const float scale_y = x * 1.27323954473516f;
    /* cvttps2dq: out-of-range -> 0x80000000 */
    const int32_t emm2_i = (scale_y < 2147483648.0f) ? (int32_t)scale_y : INT32_MIN;
    const uint32_t emm2_add = (uint32_t)emm2_i + 1u;
    const uint32_t emm2_and = emm2_add & ~1u;
    const float y = (float)(int32_t)emm2_and;
    const uint32_t sign_bit = (ai ^ (emm2_add << 29)) & 0x80000000u;
    const int use_sin_poly = (emm2_and & 2u) == 0;
    x = fmaf(y, -0.78515625f, x);
    x = fmaf(y, -2.4187564849853515625e-4f, x);
    x = fmaf(y, -3.77489497744594108e-8f, x);
    const float z = x * x;
    float yc = fmaf(z, 2.443315711809948E-005f, -1.388731625493765E-003f);
    yc = fmaf(yc, z, 4.166664568298827E-002f);
    yc = yc * z;
    yc = yc * z;
    yc = yc - z * 0.5f;
    yc = yc + 1.0f;
    float ys = fmaf(z, -1.9515295891E-4f, 8.3321608736E-3f);
    ys = fmaf(ys, z, -1.6666654611E-1f);
    ys = ys * z;
    ys = fmaf(ys, x, x);
    float r = use_sin_poly ? ys : yc;
    uint32_t ri; memcpy(&ri, &r, 4);
    ri ^= sign_bit;
    memcpy(&r, &ri, 4);
    r = (r < -1.0f) ? -1.0f : r;
    r = (r > 1.0f) ? 1.0f : r;
    return r;
}

/* ---------------------------------------------------------------- rand.comp */
/* rand.comp:3 */
static inline float Rand1(float co) { return fract(vrt_sinf(co * 91.3458f) * 47453.5453f); }
/* rand.comp:4 */
static inline float Rand2(float cx, float cy) {
    const float d = DOT2(cx, cy, 12.9898f, 78.233f);
    return fract(vrt_sinf(d) * 43758.5453f);
}
/* rand.comp:5 */
static inline float Rand3(v3 co) { const float r = Rand1(co.z); return Rand2(co.x + r, co.y + r); }
/* rand.comp:6-8 */
static inline float Rand2mm(float cx, float cy, float mn, float mx) { return mn + (mx - mn) * Rand2(cx, cy); }
/* rand.comp:15-20 */
static inline v3 RandVec3mm(float cx, float cy, float mn, float mx) {
    const float x = Rand2mm(cx, cy, mn, mx);
    const float y = Rand2mm(cx + x, cy + x, mn, mx);
    const float z = Rand2mm(cx + y, cy + y, mn, mx);
    return V3(x, y, z);
}
/* rand.comp:22-26 */
static inline float hash12(float px, float py) {
    v3 p3 = V3(fract(px * .1031f), fract(py * .1031f), fract(px * .1031f));
    const v3 q = V3(p3.y + 33.33f, p3.z + 33.33f, p3.x + 33.33f);
#ifndef ORACLE_LOWERING_FUSED
    /* p3.z == p3.x (p.xyx), so the dot is A*(B+k) + B*(A+k) + A*(A+k); Mesa's nir_opt_algebraic factors the
     * inexact a*b + a*c -> a*(b+c) out of the first two terms of its reduction (measured, tests/test_ref_gl.py) */
    (void)q;
    const float d = (p3.x + p3.y) * (p3.x + 33.33f) + p3.x * (p3.y + 33.33f);
#else
    const float d = vdot(p3, q);
#endif
    p3 = V3(p3.x + d, p3.y + d, p3.z + d);
    return fract((p3.x + p3.y) * p3.z);
}

/* comp:167,169: hash12(vec2(ax, ay) * 0.2 * float(sample_i > 0)) */
static inline float hash12_jitter(float ax, float ay, float flag) {
#ifndef ORACLE_LOWERING_FUSED
    /* Mesa folds ((a * 0.2) * flag) * .1031 of the inlined hash12 into a * (0.2 * .1031) (flag == 1; the product is 0
     * for flag == 0) and factors the dot as in hash12 above (measured, tests/test_ref_gl.py) */
    if (flag == 0.0f) return hash12(0.0f, 0.0f);
    const float k = 0.2f * .1031f;
    const float A = fract(ax * k), B = fract(ay * k);
    const float d = (A + B) * (A + 33.33f) + A * (B + 33.33f);
    return fract(((A + d) + (B + d)) * (A + d));
#else
    return hash12((ax * 0.2f) * flag, (ay * 0.2f) * flag);
#endif
}

/* exported KAT hooks */
float oracle_sinf(float x) { return vrt_sinf(x); }
float oracle_hash12(float px, float py) { return hash12(px, py); }
float oracle_rand2(float cx, float cy) { return Rand2(cx, cy); }
float oracle_rand3(float x, float y, float z) { return Rand3(V3(x, y, z)); }
void oracle_randvec3(float cx, float cy, float mn, float mx, float out[3]) {
    const v3 r = RandVec3mm(cx, cy, mn, mx);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* -------------------------------------------------------- brick_raytracer.comp */
/* comp:180-184 */
static inline Ray CreateRay(v3 origin, v3 direction) {
    Ray r; r.origin = origin; r.direction = vnormalize(direction);
    r.internal_reflection = 1.0f; r.ignore_type_material = MAT_NONE; return r;
}
/* comp:186-190 */
static inline Ray CreateShadowRay(const Env *e, v3 origin, v3 direction) {
    Ray r; r.origin = origin; r.direction = vnormalize(direction);
    r.internal_reflection = 1.0f;
    r.ignore_type_material = (e->pc->sun_enabled > 0) ? MAT_NONE : MAT_DIELECTRIC; return r;
}
/* comp:192-195 */
static inline v3 RayAt(const Ray *r, float t) { return vfma(v3s(t), r->direction, r->origin); }
/* comp:197-201 */
static inline v3 BackgroundColor(const Ray *r) {
    const float t = 0.5f * (r->direction.y + 1.0f);
    return vfma(v3s(1.0f - t), v3s(1.0f), vscale(V3(0.5f, 0.7f, 1.0f), t));
}
/* comp:267-268 */
static inline float safeInverse(float x) { return (x == 0.0f) ? 1e12f : (1.0f / x); }

/* comp:501-503 */
static inline int indexOfMaxComponent(v3 v) {
    return (int)(v.y > v.x && v.y > v.z) + (int)(v.z > v.x && v.z > v.y) * 2;
}

/* comp:522-536 */
static int AdvNormIntersect(v3 bmin, v3 bmax, const Ray *r, v3 inv, v3 *normal, float *t_min, float *t_max) {
    const v3 t_lower = vmul(vsub(bmin, r->origin), inv);
    const v3 t_upper = vmul(vsub(bmax, r->origin), inv);
    const v3 t_mins = V3(gmin(t_lower.x, t_upper.x), gmin(t_lower.y, t_upper.y), gmin(t_lower.z, t_upper.z));
    const v3 t_maxes = V3(gmax(t_lower.x, t_upper.x), gmax(t_lower.y, t_upper.y), gmax(t_lower.z, t_upper.z));
    const int i = indexOfMaxComponent(t_mins);
    *normal = V3(0, 0, 0);
    const float sg = fsign(vidx(inv, i));
    if (i == 0) normal->x = sg; else if (i == 1) normal->y = sg; else normal->z = sg;
    *t_min = gmax(*t_min, vidx(t_mins, i));
    *t_max = gmin(*t_max, gmin(gmin(t_maxes.x, t_maxes.y), t_maxes.z));
    return *t_min <= *t_max;
}

int oracle_adv_norm_intersect(const float bmin[3], const float bmax[3], const float origin[3], const float dir[3],
                              float normal[3], float *t_min, float *t_max) {
    Ray r; r.origin = V3(origin[0], origin[1], origin[2]); r.direction = V3(dir[0], dir[1], dir[2]);
    v3 inv = V3(safeInverse(dir[0]), safeInverse(dir[1]), safeInverse(dir[2]));
    v3 n;
    int ok = AdvNormIntersect(V3(bmin[0], bmin[1], bmin[2]), V3(bmax[0], bmax[1], bmax[2]), &r, inv, &n, t_min, t_max);
    normal[0] = n.x; normal[1] = n.y; normal[2] = n.z;
    return ok;
}

/* comp:378-471 */
static int BrickHit(const Env *e, const Ray *r, float t_min, float t_max, v3 ray_delta, const int ray_step[3],
                    v3 g_scale, uint32_t brick_index, v3 *brick_position, HitRecord *hit) {
    const oracle_scene *s = e->s;
    const int bd = s->brick_dimensions;
    const v3 voxel_scale = vmul(g_scale, v3s(s->brick_voxel_scale));
    const uint32_t solid_mask_base_index = brick_index * s->brick_bytes;
    const v3 fstep = V3((float)ray_step[0], (float)ray_step[1], (float)ray_step[2]);

    const v3 fposition = vdiv(vsub(RayAt(r, hit->t), *brick_position), voxel_scale);
    const v3 intersection_delta = vsub(vfloor(fposition), fposition);
    v3 side_dist = vmul(vfma(fstep, intersection_delta, vadd(vscale(fstep, 0.5f), v3s(0.5f))), ray_delta);

    const v3 normal_axis = V3(ray_step[0] < 0 ? 1.f : -1.f, ray_step[1] < 0 ? 1.f : -1.f, ray_step[2] < 0 ? 1.f : -1.f);

    int lx = vrt_f2i(floorf(fposition.x + 0.f)), ly = vrt_f2i(floorf(fposition.y + 0.f)), lz = vrt_f2i(floorf(fposition.z + 0.f));
    const float local_t_max = t_max - hit->t;
    (void)t_min; /* local_t_min is computed but never read in the shader (comp:404) */
    float t_value = 0;
    int zero_budget[3] = {3 * bd + 8, 3 * bd + 8, 3 * bd + 8}; /* hang guard, see header */
    while (lx >= 0 && ly >= 0 && lz >= 0 && lx < bd && ly < bd && lz < bd && t_value <= local_t_max) {
        const int voxel_index = lx + bd * (lz + bd * ly);
        const uint8_t mask_index = (uint8_t)(voxel_index / 8);
        const uint8_t mask_offset = (uint8_t)(voxel_index % 8);
        const uint8_t entry = s->brick_solid_mask[solid_mask_base_index + mask_index];
        if (e->c) e->c->voxel_steps++;
        if ((entry >> mask_offset) & 1u) {
            const uint32_t brick_material_index = s->brick_type_and_index[brick_index] & 0x7FFFFFFFu;
            hit->index = s->material_indices[brick_material_index + (uint32_t)voxel_index];
            const oracle_material *m = &s->materials[hit->index];
            const int ignore_brick = (m->type == r->ignore_type_material) && (r->internal_reflection == m->type_data);
            if (e->c) e->c->hits++;
            if (!ignore_brick) {
                const float t_offset = voxel_scale.x * 0.05f;
                hit->t += t_value - t_offset;
                hit->point = vadd(RayAt(r, hit->t), vscale(hit->normal, t_offset));
                *brick_position = vadd(vmul(V3((float)lx, (float)ly, (float)lz), voxel_scale), *brick_position);
                return 1;
            }
        }
        int axis;
        if (side_dist.x < side_dist.y) {
            if (side_dist.x < side_dist.z) {
                t_value = side_dist.x * voxel_scale.x; side_dist.x += ray_delta.x; lx += ray_step[0];
                hit->normal = V3(normal_axis.x, 0, 0); axis = 0;
            } else {
                t_value = side_dist.z * voxel_scale.z; side_dist.z += ray_delta.z; lz += ray_step[2];
                hit->normal = V3(0, 0, normal_axis.z); axis = 2;
            }
        } else {
            if (side_dist.y < side_dist.z) {
                t_value = side_dist.y * voxel_scale.y; side_dist.y += ray_delta.y; ly += ray_step[1];
                hit->normal = V3(0, normal_axis.y, 0); axis = 1;
            } else {
                t_value = side_dist.z * voxel_scale.z; side_dist.z += ray_delta.z; lz += ray_step[2];
                hit->normal = V3(0, 0, normal_axis.z); axis = 2;
            }
        }
        if (ray_step[axis] == 0 && --zero_budget[axis] < 0) break; /* hang guard */
    }
    return 0;
}

/* comp:271-376 */
static int GridHit(const Env *e, const Ray *r, float t_min, float t_max, v3 *hit_min, HitRecord *hit) {
    const oracle_scene *s = e->s;
    const oracle_grid_state *g = s->grid;
    const v3 g_min = V3(g->min_point_base_t[0], g->min_point_base_t[1], g->min_point_base_t[2]);
    const v3 g_max = V3(g->max_point_scale[0], g->max_point_scale[1], g->max_point_scale[2]);
    const v3 g_scale = v3s(g->max_point_scale[3]);
    const int dx = (int)g->dim_x, dy = (int)g->dim_y, dz = (int)g->dim_z;

    if (e->c) e->c->rays++;
    const v3 inv_ray_dir = V3(safeInverse(r->direction.x), safeInverse(r->direction.y), safeInverse(r->direction.z));

    float grid_t_min = t_min;
    float grid_t_max = t_max;
    if (!AdvNormIntersect(g_min, g_max, r, inv_ray_dir, &hit->normal, &grid_t_min, &grid_t_max)) return 0;

    float global_t_value = grid_t_min + 0.0001f * g_scale.x;

    const v3 ray_delta = vabs(inv_ray_dir);
    const int ray_step[3] = {(int)fsign(r->direction.x), (int)fsign(r->direction.y), (int)fsign(r->direction.z)};
    const v3 fstep = V3((float)ray_step[0], (float)ray_step[1], (float)ray_step[2]);

    const v3 hit_point = RayAt(r, global_t_value);
    const v3 fposition = vdiv(vsub(hit_point, g_min), g_scale);
    const v3 intersection_delta = vsub(vfloor(fposition), fposition);
    v3 side_dist = vmul(vfma(fstep, intersection_delta, vadd(vscale(fstep, 0.5f), v3s(0.5f))), ray_delta);

    uint32_t brick_type_index = ~0u;
    uint32_t brick_bits = 0;
    const v3 normal_axis = V3(ray_step[0] < 0 ? 1.f : -1.f, ray_step[1] < 0 ? 1.f : -1.f, ray_step[2] < 0 ? 1.f : -1.f);

    float t_value = 0;
    int lx = vrt_f2i(floorf(fposition.x + 0.f)), ly = vrt_f2i(floorf(fposition.y + 0.f)), lz = vrt_f2i(floorf(fposition.z + 0.f));
    int zero_budget[3] = {dx + dy + dz + 8, dx + dy + dz + 8, dx + dy + dz + 8}; /* hang guard, see header */
    while (lx >= 0 && ly >= 0 && lz >= 0 && lx < dx && ly < dy && lz < dz && global_t_value <= t_max) {
        if (e->c) e->c->grid_steps++;
        const uint32_t grid_index = (uint32_t)(lx + dx * (lz + dz * ly));
        const uint32_t new_brick_type_index = grid_index / 32;
        const int brick_type_offset = (int)(grid_index % 32);
        if (brick_type_index != new_brick_type_index) {
            brick_bits = s->brick_type_bits[new_brick_type_index];
            brick_type_index = new_brick_type_index;
            if (e->c) e->c->status_loads++;
        }
        const uint32_t entry_type = brick_bits & (1u << brick_type_offset);
        if (entry_type != 0) {
            v3 brick_min = vfma(V3((float)lx, (float)ly, (float)lz), g_scale, g_min);
            global_t_value = t_value + grid_t_min + 0.01f * g_scale.x;
            hit->t = global_t_value;
            const uint32_t brick_index = s->brick_indices[grid_index];
            if (e->c) e->c->bricks_entered++;
            if (BrickHit(e, r, t_min, grid_t_max, ray_delta, ray_step, g_scale, brick_index, &brick_min, hit)) {
                *hit_min = brick_min;
                return 1;
            }
        }
        int axis;
        if (side_dist.x < side_dist.y) {
            if (side_dist.x < side_dist.z) {
                t_value = side_dist.x * g_scale.x; side_dist.x += ray_delta.x; lx += ray_step[0];
                hit->normal = V3(normal_axis.x, 0, 0); axis = 0;
            } else {
                t_value = side_dist.z * g_scale.z; side_dist.z += ray_delta.z; lz += ray_step[2];
                hit->normal = V3(0, 0, normal_axis.z); axis = 2;
            }
        } else {
            if (side_dist.y < side_dist.z) {
                t_value = side_dist.y * g_scale.y; side_dist.y += ray_delta.y; ly += ray_step[1];
                hit->normal = V3(0, normal_axis.y, 0); axis = 1;
            } else {
                t_value = side_dist.z * g_scale.z; side_dist.z += ray_delta.z; lz += ray_step[2];
                hit->normal = V3(0, 0, normal_axis.z); axis = 2;
            }
        }
        if (ray_step[axis] == 0 && --zero_budget[axis] < 0) break; /* hang guard */
    }
    return 0;
}

/* reflect(I,N) = I - 2*dot(N,I)*N */
static inline v3 reflect3(v3 I, v3 N) { const float k = 2.0f * vdot(N, I); return vsub(I, vscale(N, k)); }

/* comp:539-544 */
static int ScatterLambertian(const HitRecord *hit, Ray *scattered) {
    const v3 rv = RandVec3mm(hit->point.x + hit->point.z, hit->point.y + hit->point.z, -0.4f, 0.4f);
    const v3 scatter_dir = vnormalize(vadd(hit->normal, rv));
    *scattered = CreateRay(hit->point, scatter_dir);
    return 1;
}
/* comp:546-551 */
static int ScatterMetal(const oracle_material *m, const Ray *r_in, const HitRecord *hit, Ray *scattered) {
    const v3 reflected = reflect3(r_in->direction, hit->normal);
    const float fuzz = m->type_data;
    const v3 rv = RandVec3mm(hit->point.x + hit->point.z, hit->point.y + hit->point.z, -fuzz, fuzz);
    *scattered = CreateRay(hit->point, vadd(reflected, rv));
    return vdot(scattered->direction, hit->normal) > 0;
}
/* comp:564-574 */
static int transmissionDirection(float n1, float n2, v3 ray_dir, v3 normal, v3 *refrac_dir) {
    const float eta = n1 / n2;
    const float c1 = -vdot(ray_dir, normal);
    const float w = eta * c1;
    const float c2m = (w - eta) * (w + eta);
    if (c2m < -1.0f) return 0;
    *refrac_dir = vfma(v3s(eta), ray_dir, vscale(normal, w - sqrtf(1.0f + c2m)));
    return 1;
}
/* comp:576-596 */
static int ScatterDielectric(const oracle_material *m, const Ray *r_in, const HitRecord *hit, Ray *scattered) {
    const float ir = m->type_data;
    const v3 rv = RandVec3mm(hit->point.x + hit->point.z, hit->point.y + hit->point.z, -0.05f, 0.05f);
    const v3 normal = vnormalize(vadd(hit->normal, rv));
    v3 direction = V3(0, 0, 0);
    const int should_refract = transmissionDirection(ir, r_in->internal_reflection, r_in->direction, normal, &direction);
    if (should_refract && Rand3(hit->point) > 0.5f) {
        *scattered = CreateRay(hit->point, direction);
        scattered->ignore_type_material = MAT_DIELECTRIC;
        scattered->internal_reflection = ir;
    } else {
        direction = reflect3(r_in->direction, normal);
        *scattered = CreateRay(hit->point, direction);
    }
    return 1;
}

/* comp:203-265 */
static v3 RayColor(const Env *e, Ray r) {
    const oracle_push *pc = e->pc;
    const int sun_enabled = pc->sun_enabled > 0;
    const v3 sun_color = V3(pc->sun_color[0], pc->sun_color[1], pc->sun_color[2]);
    const v3 sun_position = V3(pc->sun_position[0], pc->sun_position[1], pc->sun_position[2]);
    HitRecord hit; memset(&hit, 0, sizeof hit);
    HitRecord shadow_hit; memset(&shadow_hit, 0, sizeof shadow_hit);
    Ray current_ray = r;
    int loop_count = 0;
    v3 color = V3(0, 0, 0);
    v3 hit_v_min = V3(0, 0, 0);

    while (loop_count < pc->max_bounce && GridHit(e, &current_ray, 0.00001f, INFINITY, &hit_v_min, &hit)) {
        loop_count += 1;
        Ray scattered = current_ray;
        int result = 0;
        const oracle_material material = e->s->materials[hit.index];
        const v3 attenuation = V3(material.albedo_r, material.albedo_g, material.albedo_b);
        switch (material.type) {
            case MAT_LAMBERTIAN: result = ScatterLambertian(&hit, &scattered); break;
            case MAT_METAL: result = ScatterMetal(&material, &current_ray, &hit, &scattered); break;
            case MAT_DIELECTRIC: result = ScatterDielectric(&material, &current_ray, &hit, &scattered); break;
            default: loop_count -= 1; result = 0; break;
        }
        if (sun_enabled) {
            const v3 rv = RandVec3mm(current_ray.direction.x + current_ray.direction.z,
                                     current_ray.direction.y + current_ray.direction.z, -pc->sun_radius, pc->sun_radius);
            const v3 sun_sample_position = vadd(sun_position, rv);
            const v3 shadow_ray_dir = vsub(sun_sample_position, hit.point);
            const Ray shadow_ray = CreateShadowRay(e, hit.point, shadow_ray_dir);
            if (!GridHit(e, &shadow_ray, 0.00001f, INFINITY, &hit_v_min, &shadow_hit)) {
                color = vadd(color, vmul(attenuation, sun_color));
            }
        } else {
            color = vadd(color, attenuation);
        }
        if (!result) break;
        current_ray = scattered;
    }
    if (loop_count == 0) {
        const v3 k = sun_enabled ? sun_color : v3s(1.0f);
        color = vadd(color, vmul(BackgroundColor(&current_ray), k));
    }
    return vdiv(color, vadd(color, v3s(1.0f)));
}

/* comp:474-477 */
static Ray CameraGetRay(const oracle_push *pc, float u, float v) {
    const v3 horizontal = V3(pc->horizontal[0], pc->horizontal[1], pc->horizontal[2]);
    const v3 vertical = V3(pc->vertical[0], pc->vertical[1], pc->vertical[2]);
    const v3 llc = V3(pc->lower_left_corner[0], pc->lower_left_corner[1], pc->lower_left_corner[2]);
    const v3 origin = V3(pc->origin[0], pc->origin[1], pc->origin[2]);
    const v3 ray_dir = vadd(vfma(horizontal, v3s(u), llc), vfma(v3s(v), vertical, vneg(origin)));
    return CreateRay(origin, ray_dir);
}

static inline uint8_t unorm8(float c) {
    if (!(c > 0.0f)) c = 0.0f; /* also NaN -> 0 */
    if (c > 1.0f) c = 1.0f;
    return (uint8_t)rintf(c * 255.0f);
}

/* comp:153-178 for one pixel */
static void shade_pixel(const Env *e, int px, int py, float out_f[4], uint8_t out_u[4]) {
    const oracle_push *pc = e->pc;
    v3 color = V3(0, 0, 0);
    for (int sample_i = 0; sample_i < pc->samples_per_pixel; sample_i++) {
        const float x = (float)px;
        const float y = (float)py;
        const float flag = (sample_i > 0) ? 1.0f : 0.0f;
        const float noise_x = hash12_jitter(x + (float)sample_i, y, flag);
        const float u = (x + noise_x) / (float)(pc->image_width - 1u);
        const float noise_y = hash12_jitter(x, y + (float)sample_i, flag);
        const float v = (y + noise_y) / (float)(pc->image_height - 1u);
        Ray ray = CameraGetRay(pc, u, v);
        color = vadd(color, RayColor(e, ray));
    }
    const float spp = (float)pc->samples_per_pixel;
    color = V3(sqrtf(color.x / spp), sqrtf(color.y / spp), sqrtf(color.z / spp));
    if (out_f) { out_f[0] = color.x; out_f[1] = color.y; out_f[2] = color.z; out_f[3] = 1.0f; }
    if (out_u) { out_u[0] = unorm8(color.x); out_u[1] = unorm8(color.y); out_u[2] = unorm8(color.z); out_u[3] = 255; }
}

/*
 * Render image rows [y0,y1) of the frame (all columns).  rgba32f / rgba8 are
 * full-frame row-major targets (either may be NULL).  counters may be NULL;
 * when given, it is ACCUMULATED into (caller zeroes it), and must not be
 * shared between concurrent callers.
 */
void oracle_render_rows(const oracle_scene *scene, const oracle_push *pc, int y0, int y1,
                        float *rgba32f, uint8_t *rgba8, oracle_counters *counters) {
    Env e = {scene, pc, counters};
    const int W = (int)pc->image_width;
    for (int y = y0; y < y1; y++) {
        for (int x = 0; x < W; x++) {
            const size_t o = ((size_t)y * (size_t)W + (size_t)x) * 4;
            shade_pixel(&e, x, y, rgba32f ? rgba32f + o : 0, rgba8 ? rgba8 + o : 0);
        }
    }
}

/* Render an arbitrary list of pixels (x,y pairs); outputs are packed per pixel. */
void oracle_render_pixels(const oracle_scene *scene, const oracle_push *pc, const int32_t *xy, uint64_t n,
                          float *rgba32f, uint8_t *rgba8, oracle_counters *counters) {
    Env e = {scene, pc, counters};
    for (uint64_t i = 0; i < n; i++)
        shade_pixel(&e, xy[2 * i], xy[2 * i + 1], rgba32f ? rgba32f + 4 * i : 0, rgba8 ? rgba8 + 4 * i : 0);
}

/* Single-ray probe for known-answer tests: returns hit flag, fills the record. */
int oracle_grid_hit(const oracle_scene *scene, const oracle_push *pc, const float origin[3], const float dir[3],
                    float out_point[3], float out_normal[3], float *out_t, uint32_t *out_index, oracle_counters *counters) {
    Env e = {scene, pc, counters};
    Ray r = CreateRay(V3(origin[0], origin[1], origin[2]), V3(dir[0], dir[1], dir[2]));
    HitRecord hit; memset(&hit, 0, sizeof hit);
    v3 hmin = V3(0, 0, 0);
    const int ok = GridHit(&e, &r, 0.00001f, INFINITY, &hmin, &hit);
    out_point[0] = hit.point.x; out_point[1] = hit.point.y; out_point[2] = hit.point.z;
    out_normal[0] = hit.normal.x; out_normal[1] = hit.normal.y; out_normal[2] = hit.normal.z;
    *out_t = hit.t; *out_index = hit.index;
    return ok;
}

/* The same probe with the ray given as is (no CreateRay: the direction is used unnormalised, and the two
 * fields of comp:427's ignore test are the caller's).  For the cross-check against the independent literal
 * restatement in tests/literal_port.py. */
int oracle_grid_hit_raw(const oracle_scene *scene, const oracle_push *pc, const float origin[3], const float dir[3],
                        uint32_t ignore_type_material, float internal_reflection, float out_point[3], float out_normal[3],
                        float *out_t, uint32_t *out_index, oracle_counters *counters) {
    Env e = {scene, pc, counters};
    Ray r;
    r.origin = V3(origin[0], origin[1], origin[2]);
    r.direction = V3(dir[0], dir[1], dir[2]);
    r.internal_reflection = internal_reflection;
    r.ignore_type_material = ignore_type_material;
    HitRecord hit; memset(&hit, 0, sizeof hit);
    v3 hmin = V3(0, 0, 0);
    const int ok = GridHit(&e, &r, 0.00001f, INFINITY, &hmin, &hit);
    out_point[0] = hit.point.x; out_point[1] = hit.point.y; out_point[2] = hit.point.z;
    out_normal[0] = hit.normal.x; out_normal[1] = hit.normal.y; out_normal[2] = hit.normal.z;
    *out_t = hit.t; *out_index = hit.index;
    return ok;
}
