/*
 * denoise_oracle.c — CPU restatement of the reference's present/denoise fragment shader.
 *
 * TEST INFRASTRUCTURE ONLY (parity oracle of SURVEY.md §8(f) #3).  Follows
 * /root/reference/assets/shaders/image.frag:18-78 ("sirBird" spiral denoiser) sampled through the
 * sampler of src/modules/voxel_rt/Pipeline.zig:194-211 (linear filter, repeat addressing) on the
 * fullscreen quad of GraphicsPipeline.zig:20-25 / image.vert.  PINNED within tolerance: image.vert + image.frag themselves run
 * under Mesa llvmpipe (oracle/ref_gl, float texture filtering) and this restatement agrees with their frames to <= 1.1e-5
 * (tests/golden/ref/present_*.npz, tests/test_ref_gl.py).  Not bit for bit: pow() precision and the bilinear filter's weights are
 * implementation-defined, so the HIP kernel too is compared under north_star's 1e-4.  Where all four texels of a tap are black
 * the result is NaN only if the weights come out exactly 0 / 1: a handful of pixels where implementations differ.
 *
 * Lowering: texture() = bilinear over RGBA8 UNORM texels with exact float weights
 *   (s = u*W - 0.5, i0 = floor(s), a = s - i0, repeat wrap), mix(x,y,a) = x*(1-a) + y*a;
 *   pow(a,b) = powf(max(a,0),b) (image.frag:27 macro; GLSL max(x,y) = x<y ? y : x);
 *   normalize(v) = v * (1/sqrt(dot)), length = sqrt(dot), dot as in vrt_oracle.c; rotation constants
 *   cos/sin(2.3999632) as float literals.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

typedef struct { int32_t samples; float distribution_bias, pixel_multiplier, inverse_hue_tolerance; } denoise_params;
typedef struct { float x, y, z; } v3;

static inline float gmax(float x, float y) { return (x < y) ? y : x; }
static inline float ppow(float a, float b) { return powf(gmax(a, 0.f), b); }
static inline float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 norm3(v3 a) { const float inv = 1.0f / sqrtf(dot3(a, a)); v3 r = {a.x * inv, a.y * inv, a.z * inv}; return r; }
static inline float len3(v3 a) { return sqrtf(dot3(a, a)); }
static inline float mixf(float x, float y, float a) { return x * (1.0f - a) + y * a; }

static inline int wrap(int i, int n) { int m = i % n; return m < 0 ? m + n : m; }

static v3 sample_bilinear(const uint8_t *img, int W, int H, float u, float v) {
    const float s = u * (float)W - 0.5f, t = v * (float)H - 0.5f;
    const float fs = floorf(s), ft = floorf(t);
    const float a = s - fs, b = t - ft;
    const int i0 = wrap((int)fs, W), i1 = wrap((int)fs + 1, W), j0 = wrap((int)ft, H), j1 = wrap((int)ft + 1, H);
    const uint8_t *p00 = img + 4 * ((size_t)j0 * W + i0), *p10 = img + 4 * ((size_t)j0 * W + i1);
    const uint8_t *p01 = img + 4 * ((size_t)j1 * W + i0), *p11 = img + 4 * ((size_t)j1 * W + i1);
    float c[3];
    for (int k = 0; k < 3; k++) {
        const float t00 = (float)p00[k] / 255.0f, t10 = (float)p10[k] / 255.0f, t01 = (float)p01[k] / 255.0f, t11 = (float)p11[k] / 255.0f;
        c[k] = mixf(mixf(t00, t10, a), mixf(t01, t11, a), b);
    }
    v3 r = {c[0], c[1], c[2]};
    return r;
}

static inline uint8_t unorm8(float c) {
    if (!(c > 0.0f)) c = 0.0f;
    if (c > 1.0f) c = 1.0f;
    return (uint8_t)rintf(c * 255.0f);
}

/* image.frag:31-71 for output rows [y0,y1) of an out_w x out_h target */
void oracle_denoise_rows(const uint8_t *img, int W, int H, const denoise_params *pc, int out_w, int out_h, int y0, int y1,
                         float *out_f32, uint8_t *out_u8) {
    const float cosg = -0.7373688f, sing = 0.6754904f; /* cos/sin(GOLDEN_ANGLE = 2.3999632) */
    for (int oy = y0; oy < y1; oy++)
        for (int ox = 0; ox < out_w; ox++) {
            const float u = ((float)ox + 0.5f) / (float)out_w, v = ((float)oy + 0.5f) / (float)out_h;
            v3 denoised = {0.f, 0.f, 0.f};
            const float sample_radius = sqrtf((float)pc->samples);
            const float sample_true_radius = 0.5f / (sample_radius * sample_radius);
            const float spx = 1.0f / (float)W, spy = 1.0f / (float)H;
            const v3 center = sample_bilinear(img, W, H, u, v);
            const v3 center_norm = norm3(center);
            const float center_sat = len3(center);
            float influence_sum = 0.0f;
            float rx = 0.f, ry = 1.f;
            for (float x = 0.0f; x <= (float)pc->samples; x++) {
                /* pixelRotated *= sample2D: row vector times mat2(c, s, -s, c) (columns (c,s) and (-s,c)) */
                const float nx = rx * cosg + ry * sing, ny = rx * (-sing) + ry * cosg;
                rx = nx; ry = ny;
                const float sq = sqrtf(x);
                float px = ((pc->pixel_multiplier * rx) * sq) * 0.5f, py = ((pc->pixel_multiplier * ry) * sq) * 0.5f;
                float influence = 1.0f - sample_true_radius * ppow(fmaf(py, py, px * px), pc->distribution_bias);
                px *= spx; py *= spy;
                const v3 c = sample_bilinear(img, W, H, u + px, v + py);
                influence *= influence * influence;
                influence *= ppow(0.5f + 0.5f * dot3(center_norm, norm3(c)), pc->inverse_hue_tolerance) *
                             ppow(1.0f - fabsf(len3(c) - fabsf(center_sat)), 8.f);
                influence_sum += influence;
                denoised.x += c.x * influence; denoised.y += c.y * influence; denoised.z += c.z * influence;
            }
            const float r = denoised.x / influence_sum, g = denoised.y / influence_sum, b = denoised.z / influence_sum;
            const size_t o = 4 * ((size_t)oy * out_w + ox);
            if (out_f32) { out_f32[o] = r; out_f32[o + 1] = g; out_f32[o + 2] = b; out_f32[o + 3] = 1.0f; }
            if (out_u8) { out_u8[o] = unorm8(r); out_u8[o + 1] = unorm8(g); out_u8[o + 2] = unorm8(b); out_u8[o + 3] = 255; }
        }
}
