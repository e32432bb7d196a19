// vrt_post.hip — the present/denoise pass that follows the traversal path, as a HIP kernel for gfx950.
//
// Replaces the fullscreen-quad fragment pass of the reference for this repo's image hand-off:
// assets/shaders/image.frag:18-78 ("sirBird" spiral denoiser, https://www.shadertoy.com/view/7d2SDD)
// sampling the traced image through the sampler of src/modules/voxel_rt/Pipeline.zig:194-211 (linear,
// repeat) at the output resolution (GraphicsPipeline.zig:20-39).  One lane = one output pixel; a pure
// gather: samples+2 bilinear taps (4 texels each) per pixel from an 8 MB image that stays in L2.
// Bound: texel gathers + pow()/sqrt() transcendental rate; algorithmic bytes per output pixel
// = (samples + 2) * 16 B read + 4 B written.
// Arithmetic lowering is stated in the oracle's restatement header; pow() is ocml's here, so the
// parity test uses the 1e-4 tolerance instead of bit equality.
#include <hip/hip_runtime.h>
#include "vrt_ctx.h"
#include "vrt_math.h"

namespace vrt {

struct DenoiseParams { // GraphicsPipeline.PushConstant, GraphicsPipeline.zig:27-32
    int samples;
    float distribution_bias, pixel_multiplier, inverse_hue_tolerance;
};

VRT_DI float ppow(float a, float b) { return __builtin_powf(gl_max(a, 0.0f), b); } // image.frag:27
VRT_DI float mix1(float x, float y, float a) { return x * (1.0f - a) + y * a; }
VRT_DI int wrap_repeat(int i, int n) {
    const int m = i % n;
    return m < 0 ? m + n : m;
}

VRT_DI f3 sample_bilinear(const uchar4 *__restrict__ img, int W, int H, float u, float v) {
    const float s = u * (float)W - 0.5f, t = v * (float)H - 0.5f;
    const float fs = __builtin_floorf(s), ft = __builtin_floorf(t);
    const float a = s - fs, b = t - ft;
    const int i0 = wrap_repeat((int)fs, W), i1 = wrap_repeat((int)fs + 1, W);
    const int j0 = wrap_repeat((int)ft, H), j1 = wrap_repeat((int)ft + 1, H);
    const uchar4 p00 = img[(size_t)j0 * W + i0], p10 = img[(size_t)j0 * W + i1];
    const uchar4 p01 = img[(size_t)j1 * W + i0], p11 = img[(size_t)j1 * W + i1];
    const float k = 255.0f;
    return mk3(mix1(mix1((float)p00.x / k, (float)p10.x / k, a), mix1((float)p01.x / k, (float)p11.x / k, a), b),
               mix1(mix1((float)p00.y / k, (float)p10.y / k, a), mix1((float)p01.y / k, (float)p11.y / k, a), b),
               mix1(mix1((float)p00.z / k, (float)p10.z / k, a), mix1((float)p01.z / k, (float)p11.z / k, a), b));
}

VRT_DI float length3(f3 a) { return __builtin_sqrtf(dot3(a, a)); }

VRT_DI uint32_t unorm8p(float c) {
    c = (c > 0.0f) ? c : 0.0f;
    c = (c > 1.0f) ? 1.0f : c;
    return (uint32_t)__builtin_rintf(c * 255.0f);
}

// Round 4: the pass took 0.40 ms per 1080p frame — longer than the headline's trace (0.13 ms along the reference's benchmark path) — at
// ~500 instructions per tap: two integer `%` per texel coordinate (wrap), twelve IEEE divisions by 255 per tap, three generic pow()
// per tap, and the spiral's per-sample arithmetic (rotation, sqrt, a pow) recomputed by every pixel although it depends on the sample
// index alone.  Now: the per-sample constants (texel offset, cubed distance weight) are computed once per workgroup into LDS by the
// SAME operations (bit-equal); coordinates wrap by one conditional add / subtract (a tap lies within a few texels of the image);
// c / 255 is c * (1 / 255) (<= 1 ulp); pow(x, 8) is three squarings and pow(x, t) for integer t <= 64 a chain of squarings, v_exp /
// v_log otherwise (<= 2e-6 relative).  The tolerance of the pass is 1e-4 per channel (tests/test_denoise.py, tests/test_ref_gl.py).
constexpr int kDenoiseTable = 256; // per-sample constants held in LDS (more samples: computed in place)

VRT_DI int wrap_near(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); } // i in [-n, 2n)
VRT_DI f3 texel_rgb(const uchar4 *__restrict__ img, int W, int j, int i) {
    const uchar4 p = img[(size_t)j * W + i];
    const float k = 1.0f / 255.0f;
    return mk3((float)p.x * k, (float)p.y * k, (float)p.z * k);
}
template <bool NEAR>
VRT_DI f3 sample_bilinear_fast(const uchar4 *__restrict__ img, int W, int H, float u, float v) {
    const float s = u * (float)W - 0.5f, t = v * (float)H - 0.5f;
    const float fs = __builtin_floorf(s), ft = __builtin_floorf(t);
    const float a = s - fs, b = t - ft;
    const int i0 = NEAR ? wrap_near((int)fs, W) : wrap_repeat((int)fs, W), i1 = NEAR ? wrap_near((int)fs + 1, W) : wrap_repeat((int)fs + 1, W);
    const int j0 = NEAR ? wrap_near((int)ft, H) : wrap_repeat((int)ft, H), j1 = NEAR ? wrap_near((int)ft + 1, H) : wrap_repeat((int)ft + 1, H);
    const f3 p00 = texel_rgb(img, W, j0, i0), p10 = texel_rgb(img, W, j0, i1), p01 = texel_rgb(img, W, j1, i0), p11 = texel_rgb(img, W, j1, i1);
    return mk3(mix1(mix1(p00.x, p10.x, a), mix1(p01.x, p11.x, a), b), mix1(mix1(p00.y, p10.y, a), mix1(p01.y, p11.y, a), b),
               mix1(mix1(p00.z, p10.z, a), mix1(p01.z, p11.z, a), b));
}
// max(a, 0) ^ e: by squarings when e is a small whole number (wave-uniform), else exp2(e * log2(a))
VRT_DI float pow_whole(float a, int e) {
    a = gl_max(a, 0.0f);
    float r = 1.0f, q = a;
    for (int k = e; k > 0; k >>= 1) { // (uniform trip count)
        if (k & 1) r *= q;
        q *= q;
    }
    return r;
}
// normalize() and length() of a tap through the hardware's reciprocal square root / square root (<= 1 ulp; 0 -> inf -> NaN as in IEEE)
VRT_DI f3 normalize_fast(f3 a) { return a * __builtin_amdgcn_rsqf(dot3(a, a)); }
VRT_DI float length_fast(f3 a) { return __builtin_amdgcn_sqrtf(dot3(a, a)); }
VRT_DI float pow_fast(float a, float e) { return __builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf(gl_max(a, 0.0f))); }

// The taps of a workgroup's 16 x 16 output pixels fall into a small box of the traced image (25 x 25 texels at the reference's defaults,
// 1 : 1): vrt_denoise_tile_kernel converts that box to floats ONCE into LDS (32 x 32 texels, 16 KiB; coordinates wrapped there) and a
// tap is four LDS reads — no global loads, no unpacking, no wrapping per tap (round 4: 163 -> ~120 us per 1080p frame).  The launcher
// takes it when every workgroup's box fits.
constexpr int kDenoiseTile = 32;
VRT_DI f3 sample_bilinear_tile(const float4 *tile, int x0, int y0, int W, int H, float u, float v) {
    const float s = u * (float)W - 0.5f, t = v * (float)H - 0.5f;
    const float fs = __builtin_floorf(s), ft = __builtin_floorf(t);
    const float a = s - fs, b = t - ft;
    const int i0 = (int)fs - x0, j0 = (int)ft - y0; // (inside [0, kDenoiseTile - 2] by the launcher's bound on the box)
    const float4 p00 = tile[j0 * kDenoiseTile + i0], p10 = tile[j0 * kDenoiseTile + i0 + 1], p01 = tile[(j0 + 1) * kDenoiseTile + i0],
                 p11 = tile[(j0 + 1) * kDenoiseTile + i0 + 1];
    return mk3(mix1(mix1(p00.x, p10.x, a), mix1(p01.x, p11.x, a), b), mix1(mix1(p00.y, p10.y, a), mix1(p01.y, p11.y, a), b),
               mix1(mix1(p00.z, p10.z, a), mix1(p01.z, p11.z, a), b));
}

// NEAR: every tap lies within one image width / height of the image (the launcher checks the spiral's radius): one conditional wrap
template <bool NEAR>
__global__ __launch_bounds__(256) void vrt_denoise_kernel(const uchar4 *__restrict__ img, int W, int H, DenoiseParams pc, int out_w, int out_h,
                                                          uint32_t *__restrict__ out_u8, float4 *__restrict__ out_f32) {
    __shared__ float tab_x[kDenoiseTable], tab_y[kDenoiseTable], tab_w[kDenoiseTable];
    const float cosg = -0.7373688f, sing = 0.6754904f; // cos/sin(GOLDEN_ANGLE), image.frag:25,29
    const float sample_radius = __builtin_sqrtf((float)pc.samples);
    const float sample_true_radius = 0.5f / (sample_radius * sample_radius);
    const float spx = 1.0f / (float)W, spy = 1.0f / (float)H;
    // sample k of the spiral (image.frag:45-55): its texel offset and its distance weight, cubed — the sample index is all they depend on
    auto spiral = [&](int k, float &rx, float &ry, float &ox, float &oy, float &wgt) {
        const float nx = rx * cosg + ry * sing, ny = rx * (-sing) + ry * cosg;
        rx = nx;
        ry = ny;
        const float sq = __builtin_sqrtf((float)k);
        const float px = ((pc.pixel_multiplier * rx) * sq) * 0.5f, py = ((pc.pixel_multiplier * ry) * sq) * 0.5f;
        float influence = 1.0f - sample_true_radius * ppow(__builtin_fmaf(py, py, px * px), pc.distribution_bias);
        influence *= influence * influence;
        ox = px * spx, oy = py * spy, wgt = influence;
    };
    const int n = pc.samples + 1; // x = 0 .. samples inclusive (image.frag:45)
    if (threadIdx.x == 0) {
        float rx = 0.0f, ry = 1.0f;
        for (int k = 0; k < n && k < kDenoiseTable; k++) spiral(k, rx, ry, tab_x[k], tab_y[k], tab_w[k]);
    }
    __syncthreads();
    // 16x16 output pixels per workgroup: neighbouring lanes fetch neighbouring texels
    const int ox = (int)(blockIdx.x * 16u + (threadIdx.x & 15u));
    const int oy = (int)(blockIdx.y * 16u + (threadIdx.x >> 4));
    if (ox >= out_w || oy >= out_h) return;
    const float u = ((float)ox + 0.5f) / (float)out_w, v = ((float)oy + 0.5f) / (float)out_h;
    auto tap = [&](float tu, float tv) { return sample_bilinear_fast<NEAR>(img, W, H, tu, tv); };
    const f3 center = tap(u, v);
    const f3 center_norm = normalize3(center);
    const float center_sat = length3(center);
    const int hue_whole = (pc.inverse_hue_tolerance >= 0.0f && pc.inverse_hue_tolerance <= 64.0f && pc.inverse_hue_tolerance == __builtin_floorf(pc.inverse_hue_tolerance))
                              ? (int)pc.inverse_hue_tolerance : -1;
    f3 denoised = mk3(0, 0, 0);
    float influence_sum = 0.0f;
    float rx = 0.0f, ry = 1.0f; // (only walked beyond the table)
    for (int k = 0; k < n; k++) {
        float px, py, influence;
        if (k < kDenoiseTable) px = tab_x[k], py = tab_y[k], influence = tab_w[k];
        else {
            if (k == kDenoiseTable) { // catch up with the spiral's rotation
                rx = 0.0f, ry = 1.0f;
                float a, b, c;
                for (int q = 0; q < kDenoiseTable; q++) spiral(q, rx, ry, a, b, c);
            }
            spiral(k, rx, ry, px, py, influence);
        }
        const f3 c = tap(u + px, v + py);
        const float hue = 0.5f + 0.5f * dot3(center_norm, normalize_fast(c));
        const float sat = 1.0f - __builtin_fabsf(length_fast(c) - __builtin_fabsf(center_sat));
        influence *= (hue_whole >= 0 ? pow_whole(hue, hue_whole) : pow_fast(hue, pc.inverse_hue_tolerance)) * pow_whole(sat, 8);
        influence_sum += influence;
        denoised = denoised + c * influence;
    }
    const float r = denoised.x / influence_sum, g = denoised.y / influence_sum, b = denoised.z / influence_sum;
    const size_t o = (size_t)oy * out_w + ox;
    out_u8[o] = unorm8p(r) | (unorm8p(g) << 8) | (unorm8p(b) << 16) | (255u << 24);
    if (out_f32) out_f32[o] = make_float4(r, g, b, 1.0f);
}

// Round 5: the staged path of the reference's default hue exponent, written for the vector unit's issue rate — the pass is bound by it
// (round 4: ~110 vector instructions per tap, 428 cycles per wave-tap measured = 4 x 107).  Per tap, inside the pass's 1e-4 tolerance:
// the bilinear blend and the running sums as fused multiply-adds, two colour channels per instruction (v_pk_mul_f32 / v_pk_fma_f32 on
// the register pair ds_read leaves x and y in); one reciprocal square root serves normalize() and length() (length = d * rsq(d));
// dot(centre_n, normalize(c)) = dot(centre_n, c) * rsq; hue ^ 20 as five multiplications (hue >= -1e-7: its twentieth power is 0 with
// or without the shader's max(., 0)); the per-sample constants straight from the table (the launcher guarantees it holds them all).
// What a texel's coordinates are computed from and how — (u + offset) * W - 0.5, floor, fraction — is untouched: that decides WHICH
// texels a tap blends.  NaN where the shader has NaN: a black tap's rsq(0) = inf poisons its hue, and through it the pixel.
typedef float f2v __attribute__((ext_vector_type(2)));
VRT_DI f2v splat2(float a) { return f2v{a, a}; }
VRT_DI f2v fma2(f2v a, f2v b, f2v c) { return __builtin_elementwise_fma(a, b, c); }
struct Rgb2 {
    f2v xy;
    float z;
};
VRT_DI Rgb2 tile_tap(const float4 *tile, int base, float Wf, float Hf, float u, float v) {
    const float s = u * Wf - 0.5f, t = v * Hf - 0.5f;
    const float fs = __builtin_floorf(s), ft = __builtin_floorf(t);
    const float a = s - fs, b = t - ft, ia = 1.0f - a, ib = 1.0f - b;
    // (the texel's place in the staged box, formed in floats — whole numbers below 2^24, exact — and converted once; inside the box by the
    // launcher's bound)
    const float4 *q = tile + ((int)__builtin_fmaf(ft, (float)kDenoiseTile, fs) + base);
    const float4 p00 = q[0], p10 = q[1], p01 = q[kDenoiseTile], p11 = q[kDenoiseTile + 1];
    const f2v top = fma2(f2v{p10.x, p10.y}, splat2(a), f2v{p00.x, p00.y} * splat2(ia));
    const f2v bot = fma2(f2v{p11.x, p11.y}, splat2(a), f2v{p01.x, p01.y} * splat2(ia));
    const float topz = __builtin_fmaf(p10.z, a, p00.z * ia), botz = __builtin_fmaf(p11.z, a, p01.z * ia);
    return Rgb2{fma2(bot, splat2(b), top * splat2(ib)), __builtin_fmaf(botz, b, topz * ib)};
}
template <int HUE>
__global__ __launch_bounds__(256) void vrt_denoise_tile_kernel(const uchar4 *__restrict__ img, int W, int H, DenoiseParams pc, int out_w, int out_h,
                                                               uint32_t *__restrict__ out_u8, float4 *__restrict__ out_f32) {
    // HUE 20: the reference's default exponent (GraphicsPipeline.zig:38); 0: pc.inverse_hue_tolerance, whatever it is
    static_assert(HUE == 20 || HUE == 0, "");
    __shared__ float4 tab[kDenoiseTable]; // per sample: texel offset x, y, the cubed distance weight
    __shared__ float4 tile[kDenoiseTile * kDenoiseTile];
    const float cosg = -0.7373688f, sing = 0.6754904f; // cos/sin(GOLDEN_ANGLE), image.frag:25,29
    const float sample_radius = __builtin_sqrtf((float)pc.samples);
    const float sample_true_radius = 0.5f / (sample_radius * sample_radius);
    const float spx = 1.0f / (float)W, spy = 1.0f / (float)H;
    const int n = pc.samples + 1; // x = 0 .. samples inclusive (image.frag:45)
    if ((int)threadIdx.x < n) {
        // sample k by thread k — the same operations in the same order as vrt_denoise_kernel's `spiral` (its rotation is a recurrence:
        // thread k repeats the k + 1 turns, four operations each; the square root and the pow() are then done side by side instead of
        // one sample after the other by one thread while its workgroup waits: that was 30 % of the pass)
        const int k = (int)threadIdx.x;
        float rx = 0.0f, ry = 1.0f;
        for (int q = 0; q <= k; q++) {
            const float nx = rx * cosg + ry * sing, ny = rx * (-sing) + ry * cosg;
            rx = nx;
            ry = ny;
        }
        const float sq = __builtin_sqrtf((float)k);
        const float px = ((pc.pixel_multiplier * rx) * sq) * 0.5f, py = ((pc.pixel_multiplier * ry) * sq) * 0.5f;
        float influence = 1.0f - sample_true_radius * ppow(__builtin_fmaf(py, py, px * px), pc.distribution_bias);
        influence *= influence * influence;
        tab[k] = make_float4(px * spx, py * spy, influence, 0.0f);
    }
    const float reach = __builtin_fabsf(pc.pixel_multiplier) * sample_radius * 0.5f;
    const int x0 = (int)__builtin_floorf(((float)(blockIdx.x * 16u) + 0.5f) / (float)out_w * (float)W - 0.5f - reach) - 1;
    const int y0 = (int)__builtin_floorf(((float)(blockIdx.y * 16u) + 0.5f) / (float)out_h * (float)H - 0.5f - reach) - 1;
    // (the box starts less than its own size outside the image and the image is at least as large — the launcher's conditions —: a
    // coordinate wraps by one conditional add / subtract, not by two integer remainders per texel)
    for (int i = (int)threadIdx.x; i < kDenoiseTile * kDenoiseTile; i += 256) {
        const f3 c = texel_rgb(img, W, wrap_near(y0 + i / kDenoiseTile, H), wrap_near(x0 + i % kDenoiseTile, W));
        tile[i] = make_float4(c.x, c.y, c.z, 0.0f);
    }
    __syncthreads();
    const int ox = (int)(blockIdx.x * 16u + (threadIdx.x & 15u));
    const int oy = (int)(blockIdx.y * 16u + (threadIdx.x >> 4));
    if (ox >= out_w || oy >= out_h) return;
    const float u = ((float)ox + 0.5f) / (float)out_w, v = ((float)oy + 0.5f) / (float)out_h;
    const float Wf = (float)W, Hf = (float)H;
    const int base = -(y0 * kDenoiseTile + x0);
    const f3 center = sample_bilinear_tile(tile, x0, y0, W, H, u, v);
    const f3 cn = normalize3(center);
    const float center_sat = __builtin_fabsf(length3(center));
    const f2v cn_xy = f2v{cn.x, cn.y};
    [[maybe_unused]] const int hue_whole = (pc.inverse_hue_tolerance >= 0.0f && pc.inverse_hue_tolerance <= 64.0f && pc.inverse_hue_tolerance == __builtin_floorf(pc.inverse_hue_tolerance))
                                               ? (int)pc.inverse_hue_tolerance : -1;
    f2v acc_xy = splat2(0.0f);
    float acc_z = 0.0f, influence_sum = 0.0f;
    for (int k = 0; k < n; k++) {
        const float4 sk = tab[k];
        const Rgb2 c = tile_tap(tile, base, Wf, Hf, u + sk.x, v + sk.y);
        const f2v sq = c.xy * c.xy;
        const float d = __builtin_fmaf(c.z, c.z, sq.x + sq.y);
        const float rs = __builtin_amdgcn_rsqf(d);
        const f2v cd = cn_xy * c.xy;
        const float hue = __builtin_fmaf(0.5f, __builtin_fmaf(cn.z, c.z, cd.x + cd.y) * rs, 0.5f);
        float hue_w;
        if constexpr (HUE == 20) {
            const float h2 = hue * hue, h4 = h2 * h2, h8 = h4 * h4;
            hue_w = (h8 * h8) * h4;
        } else {
            hue_w = hue_whole >= 0 ? pow_whole(hue, hue_whole) : pow_fast(hue, pc.inverse_hue_tolerance);
        }
        const float sat = __builtin_fmaxf(1.0f - __builtin_fabsf(d * rs - center_sat), 0.0f); // (NaN only where hue is NaN already)
        const float s2 = sat * sat, s4 = s2 * s2;
        const float influence = (sk.z * hue_w) * (s4 * s4);
        influence_sum += influence;
        acc_xy = fma2(c.xy, splat2(influence), acc_xy);
        acc_z = __builtin_fmaf(c.z, influence, acc_z);
    }
    const float r = acc_xy.x / influence_sum, g = acc_xy.y / influence_sum, b = acc_z / influence_sum;
    const size_t o = (size_t)oy * out_w + ox;
    out_u8[o] = unorm8p(r) | (unorm8p(g) << 8) | (unorm8p(b) << 16) | (255u << 24);
    if (out_f32) out_f32[o] = make_float4(r, g, b, 1.0f);
}

hipError_t launch_denoise(const void *img, int W, int H, int samples, float bias, float mult, float tol, int out_w, int out_h, void *out_u8,
                          void *out_f32, hipStream_t stream) {
    const dim3 grid((out_w + 15) / 16, (out_h + 15) / 16);
    const DenoiseParams pc{samples, bias, mult, tol};
    // the farthest tap: |pixel_multiplier| * sqrt(samples) / 2 texels from the pixel's own (image.frag:49-50), + the bilinear footprint
    const float reach = __builtin_fabsf(mult) * __builtin_sqrtf((float)samples) * 0.5f + 3.0f;
    // the box of a workgroup: 15 output pixels' worth of texels + the spiral both ways + the bilinear footprint and slack (kernel: x0 .. x0 + 31)
    const float span_x = 15.0f * (float)W / (float)out_w + 2.0f * (reach - 3.0f) + 5.0f, span_y = 15.0f * (float)H / (float)out_h + 2.0f * (reach - 3.0f) + 5.0f;
    const bool staged = span_x <= (float)kDenoiseTile && span_y <= (float)kDenoiseTile && samples < kDenoiseTable && W >= 2 * kDenoiseTile && H >= 2 * kDenoiseTile;
    if (staged && tol == 20.0f)
        VRT_LAUNCH(vrt_denoise_tile_kernel<20>, grid, dim3(256), 0, stream, (const uchar4 *)img, W, H, pc, out_w, out_h, (uint32_t *)out_u8, (float4 *)out_f32);
    else if (staged)
        VRT_LAUNCH(vrt_denoise_tile_kernel<0>, grid, dim3(256), 0, stream, (const uchar4 *)img, W, H, pc, out_w, out_h, (uint32_t *)out_u8, (float4 *)out_f32);
    else if (reach < (float)(W < H ? W : H))
        VRT_LAUNCH(vrt_denoise_kernel<true>, grid, dim3(256), 0, stream, (const uchar4 *)img, W, H, pc, out_w, out_h, (uint32_t *)out_u8, (float4 *)out_f32);
    else
        VRT_LAUNCH(vrt_denoise_kernel<false>, grid, dim3(256), 0, stream, (const uchar4 *)img, W, H, pc, out_w, out_h, (uint32_t *)out_u8, (float4 *)out_f32);
    return hipGetLastError();
}

} // namespace vrt

// ---- the C ABI of the pass (include/vrt_hip.h: vrt_denoise, vrt_read_denoised_*, vrt_last_denoise_ms) ----
using namespace vrt_impl;

extern "C" {

int vrt_denoise(vrt_ctx *ctx, const vrt_denoise_config *cfg, uint32_t out_w, uint32_t out_h, uint32_t want_float) {
    if (!ctx || out_w == 0 || out_h == 0) return ctx ? fail(ctx, VRT_E_INVALID_ARG, "zero output size") : VRT_E_INVALID_ARG;
    if (ctx->shard.shard_count > 1u) return fail(ctx, VRT_E_STATE, "vrt_denoise needs the whole frame (unsharded context)");
    vrt_denoise_config c{20, 0.6f, 1.5f, 20.0f}; // GraphicsPipeline.Config, GraphicsPipeline.zig:34-39
    if (cfg) c = *cfg;
    if (c.samples < 0 || c.samples > 4096) return fail(ctx, VRT_E_INVALID_ARG, "samples out of range");
    DeviceGuard dg(ctx->device);
    // On the stream that rendered the most recent frame, so it is ordered after that frame; with two frames in flight the next frame — on
    // the other stream, into the other target — traces underneath it.  VRT_TUNE_PRESENT_OWN_STREAM: on the present stream, behind an event
    // of that frame (the reference's arrangement: graphics queue behind the compute queue's semaphore, Pipeline.zig:494-517).
    const int slot = ctx->last_slot == 1 ? 1 : 0;
    const hipStream_t frame_stream = slot ? ctx->stream_b : ctx->stream;
    // (... measured, round 6: the pass on the frame's OWN stream overlaps the next frame's trace just as well — that frame runs on the
    // other stream — and the present stream's two extra cross-stream waits per frame cost 3-12 %: headline fly-through, two frames in
    // flight, 0.165 ms per frame on the frame's stream against 0.184 on the present stream (one frame at a time: 0.198); the arrangement
    // stays behind VRT_TUNE_PRESENT_OWN_STREAM, profiles/r06_present_overlap.txt)
    const bool overlap = ctx->stream_b && (ctx->cfg.tuning_flags & VRT_TUNE_PRESENT_OWN_STREAM);
    hipStream_t s = frame_stream;
    if (overlap) {
        if (!ctx->stream_post) {
            VRT_HIP(ctx, ctx->res.stream(&ctx->stream_post));
            for (int k = 0; k < 2; k++) {
                VRT_HIP(ctx, ctx->res.event(&ctx->ev_post_src[k], hipEventDisableTiming));
                VRT_HIP(ctx, ctx->res.event(&ctx->ev_post_done[k], hipEventDisableTiming));
            }
        }
        s = ctx->stream_post;
    }
    if (ctx->denoised_w != out_w || ctx->denoised_h != out_h || (want_float && !ctx->d_denoised32f)) {
        VRT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->stream_b) VRT_HIP(ctx, hipStreamSynchronize(ctx->stream_b));
        if (ctx->stream_post) VRT_HIP(ctx, hipStreamSynchronize(ctx->stream_post));
        ctx->res.drop(ctx->d_denoised8);
        ctx->res.drop(ctx->d_denoised32f);
        VRT_HIP(ctx, ctx->res.device(&ctx->d_denoised8, (size_t)out_w * out_h * 4u));
        if (want_float) VRT_HIP(ctx, ctx->res.device(&ctx->d_denoised32f, (size_t)out_w * out_h * 16u));
        ctx->denoised_w = out_w;
        ctx->denoised_h = out_h;
    }
    const void *img = slot ? ctx->target8_b : ctx->target8;
    if (overlap) {
        VRT_HIP(ctx, hipEventRecord(ctx->ev_post_src[slot], frame_stream));
        VRT_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_post_src[slot], 0));
    }
    VRT_HIP(ctx, hipEventRecord(ctx->ev_post_start, s));
    VRT_HIP(ctx, vrt::launch_denoise(img, (int)ctx->cfg.width, (int)ctx->cfg.height, c.samples, c.distribution_bias, c.pixel_multiplier,
                                     c.inverse_hue_tolerance, (int)out_w, (int)out_h, ctx->d_denoised8, want_float ? ctx->d_denoised32f : nullptr, s));
    VRT_HIP(ctx, hipEventRecord(ctx->ev_post_stop, s));
    if (overlap) { // (the next frame into this target waits until the pass has read it: do_dispatch)
        VRT_HIP(ctx, hipEventRecord(ctx->ev_post_done[slot], s));
        ctx->post_pending[slot] = true;
    }
    ctx->post_timed = true;
    ctx->denoised_stream = s;
    return VRT_OK;
}

double vrt_last_denoise_ms(vrt_ctx *ctx) {
    if (!ctx || !ctx->post_timed) return -1.0;
    DeviceGuard dg(ctx->device);
    if (wait_event(ctx->ev_post_stop) != hipSuccess) return -1.0;
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ctx->ev_post_start, ctx->ev_post_stop) != hipSuccess) return -1.0;
    return (double)ms;
}

static int read_denoised(vrt_ctx *ctx, void *dst, uint64_t nbytes, const void *src, uint64_t avail) {
    if (!ctx || !dst) return VRT_E_INVALID_ARG;
    if (!src) return fail(ctx, VRT_E_STATE, "no denoised image (call vrt_denoise first; want_float for the float image)");
    if (nbytes > avail) return fail(ctx, VRT_E_OUT_OF_RANGE, "read exceeds the denoised image");
    DeviceGuard dg(ctx->device);
    VRT_HIP(ctx, hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToHost, ctx->denoised_stream));
    VRT_HIP(ctx, hipStreamSynchronize(ctx->denoised_stream));
    return VRT_OK;
}
int vrt_read_denoised_rgba8(vrt_ctx *ctx, void *dst, uint64_t nbytes) {
    return read_denoised(ctx, dst, nbytes, ctx ? ctx->d_denoised8 : nullptr, ctx ? (uint64_t)ctx->denoised_w * ctx->denoised_h * 4u : 0);
}
int vrt_read_denoised_rgba32f(vrt_ctx *ctx, void *dst, uint64_t nbytes) {
    return read_denoised(ctx, dst, nbytes, ctx ? ctx->d_denoised32f : nullptr, ctx ? (uint64_t)ctx->denoised_w * ctx->denoised_h * 16u : 0);
}
void *vrt_device_denoised_rgba8(vrt_ctx *ctx) { return ctx ? ctx->d_denoised8 : nullptr; }

} // extern "C"
