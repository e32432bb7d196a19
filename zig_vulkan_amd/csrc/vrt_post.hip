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
#include "vrt_internal.h"
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

__global__ __launch_bounds__(256) void vrt_denoise_kernel(const uchar4 *__restrict__ img, int W, int H, DenoiseParams pc, int out_w, int out_h,
                                                          uint32_t *__restrict__ out_u8, float4 *__restrict__ out_f32) {
    // 16x16 output pixels per workgroup: neighbouring lanes fetch neighbouring texels
    const int ox = (int)(blockIdx.x * 16u + (threadIdx.x & 15u));
    const int oy = (int)(blockIdx.y * 16u + (threadIdx.x >> 4));
    if (ox >= out_w || oy >= out_h) return;
    const float cosg = -0.7373688f, sing = 0.6754904f; // cos/sin(GOLDEN_ANGLE), image.frag:25,29
    const float u = ((float)ox + 0.5f) / (float)out_w, v = ((float)oy + 0.5f) / (float)out_h;
    const float sample_radius = __builtin_sqrtf((float)pc.samples);
    const float sample_true_radius = 0.5f / (sample_radius * sample_radius);
    const float spx = 1.0f / (float)W, spy = 1.0f / (float)H;
    const f3 center = sample_bilinear(img, W, H, u, v);
    const f3 center_norm = normalize3(center);
    const float center_sat = length3(center);
    f3 denoised = mk3(0, 0, 0);
    float influence_sum = 0.0f;
    float rx = 0.0f, ry = 1.0f;
    for (float x = 0.0f; x <= (float)pc.samples; x++) { // image.frag:45
        const float nx = rx * cosg + ry * sing, ny = rx * (-sing) + ry * cosg;
        rx = nx;
        ry = ny;
        const float sq = __builtin_sqrtf(x);
        float px = ((pc.pixel_multiplier * rx) * sq) * 0.5f, py = ((pc.pixel_multiplier * ry) * sq) * 0.5f;
        float influence = 1.0f - sample_true_radius * ppow(__builtin_fmaf(py, py, px * px), pc.distribution_bias);
        px *= spx;
        py *= spy;
        const f3 c = sample_bilinear(img, W, H, u + px, v + py);
        influence *= influence * influence;
        influence *= ppow(0.5f + 0.5f * dot3(center_norm, normalize3(c)), pc.inverse_hue_tolerance) *
                     ppow(1.0f - __builtin_fabsf(length3(c) - __builtin_fabsf(center_sat)), 8.0f);
        influence_sum += influence;
        denoised = denoised + c * influence;
    }
    const float r = denoised.x / influence_sum, g = denoised.y / influence_sum, b = denoised.z / influence_sum;
    const size_t o = (size_t)oy * out_w + ox;
    out_u8[o] = unorm8p(r) | (unorm8p(g) << 8) | (unorm8p(b) << 16) | (255u << 24);
    if (out_f32) out_f32[o] = make_float4(r, g, b, 1.0f);
}

hipError_t launch_denoise(const void *img, int W, int H, int samples, float bias, float mult, float tol, int out_w, int out_h, void *out_u8,
                          void *out_f32, hipStream_t stream) {
    const dim3 grid((out_w + 15) / 16, (out_h + 15) / 16);
    const DenoiseParams pc{samples, bias, mult, tol};
    hipLaunchKernelGGL(vrt_denoise_kernel, grid, dim3(256), 0, stream, (const uchar4 *)img, W, H, pc, out_w, out_h, (uint32_t *)out_u8,
                       (float4 *)out_f32);
    return hipGetLastError();
}

} // namespace vrt
