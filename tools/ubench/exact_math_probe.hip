// exact_math_probe.hip — exhaustive check on the GPU (every binary32 bit pattern) of short instruction sequences against the IEEE
// operations they would replace in the traversal kernels: 1/x, c/(c+1), sqrt(x).  Prints, per candidate, the number of inputs whose
// result differs in any bit (NaNs compared as NaN == NaN) inside the guarded range, and the first few.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-gpu-flush-denormals-to-zero exact_math_probe.hip -o exact_math_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ bool same(float a, float b) { return f2u(a) == f2u(b) || (a != a && b != b); }

// candidates
__device__ __forceinline__ float rcp1(float x) { // rcp + one Newton step
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float rcp2(float x) { // ... + the residual correction of the quotient 1/x
    const float y1 = rcp1(x);
    const float r = __builtin_fmaf(-x, y1, 1.0f);
    return __builtin_fmaf(r, y1, y1);
}
__device__ __forceinline__ float div1(float c) { // c / (c + 1)
    const float d = c + 1.0f;
    const float y1 = rcp1(d);
    const float q0 = c * y1;
    const float r = __builtin_fmaf(-d, q0, c);
    return __builtin_fmaf(r, y1, q0);
}
__device__ __forceinline__ float sqrt1(float x) { // v_sqrt_f32 + the two one-ulp candidates (the compiler's own core, without scaling)
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = u2f(f2u(s) - 1u), sp = u2f(f2u(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    float r = (rm <= 0.0f) ? sm : s;
    r = (rp > 0.0f) ? sp : r;
    return r;
}

__global__ void probe(unsigned long long *bad, uint32_t *first, float lo, float hi) {
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned long long n[4] = {0, 0, 0, 0};
    for (uint64_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const float x = u2f((uint32_t)i);
        const float ax = __builtin_fabsf(x);
        const bool in = ax >= lo && ax <= hi;
        if (in) {
            const float want = 1.0f / x;
            if (!same(rcp1(x), want)) { if (n[0]++ == 0) first[0] = (uint32_t)i; }
            if (!same(rcp2(x), want)) { if (n[1]++ == 0) first[1] = (uint32_t)i; }
        }
        if (x >= 0.0f && x <= hi) { // colours: 0 and up
            const float want = x / (x + 1.0f);
            if (!same(div1(x), want)) { if (n[2]++ == 0) first[2] = (uint32_t)i; }
        }
        if (x >= lo && x <= hi) {
            const float want = __builtin_sqrtf(x);
            if (!same(sqrt1(x), want)) { if (n[3]++ == 0) first[3] = (uint32_t)i; }
        }
    }
    for (int k = 0; k < 4; k++)
        if (n[k]) atomicAdd(&bad[k], n[k]);
}

int main() {
    unsigned long long *bad;
    uint32_t *first;
    hipMalloc(&bad, 4 * sizeof *bad);
    hipMalloc(&first, 4 * sizeof *first);
    const float ranges[3][2] = {{0x1p-100f, 0x1p100f}, {0x1p-120f, 0x1p120f}, {0x1p-126f, 0x1p126f}};
    for (auto &r : ranges) {
        hipMemset(bad, 0, 4 * sizeof *bad);
        hipMemset(first, 0, 4 * sizeof *first);
        hipLaunchKernelGGL(probe, dim3(4096), dim3(256), 0, 0, bad, first, r[0], r[1]);
        unsigned long long h[4];
        uint32_t f[4];
        hipMemcpy(h, bad, sizeof h, hipMemcpyDeviceToHost);
        hipMemcpy(f, first, sizeof f, hipMemcpyDeviceToHost);
        const char *names[4] = {"1/x: rcp + 1 Newton step", "1/x: rcp + Newton + residual", "c/(c+1): rcp1, q0, residual", "sqrt: v_sqrt + one-ulp candidates"};
        printf("|x| in [%a, %a]:\n", r[0], r[1]);
        for (int k = 0; k < 4; k++) printf("  %-36s %llu inputs differ (first 0x%08x)\n", names[k], h[k], f[k]);
    }
    return 0;
}
