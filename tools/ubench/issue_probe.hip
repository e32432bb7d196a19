// issue_probe.hip — how many scalar and vector instructions a gfx950 CU issues per clock, alone and mixed ("issue slots")
// and whether the scalar unit is shared by the CU's four SIMDs (measurement tool; hipcc --offload-arch=gfx950 -O2 issue_probe.hip -o issue_probe).
// Each wave runs 20 000 iterations of a block of ~32 independent adds (four chains per kind): scalar only, vector only, and mixes
// 1:1, 1:2, 1:3, 2:1.  One or two workgroups per CU (an LDS request makes exactly that many fit) of 1-4 waves per SIMD.  Rates are per
// tick of s_memtime over a wave's own loop (printed beside the 100 MHz counter: the tick is the 2.4 GHz shader clock); the waves of a
// CU do not run side by side for the whole launch (the oldest is served first), so for rates of the WHOLE launch divide the instruction
// totals by the event time instead: vector only 0.42 / clock / SIMD, scalar only 0.92 / clock / CU, 1:2 -> 0.70 per CU + 0.35 per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define S4 "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n"
#define V4 "v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1\n"
#define SV4 "s_add_u32 %0, %0, 1\n v_add_u32 %4, %4, 1\n s_add_u32 %1, %1, 1\n v_add_u32 %5, %5, 1\n s_add_u32 %2, %2, 1\n v_add_u32 %6, %6, 1\n s_add_u32 %3, %3, 1\n v_add_u32 %7, %7, 1\n"
#define SVV4 "s_add_u32 %0, %0, 1\n v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n s_add_u32 %1, %1, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1\n"
#define SVVV4 "s_add_u32 %0, %0, 1\n v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n s_add_u32 %1, %1, 1\n v_add_u32 %7, %7, 1\n v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n"
#define SSV4 "s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n v_add_u32 %4, %4, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n v_add_u32 %5, %5, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 1\n v_add_u32 %6, %6, 1\n s_add_u32 %2, %2, 1\n s_add_u32 %3, %3, 1\n v_add_u32 %7, %7, 1\n"
#define OPS "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)

template <int MODE>
__global__ __launch_bounds__(1024) void probe(unsigned long long *out, int iters) {
    extern __shared__ unsigned lds_hold[]; // sized by the host so that exactly `groups per CU` workgroups fit a CU
    if (iters < 0) lds_hold[threadIdx.x] = 0;
    unsigned s0 = 0, s1 = 1, s2 = 2, s3 = 3;
    unsigned v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3;
    const unsigned long long t0 = __builtin_readcyclecounter(); // s_memtime
    const unsigned long long r0 = wall_clock64();             // s_memrealtime: 100 MHz
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) asm volatile(S4 S4 S4 S4 S4 S4 S4 S4 : OPS : : "scc");
        if (MODE == 1) asm volatile(V4 V4 V4 V4 V4 V4 V4 V4 : OPS : : "scc");
        if (MODE == 2) asm volatile(SV4 SV4 SV4 SV4 : OPS : : "scc");
        if (MODE == 3) asm volatile(SVV4 SVV4 SVV4 SVV4 SVV4 : OPS : : "scc"); // 10 scalar + 20 vector
        if (MODE == 4) asm volatile(SVVV4 SVVV4 SVVV4 SVVV4 : OPS : : "scc");  // 8 scalar + 24 vector
        if (MODE == 5) asm volatile(SSV4 SSV4 SSV4 SSV4 : OPS : : "scc");      // 16 scalar + 8 vector
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = wall_clock64();
    if (v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3 == 0x7fffffffu) out[1] = 1;
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out, t1 - t0); // sum of the waves' loop times
        atomicAdd(out + 2, r1 - r0);
    }
}

template <int MODE>
static void run(const char *label, int scalar_per_iter, int vector_per_iter, unsigned long long *d) {
    const int iters = 20000;
    // {waves per SIMD per workgroup, workgroups per CU}: the LDS request makes exactly that many fit, so every CU holds the same
    const int shapes[][2] = {{1, 1}, {2, 1}, {4, 1}, {3, 2}, {4, 2}};
    for (auto &sh : shapes) {
        const int kk = sh[0], per_cu = sh[1], k = kk * per_cu;
        const size_t lds = per_cu == 1 ? 96 * 1024 : 72 * 1024;
        hipFuncSetAttribute(reinterpret_cast<const void *>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        probe<MODE><<<256 * per_cu, 256 * kk, lds>>>(d, 100);
        hipMemsetAsync(d, 0, 32);
        hipEventRecord(a);
        probe<MODE><<<256 * per_cu, 256 * kk, lds>>>(d, iters);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        unsigned long long sums[4] = {0, 0, 0, 0};
        hipMemcpy(sums, d, 32, hipMemcpyDeviceToHost);
        const unsigned long long sum = sums[0];
        const double real_ms = (double)sums[2] / (256.0 * 4.0 * k) / 100e6 * 1e3; // a wave's loop by the 100 MHz counter
        const double cycles = (double)sum / (256.0 * 4.0 * k); // s_memtime ticks of a wave's loop (all waves run side by side)
        const double waves_per_cu = 4.0 * k;
        const double s_cu = waves_per_cu * iters * scalar_per_iter / cycles, v_simd = k * (double)iters * vector_per_iter / cycles;
        printf("%-22s %d waves/SIMD: %7.3f ms (wave loop %.3f ms), %.2f G ticks/s  scalar %.2f /tick/CU  vector %.2f /tick/SIMD  per SIMD: 2V + S = %.2f\n", label, k, ms, real_ms,
               cycles / (real_ms * 1e6), s_cu, v_simd, 2.0 * v_simd + s_cu / 4.0);
    }
}

int main() {
    unsigned long long *d;
    hipMalloc(&d, 32);
    run<0>("scalar only", 32, 0, d);
    run<1>("vector only", 0, 32, d);
    run<2>("1 scalar : 1 vector", 16, 16, d);
    run<3>("1 scalar : 2 vector", 10, 20, d);
    run<4>("1 scalar : 3 vector", 8, 24, d);
    run<5>("2 scalar : 1 vector", 32, 16, d);
    return 0;
}
