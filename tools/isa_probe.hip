// Hardware-assumption probe for the hand-written gfx950 DDA step (vrt_trace.hip, dda_step_regions):
// a VOP3 carry-out (v_sub_co_u32 with an SGPR-pair destination) executed under a partial EXEC mask must
// write ZERO for the inactive lanes' bits, like v_cmp does.  Build: hipcc --offload-arch=gfx950 -O2 -o isa_probe isa_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void probe(const uint32_t *vals, unsigned long long region, unsigned long long *out) {
    uint32_t r = vals[threadIdx.x];
    unsigned long long borrow = ~0ull, cmp = ~0ull, save;
    asm volatile("s_mov_b64 %[save], exec\n\t"
                 "s_mov_b64 %[b], -1\n\t"
                 "s_mov_b64 %[c], -1\n\t"
                 "s_and_b64 exec, %[save], %[reg]\n\t"
                 "v_sub_co_u32_e64 %[r], %[b], %[r], 1\n\t"
                 "v_cmp_eq_u32_e64 %[c], %[r], %[r]\n\t"
                 "s_mov_b64 exec, %[save]"
                 : [r] "+v"(r), [b] "=&s"(borrow), [c] "=&s"(cmp), [save] "=&s"(save)
                 : [reg] "s"(region));
    if (threadIdx.x == 0) {
        out[0] = borrow;
        out[1] = cmp;
    }
    out[2 + threadIdx.x] = r;
}

int main() {
    uint32_t h[64];
    for (int i = 0; i < 64; ++i) h[i] = (i % 3 == 0) ? 0u : (uint32_t)i; // lanes 0,3,6,... borrow when active
    uint32_t *d;
    unsigned long long *o, ho[66];
    if (hipMalloc(&d, sizeof h) != hipSuccess || hipMalloc(&o, sizeof ho) != hipSuccess) { printf("no device\n"); return 2; }
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    const unsigned long long region = 0x00FF00FFF0F0A5A5ull;
    probe<<<1, 64>>>(d, region, o);
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    unsigned long long expect = 0;
    for (int i = 0; i < 64; ++i) if (((region >> i) & 1) && h[i] == 0) expect |= 1ull << i;
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const uint32_t want = ((region >> i) & 1) ? h[i] - 1u : h[i];
        if ((uint32_t)ho[2 + i] != want) bad++;
    }
    printf("borrow %016llx expect %016llx  cmp %016llx expect %016llx  vgpr_mismatch %d\n", ho[0], expect, ho[1], region, bad);
    const bool ok = ho[0] == expect && ho[1] == region && bad == 0;
    printf(ok ? "ISA_PROBE_OK\n" : "ISA_PROBE_FAIL\n");
    return ok ? 0 : 1;
}
