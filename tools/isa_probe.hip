// Hardware-assumption probes for the hand-written gfx950 DDA loop (vrt_trace.hip):
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

// Second assumption: buffer_load_dword ... idxen with a stride-4 resource returns word[index] and ZERO for
// index >= num_records (a lane that has left the grid carries an arbitrary cell index).
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
__global__ void probe_idxen(const uint32_t *words, uint32_t n, const uint32_t *indices, uint32_t *out) {
    const unsigned long long a = (unsigned long long)words;
    u4 r;
    r.x = (uint32_t)a; r.y = (uint32_t)(a >> 32) | (4u << 16); r.z = n; r.w = 0x00020000u;
    uint32_t v;
    asm volatile("buffer_load_dword %0, %1, %2, 0 idxen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(indices[threadIdx.x]), "s"(r));
    out[threadIdx.x] = v;
}

static int check_idxen() {
    const uint32_t n = 100;
    uint32_t hw[128], hi[64], ho[64];
    for (int i = 0; i < 128; ++i) hw[i] = 0xA5000000u + i;   // words [100,128) lie beyond num_records
    for (int i = 0; i < 64; ++i) hi[i] = (i < 40) ? (uint32_t)(i * 2) : (i < 50 ? 99u + (i - 40) : (i < 60 ? 0x07FFFFF0u + i : 0xFFFFFFFFu - i));
    uint32_t *dw, *di, *dout;
    hipMalloc(&dw, sizeof hw); hipMalloc(&di, sizeof hi); hipMalloc(&dout, sizeof ho);
    hipMemcpy(dw, hw, sizeof hw, hipMemcpyHostToDevice); hipMemcpy(di, hi, sizeof hi, hipMemcpyHostToDevice);
    probe_idxen<<<1, 64>>>(dw, n, di, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const uint32_t want = hi[i] < n ? hw[hi[i]] : 0u;
        if (ho[i] != want) { if (bad < 4) printf("idxen lane %d index %u got %08x want %08x\n", i, hi[i], ho[i], want); bad++; }
    }
    printf("idxen mismatches %d\n", bad);
    return bad;
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
    const bool ok = ho[0] == expect && ho[1] == region && bad == 0 && check_idxen() == 0;
    printf(ok ? "ISA_PROBE_OK\n" : "ISA_PROBE_FAIL\n");
    return ok ? 0 : 1;
}
