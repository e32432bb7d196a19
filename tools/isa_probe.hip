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

// Third assumption (skip_to_box): v_cmpx narrows EXEC for the very next vector instruction and for s_cbranch_execnz, with no
// wait states inserted by hand; a carry-out under partial EXEC narrows it the same way.  The two loops of vrt_trace.hip, verbatim
// in structure, against plain C loops.
__global__ void probe_skip(const float *c0, const float *d, const float *lim, const int *left, const int *count, float *c_out, int *n_out, float *t_out) {
    float c = c0[threadIdx.x], t = c0[threadIdx.x];
    int n = 0, cnt = count[threadIdx.x];
    unsigned long long save;
#define STEP "v_cmpx_le_f32_e32 vcc, %[c], %[lim]\n\tv_add_f32_e64 %[c], %[c], |%[d]|\n\tv_add_u32_e32 %[n], 1, %[n]\n\t"
    asm volatile("s_mov_b64 %[save], exec\n\t"
                 "0:\n\t" STEP STEP STEP STEP
                 "v_cmpx_ge_i32_e32 vcc, %[r], %[n]\n\t"
                 "s_cbranch_execnz 0b\n\t"
                 "s_mov_b64 exec, %[save]"
                 : [c] "+v"(c), [n] "+v"(n), [save] "=&s"(save)
                 : [d] "v"(d[threadIdx.x]), [lim] "v"(lim[threadIdx.x]), [r] "v"(left[threadIdx.x])
                 : "vcc", "scc");
#undef STEP
#define STEP "v_add_co_u32_e32 %[n], vcc, -1, %[n]\n\ts_and_b64 exec, exec, vcc\n\tv_add_f32_e64 %[t], %[t], |%[d]|\n\t"
    asm volatile("s_mov_b64 %[save], exec\n\t"
                 "0:\n\t" STEP STEP STEP STEP
                 "s_cbranch_execnz 0b\n\t"
                 "s_mov_b64 exec, %[save]"
                 : [t] "+v"(t), [n] "+v"(cnt), [save] "=&s"(save)
                 : [d] "v"(d[threadIdx.x])
                 : "vcc", "scc");
#undef STEP
    c_out[threadIdx.x] = c;
    n_out[threadIdx.x] = n;
    t_out[threadIdx.x] = t;
}

static int check_skip_loops() {
    float hc[64], hd[64], hl[64], oc[64], ot[64];
    int hr[64], hk[64], on[64];
    uint32_t seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (seed >> 8) * (1.0f / 16777216.0f); };
    for (int i = 0; i < 64; ++i) {
        hd[i] = (i & 1 ? -1.0f : 1.0f) * (1.0f + 3.0f * rnd());   // sign ignored (|d|)
        hc[i] = rnd() * 1.5f;
        hl[i] = (i % 7 == 0) ? hc[i] - 0.25f : hc[i] + 40.0f * rnd(); // some lanes consume nothing
        hr[i] = (i % 5 == 0) ? (int)(3 * rnd()) : 64;                 // some lanes run out of steps
        hk[i] = (i % 4 == 0) ? 0 : (int)(37 * rnd());
    }
    float *dc, *dd, *dl, *doc, *dot; int *dr, *dk, *don;
    hipMalloc(&dc, 256); hipMalloc(&dd, 256); hipMalloc(&dl, 256); hipMalloc(&doc, 256); hipMalloc(&dot, 256);
    hipMalloc(&dr, 256); hipMalloc(&dk, 256); hipMalloc(&don, 256);
    hipMemcpy(dc, hc, 256, hipMemcpyHostToDevice); hipMemcpy(dd, hd, 256, hipMemcpyHostToDevice); hipMemcpy(dl, hl, 256, hipMemcpyHostToDevice);
    hipMemcpy(dr, hr, 256, hipMemcpyHostToDevice); hipMemcpy(dk, hk, 256, hipMemcpyHostToDevice);
    probe_skip<<<1, 64>>>(dc, dd, dl, dr, dk, doc, don, dot);
    hipMemcpy(oc, doc, 256, hipMemcpyDeviceToHost); hipMemcpy(on, don, 256, hipMemcpyDeviceToHost); hipMemcpy(ot, dot, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        volatile float c = hc[i], t = hc[i];
        const float ad = hd[i] < 0 ? -hd[i] : hd[i];
        int n = 0;
        while (n <= hr[i] && c <= hl[i]) { c = c + ad; n++; }
        for (int k = 0; k < hk[i]; ++k) t = t + ad;
        const bool dead = n > hr[i];
        const bool ok = (dead ? on[i] > hr[i] : (on[i] == n && oc[i] == c)) && ot[i] == t;
        if (!ok) { if (bad < 4) printf("skip lane %d: n %d want %d  c %g want %g  t %g want %g\n", i, on[i], n, oc[i], (float)c, ot[i], (float)t); bad++; }
    }
    printf("skip-loop mismatches %d\n", bad);
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
    const bool ok = ho[0] == expect && ho[1] == region && bad == 0 && check_idxen() == 0 && check_skip_loops() == 0;
    printf(ok ? "ISA_PROBE_OK\n" : "ISA_PROBE_FAIL\n");
    return ok ? 0 : 1;
}
