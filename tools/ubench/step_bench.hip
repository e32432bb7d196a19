// Micro-benchmark of candidate DDA-step instruction sequences on gfx950: cycles per trip for ONE wave alone on a
// CU (latency) and for 8 waves per SIMD (throughput).  Build: hipcc --offload-arch=gfx950 -O2 -o step_bench step_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

struct State { float sx, sy, sz, ix, iy, iz; int rx, ry, rz; uint32_t idx, stx, sty, stz; };

#define LOAD_STATE \
    State s = in[threadIdx.x & 63]; \
    float sx = s.sx, sy = s.sy, sz = s.sz; const float ix = s.ix, iy = s.iy, iz = s.iz; \
    int rx = s.rx, ry = s.ry, rz = s.rz; uint32_t idx = s.idx; const uint32_t stx = s.stx, sty = s.sty, stz = s.stz; \
    float ts = 0.f; unsigned long long mx, my, mxy, save, ex, by, bz; int ax = 0; float nx, ny, nz; \
    (void)ax; (void)nx; (void)ny; (void)nz; (void)save; (void)ex; (void)by; (void)bz; (void)mxy; (void)my; (void)mx;

#define STORE_STATE \
    out[blockIdx.x * blockDim.x + threadIdx.x] = sx + sy + sz + ts + (float)(rx + ry + rz) + (float)idx + (float)ax; \
    if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = w1 - w0; }

#define TIMED_LOOP(BODY) \
    unsigned long long w0 = wall_clock64(); unsigned long long t0 = clock64(); \
    for (int i = 0; i < iters; ++i) { BODY } \
    unsigned long long t1 = clock64(); unsigned long long w1 = wall_clock64();

// 0: empty loop (overhead)
__global__ void k_empty(const State *in, float *out, unsigned long long *cyc, int iters) {
    LOAD_STATE
    TIMED_LOOP(asm volatile("" : "+v"(sx));)
    STORE_STATE
}
// 1: select form (22 VALU + 3 SALU)
__global__ void k_select(const State *in, float *out, unsigned long long *cyc, int iters) {
    LOAD_STATE
    TIMED_LOOP(
        asm volatile("v_cmp_lt_f32_e64 %[mx], %[sdx], %[sdy]\n\t"
        "v_cmp_lt_f32_e64 %[mxy], %[sdx], %[sdz]\n\t"
        "v_cmp_lt_f32_e64 %[my], %[sdy], %[sdz]\n\t"
        "v_add_f32_e64 %[nx], %[sdx], |%[ix]|\n\t"
        "v_add_f32_e64 %[ny], %[sdy], |%[iy]|\n\t"
        "v_add_f32_e64 %[nz], %[sdz], |%[iz]|\n\t"
        "s_andn2_b64 %[my], %[my], %[mx]\n\t"
        "s_and_b64 %[mx], %[mx], %[mxy]\n\t"
        "s_or_b64 %[mxy], %[mx], %[my]\n\t"
        "v_cndmask_b32_e64 %[ts], %[sdz], %[sdy], %[my]\n\t"
        "v_cndmask_b32_e64 %[ts], %[ts], %[sdx], %[mx]\n\t"
        "v_cndmask_b32_e64 %[sdx], %[sdx], %[nx], %[mx]\n\t"
        "v_cndmask_b32_e64 %[sdy], %[sdy], %[ny], %[my]\n\t"
        "v_cndmask_b32_e64 %[sdz], %[nz], %[sdz], %[mxy]\n\t"
        "v_subbrev_co_u32_e64 %[rx], %[cc], 0, %[rx], %[mx]\n\t"
        "v_subbrev_co_u32_e64 %[ry], %[cc], 0, %[ry], %[my]\n\t"
        "v_addc_co_u32_e64 %[rz], %[cc], -1, %[rz], %[mxy]\n\t"
        "v_cndmask_b32_e64 %[nx], %[stz], %[sty], %[my]\n\t"
        "v_cndmask_b32_e64 %[nx], %[nx], %[stx], %[mx]\n\t"
        "v_add_u32_e32 %[idx], %[idx], %[nx]\n\t"
        "v_cndmask_b32_e64 %[ax], 2, 1, %[my]\n\t"
        "v_cndmask_b32_e64 %[ax], %[ax], 0, %[mx]"
        : [sdx] "+v"(sx), [sdy] "+v"(sy), [sdz] "+v"(sz), [rx] "+v"(rx), [ry] "+v"(ry), [rz] "+v"(rz),
          [idx] "+v"(idx), [ts] "=&v"(ts), [ax] "=&v"(ax), [nx] "=&v"(nx), [ny] "=&v"(ny), [nz] "=&v"(nz), [mx] "=&s"(mx),
          [my] "=&s"(my), [mxy] "=&s"(mxy), [cc] "=&s"(ex)
        : [ix] "v"(ix), [iy] "v"(iy), [iz] "v"(iz), [stx] "v"(stx), [sty] "v"(sty), [stz] "v"(stz) : "scc");
    )
    STORE_STATE
}
// 2: region form (14 VALU + 10 SALU), as shipped
#define REGION_ASM \
        "v_cmp_lt_f32_e64 %[mx], %[sdx], %[sdy]\n\t" \
        "v_cmp_lt_f32_e64 %[mxy], %[sdx], %[sdz]\n\t" \
        "v_cmp_lt_f32_e64 %[my], %[sdy], %[sdz]\n\t" \
        "s_mov_b64 %[save], exec\n\t" \
        "s_andn2_b64 %[my], %[my], %[mx]\n\t" \
        "s_and_b64 %[mx], %[mx], %[mxy]\n\t" \
        "s_or_b64 %[mxy], %[mx], %[my]\n\t" \
        "v_cndmask_b32_e64 %[ts], %[sdz], %[sdy], %[my]\n\t" \
        "v_cndmask_b32_e64 %[ts], %[ts], %[sdx], %[mx]\n\t" \
        "s_mov_b64 exec, %[mx]\n\t" \
        "v_add_f32_e64 %[sdx], %[sdx], |%[ix]|\n\t" \
        "v_sub_co_u32_e64 %[rx], %[ex], %[rx], 1\n\t" \
        "v_add_u32_e32 %[idx], %[idx], %[stx]\n\t" \
        "s_mov_b64 exec, %[my]\n\t" \
        "v_add_f32_e64 %[sdy], %[sdy], |%[iy]|\n\t" \
        "v_sub_co_u32_e64 %[ry], %[by], %[ry], 1\n\t" \
        "v_add_u32_e32 %[idx], %[idx], %[sty]\n\t" \
        "s_andn2_b64 exec, %[save], %[mxy]\n\t" \
        "v_add_f32_e64 %[sdz], %[sdz], |%[iz]|\n\t" \
        "v_sub_co_u32_e64 %[rz], %[bz], %[rz], 1\n\t" \
        "v_add_u32_e32 %[idx], %[idx], %[stz]\n\t" \
        "s_mov_b64 exec, %[save]\n\t" \
        "s_or_b64 %[ex], %[ex], %[by]\n\t" \
        "s_or_b64 %[ex], %[ex], %[bz]"
#define REGION_OPS \
        : [sdx] "+v"(sx), [sdy] "+v"(sy), [sdz] "+v"(sz), [rx] "+v"(rx), [ry] "+v"(ry), [rz] "+v"(rz), \
          [idx] "+v"(idx), [ts] "=&v"(ts), [mx] "=&s"(mx), [my] "=&s"(my), [mxy] "=&s"(mxy), [save] "=&s"(save), [ex] "=&s"(ex), \
          [by] "=&s"(by), [bz] "=&s"(bz) \
        : [ix] "v"(ix), [iy] "v"(iy), [iz] "v"(iz), [stx] "v"(stx), [sty] "v"(sty), [stz] "v"(stz) : "scc"
__global__ void k_region(const State *in, float *out, unsigned long long *cyc, int iters) {
    LOAD_STATE
    TIMED_LOOP(asm volatile(REGION_ASM REGION_OPS);)
    STORE_STATE
}
// 3: region form, borrow ORs dropped and SALU mask math interleaved with independent VALU
__global__ void k_region_b(const State *in, float *out, unsigned long long *cyc, int iters) {
    LOAD_STATE
    TIMED_LOOP(
        asm volatile("v_cmp_lt_f32_e64 %[mx], %[sdx], %[sdy]\n\t"
        "v_cmp_lt_f32_e64 %[mxy], %[sdx], %[sdz]\n\t"
        "v_cmp_lt_f32_e64 %[my], %[sdy], %[sdz]\n\t"
        "s_mov_b64 %[save], exec\n\t"
        "s_and_b64 %[mxy], %[mx], %[mxy]\n\t"      // x crossed
        "s_mov_b64 exec, %[mxy]\n\t"
        "v_mov_b32_e32 %[ts], %[sdx]\n\t"
        "v_add_f32_e64 %[sdx], %[sdx], |%[ix]|\n\t"
        "v_sub_co_u32_e64 %[rx], %[ex], %[rx], 1\n\t"
        "v_add_u32_e32 %[idx], %[idx], %[stx]\n\t"
        "s_andn2_b64 %[my], %[my], %[mx]\n\t"      // y crossed
        "s_mov_b64 exec, %[my]\n\t"
        "v_mov_b32_e32 %[ts], %[sdy]\n\t"
        "v_add_f32_e64 %[sdy], %[sdy], |%[iy]|\n\t"
        "v_sub_co_u32_e64 %[ry], %[by], %[ry], 1\n\t"
        "v_add_u32_e32 %[idx], %[idx], %[sty]\n\t"
        "s_or_b64 %[mx], %[mxy], %[my]\n\t"
        "s_andn2_b64 exec, %[save], %[mx]\n\t"
        "v_mov_b32_e32 %[ts], %[sdz]\n\t"
        "v_add_f32_e64 %[sdz], %[sdz], |%[iz]|\n\t"
        "v_sub_co_u32_e64 %[rz], %[bz], %[rz], 1\n\t"
        "v_add_u32_e32 %[idx], %[idx], %[stz]\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        "s_or_b64 %[ex], %[ex], %[by]\n\t"
        "s_or_b64 %[ex], %[ex], %[bz]"
        : [sdx] "+v"(sx), [sdy] "+v"(sy), [sdz] "+v"(sz), [rx] "+v"(rx), [ry] "+v"(ry), [rz] "+v"(rz),
          [idx] "+v"(idx), [ts] "+v"(ts), [mx] "=&s"(mx), [my] "=&s"(my), [mxy] "=&s"(mxy), [save] "=&s"(save), [ex] "=&s"(ex),
          [by] "=&s"(by), [bz] "=&s"(bz)
        : [ix] "v"(ix), [iy] "v"(iy), [iz] "v"(iz), [stx] "v"(stx), [sty] "v"(sty), [stz] "v"(stz) : "scc");
    )
    STORE_STATE
}
// 4..: primitive chains, 16 instructions per trip
__global__ void k_chain_valu(const State *in, float *out, unsigned long long *cyc, int iters) {
    LOAD_STATE
    TIMED_LOOP(asm volatile(
        "v_add_f32_e32 %0, %0, %1\n\tv_add_f32_e32 %0, %0, %1\n\tv_add_f32_e32 %0, %0, %1\n\tv_add_f32_e32 %0, %0, %1\n\t"
        "v_add_f32_e32 %0, %0, %1\n\tv_add_f32_e32 %0, %0, %1\n\tv_add_f32_e32 %0, %0, %1\n\tv_add_f32_e32 %0, %0, %1\n\t"
        "v_add_f32_e32 %0, %0, %1\n\tv_add_f32_e32 %0, %0, %1\n\tv_add_f32_e32 %0, %0, %1\n\tv_add_f32_e32 %0, %0, %1\n\t"
        "v_add_f32_e32 %0, %0, %1\n\tv_add_f32_e32 %0, %0, %1\n\tv_add_f32_e32 %0, %0, %1\n\tv_add_f32_e32 %0, %0, %1" : "+v"(sx) : "v"(ix));)
    STORE_STATE
}
__global__ void k_indep_valu(const State *in, float *out, unsigned long long *cyc, int iters) {
    LOAD_STATE
    TIMED_LOOP(asm volatile(
        "v_add_f32_e32 %0, %0, %3\n\tv_add_f32_e32 %1, %1, %3\n\tv_add_f32_e32 %2, %2, %3\n\tv_add_f32_e32 %4, %4, %3\n\t"
        "v_add_f32_e32 %0, %0, %3\n\tv_add_f32_e32 %1, %1, %3\n\tv_add_f32_e32 %2, %2, %3\n\tv_add_f32_e32 %4, %4, %3\n\t"
        "v_add_f32_e32 %0, %0, %3\n\tv_add_f32_e32 %1, %1, %3\n\tv_add_f32_e32 %2, %2, %3\n\tv_add_f32_e32 %4, %4, %3\n\t"
        "v_add_f32_e32 %0, %0, %3\n\tv_add_f32_e32 %1, %1, %3\n\tv_add_f32_e32 %2, %2, %3\n\tv_add_f32_e32 %4, %4, %3" : "+v"(sx), "+v"(sy), "+v"(sz) : "v"(ix), "v"(ts));)
    STORE_STATE
}
__global__ void k_chain_salu(const State *in, float *out, unsigned long long *cyc, int iters) {
    LOAD_STATE
    mx = 1;
    TIMED_LOOP(asm volatile(
        "s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\t"
        "s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\t"
        "s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\t"
        "s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1" : "+s"(ax) : : "scc");)
    STORE_STATE
}
// v_cmp -> s_and -> v_cndmask round trips (4 per trip = 12 instructions)
__global__ void k_chain_mask(const State *in, float *out, unsigned long long *cyc, int iters) {
    LOAD_STATE
    TIMED_LOOP(asm volatile(
        "v_cmp_lt_f32_e64 %[m], %[a], %[b]\n\ts_and_b64 %[m], %[m], exec\n\tv_cndmask_b32_e64 %[a], %[a], %[c], %[m]\n\t"
        "v_cmp_lt_f32_e64 %[m], %[a], %[b]\n\ts_and_b64 %[m], %[m], exec\n\tv_cndmask_b32_e64 %[a], %[a], %[c], %[m]\n\t"
        "v_cmp_lt_f32_e64 %[m], %[a], %[b]\n\ts_and_b64 %[m], %[m], exec\n\tv_cndmask_b32_e64 %[a], %[a], %[c], %[m]\n\t"
        "v_cmp_lt_f32_e64 %[m], %[a], %[b]\n\ts_and_b64 %[m], %[m], exec\n\tv_cndmask_b32_e64 %[a], %[a], %[c], %[m]"
        : [a] "+v"(sx), [m] "=&s"(mx) : [b] "v"(sy), [c] "v"(sz) : "scc");)
    STORE_STATE
}
// v_cmp -> exec -> v_add round trips (4 per trip = 12 instructions + restore)
__global__ void k_chain_exec(const State *in, float *out, unsigned long long *cyc, int iters) {
    LOAD_STATE
    TIMED_LOOP(asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "v_cmp_lt_f32_e64 %[m], %[a], %[b]\n\ts_mov_b64 exec, %[m]\n\tv_add_f32_e32 %[a], %[a], %[c]\n\ts_mov_b64 exec, %[sv]\n\t"
        "v_cmp_lt_f32_e64 %[m], %[a], %[b]\n\ts_mov_b64 exec, %[m]\n\tv_add_f32_e32 %[a], %[a], %[c]\n\ts_mov_b64 exec, %[sv]\n\t"
        "v_cmp_lt_f32_e64 %[m], %[a], %[b]\n\ts_mov_b64 exec, %[m]\n\tv_add_f32_e32 %[a], %[a], %[c]\n\ts_mov_b64 exec, %[sv]\n\t"
        "v_cmp_lt_f32_e64 %[m], %[a], %[b]\n\ts_mov_b64 exec, %[m]\n\tv_add_f32_e32 %[a], %[a], %[c]\n\ts_mov_b64 exec, %[sv]"
        : [a] "+v"(sx), [m] "=&s"(mx), [sv] "=&s"(save) : [b] "v"(sy), [c] "v"(sz));)
    STORE_STATE
}
// L1-hit dependent load + bfe + cmp chain (pointer chase through a small table): 1 load per trip
__global__ void k_chain_load(const State *in, float *out, unsigned long long *cyc, int iters, const uint32_t *table) {
    LOAD_STATE
    TIMED_LOOP(idx = table[idx & 1023u];)
    STORE_STATE
}

typedef void (*kern_t)(const State *, float *, unsigned long long *, int);

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int iters = 4000;
    std::vector<State> h(64);
    for (int i = 0; i < 64; ++i) {
        h[i] = State{0.1f + 0.013f * i, 0.2f + 0.007f * i, 0.15f + 0.011f * i, 1.0f + 0.01f * i, -1.3f, 0.9f, 1 << 30, 1 << 30, 1 << 30, 0u, 1u, 4096u, 64u};
    }
    State *d_in; float *d_out; unsigned long long *d_cyc; uint32_t *d_tab;
    const int max_blocks = 256 * 8;
    if (hipMalloc(&d_in, sizeof(State) * 64) != hipSuccess) { printf("no device\n"); return 2; }
    hipMalloc(&d_out, sizeof(float) * 256 * max_blocks); hipMalloc(&d_cyc, 16 * max_blocks); hipMalloc(&d_tab, 4096);
    std::vector<uint32_t> tab(1024); for (int i = 0; i < 1024; ++i) tab[i] = (i * 37 + 11) & 1023;
    hipMemcpy(d_tab, tab.data(), 4096, hipMemcpyHostToDevice);
    hipMemcpy(d_in, h.data(), sizeof(State) * 64, hipMemcpyHostToDevice);
    struct { const char *name; kern_t k; int instrs; } ks[] = {
        {"empty loop", k_empty, 0}, {"select step 22V+3S", k_select, 25}, {"region step 14V+10S", k_region, 24}, {"region step v_mov ts, interleaved 15V+10S", k_region_b, 25},
        {"16 dependent v_add", k_chain_valu, 16}, {"16 independent v_add (4 chains)", k_indep_valu, 16}, {"16 dependent s_add", k_chain_salu, 16},
        {"4x v_cmp->s_and->v_cndmask", k_chain_mask, 12}, {"4x v_cmp->exec->v_add->restore", k_chain_exec, 17}};
    for (auto &e : ks) {
        for (int mode = 0; mode < 2; ++mode) {
            const int blocks = mode == 0 ? 1 : max_blocks, threads = mode == 0 ? 64 : 256; // 1 wave alone | 8 waves per SIMD on every CU
            e.k<<<blocks, threads>>>(d_in, d_out, d_cyc, 16); hipDeviceSynchronize();
            e.k<<<blocks, threads>>>(d_in, d_out, d_cyc, iters);
            hipDeviceSynchronize();
            std::vector<unsigned long long> c(2 * blocks);
            hipMemcpy(c.data(), d_cyc, 16 * blocks, hipMemcpyDeviceToHost);
            double cyc = 0, wall = 0; for (int b = 0; b < blocks; ++b) { cyc += c[2 * b]; wall += c[2 * b + 1]; }
            cyc /= blocks; wall /= blocks;
            printf("%-44s %-18s clock64 %.1f /trip  wall %.2f ns/trip%s\n", e.name, mode == 0 ? "1 wave alone" : "8 waves/SIMD full", cyc / iters, wall * 10.0 / iters,
                   mode == 1 ? "  (per wave; x1/8 = per-SIMD throughput)" : "");
        }
    }
    for (int mode = 0; mode < 2; ++mode) {
        const int blocks = mode == 0 ? 1 : max_blocks, threads = mode == 0 ? 64 : 256;
        k_chain_load<<<blocks, threads>>>(d_in, d_out, d_cyc, iters, d_tab); hipDeviceSynchronize();
        std::vector<unsigned long long> c(2 * blocks);
        hipMemcpy(c.data(), d_cyc, 16 * blocks, hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0; for (int b = 0; b < blocks; ++b) { cyc += c[2 * b]; wall += c[2 * b + 1]; }
        printf("%-44s %-18s clock64 %.1f /trip  wall %.2f ns/trip\n", "dependent L1-hit global_load chain", mode == 0 ? "1 wave alone" : "8 waves/SIMD full", cyc / blocks / iters, wall / blocks * 10.0 / iters);
    }
    return 0;
}
