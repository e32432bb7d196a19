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

// The shipped trip (vrt_trace.hip VRT_TRIP, brick level) on an all-empty bitmap, four trips per back edge.
typedef __attribute__((ext_vector_type(4))) unsigned int u4;
#define TSFULL(TS, MX, MY) "v_cndmask_b32_e64 %[" TS "], %[sdz], %[sdy], %[" MY "]\n\t" "v_cndmask_b32_e64 %[" TS "], %[" TS "], %[sdx], %[" MX "]\n\t"
#define TSNONE(TS, MX, MY)
#define TSPART TSFULL
#define TRIP(TS, MX, MY, MXY, IDX, IDXN, WORD, WORDN)                      \
    "v_cmp_lt_f32_e64 %[" MX "], %[sdx], %[sdy]\n\t"                      \
    "v_cmp_lt_f32_e64 %[" MXY "], %[sdx], %[sdz]\n\t"                     \
    "v_cmp_lt_f32_e64 %[" MY "], %[sdy], %[sdz]\n\t"                      \
    "v_add_f32_e64 %[t0], %[sdx], |%[ix]|\n\t"                            \
    "v_add_f32_e64 %[t1], %[sdy], |%[iy]|\n\t"                            \
    "v_add_f32_e64 %[t2], %[sdz], |%[iz]|\n\t"                            \
    "s_andn2_b64 %[" MY "], %[" MY "], %[" MX "]\n\t"                     \
    "s_and_b64 %[" MX "], %[" MX "], %[" MXY "]\n\t"                      \
    "s_or_b64 %[" MXY "], %[" MX "], %[" MY "]\n\t"                       \
    TSPART(TS, MX, MY)                                                    \
    "v_cndmask_b32_e64 %[sdx], %[sdx], %[t0], %[" MX "]\n\t"              \
    "v_cndmask_b32_e64 %[sdy], %[sdy], %[t1], %[" MY "]\n\t"              \
    "v_cndmask_b32_e64 %[sdz], %[t2], %[sdz], %[" MXY "]\n\t"             \
    "v_cndmask_b32_e64 %[t0], %[stz], %[sty], %[" MY "]\n\t"              \
    "v_cndmask_b32_e64 %[t0], %[t0], %[stx], %[" MX "]\n\t"               \
    "v_add_u32_e32 %[" IDXN "], %[" IDX "], %[t0]\n\t"                    \
    "v_lshrrev_b32_e32 %[t2], 5, %[" IDXN "]\n\t"                         \
    LOADPART(WORDN)                                                       \
    "v_bfe_u32 %[t1], %[" WORD "], %[" IDX "], 1\n\t"                     \
    "v_cmp_ne_u32_e32 vcc, 0, %[t1]\n\t"                                  \
    "v_subbrev_co_u32_e64 %[rx], %[ex], 0, %[rx], %[" MX "]\n\t"          \
    "v_subbrev_co_u32_e64 %[ry], %[by], 0, %[ry], %[" MY "]\n\t"          \
    "v_addc_co_u32_e64 %[rz], %[cz], -1, %[rz], %[" MXY "]\n\t"           \
    "s_or_b64 %[ex], %[ex], %[by]\n\t"                                    \
    "s_orn2_b64 %[ex], %[ex], %[cz]\n\t"                                  \
    EXECPART                                                              \
    BRANCHPART
#define KFULL(NAME) \
__global__ void NAME(const State *in, float *out, unsigned long long *cyc, int iters, const uint32_t *bitmap, uint32_t nwords) { \
    LOAD_STATE \
    const unsigned long long a = (unsigned long long)bitmap; \
    u4 rsrc; rsrc.x = (uint32_t)a; rsrc.y = (uint32_t)(a >> 32) | (4u << 16); rsrc.z = nwords; rsrc.w = 0x00020000u; \
    uint32_t idxa = idx & 0xFFFFu, idxb = 0, worda = 0, wordb = 0; float tsa = 0, tsb = 0, t0, t1, t2; \
    unsigned long long mxa, mya, mxya, mxb, myb, mxyb, cz; \
    int n = iters / 4; \
    unsigned long long w0 = wall_clock64(); unsigned long long tt0 = clock64(); \
    asm volatile("s_mov_b64 %[save], exec\n\t" \
        "buffer_load_dword %[worda], %[idxa], %[rsrc], 0 idxen\n\t" \
        "0:\n\t" \
        TRIP("tsa", "mxa", "mya", "mxya", "idxa", "idxb", "worda", "wordb") \
        TRIP("tsb", "mxb", "myb", "mxyb", "idxb", "idxa", "wordb", "worda") \
        TRIP("tsa", "mxa", "mya", "mxya", "idxa", "idxb", "worda", "wordb") \
        TRIP("tsb", "mxb", "myb", "mxyb", "idxb", "idxa", "wordb", "worda") \
        "s_sub_u32 %[n], %[n], 1\n\t" \
        "s_cmp_lg_u32 %[n], 0\n\t" \
        "s_cbranch_scc1 0b\n\t" \
        "9:\n\t" \
        "s_waitcnt vmcnt(0)\n\t" \
        "s_mov_b64 exec, %[save]" \
        : [sdx] "+v"(sx), [sdy] "+v"(sy), [sdz] "+v"(sz), [rx] "+v"(rx), [ry] "+v"(ry), [rz] "+v"(rz), [idxa] "+v"(idxa), [idxb] "+v"(idxb), \
          [worda] "+v"(worda), [wordb] "+v"(wordb), [tsa] "+v"(tsa), [tsb] "+v"(tsb), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), \
          [mxa] "=&s"(mxa), [mya] "=&s"(mya), [mxya] "=&s"(mxya), [mxb] "=&s"(mxb), [myb] "=&s"(myb), [mxyb] "=&s"(mxyb), [ex] "=&s"(ex), [by] "=&s"(by), \
          [cz] "=&s"(cz), [save] "=&s"(save), [n] "+s"(n) \
        : [ix] "v"(ix), [iy] "v"(iy), [iz] "v"(iz), [stx] "v"(stx), [sty] "v"(sty), [stz] "v"(stz), [rsrc] "s"(rsrc) \
        : "vcc", "scc"); \
    unsigned long long t1c = clock64(); unsigned long long w1 = wall_clock64(); \
    ts = tsa + tsb; idx = idxa + idxb + worda + wordb; \
    unsigned long long t0c = tt0; \
    out[blockIdx.x * blockDim.x + threadIdx.x] = sx + sy + sz + ts + (float)(rx + ry + rz) + (float)idx; \
    if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t1c - t0c; cyc[blockIdx.x * 2 + 1] = w1 - w0; } \
}

#define LOADPART(WORDN) "buffer_load_dword %[" WORDN "], %[t2], %[rsrc], 0 idxen\n\t" "s_waitcnt vmcnt(1)\n\t"
#define EXECPART "s_andn2_b64 exec, exec, %[ex]\n\t"
#define BRANCHPART "s_cbranch_vccnz 9f\n\t"
KFULL(k_full)
#undef LOADPART
#define LOADPART(WORDN)
KFULL(k_full_noload)
#undef LOADPART
#define LOADPART(WORDN) "buffer_load_dword %[" WORDN "], %[t2], %[rsrc], 0 idxen\n\t" "s_waitcnt vmcnt(1)\n\t"
#undef EXECPART
#define EXECPART
KFULL(k_full_noexec)
#undef EXECPART
#define EXECPART "s_andn2_b64 exec, exec, %[ex]\n\t"
#undef BRANCHPART
#define BRANCHPART
KFULL(k_full_nobranch)
#undef LOADPART
#define LOADPART(WORDN)
#undef EXECPART
#define EXECPART
KFULL(k_full_bare)
#undef LOADPART
#undef EXECPART
#undef BRANCHPART
#define LOADPART(WORDN) "buffer_load_dword %[" WORDN "], %[t2], %[rsrc], 0 idxen\n\t" "s_waitcnt vmcnt(1)\n\t"
#define EXECPART "s_andn2_b64 exec, exec, %[ex]\n\t"
#define BRANCHPART "s_cbranch_vccnz 9f\n\t"
#undef TSPART
#define TSPART TSNONE
KFULL(k_full_nots)
#undef TSPART
#define TSPART TSFULL

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
    {
        // address pattern of the trip's load: scattered (default state), every lane the same word, 8x8 block of neighbouring cells
        std::vector<State> h2(64), h3(64);
        for (int i = 0; i < 64; ++i) { h2[i] = h[i]; h2[i].stx = h2[i].sty = h2[i].stz = 0; h2[i].idx = 4096; }
        for (int i = 0; i < 64; ++i) { h3[i] = h[i]; h3[i].stx = h3[i].sty = h3[i].stz = 0; h3[i].idx = 64 * 64 * 8 + (i & 7) * 64 + (i >> 3) * 4096 + 17; } // 8 z-rows x 8 y-layers
        State *d_in2, *d_in3; hipMalloc(&d_in2, sizeof(State) * 64); hipMalloc(&d_in3, sizeof(State) * 64);
        hipMemcpy(d_in2, h2.data(), sizeof(State) * 64, hipMemcpyHostToDevice); hipMemcpy(d_in3, h3.data(), sizeof(State) * 64, hipMemcpyHostToDevice);
        uint32_t *d_bm2; hipMalloc(&d_bm2, 8192 * 4); hipMemset(d_bm2, 0, 8192 * 4);
        // more patterns inside ONE y-layer: 2 / 8 / 32 distinct words of the same 128-byte line, and 2 y-layers x 2 words
        State *d_more[4]; const int nw[4] = {2, 8, 32, 4};
        for (int m = 0; m < 4; ++m) {
            std::vector<State> hm(64);
            for (int i = 0; i < 64; ++i) {
                hm[i] = h[i]; hm[i].stx = hm[i].sty = hm[i].stz = 0;
                hm[i].idx = (m < 3) ? 64 * 64 * 8 + (i % nw[m]) * 32 + 5 : 64 * 64 * 8 + (i & 1) * 64 + ((i >> 1) & 1) * 4096 + 5;
            }
            hipMalloc(&d_more[m], sizeof(State) * 64); hipMemcpy(d_more[m], hm.data(), sizeof(State) * 64, hipMemcpyHostToDevice);
        }
        const State *ins[7] = {d_in, d_in2, d_in3, d_more[0], d_more[1], d_more[2], d_more[3]};
        const char *pn[7] = {"scattered/out of range", "all lanes one word", "8 z-rows x 8 y-layers (8 lines)", "2 words of one line", "8 words of one line",
                             "32 words of one line", "2 y-layers x 2 z-rows (2 lines)"};
        for (int q = 0; q < 7; ++q) {
            k_full<<<256 * 6, 256>>>(ins[q], d_out, d_cyc, 16, d_bm2, 8192); hipDeviceSynchronize();
            k_full<<<256 * 6, 256>>>(ins[q], d_out, d_cyc, iters, d_bm2, 8192); hipDeviceSynchronize();
            const int blocks = 256 * 6; std::vector<unsigned long long> c(2 * blocks);
            hipMemcpy(c.data(), d_cyc, 16 * blocks, hipMemcpyDeviceToHost);
            double cyc = 0; for (int b = 0; b < blocks; ++b) cyc += c[2 * b];
            printf("shipped trip, 6 waves/SIMD, load pattern %-34s -> %.1f cycles per trip per SIMD\n", pn[q], cyc / blocks / iters / 6);
        }
    }
    {
        uint32_t *d_bm; const uint32_t nwords = 8192; hipMalloc(&d_bm, nwords * 4); hipMemset(d_bm, 0, nwords * 4);
        // strides small so the index stays inside the bitmap for a while, then reads 0 out of range
        const int cfgs[4][2] = {{1, 64}, {256 * 4, 256}, {256 * 6, 256}, {256 * 8, 256}};
        const char *names[4] = {"1 wave alone", "4 waves/SIMD", "6 waves/SIMD", "8 waves/SIMD"};
        typedef void (*kf_t)(const State *, float *, unsigned long long *, int, const uint32_t *, uint32_t);
        struct { const char *name; kf_t k; } fam[] = {{"shipped brick-level trip (29 instr)", k_full}, {"  without load+waitcnt", k_full_noload}, {"  without exec update", k_full_noexec},
                                                       {"  without vccnz branch", k_full_nobranch}, {"  without load, exec update, branch", k_full_bare}, {"  full trip without the two t-select instructions", k_full_nots}};
        for (auto &f : fam)
        for (int m = 0; m < 4; m += (m == 0 ? 2 : 1)) {
            f.k<<<cfgs[m][0], cfgs[m][1]>>>(d_in, d_out, d_cyc, 16, d_bm, nwords); hipDeviceSynchronize();
            f.k<<<cfgs[m][0], cfgs[m][1]>>>(d_in, d_out, d_cyc, iters, d_bm, nwords); hipDeviceSynchronize();
            const int blocks = cfgs[m][0];
            std::vector<unsigned long long> c(2 * blocks);
            hipMemcpy(c.data(), d_cyc, 16 * blocks, hipMemcpyDeviceToHost);
            double cyc = 0, wall = 0; for (int b = 0; b < blocks; ++b) { cyc += c[2 * b]; wall += c[2 * b + 1]; }
            const int waves_per_simd = m == 0 ? 1 : (m == 1 ? 4 : (m == 2 ? 6 : 8));
            printf("%-44s %-14s clock64 %.1f /trip/wave -> %.1f cycles, %.2f ns per trip per SIMD\n", f.name, names[m],
                   cyc / blocks / iters, cyc / blocks / iters / waves_per_simd, wall / blocks * 10.0 / iters / waves_per_simd);
        }
    }
    return 0;
}
