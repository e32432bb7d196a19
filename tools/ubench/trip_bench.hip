// trip_bench.hip — cycles per trip of the brick-level walk loops (the production asm of vrt_trace_kernels.h and candidate
// re-formulations) on rays that never meet an occupied cell, for 1 wave per CU (pure latency of a wave's dependent chain) up to 8
// waves per SIMD (throughput).  Round 4: the persistent-wave kernels and the tail waves of bounce frames issue one instruction per
// ~11 cycles per wave whatever the occupancy — a wave is one chain of dependent instructions — so the LATENCY of a trip is what their
// time is made of, where the headline kernel (7 waves per SIMD) pays for issue slots.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I zig_vulkan_amd/csrc -o tools/ubench/trip_bench tools/ubench/trip_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "vrt_trace_kernels.h"

using namespace vrt;

struct Out {
    unsigned long long cycles, trips;
};

// VARIANT 0: grid_walk_gfx950 (status words)   1: grid_walk_bytes_gfx950 (a byte per cell)   2: grid_walk_park_gfx950 (words, parking)
// 3: grid_walk_park_dilated_carry_gfx950 (half-block words on the dilated index, no counters: the pool kernel's loop)
template <int VARIANT>
__global__ __launch_bounds__(256) void k_walk(const uint32_t *status, uint32_t status_words, int steps, Out *out) {
    const uint32_t lane = threadIdx.x & 63u;
    Walk w;
    // every lane the same ray (+ a lane-dependent epsilon that changes no comparison): the wave's trips = the lane's
    w.side_dist = mk3(0.37f, 0.81f, 0.55f);
    const f3 inv = mk3(1.9f, -2.3f, 1.4f);
    w.rx = w.ry = w.rz = steps;
    w.t_value = 0.0f;
    uint32_t index = 5u + (lane & 3u), cell = 0u, word = 0u;
    const unsigned long long a = (unsigned long long)status;
    u32x4 rsrc;
    rsrc.x = (uint32_t)a;
    rsrc.y = (uint32_t)(a >> 32) | ((VARIANT == 1) ? 0u : (4u << 16));
    rsrc.z = (VARIANT == 1) ? status_words * 4u : status_words;
    rsrc.w = 0x00020000u;
    unsigned long long t0 = 0, t1 = 0;
    uint32_t trips = 0;
    if constexpr (VARIANT == 0 || VARIANT == 1) {
        GridWalkRegs g;
        g.alive = ~0ull;
        g.out_x = g.out_y = 0ull;
        g.t_out = 0.0f;
        t0 = __builtin_readcyclecounter();
        if constexpr (VARIANT == 0) grid_walk_gfx950(w, inv, index, cell, 1u, 64u * 64u, 64u, word, rsrc, g);
        else grid_walk_bytes_gfx950(w, inv, index, cell, 1u, 64u * 64u, 64u, word, rsrc, g);
        t1 = __builtin_readcyclecounter();
        trips = (uint32_t)((steps - w.rx) + (steps - w.ry) + (steps - w.rz));
    } else if constexpr (VARIANT == 2) {
        GridParkRegs g;
        g.alive = ~0ull;
        g.out_x = g.out_y = 0ull;
        g.t_out = g.t_in = 0.0f;
        g.code = 3u << 4;
        g.batch = 64u;
        g.min_alive = 0u;
        t0 = __builtin_readcyclecounter();
        grid_walk_park_gfx950(w, inv, index, cell, 1u, 64u * 64u, 64u, word, rsrc, g);
        t1 = __builtin_readcyclecounter();
        trips = (uint32_t)((steps - w.rx) + (steps - w.ry) + (steps - w.rz));
    } else if constexpr (VARIANT == 4) {
        // the park loop two trips ahead (grid_walk_ahead_gfx950): a ring of three cells per lane, two requests in flight
        AheadRing ring{index, index + 1u, index + 2u, 0u, 0u, 0u, 0u, 2u, 0.0f, 0.0f, 0.0f};
        AheadWalkRegs g;
        g.alive = ~0ull;
        g.batch = 64u;
        g.min_alive = 0u;
        t0 = __builtin_readcyclecounter();
        grid_walk_ahead_gfx950(w, inv, ring, 1u, 64u * 64u, 64u, rsrc, g);
        t1 = __builtin_readcyclecounter();
        trips = (uint32_t)((steps - w.rx) + (steps - w.ry) + (steps - w.rz));
        index = ring.q0 + ring.w0;
    } else {
        // dilated index of a 2^k-cell grid: x field bits 0-1 + 5.., z bits 2-3 + .., y bit 4 + ..; the walk ends when a field overflows
        const uint32_t lx = 10u, lz = 10u, ly = 10u; // 1024^3 cells: `steps` trips never reach a face from the middle
        const uint32_t fx = 3u | (((1u << (lx - 2u)) - 1u) << 5), fz = (3u << 2) | (((1u << (lz - 2u)) - 1u) << (lx + 3u)),
                       fy = (1u << 4) | (((1u << (ly - 1u)) - 1u) << (lx + lz + 1u));
        (void)fy;
        GridParkRegs g;
        g.alive = ~0ull;
        g.out_x = g.out_y = 0ull;
        g.t_out = g.t_in = 0.0f;
        g.code = 3u << 4;
        g.batch = 64u;
        g.min_alive = 0u;
        // start at the far corner minus `steps` cells on x so that the walk ends by overflow of the x field after ~steps x-steps
        const uint32_t mx = (1u << lx) - 1u - (uint32_t)steps, mz = 8u, my = 8u;
        index = (mx & 3u) | ((mz & 3u) << 2) | ((my & 1u) << 4) | ((mx >> 2) << 5) | ((mz >> 2) << (lx + 3u)) | ((my >> 1) << (lx + lz + 1u));
        unsigned long long gone = 0ull;
        const f3 sd0 = w.side_dist;
        t0 = __builtin_readcyclecounter();
        grid_walk_park_dilated_carry_gfx950(w.side_dist, inv, index, cell, ~fx, ~fy, ~fz, word, rsrc, g, 0u, gone);
        t1 = __builtin_readcyclecounter();
        // trips = additions made to the three side distances
        trips = (uint32_t)((w.side_dist.x - sd0.x) / 1.9f + (w.side_dist.y - sd0.y) / 2.3f + (w.side_dist.z - sd0.z) / 1.4f + 0.5f);
    }
    if (lane == 0u) {
        Out &o = out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)];
        o.cycles = t1 - t0;
        o.trips = trips;
    }
    if (index == 0xFFFFFFFFu && word == 12345u) out[0].trips = cell; // (keep the results alive)
}

// ---- candidate: the trip with its selects on the vector unit (round 4) ---------------------------------------------------------
// One chain of dependent instructions per wave: what a trip costs a wave that has its SIMD (nearly) to itself is the LENGTH of
// that chain.  The production trip goes VALU -> SGPR -> three scalar mask operations -> EXEC -> VALU for the side-distance update
// (8 instructions deep, three crossings between the units).  Here the three candidate side distances are formed before the minimum
// is known and chosen by selects under the two compare masks: min3 -> compare -> select, 3-4 deep, no scalar instruction on the
// chain.  19 vector + 7 scalar instructions instead of 13 + 11 (45 issue slots for 37): for kernels bound by latency, not by slots.
#define TRIPV(IDX, IDXN, WORD, WORDN, OUT)                                     \
    "v_min3_f32 %[ts], %[sdx], %[sdy], %[sdz]\n\t"                             \
    "v_add_f32_e64 %[nx], %[sdx], |%[ix]|\n\t"                                 \
    "v_add_f32_e64 %[ny], %[sdy], |%[iy]|\n\t"                                 \
    "v_add_f32_e64 %[nz], %[sdz], |%[iz]|\n\t"                                 \
    "v_cmp_eq_f32_e64 %[mz], %[sdz], %[ts]\n\t"                                \
    "v_cmp_eq_f32_e64 %[my], %[sdy], %[ts]\n\t"                                \
    "v_cndmask_b32_e64 %[t0], %[stx], %[sty], %[my]\n\t"                       \
    "v_cndmask_b32_e64 %[t0], %[t0], %[stz], %[mz]\n\t"                        \
    "v_add_u32_e32 %[" IDXN "], %[" IDX "], %[t0]\n\t"                         \
    "buffer_load_ubyte %[" WORDN "], %[" IDXN "], %[rsrc], 0 offen\n\t"        \
    "v_cndmask_b32_e64 %[sdz], %[sdz], %[nz], %[mz]\n\t"                       \
    "v_cndmask_b32_e64 %[t1], %[sdy], %[ny], %[my]\n\t"                        \
    "v_cndmask_b32_e64 %[sdy], %[t1], %[sdy], %[mz]\n\t"                       \
    "v_cndmask_b32_e64 %[t1], %[nx], %[sdx], %[my]\n\t"                        \
    "v_cndmask_b32_e64 %[sdx], %[t1], %[sdx], %[mz]\n\t"                       \
    "s_andn2_b64 %[my], %[my], %[mz]\n\t"                                      \
    "s_or_b64 %[mx], %[my], %[mz]\n\t"                                         \
    "s_andn2_b64 %[mx], exec, %[mx]\n\t"                                       \
    "v_subbrev_co_u32_e64 %[rx], %[ex], 0, %[rx], %[mx]\n\t"                   \
    "v_subbrev_co_u32_e64 %[ry], %[by], 0, %[ry], %[my]\n\t"                   \
    "v_subbrev_co_u32_e64 %[rz], %[cz], 0, %[rz], %[mz]\n\t"                   \
    "s_waitcnt vmcnt(1)\n\t"                                                   \
    "v_cmp_ne_u32_e32 vcc, 0, %[" WORD "]\n\t"                                 \
    "s_or_b64 %[ex], %[ex], %[by]\n\t"                                         \
    "s_or_b64 %[ex], %[ex], %[cz]\n\t"                                         \
    "s_andn2_b64 exec, exec, %[ex]\n\t"                                        \
    "s_cbranch_vccnz " OUT "\n\t"

__global__ __launch_bounds__(256) void k_tripv(const uint32_t *status, uint32_t status_words, int steps, Out *out) {
    const uint32_t lane = threadIdx.x & 63u;
    float sdx = 0.37f, sdy = 0.81f, sdz = 0.55f;
    const float ix = 1.9f, iy = -2.3f, iz = 1.4f;
    int rx = steps, ry = steps, rz = steps;
    uint32_t idxa = 5u + (lane & 3u), idxb = 0u, worda = 0u, wordb = 0u;
    const uint32_t stx = 1u, sty = 64u * 64u, stz = 64u;
    const unsigned long long a = (unsigned long long)status;
    u32x4 rsrc;
    rsrc.x = (uint32_t)a;
    rsrc.y = (uint32_t)(a >> 32);
    rsrc.z = status_words * 4u;
    rsrc.w = 0x00020000u;
    float ts, nx, ny, nz, t1;
    uint32_t t0;
    unsigned long long mx, my, mz, ex, by, cz, save;
    const unsigned long long c0 = __builtin_readcyclecounter();
    asm volatile("s_mov_b64 %[save], exec\n\t"
                 "0:\n\t"
                 TRIPV("idxa", "idxb", "worda", "wordb", "1f")
                 TRIPV("idxb", "idxa", "wordb", "worda", "1f")
                 TRIPV("idxa", "idxb", "worda", "wordb", "1f")
                 TRIPV("idxb", "idxa", "wordb", "worda", "1f")
                 "s_cbranch_execnz 0b\n\t"
                 "1:\n\t"
                 "s_waitcnt vmcnt(0)\n\t"
                 "s_mov_b64 exec, %[save]"
                 : [sdx] "+v"(sdx), [sdy] "+v"(sdy), [sdz] "+v"(sdz), [rx] "+v"(rx), [ry] "+v"(ry), [rz] "+v"(rz), [idxa] "+v"(idxa), [idxb] "+v"(idxb),
                   [worda] "+v"(worda), [wordb] "+v"(wordb), [ts] "=&v"(ts), [nx] "=&v"(nx), [ny] "=&v"(ny), [nz] "=&v"(nz), [t0] "=&v"(t0), [t1] "=&v"(t1),
                   [mx] "=&s"(mx), [my] "=&s"(my), [mz] "=&s"(mz), [ex] "=&s"(ex), [by] "=&s"(by), [cz] "=&s"(cz), [save] "=&s"(save)
                 : [ix] "v"(ix), [iy] "v"(iy), [iz] "v"(iz), [stx] "v"(stx), [sty] "v"(sty), [stz] "v"(stz), [rsrc] "s"(rsrc)
                 : "vcc", "scc");
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (lane == 0u) {
        Out &o = out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)];
        o.cycles = c1 - c0;
        o.trips = (unsigned long long)((steps - rx) + (steps - ry) + (steps - rz));
    }
    if (idxa == 0xFFFFFFFFu && worda == 12345u) out[0].trips = idxb + (uint32_t)(sdx + sdy + sdz);
}

template <int VARIANT>
static void run(const char *label, const uint32_t *d_status, uint32_t words, Out *d_out, int steps) {
    // waves per SIMD: blocks of 256 threads = one wave on each SIMD of a CU; LDS pins how many blocks share a CU
    const int configs[][2] = {{1, 0}, {256, 1}, {256, 2}, {256, 4}, {256, 5}, {256, 8}}; // {blocks (1: a single WAVE), blocks per CU}
    for (auto &cfgp : configs) {
        const int per_cu = cfgp[1];
        const int blocks = cfgp[0] == 1 ? 1 : 256 * per_cu;
        const int threads = cfgp[0] == 1 ? 64 : 256;
        const size_t lds = per_cu ? (size_t)(160 * 1024 / per_cu / 1024) * 1024 - 1024 : 0;
        hipFuncSetAttribute((const void *)k_walk<(VARIANT == 100 ? 0 : VARIANT)>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (VARIANT == 100) hipFuncSetAttribute((const void *)k_tripv, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int rep = 0; rep < 2; rep++) {
            if (VARIANT == 100) hipLaunchKernelGGL(k_tripv, dim3(blocks), dim3(threads), lds, 0, d_status, words, steps, d_out);
            else hipLaunchKernelGGL(k_walk<(VARIANT == 100 ? 0 : VARIANT)>, dim3(blocks), dim3(threads), lds, 0, d_status, words, steps, d_out);
        }
        hipDeviceSynchronize();
        const int waves = blocks * threads / 64;
        std::vector<Out> h(waves);
        hipMemcpy(h.data(), d_out, waves * sizeof(Out), hipMemcpyDeviceToHost);
        double cyc = 0, trips = 0;
        for (auto &o : h) cyc += (double)o.cycles, trips += (double)o.trips;
        printf("%-44s %s: %7.1f cycles per trip (%.0f trips per wave)\n", label,
               cfgp[0] == 1 ? "1 wave alone      " : (per_cu == 1 ? "1 wave per SIMD   " : (per_cu == 2 ? "2 waves per SIMD  " : (per_cu == 4 ? "4 waves per SIMD  " : (per_cu == 5 ? "5 waves per SIMD  " : "8 waves per SIMD  ")))),
               cyc / trips, trips / waves);
    }
}

int main() {
    const uint32_t words = 64u * 64u * 64u / 32u; // a 64^3-cell grid's status bits (32 KiB): L1 / L2 resident
    uint32_t *d_status;
    Out *d_out;
    hipMalloc(&d_status, words * 4u * 32u);
    hipMemset(d_status, 0, words * 4u * 32u);
    hipMalloc(&d_out, sizeof(Out) * 256 * 8 * 4);
    const int steps = 600;
    run<0>("grid_walk_gfx950 (words)", d_status, words, d_out, steps);
    run<1>("grid_walk_bytes_gfx950 (byte per cell)", d_status, words, d_out, steps);
    run<2>("grid_walk_park_gfx950 (words, parking)", d_status, words, d_out, steps);
    run<3>("grid_walk_park_dilated_carry_gfx950 (pool)", d_status, words, d_out, steps);
    run<4>("grid_walk_ahead_gfx950 (words, two ahead)", d_status, words, d_out, steps);
    run<100>("candidate: selects on the vector unit (bytes)", d_status, words, d_out, steps);
    return 0;
}
