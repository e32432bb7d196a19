// vrt_pool_kernel.h — frames with bounces on scenes larger than the caches, round 4: every wave keeps a POOL of rays.
// Included by vrt_inst_path.hip only.
#pragma once
//
// vrt_path_kernel ties a ray to a lane: a lane whose ray waits for a brick round or for its next ray idles while its neighbours
// walk, and the rounds it waits for run for the lanes that happen to wait — measured on the 2048^3 path trace
// (profiles/r03_cfg4_phase_profile_dilated.txt): 54 lanes enter a call of the walk loop and 26 still move when it returns, a brick
// round serves 25 lanes, a round of transitions 24, and a wave instruction costs the same for one lane as for 64.  Every batching
// parameter of that kernel sits on a plateau; what is left is to take the ray OFF the lane.
//
// Here a wave owns 128 paths.  64 rays are in the registers of its lanes, the other 64 lie in ray records in the wave's own LDS
// (18 dwords each — 21 until round 5 —, kept by field: dword k of slot j at rec[S k + j], so that exchanging a lane's ray with a slot's is 21 reads and
// 21 writes without bank conflicts).  The wave's loop picks a phase — WALK (comp:314-375, the hand-written park loop), BRICK
// (comp:378-471 for the rays that stand in front of an occupied cell) or TRANSITION (comp:153-265 around GridHit: shade, scatter,
// shadow ray, next sample, next pixel, ray set-up) — by how many of its 128 rays wait for each; lanes whose ray is in another
// state exchange it with a slot whose ray is in that state; the phase then runs on (nearly) 64 lanes.  The walk loop returns once
// `pool_walk_k` of its lanes have parked or left, and is re-entered with fresh rays in those lanes.
// What belongs to the PATH rather than to its current ray (pixel, sample index, the sample sum, RayColor's locals) is needed by
// the transitions only and lies in global memory (16 dwords per path, by field, wave-private: TraceParams::pool_paths), loaded and
// stored once per transition.
// Per ray the sequence of arithmetic operations is ray_color's / main()'s exactly as in vrt_path_kernel (the same functions are
// called).  The unit of work a path takes from the counter is ONE SAMPLE of a pixel (unit = pixel * spp + sample): its term of the sample
// loop's sum, color / (color + 1) (comp:173), goes to TraceParams::pool_samples[unit], and vrt_pool_resolve_kernel adds a pixel's terms
// in the sample loop's order, tone-maps and stores (comp:173-177): frames are bit-identical.  (A path that owned a whole PIXEL — 16
// samples one after the other — left the frame a drain of 11 % of its time: when the counter ran out every path still had half a pixel
// to go, and the pools emptied for 12.6 ms of a 113 ms frame, tools/path_profile.py.  With samples as units the drain is one sample long,
// and the 16 rays of a pixel start side by side in one wave.)
// No cross-wave communication (no barrier, no atomics but the unit counter): a wave's LDS is its own.
// LDS per wave (round 4: 64 slots): 21 x 256 B records + 256 B slot states + 256 B scratch = 5 888 B, and 4 KiB of staged bricks for a wave in a brick
// round.  With a staging area per wave that is four 256-thread workgroups per CU (156 of the CU's 160 KiB), four waves per SIMD.
// Measured, the kernel issues one instruction per ~11 cycles per wave whatever the wave count — a wave is one chain of dependent
// instructions — so occupancy is worth what it is in a latency-bound kernel: the waves of a workgroup SHARE two staging areas
// (STAGES; a wave locks one for the length of its brick round), five workgroups fit, five waves per SIMD — and, round 5, six with 48
// records per wave (the kernel has come down to 80 VGPRs): SLOTS 48, 112 paths per wave.
// Only the configuration the 2048^3 path trace runs: 8^3 bricks staged in LDS, the counter-free dilated-index walk (all three grid
// dimensions powers of two, the walk ends at the grid's face: vrt_path_kernel<..., DIL 2>'s loop).  Everything else keeps
// vrt_path_kernel.
#include "vrt_path_kernel.h"

namespace vrt {

// (kPoolRecDwords, kPoolStageBytes, kPoolWaveLdsBytes, kPoolPaths, kPoolPathDwords: vrt_kernels.h — the launcher sizes LDS and the
// paths' buffer by them)

// state of a ray.  class 0 (<= 2): waits for a transition; 1: walks; 2: waits in front of an occupied cell; 3: its path is over
enum : uint32_t { kRayFetch = 0u, kRayMiss = 1u, kRayHit = 2u, kRayWalk = 3u, kRayParked = 4u, kRayExit = 5u };
VRT_DI uint32_t pool_class(uint32_t st) { return st <= 2u ? 0u : st - 2u; }

VRT_DI uint32_t pool_mbcnt(unsigned long long m) { // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
VRT_DI uint32_t f2u(float v) { return __builtin_bit_cast(uint32_t, v); }
VRT_DI float u2f(uint32_t v) { return __builtin_bit_cast(float, v); }

// SLOTS: ray records in LDS per wave (the wave owns 64 + SLOTS paths).  STAGES: 4 KiB staging areas for bricks per workgroup of four
// waves — 4: one per wave; fewer: a wave takes one for the length of a brick round (try-lock in LDS; if none is free it serves another
// queue), which is what lets five workgroups of 128-path waves — or six of 112-path waves — fit a CU's LDS.
template <int B, int MIN_WAVES, int SLOTS = 64, int STAGES = 4>
__global__ __launch_bounds__(256, MIN_WAVES) void vrt_pool_kernel(const TraceParams p) {
    // B = 8: a brick (64 bytes) is staged in LDS for its voxel-level walk (STAGES areas per workgroup).  B = 4 (round 5: the reference's own
    // brick size, State.zig:5): a brick is two words, its walk asks for them as it goes — no staging area, no lock (STAGES = 0)
    static_assert((B == 8 && STAGES >= 1) || (B == 4 && STAGES == 0), "8^3 bricks are staged in LDS, 4^3 bricks are not");
    extern __shared__ __attribute__((aligned(16))) uint32_t pool_lds[];
    // (the wave's number through readfirstlane: its LDS and its block of path records are then scalar addresses, not per-lane registers)
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // the workgroup's LDS: 16 lock words | STAGES staging areas of 4 KiB | per wave: records, slot states, scratch
    constexpr uint32_t S = (uint32_t)SLOTS, kWaveDwords = (kPoolRecDwords + 1u) * S;
    uint32_t *const locks = pool_lds;
    // the staging areas for bricks (LDS byte address; layout inside one dictated by global_load_lds: vrt_trace_kernels.h)
    const uint32_t stage0 = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)pool_lds + 64u;
    uint32_t *const rec = pool_lds + 16u + (uint32_t)STAGES * (kPoolStageBytes / 4u) + wave * kWaveDwords; // rec[S k + j]: dword k of the ray in slot j
    uint32_t *const sstate = rec + kPoolRecDwords * S; // bits 0-7: state of the ray in slot j; bits 8-15: its walk code (GridParkRegs::code)
    // (LDS byte addresses for the exchange's instructions; scalar: the wave's number is)
    const uint32_t rec_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)rec);
    if (threadIdx.x < 16u) locks[threadIdx.x] = 0u; // (staging locks free; every wave's chunk empty, the counter not yet run out)
    __syncthreads();
    uint32_t *const path = p.pool_paths + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4u + wave) * (size_t)(kPoolPaths * kPoolPathDwords);

    const PushConstants &pc = p.pcs[blockIdx.y];
    const uint32_t uspp = (uint32_t)max(1, pc.cam.samples_per_pixel);
    const uint32_t total = p.owned_tiles * (uint32_t)(kTileW * kTileH) * uspp; // units: samples (vrt_api.hip keeps this below 2^32)
    uint32_t *const counter = p.work_counter + blockIdx.y;
    const bool sun_enabled = pc.sun.enabled > 0;
    const int max_bounce = pc.cam.max_bounce;
    const float t_max = __builtin_inff();
    const f3 g_min = mk3(p.grid.min_point_base_t[0], p.grid.min_point_base_t[1], p.grid.min_point_base_t[2]);
    const float g_scale = p.grid.max_point_scale[3];
    const int dx = (int)p.grid.dim_x, dy = (int)p.grid.dim_y, dz = (int)p.grid.dim_z;
    int lox = 0, loy = 0, loz = 0, hix = dx - 1, hiy = dy - 1, hiz = dz - 1;
    if (p.cell_bounds) {
        lox = -p.cell_bounds[0], loy = -p.cell_bounds[1], loz = -p.cell_bounds[2];
        hix = p.cell_bounds[3], hiy = p.cell_bounds[4], hiz = p.cell_bounds[5];
    }
    const int zero_budget = dx + dy + dz + 8;
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    // the dilated cell index (vrt_trace_kernels.h, grid_walk_park_dilated_gfx950): the three axes' bit fields
    const uint32_t lx = uni(31u - (uint32_t)__builtin_clz(p.grid.dim_x)), lz = uni(31u - (uint32_t)__builtin_clz(p.grid.dim_z)),
                   ly = uni(31u - (uint32_t)__builtin_clz(p.grid.dim_y));
    const uint32_t fx = uni(3u | (((1u << (lx - 2u)) - 1u) << 5)), fz = uni((3u << 2) | (((1u << (lz - 2u)) - 1u) << (lx + 3u))),
                   fy = uni((1u << 4) | (((1u << (ly - 1u)) - 1u) << (lx + lz + 1u)));
    u32x4 hb_rsrc;
    {
        const unsigned long long a = (unsigned long long)p.status_halfblocks;
        hb_rsrc.x = uni((uint32_t)a);
        hb_rsrc.y = uni((uint32_t)(a >> 32) | (4u << 16));
        hb_rsrc.z = uni(p.status_words);
        hb_rsrc.w = 0x00020000u;
    }
    const bool by_cell = p.cell_occupancy != nullptr, has_cell_material = p.cell_material != nullptr;
    const bool start_is_slot = p.start_is_slot != nullptr && __builtin_amdgcn_readfirstlane((int)*p.start_is_slot) != 0;
    const uint32_t walk_k = max(1u, p.pool_walk_k), brick_thr = p.pool_brick_thr, trans_thr = p.pool_trans_thr, walk_min = p.pool_walk_min;

    // ---- the ray in this lane's registers (the record's 18 dwords) + its state and walk code ----
    f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 1), inv = mk3(1, 1, 1), sd = mk3(0, 0, 0);
    uint32_t idx = 0u;  // the cell the walk stands on (dilated index; axes walked down mirrored)
    uint32_t cw = 0u;   // kRayWalk: the half-block word of that cell; kRayParked: the occupied cell left behind (dilated index)
    float t_in = 0.0f;  // crossed distance of the step into the parked cell; kRayHit: hit.t
    float t_out = 0.0f; // crossed distance of the ray's last step; kRayHit: hit.index (bits)
    float ir = 1.0f; // (the slab distances of GridHit, comp:273-286, are not carried: the brick round forms them again from origin and 1 / dir)
    // flags: bits 0-6 the path (slot of `path`), 8-9 / 10-11 / 12-13 ray step x / y / z + 1, 14-17 slab-entry code, 18-19 ignored
    // material type, 20 the step out of the parked cell left the grid, 21-22 kRayHit: the face of the voxel hit, 23 the ray is a shadow ray
    // (RayColor's `kind`, kept with the ray: the transition that finds it finished knows which fields of the path it needs before loading any)
    uint32_t fl = lane;
    uint32_t code = 3u << 4; // GridParkRegs::code
    uint32_t st = kRayFetch;
    if (lane < S) {
        sstate[lane] = kRayFetch;
#pragma unroll
        for (uint32_t k = 0; k < kPoolRecDwords; k++) rec[S * k + lane] = (k == 17u) ? 64u + lane : 0u;
    }

    // tile / tiles_x by the host's reciprocal (TraceParams::pool_tiles_x_magic): the compiler's own would be a per-lane register kept
    // for the length of the kernel
    auto tile_row = [&](uint32_t tile) { return p.pool_tiles_x_magic ? __umulhi(tile, p.pool_tiles_x_magic) : tile; };
    auto sx_of = [](uint32_t f) { return (int)((f >> 8) & 3u) - 1; };
    auto sy_of = [](uint32_t f) { return (int)((f >> 10) & 3u) - 1; };
    auto sz_of = [](uint32_t f) { return (int)((f >> 12) & 3u) - 1; };

    // lanes whose ray is not of class X take the ray of a slot that is, as many as there are on either side
    auto exchange = [&](uint32_t X) {
        const uint32_t sst = lane < S ? (sstate[lane] & 0xFFu) : (uint32_t)kRayExit;
        const unsigned long long offer = __builtin_amdgcn_ballot_w64(pool_class(sst) == X);
        const unsigned long long want = __builtin_amdgcn_ballot_w64(pool_class(st) != X);
        const uint32_t n_offer = (uint32_t)__builtin_popcountll(offer);
        const uint32_t n = min(n_offer, (uint32_t)__builtin_popcountll(want));
        const uint32_t q = pool_mbcnt(offer), r = pool_mbcnt(want);
        // the r-th wanting lane takes the r-th offering slot.  Which slot that is comes through two cross-lane permutes (round 5; until then
        // through a scratch row in LDS, a dword per slot): first every lane sends its number to its place in "offering lanes first, in
        // order" — a permutation of the 64 lanes —, then the wanting lane of rank r reads place r.
        const uint32_t place = ((offer >> lane) & 1ull) ? q : n_offer + (lane - q);
        const uint32_t by_rank = (uint32_t)__builtin_amdgcn_ds_permute((int)(place << 2), (int)lane);
        const uint32_t slot = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(r << 2), (int)by_rank);
        {
            // the lane's 18 dwords and its state + code against the slot's, field by field: ds_wrxchg writes the register and returns what
            // was there into the same register — one LDS instruction per field, nothing copied.  The taking lanes are selected by EXEC inside
            // the block: to the compiler this is straight-line code on the ray's registers (as a branch it copies all of them to
            // another set of registers before every exchange: the loaded values are the branch's, the old ones the other path's)
            const unsigned long long take = __builtin_amdgcn_ballot_w64(((want >> lane) & 1ull) && r < n);
            uint32_t sc = st | (code << 8);
            uint32_t at = rec_lds + (slot << 2);
            unsigned long long saved;
#define VRT_X(k) "ds_wrxchg_rtn_b32 %[f" #k "], %[at], %[f" #k "] offset:%[o" #k "]\n\t"
            asm volatile("s_and_saveexec_b64 %[saved], %[take]\n\t"
                         VRT_X(0) VRT_X(1) VRT_X(2) VRT_X(3) VRT_X(4) VRT_X(5) VRT_X(6) VRT_X(7) VRT_X(8) VRT_X(9) VRT_X(10) VRT_X(11) VRT_X(12) VRT_X(13)
                         VRT_X(14) VRT_X(15) VRT_X(16) VRT_X(17) VRT_X(18)
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "s_mov_b64 exec, %[saved]"
                         : [f0] "+v"(ro.x), [f1] "+v"(ro.y), [f2] "+v"(ro.z), [f3] "+v"(rd.x), [f4] "+v"(rd.y), [f5] "+v"(rd.z), [f6] "+v"(inv.x),
                           [f7] "+v"(inv.y), [f8] "+v"(inv.z), [f9] "+v"(sd.x), [f10] "+v"(sd.y), [f11] "+v"(sd.z), [f12] "+v"(idx), [f13] "+v"(cw),
                           [f14] "+v"(t_in), [f15] "+v"(t_out), [f16] "+v"(ir), [f17] "+v"(fl), [f18] "+v"(sc), [saved] "=&s"(saved)
                         : [take] "s"(take), [at] "v"(at), [o0] "n"(0), [o1] "n"(4 * S), [o2] "n"(8 * S), [o3] "n"(12 * S),
                           [o4] "n"(16 * S), [o5] "n"(20 * S), [o6] "n"(24 * S), [o7] "n"(28 * S), [o8] "n"(32 * S), [o9] "n"(36 * S), [o10] "n"(40 * S),
                           [o11] "n"(44 * S), [o12] "n"(48 * S), [o13] "n"(52 * S), [o14] "n"(56 * S), [o15] "n"(60 * S), [o16] "n"(64 * S),
                           [o17] "n"(68 * S), [o18] "n"(72 * S)
                         : "memory", "scc");
#undef VRT_X
            st = sc & 0xFFu;
            code = sc >> 8;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // the wave's chunk of units: [next, end) and "the counter has run out", in three of the workgroup's spare lock words — the lanes
    // that make a round of transitions are not the lanes that made the last one, so the state cannot live in their registers
#ifndef VRT_POOL_CHUNK
#define VRT_POOL_CHUNK 512u /* 64 / 128 / 256 / 512 / 1024: 2048^3 path trace 92.7 / 92.3 / 92.3 / 92.3 / 93.6 ms, from outside the field 33.6 / 26.8 / 24.5 / 23.6 / 24.3 */
#endif
    // (round 5: a frame of fewer units than the launch's waves take in chunks of 512 — the reference app's 1024 x 576 x 2 samples: 1.18 M
    // units for 5 120 waves — left more than half of the waves without work and the others four generations of their pools to go through
    // one after the other, the counter exhausted at 1.3 % of the kernel: the chunk shrinks until every wave gets sixteen, 64 at least)
    const uint32_t kPoolChunk = uni(max(64u, min((uint32_t)VRT_POOL_CHUNK, total / (gridDim.x * 64u))));
    uint32_t *const chunk = locks + 4u + wave * 3u;
#ifdef VRT_DEV_PROFILE
    if (threadIdx.x < 8) vrt_prof[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned long long pf_t[3] = {0ull, 0ull, 0ull};
    unsigned long long pf_n[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
    const unsigned long long pf_begin = wall_clock64();
    unsigned long long pf_dry = 0ull; // when this wave first found the pixel counter exhausted
#define VRT_PF_T(k, t0) pf_t[k] += __builtin_readcyclecounter() - (t0)
#define VRT_PF_N(k, v) pf_n[k] += (unsigned long long)(v)
#define VRT_PF_NOW() __builtin_readcyclecounter()
#else
#define VRT_PF_T(k, t0)
#define VRT_PF_N(k, v)
#define VRT_PF_NOW() 0ull
#endif
    // (the guard bounds a wave whose state machine stalls — a bug — to a finite run instead of a hung GPU; a wave of the 4K /
    // 2048^3 / 16 spp frame takes ~40 000 rounds)
    for (uint32_t round = 0u; round < (1u << 23); round++) {
        // how many of the wave's rays wait for what
        VRT_PROF_BEGIN(tpd);
        const uint32_t sst = lane < S ? (sstate[lane] & 0xFFu) : (uint32_t)kRayExit;
        const uint32_t n_walk = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(st == kRayWalk)) +
                                (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(sst == kRayWalk));
        const uint32_t n_brick = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(st == kRayParked)) +
                                 (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(sst == kRayParked));
        const uint32_t n_trans = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(st <= kRayHit)) +
                                 (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(sst <= kRayHit));
        if (n_walk + n_brick + n_trans == 0u) break;
        uint32_t phase; // 0 transitions, 1 walk, 2 bricks
        if (n_brick >= brick_thr) phase = 2u;
        else if (n_trans >= trans_thr) phase = 0u;
        else if (n_walk >= walk_min) phase = 1u;
        else if (n_brick != 0u && n_brick >= n_trans) phase = 2u;
        else if (n_trans != 0u) phase = 0u;
        else phase = 1u;
        [[maybe_unused]] int stage = (int)wave;
        if constexpr (STAGES > 0 && STAGES < 4) {
            if (phase == 2u) {
                // a staging area for the length of the round; all taken by other waves of the workgroup: serve another queue
                stage = -1;
                if (lane == 0u) {
                    for (int i = 0; i < STAGES && stage < 0; i++)
                        if (atomicCAS(&locks[i], 0u, 1u) == 0u) stage = i;
                }
                stage = __builtin_amdgcn_readfirstlane(stage);
                VRT_PF_N(7, stage < 0 ? 1 : 0);
                if (stage < 0) phase = n_walk != 0u ? 1u : (n_trans != 0u ? 0u : 3u);
            }
        }
        VRT_PROF_END(3, tpd);
        // (three separate `if`s on the wave-uniform phase, no `else`, no `continue`: an if / else-if chain is linearised by the compiler's
        // CFG structuriser with a join per arm, and every join copies the 22 values of the ray from one set of registers to another)
        if (phase == 3u) __builtin_amdgcn_s_sleep(16);
        if (phase == 0u) {
            [[maybe_unused]] const unsigned long long pf0 = VRT_PF_NOW();
            VRT_PROF_BEGIN(tpz);
            exchange(0u);
            VRT_PROF_END(7, tpz);
            VRT_PF_N(0, 1);
            VRT_PF_N(1, __builtin_popcountll(__builtin_amdgcn_ballot_w64(st <= kRayHit)));
            if (st <= kRayHit) {
                const uint32_t ps = fl & 127u;
                uint32_t *const pr = path + ps;
                // the path: pixel, sample index, the sample sum (comp:173), RayColor's locals (comp:203-216) and what it keeps while the
                // shadow ray is walked (comp:221-239).
                // Round 5: only the fields this transition reads are loaded, only those it changes are stored.  The records of a wave are 6.6 KB,
                // those of the 640 waves of an XCD 4.3 MB beside a 4 MiB L2 that the bricks stream through: a record was written back to the
                // fabric and fetched again between two transitions of its path — 61 GB written and as much read per 4K frame of 16 samples, half
                // of the kernel's fabric traffic (profiles/r05_cfg4_pool_counters.txt).  What a transition needs follows from what kind of ray
                // has just finished, which is kept WITH THE RAY (bit 23 of its flags) so that no load waits for another:
                //   a camera / scattered ray that hit, sun on: shades and starts the shadow ray — reads the flags word, writes it and the
                //     the scattered ray's direction (its refraction index and the hit ray's direction.y only where they will be
                //     read; the attenuation is the hit material's albedo: its id rides in the flags word);
                //   a shadow ray: reads those, the colour and the unit — writes the flags and, if the sun reached the hit, the colour;
                //   a ray that missed: the path is over — reads the colour and the unit; the unit is written when a new one is taken.
                const bool fresh = st == kRayFetch;                       // (no path yet: every field it will use is set before it is read)
                const bool was_shadow = !fresh && ((fl >> 23) & 1u) != 0u;
                const bool shades_only = !fresh && !was_shadow && st == kRayHit && sun_enabled;
                uint32_t work = 0u, pf = 0u;
                f3 color = mk3(0, 0, 0), sc_dir = mk3(0, 0, 1), attenuation = mk3(0, 0, 0);
                float cur_dir_y = 0.0f, sc_ir = 1.0f;
                if (!fresh) pf = pr[128];
                if (!fresh && !shades_only) {
                    work = pr[0]; // the path's unit: pixel * spp + sample
                    color = mk3(u2f(pr[5 * 128]), u2f(pr[6 * 128]), u2f(pr[7 * 128]));
                }
                if (was_shadow) {
                    sc_dir = mk3(u2f(pr[9 * 128]), u2f(pr[10 * 128]), u2f(pr[11 * 128]));
                    // (two words that hardly ever matter, behind the flags word: the hit ray's direction.y is read again only by a path that
                    // ends with loop_count 0 — its first hit was a material of unknown type, comp:235-237 —, the scattered ray's refraction
                    // index only where it is not 1: inside glass)
                    if (((pf >> 16) & 15u) == 0u) cur_dir_y = u2f(pr[8 * 128]);
                    if (pf & 1u) sc_ir = u2f(pr[12 * 128]);
                }
                bool color_changed = false, work_changed = false;
                int loop_count = (int)((pf >> 16) & 15u);
                int kind = (int)((pf >> 20) & 1u);
                bool scattered_ok = ((pf >> 21) & 1u) != 0u;
                uint32_t sc_ignore = (pf >> 22) & 3u;
                uint32_t hit_mat = pf >> 24; // the material of the hit the shadow ray left from: its albedo is the attenuation (comp:223-226), read again from the table

                Ray r = Ray{ro, rd, ir, (fl >> 18) & 3u};
                RaySetup s;
                s.inv_dir = inv;
                s.entry_code = (int)((fl >> 14) & 15u);
                s.sx = sx_of(fl), s.sy = sy_of(fl), s.sz = sz_of(fl);
                s.grid_t_min = 0.0f, s.grid_t_max = 0.0f; // (of the finished ray: not read again)
                const bool found = st == kRayHit;
                int ls = (st == kRayFetch) ? kLaneFetch : kLaneDone;
                // (1) a ray has finished: comp:218-258 from the loop condition's GridHit onwards
                if (ls == kLaneDone) {
                    bool after_shadow = false;
                    if (kind == 0) {
                        if (found) {
                            Hit hit;
                            hit.t = t_in;
                            hit.index = f2u(t_out);
                            if (hit.index & kDeferredHit) {
                                // the brick round left comp:337 / :422 / :425 to this round: the parked cell (still in cw) -> its brick ->
                                // the brick's first material entry -> the voxel's material
                                const uint32_t flipped = (sx_of(fl) < 0 ? fx : 0u) | (sy_of(fl) < 0 ? fy : 0u) | (sz_of(fl) < 0 ? fz : 0u);
                                const uint32_t real = cw ^ flipped;
                                const uint32_t hx = (real & 3u) | ((real >> 3) & (((1u << (lx - 2u)) - 1u) << 2));
                                const uint32_t hz = ((real >> 2) & 3u) | ((real >> (lx + 1u)) & (((1u << (lz - 2u)) - 1u) << 2));
                                const uint32_t hy = ((real >> 4) & 1u) | ((real >> (lx + lz + 1u)) << 1);
                                const uint32_t hcell = hx + (uint32_t)dx * (hz + (uint32_t)dz * hy);
                                // (round 5) a brick of ONE material — every solid voxel's entry of binding 7 the same — has that id in a byte
                                // per cell: one look-up in an array of `cells` bytes instead of two dependent ones in brick_index and the
                                // 128 times larger material_index (TraceParams::cell_material; 0xFF: look it up)
                                uint32_t hmat = has_cell_material ? (uint32_t)p.cell_material[hcell] : 0xFFu;
                                if (hmat == 0xFFu) {
                                    const uint32_t hbrick = p.brick_index[hcell];
                                    const uint32_t hstart = start_is_slot ? hbrick * (uint32_t)(B * B * B) : (p.brick_start_index[hbrick] & 0x7FFFFFFFu);
                                    hmat = p.material_index[hstart + (hit.index & ~kDeferredHit)];
                                }
                                hit.index = hmat;
                            }
                            const float t_offset = (g_scale * (1.0f / (float)B)) * 0.05f;
                            hit.normal = axis_normal(s, (int)((fl >> 21) & 3u));
                            hit.point = ray_at(r, hit.t) + hit.normal * t_offset;
                            loop_count += 1;
                            Ray scattered = r;
                            bool result = false;
                            const vrt_material *m = p.materials + hit.index;
                            const uint32_t mtype = m->type;
                            attenuation = mk3(m->albedo_r, m->albedo_g, m->albedo_b);
                            hit_mat = hit.index & 0xFFu;
                            const float mdata = m->type_data;
                            switch (mtype) {
                                case MAT_LAMBERTIAN: result = scatter_lambertian(hit, scattered); break;
                                case MAT_METAL: result = scatter_metal(mdata, r, hit, scattered); break;
                                case MAT_DIELECTRIC: result = scatter_dielectric(mdata, r, hit, scattered); break;
                                default:
                                    loop_count -= 1;
                                    result = false;
                                    break;
                            }
                            scattered_ok = result;
                            sc_dir = scattered.direction;
                            sc_ir = scattered.internal_reflection;
                            sc_ignore = scattered.ignore_type_material;
                            cur_dir_y = r.direction.y;
                            if (sun_enabled) {
                                const f3 sun_position = mk3(pc.sun.position[0], pc.sun.position[1], pc.sun.position[2]);
                                const f3 rv = rand_vec3_range(r.direction.x + r.direction.z, r.direction.y + r.direction.z, -pc.sun.radius, pc.sun.radius);
                                const f3 shadow_ray_dir = (sun_position + rv) - hit.point;
                                r = create_ray(hit.point, shadow_ray_dir); // CreateShadowRay, comp:186-190 (ignore type MAT_NONE)
                                kind = 1;
                                ls = kLaneStart;
                            } else {
                                color = color + attenuation;
                                color_changed = true;
                                r.origin = hit.point; // the scattered ray starts where the shadow ray would have
                                after_shadow = true;
                            }
                        } else {
                            cur_dir_y = r.direction.y;
                            ls = kLaneEnd; // the while condition failed (comp:218)
                        }
                    } else {
                        if (!found) {
                            const vrt_material *hm = p.materials + hit_mat; // (5 KiB table: resident; three words of the record less to carry)
                            attenuation = mk3(hm->albedo_r, hm->albedo_g, hm->albedo_b);
                            color = color + attenuation * mk3(pc.sun.color[0], pc.sun.color[1], pc.sun.color[2]);
                            color_changed = true;
                        }
                        after_shadow = true;
                    }
                    if (after_shadow) {
                        if (!scattered_ok) {
                            ls = kLaneEnd; // comp:253-255
                        } else {
                            r.direction = sc_dir; // current_ray = scattered (its origin, hit.point, is r.origin already)
                            r.internal_reflection = sc_ir;
                            r.ignore_type_material = sc_ignore;
                            cur_dir_y = sc_dir.y;
                            kind = 0;
                            ls = (loop_count < max_bounce) ? kLaneStart : kLaneEnd;
                        }
                    }
                }
                // (2) the path is over: comp:260-264, then the sample loop's accumulation (comp:173)
                if (ls == kLaneEnd) {
                    if (loop_count == 0) {
                        const f3 sun_color = mk3(pc.sun.color[0], pc.sun.color[1], pc.sun.color[2]);
                        const float t = 0.5f * (cur_dir_y + 1.0f);
                        const f3 bg = fma3(splat3(1.0f - t), splat3(1.0f), mk3(0.5f, 0.7f, 1.0f) * t);
                        color = color + bg * (sun_enabled ? sun_color : splat3(1.0f));
                    }
                    // (3) the sample's term of the sum; the pixel is finished by vrt_pool_resolve_kernel (comp:173-177)
                    // (stored by groups of 64 pixels, sample-major inside a group: the resolve pass then reads whole lines — by unit it
                    // read 64 lines per instruction for 64 pixels, 1.9 ms per 4K frame of 16 samples instead of 0.5)
                    const f3 term = color / (color + splat3(1.0f));
                    const uint32_t tpx = work / uspp, tk = work - tpx * uspp;
                    p.pool_samples[((size_t)(tpx >> 6) * uspp + tk) * 64u + (tpx & 63u)] = make_float4(term.x, term.y, term.z, 0.0f);
                    ls = kLaneFetch;
                }
                // (4) next unit, from the wave's chunk of kPoolChunk consecutive units; one atomic per chunk (one per round of
                // transitions, 90 M a second on one address, made every transition wait for the counter: the frame from outside the
                // field 34 -> 46 ms).  A lane that asks where the chunk ends and the next has not come yet asks again next round.
                {
                    const unsigned long long asking = __builtin_amdgcn_ballot_w64(ls == kLaneFetch);
                    if (asking != 0ull) {
                        uint32_t chunk_next = chunk[0], chunk_end = chunk[1];
                        bool more_chunks = chunk[2] == 0u;
                        bool work_left = more_chunks || chunk_next < chunk_end;
                        if (work_left) {
                            const uint32_t n = (uint32_t)__builtin_popcountll(asking), rank = pool_mbcnt(asking);
                            const uint32_t take = min(chunk_end - chunk_next, n);
                            if (ls == kLaneFetch && rank < take) {
                                work = chunk_next + rank;
                                ls = kLaneSample;
                            }
                            chunk_next += take;
                            if (take < n && more_chunks) {
                                uint32_t first = 0u;
                                if (lane == (uint32_t)__builtin_ctzll(asking)) first = atomicAdd(counter, kPoolChunk);
                                first = (uint32_t)__builtin_amdgcn_readlane((int)first, __builtin_ctzll(asking));
                                more_chunks = first < total;
                                chunk_next = more_chunks ? first : 0u;
                                chunk_end = more_chunks ? min(first + kPoolChunk, total) : 0u;
                                const uint32_t take2 = min(chunk_end - chunk_next, n - take);
                                if (ls == kLaneFetch && rank >= take && rank - take < take2) {
                                    work = chunk_next + (rank - take);
                                    ls = kLaneSample;
                                }
                                chunk_next += take2;
                            }
                            work_left = more_chunks || chunk_next < chunk_end;
                            if (lane == (uint32_t)__builtin_ctzll(asking)) chunk[0] = chunk_next, chunk[1] = chunk_end, chunk[2] = more_chunks ? 0u : 1u;
#ifdef VRT_DEV_PROFILE
                            if (!work_left && pf_dry == 0ull) pf_dry = wall_clock64();
#endif
                        }
                        if (!work_left && ls == kLaneFetch) ls = kLaneExit;
                    }
                }
                // (5) the unit's sample of its pixel: comp:162-171
                if (ls == kLaneSample) {
                    const uint32_t pixel = work / uspp;
                    const int sample_i = (int)(work - pixel * uspp);
                    const uint32_t owned = p.owned_tiles - 1u - (pixel >> 8);
                    const uint32_t tile = p.own_period ? (owned / p.own_count) * p.own_period + p.own_slots[owned % p.own_count] : owned * p.shard_count + p.shard_rank;
                    const uint32_t j = pixel & 255u;
                    const uint32_t in_x = ((j >> 6) & 1u) * 8u + (j & 7u), in_y = (j >> 7) * 8u + ((j >> 3) & 7u);
                    const uint32_t tile_y = tile_row(tile), tile_x = tile - tile_y * p.tiles_x;
                    const uint32_t px = tile_x * kTileW + in_x, py = tile_y * kTileH + in_y;
                    if (px >= p.width || py >= p.height) {
                        ls = kLaneFetch; // outside the image (comp:155-159): nothing to trace, nothing to store; asks again next round
                    } else {
                        const float x = (float)px, y = (float)py;
                        const f3 horizontal = mk3(pc.cam.horizontal[0], pc.cam.horizontal[1], pc.cam.horizontal[2]);
                        const f3 vertical = mk3(pc.cam.vertical[0], pc.cam.vertical[1], pc.cam.vertical[2]);
                        const f3 llc = mk3(pc.cam.lower_left_corner[0], pc.cam.lower_left_corner[1], pc.cam.lower_left_corner[2]);
                        const f3 origin = mk3(pc.cam.origin[0], pc.cam.origin[1], pc.cam.origin[2]);
                        const float flag = (sample_i > 0) ? 1.0f : 0.0f;
                        const float noise_x = hash_12_jitter(x + (float)sample_i, y, flag);
                        const float u = (x + noise_x) / (float)(pc.cam.image_width - 1u);
                        const float noise_y = hash_12_jitter(x, y + (float)sample_i, flag);
                        const float v = (y + noise_y) / (float)(pc.cam.image_height - 1u);
                        const f3 ray_dir = fma3(horizontal, splat3(u), llc) + fma3(splat3(v), vertical, -origin);
                        r = create_ray(origin, ray_dir);
                        kind = 0;
                        loop_count = 0;
                        color = mk3(0, 0, 0);
                        color_changed = work_changed = true;
                        cur_dir_y = r.direction.y;
                        ls = (loop_count < max_bounce) ? kLaneStart : kLaneEnd;
                    }
                }
                // (6) a new ray: comp:271-312 (GridHit up to its loop)
                uint32_t nst = (ls == kLaneExit) ? kRayExit : ((ls == kLaneFetch) ? kRayFetch : kRayMiss); // (kLaneEnd: max_bounce 0, next round)
                if (ls == kLaneStart) {
                    nst = kRayMiss;
                    if (grid_slab(p, r, 0.00001f, t_max, s)) {
                        const float global_t_value = s.grid_t_min + 0.0001f * opaque_uniform(g_scale); // comp:287 (opaque: the product is not kept in a register)
                        const f3 fposition = p.scale_pow2 ? (ray_at(r, global_t_value) - g_min) * p.inv_grid_scale : (ray_at(r, global_t_value) - g_min) / splat3(g_scale);
                        Walk w;
                        w.side_dist = initial_side_dist(mk3((float)s.sx, (float)s.sy, (float)s.sz), fposition, s.ray_delta());
                        const int px = f2i_clamp(__builtin_floorf(fposition.x));
                        const int py = f2i_clamp(__builtin_floorf(fposition.y));
                        const int pz = f2i_clamp(__builtin_floorf(fposition.z));
                        w.rx = steps_left_box(s.sx, px, lox, hix, zero_budget);
                        w.ry = steps_left_box(s.sy, py, loy, hiy, zero_budget);
                        w.rz = steps_left_box(s.sz, pz, loz, hiz, zero_budget);
                        const int base_x = walk_base_box(s.sx, px, lox, hix), base_y = walk_base_box(s.sy, py, loy, hiy), base_z = walk_base_box(s.sz, pz, loz, hiz);
                        w.t_value = 0;
                        uint32_t grid_index = (uint32_t)px + (uint32_t)dx * ((uint32_t)pz + (uint32_t)dz * (uint32_t)py);
                        const uint32_t stride_x = (uint32_t)s.sx, stride_y = (uint32_t)s.sy * (uint32_t)dx * (uint32_t)dz, stride_z = (uint32_t)s.sz * (uint32_t)dx;
                        bool more = (global_t_value <= t_max) && (unsigned)px < (unsigned)dx && (unsigned)py < (unsigned)dy && (unsigned)pz < (unsigned)dz &&
                                    (w.rx | w.ry | w.rz) >= 0;
                        int in_axis = 3; // the first cell of the walk was entered through the slab test, not by a step ...
                        float skip_t = 0.0f;
                        // ... unless the ray enters the grid in front of the occupied-cell box and jumps to its near face
                        if (p.cell_bounds && p.skip_to_box)
                            skip_to_box(w, s, (int)((uint32_t)hix - (uint32_t)lox), (int)((uint32_t)hiy - (uint32_t)loy), (int)((uint32_t)hiz - (uint32_t)loz), grid_index,
                                        stride_x, stride_y, stride_z, more, in_axis, skip_t);
                        if (more) {
                            // the walk's index in dilated form, from the cell the lane stands on; axes walked down are stored mirrored
                            const uint32_t cx = (uint32_t)(base_x - __mul24(s.sx, w.rx)), cy = (uint32_t)(base_y - __mul24(s.sy, w.ry)),
                                           cz = (uint32_t)(base_z - __mul24(s.sz, w.rz));
                            const uint32_t mx = s.sx < 0 ? ((uint32_t)dx - 1u - cx) : cx, my = s.sy < 0 ? ((uint32_t)dy - 1u - cy) : cy,
                                           mz = s.sz < 0 ? ((uint32_t)dz - 1u - cz) : cz;
                            idx = (mx & 3u) | ((mz & 3u) << 2) | ((my & 1u) << 4) | ((mx >> 2) << 5) | ((mz >> 2) << (lx + 3u)) | ((my >> 1) << (lx + lz + 1u));
                            const uint32_t flip = (s.sx < 0 ? fx : 0u) | (s.sy < 0 ? fy : 0u) | (s.sz < 0 ? fz : 0u);
                            cw = p.status_halfblocks[(idx ^ flip) >> 5];
                            sd = w.side_dist;
                            t_out = skip_t;
                            t_in = 0.0f;
                            code = (uint32_t)in_axis << 4;
                            nst = kRayWalk;
                        }
                    }
                    // the ray as the walk and the next transition need it
                    ro = r.origin, rd = r.direction, ir = r.internal_reflection;
                    inv = s.inv_dir;
                    fl = ps | ((uint32_t)(s.sx + 1) << 8) | ((uint32_t)(s.sy + 1) << 10) | ((uint32_t)(s.sz + 1) << 12) | (((uint32_t)s.entry_code & 15u) << 14) |
                         ((r.ignore_type_material & 3u) << 18);
                }
                fl = (fl & ~(1u << 23)) | ((uint32_t)kind << 23); // (what kind of ray the next transition of this path will find finished)
                st = nst;
                const bool ir_kept = shades_only && f2u(sc_ir) != 0x3F800000u; // (bit 0 of the flags word: the record holds a refraction index other than 1.0f)
                pr[128] = (((uint32_t)loop_count & 15u) << 16) | ((uint32_t)kind << 20) | ((scattered_ok ? 1u : 0u) << 21) | ((sc_ignore & 3u) << 22) | (hit_mat << 24) |
                          (ir_kept ? 1u : 0u);
                if (work_changed) pr[0] = work;
                if (color_changed) pr[5 * 128] = f2u(color.x), pr[6 * 128] = f2u(color.y), pr[7 * 128] = f2u(color.z);
                if (shades_only) { // (what the end of the shadow ray started above will read)
                    pr[9 * 128] = f2u(sc_dir.x), pr[10 * 128] = f2u(sc_dir.y), pr[11 * 128] = f2u(sc_dir.z);
                    if (loop_count == 0) pr[8 * 128] = f2u(cur_dir_y);
                    if (ir_kept) pr[12 * 128] = f2u(sc_ir);
                }
            }
            VRT_PF_T(0, pf0);
        }
        if (phase == 1u) {
            // every lane that has a ray to walk walks (comp:314-375), until pool_walk_k of them have parked or left — and again while the
            // phase rule keeps saying so (two calls in three follow a call): an inner loop, whose rays stay in ONE set of registers from
            // call to call (at the outer loop's back edge the compiler copies all 22 values of the ray to another set and back)
            bool again; // wave-uniform
            do {
            [[maybe_unused]] const unsigned long long pf1 = VRT_PF_NOW();
            VRT_PROF_BEGIN(tpx);
            exchange(1u);
            VRT_PROF_END(0, tpx);
            const unsigned long long walking = __builtin_amdgcn_ballot_w64(st == kRayWalk);
            if (walking != 0ull) {
            const uint32_t n_walking = (uint32_t)__builtin_popcountll(walking);
            const int sx = sx_of(fl), sy = sy_of(fl), sz = sz_of(fl);
            const uint32_t flip = (sx < 0 ? fx : 0u) | (sy < 0 ? fy : 0u) | (sz < 0 ? fz : 0u);
            const uint32_t nm_x = sx != 0 ? ~fx : ~0u, nm_y = sy != 0 ? ~fy : ~0u, nm_z = sz != 0 ? ~fz : ~0u;
            GridParkRegs g;
            g.alive = walking;
            g.out_x = g.out_y = 0ull;
            g.t_out = t_out;
            g.t_in = t_in;
            g.code = code;
            g.batch = uni(walk_k);
            g.min_alive = uni(n_walking >= walk_k ? n_walking - walk_k + 1u : 1u);
            const float t_in_keep = t_in; // (the loop's output register: lanes outside the call keep theirs)
            uint32_t word = cw, cell;
            unsigned long long gone = 0ull;
            VRT_PROF_BEGIN(tpw);
#ifdef VRT_POOL_AHEAD
            grid_walk_park_dilated_ahead_gfx950(sd, inv, idx, cell, nm_x, nm_y, nm_z, word, hb_rsrc, g, flip, gone);
#else
            grid_walk_park_dilated_carry_gfx950(sd, inv, idx, cell, nm_x, nm_y, nm_z, word, hb_rsrc, g, flip, gone);
#endif
            VRT_PROF_END(1, tpw);
            const bool was_walking = (walking >> lane) & 1ull;
            const bool parked = __builtin_amdgcn_inverse_ballot_w64(g.parked);
            const bool moving = __builtin_amdgcn_inverse_ballot_w64(g.alive);
            VRT_PF_T(1, pf1);
            VRT_PF_N(2, 1);
            VRT_PF_N(3, n_walking);
            VRT_PF_N(4, __builtin_popcountll(g.alive));
            // (selects, not branches: the asm ran under the walking lanes' EXEC, so its in/out operands — side distances, index, t_out —
            // are the other lanes' own values still; a branch here is one more join that copies the ray)
            t_in = was_walking ? g.t_in : t_in_keep;
            t_out = g.t_out;
            const bool park = was_walking && parked, move = was_walking && !parked && moving, left = was_walking && !parked && !moving;
            const uint32_t axis_code = (__builtin_amdgcn_inverse_ballot_w64(g.out_x) ? 0u : (__builtin_amdgcn_inverse_ballot_w64(g.out_y) ? 1u : 2u)) << 4;
            st = park ? (uint32_t)kRayParked : (left ? (uint32_t)kRayMiss : st); // (left: the ray has left the grid)
            cw = park ? cell : (move ? word : cw);
            code = park ? g.code : (move ? axis_code : code); // parked: bits 0-1 the axis INTO the occupied cell, 2-3 the axis out of it
            fl = park ? ((fl & ~(1u << 20)) | (__builtin_amdgcn_inverse_ballot_w64(gone) ? (1u << 20) : 0u)) : fl;
            }
            // the phase rule's first three lines on fresh counts: neither of the other queues is full, and enough rays are left to walk
            {
                const uint32_t sst2 = lane < S ? (sstate[lane] & 0xFFu) : (uint32_t)kRayExit;
                const uint32_t w2 = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(st == kRayWalk)) +
                                    (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(sst2 == kRayWalk));
                const uint32_t b2 = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(st == kRayParked)) +
                                    (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(sst2 == kRayParked));
                const uint32_t t2 = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(st <= kRayHit)) +
                                    (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(sst2 <= kRayHit));
                again = b2 < brick_thr && t2 < trans_thr && w2 >= walk_min && ++round < (1u << 23);
            }
            } while (again);
        }
        if (phase == 2u) {
            // the rays that stand in front of an occupied cell walk its brick (comp:378-471)
            [[maybe_unused]] const unsigned long long pf2 = VRT_PF_NOW();
            const uint32_t wave_lds = stage0 + (uint32_t)stage * kPoolStageBytes;
            VRT_PROF_BEGIN(tpy);
            exchange(2u);
            VRT_PROF_END(6, tpy);
            VRT_PF_N(5, 1);
            VRT_PF_N(6, __builtin_popcountll(__builtin_amdgcn_ballot_w64(st == kRayParked)));
            if (st == kRayParked) {
                const int sx = sx_of(fl), sy = sy_of(fl), sz = sz_of(fl);
                const uint32_t flip = (sx < 0 ? fx : 0u) | (sy < 0 ? fy : 0u) | (sz < 0 ? fz : 0u);
                // the cell's position from the walk's index, un-mirrored and un-dilated
                const uint32_t real = cw ^ flip;
                const int cx = (int)((real & 3u) | ((real >> 3) & (((1u << (lx - 2u)) - 1u) << 2)));
                const int cz = (int)(((real >> 2) & 3u) | ((real >> (lx + 1u)) & (((1u << (lz - 2u)) - 1u) << 2)));
                const int cy = (int)(((real >> 4) & 1u) | ((real >> (lx + lz + 1u)) << 1));
                const uint32_t cell = (uint32_t)cx + (uint32_t)dx * ((uint32_t)cz + (uint32_t)dz * (uint32_t)cy);
                const uint32_t occ_slot = by_cell ? cell : p.brick_index[cell]; // comp:337 (by_cell: only on a solid voxel)
                if constexpr (B == 8) stage_brick_lds(p, occ_slot, by_cell, wave_lds); // (first: the arithmetic below runs while the brick arrives)
                Ray r = Ray{ro, rd, ir, (fl >> 18) & 3u};
                RaySetup s;
                s.inv_dir = inv;
                s.entry_code = (int)((fl >> 14) & 15u);
                s.sx = sx, s.sy = sy, s.sz = sz;
                {
                    // GridHit's slab distances (comp:273-286) again, by grid_slab's own operations on the ray's origin and the 1 / dir it
                    // carries: two words less in every ray record (round 5: room for six more records per wave)
                    const f3 g_max = mk3(p.grid.max_point_scale[0], p.grid.max_point_scale[1], p.grid.max_point_scale[2]);
                    const f3 t_lower = (g_min - ro) * inv, t_upper = (g_max - ro) * inv;
                    const f3 t_mins = mk3(gl_min(t_lower.x, t_upper.x), gl_min(t_lower.y, t_upper.y), gl_min(t_lower.z, t_upper.z));
                    const f3 t_maxes = mk3(gl_max(t_lower.x, t_upper.x), gl_max(t_lower.y, t_upper.y), gl_max(t_lower.z, t_upper.z));
                    const bool iy = (t_mins.y > t_mins.x) && (t_mins.y > t_mins.z), iz = (t_mins.z > t_mins.x) && (t_mins.z > t_mins.y);
                    const float tmin_i = iz ? t_mins.z : (iy ? t_mins.y : t_mins.x);
                    s.grid_t_min = gl_max(0.00001f, tmin_i);
                    s.grid_t_max = gl_min(t_max, gl_min(gl_min(t_maxes.x, t_maxes.y), t_maxes.z));
                }
                const float gtmin = s.grid_t_min;
                const f3 brick_min = fma3(mk3((float)cx, (float)cy, (float)cz), splat3(g_scale), g_min); // comp:331
                const float global_t_value = t_in * g_scale + gtmin + 0.01f * opaque_uniform(g_scale);   // comp:347 (deferred) + comp:332
                Hit hit;
                hit.t = global_t_value;
                hit.index = 0u;
                int hit_axis = 0;
                const int a = (int)(code & 3u);
                const bool hit_voxel = brick_walk_park_gfx950<B, B == 8, B == 8, true>(p, r, s, g_scale, occ_slot, cell, by_cell, start_is_slot, brick_min, hit, a, hit_axis, wave_lds);
                if (hit_voxel) {
                    st = kRayHit;
                    t_in = hit.t;
                    t_out = u2f(hit.index);
                    fl = (fl & ~(3u << 21)) | (((uint32_t)hit_axis & 3u) << 21);
                } else if (!(global_t_value <= t_max) || ((fl >> 20) & 1u)) {
                    st = kRayMiss; // t became NaN (comp:316), or the step out of this cell left the grid
                } else {
                    st = kRayWalk;
                    cw = p.status_halfblocks[(idx ^ flip) >> 5]; // (an A-trip park left the lane's word in the other register set)
                    code = ((code >> 2) & 3u) << 4;              // the axis of its last step, for its first trip in the next call
                }
            }
            if constexpr (STAGES > 0 && STAGES < 4) {
                // (every LDS read of the round has returned before the area is handed on)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0u) __hip_atomic_store(&locks[stage], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            VRT_PF_T(2, pf2);
        }
    }
#ifdef VRT_DEV_PROFILE
    if (p.wave_timeline && lane == 0u) {
        for (int k = 0; k < 3; k++) atomicAdd(&p.wave_timeline[k], pf_t[k]);
        for (int k = 0; k < 8; k++) atomicAdd(&p.wave_timeline[3 + k], pf_n[k]);
        atomicAdd(&p.wave_timeline[11], 1ull);
        // the frame's drain: first wave to find the counter exhausted, last wave to end (100 MHz ticks, stored complemented where a minimum is wanted)
        atomicMax(&p.wave_timeline[20], ~pf_begin);
        if (pf_dry) atomicMax(&p.wave_timeline[21], ~pf_dry);
        if (pf_dry) atomicMax(&p.wave_timeline[23], pf_dry);
        atomicMax(&p.wave_timeline[22], wall_clock64());
    }
    __syncthreads();
    if (p.wave_timeline && threadIdx.x < 8) atomicAdd(&p.wave_timeline[12 + threadIdx.x], vrt_prof[threadIdx.x]);
#endif
#undef VRT_PF_T
#undef VRT_PF_N
#undef VRT_PF_NOW
}

// comp:173-177 for the frames vrt_pool_kernel traced: one thread per pixel of the owned tiles adds the pixel's terms
// (TraceParams::pool_samples) in the sample loop's order, tone-maps and stores.  HBM-bound: 16 B per sample read, 4 (+ 16) B per pixel
// written; 2 GiB for a 4K frame of 16 samples, 0.5 ms.  A workgroup is one tile and its threads map to the tile's pixels as the lanes of
// vrt_trace_kernel's four waves do, so the RGB shard of the multi-GPU pipeline (TraceParams::packed_rgb, round 5) is packed by the same
// row-of-eight-lanes exchange.
__global__ __launch_bounds__(256) void vrt_pool_resolve_kernel(const TraceParams p) {
    const PushConstants &pc = p.pcs[blockIdx.y];
    const uint32_t pixel = blockIdx.x * 256u + threadIdx.x;
    if (pixel >= p.owned_tiles * (uint32_t)(kTileW * kTileH)) return; // (uniform over the workgroup)
    const uint32_t spp = (uint32_t)max(1, pc.cam.samples_per_pixel);
    const uint32_t owned = p.owned_tiles - 1u - (pixel >> 8);
    const uint32_t tile = p.own_period ? (owned / p.own_count) * p.own_period + p.own_slots[owned % p.own_count] : owned * p.shard_count + p.shard_rank;
    const uint32_t j = pixel & 255u;
    const uint32_t in_x = ((j >> 6) & 1u) * 8u + (j & 7u), in_y = (j >> 7) * 8u + ((j >> 3) & 7u);
    const uint32_t tile_y = tile / p.tiles_x, tile_x = tile - tile_y * p.tiles_x;
    const uint32_t px = tile_x * kTileW + in_x, py = tile_y * kTileH + in_y;
    uint32_t rgba = 0u; // (0 outside the image: the shard's padding, as vrt_trace_kernel leaves it)
    if (px < p.width && py < p.height) { // comp:155-159
        const float4 *s = p.pool_samples + (size_t)(pixel >> 6) * spp * 64u + (pixel & 63u); // (groups of 64 pixels, sample-major inside)
        f3 acc = mk3(0, 0, 0);
        for (uint32_t k = 0; k < spp; k++) {
            const float4 t = s[(size_t)k * 64u];
            acc = acc + mk3(t.x, t.y, t.z);
        }
        const float fspp = (float)pc.cam.samples_per_pixel;
        const f3 c = mk3(__builtin_sqrtf(acc.x / fspp), __builtin_sqrtf(acc.y / fspp), __builtin_sqrtf(acc.z / fspp));
        const size_t o = (p.shard_count > 1u || p.packed_tiles) ? (size_t)owned * (kTileW * kTileH) + in_y * kTileW + in_x : (size_t)py * p.width + px;
        rgba = unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | (255u << 24);
        if (!p.packed_rgb) reinterpret_cast<uint32_t *>(p.target_rgba8 + (size_t)blockIdx.y * p.batch_target_stride)[o] = rgba;
        if (p.target_rgba32f) reinterpret_cast<float4 *>(p.target_rgba32f)[o] = make_float4(c.x, c.y, c.z, 1.0f);
    }
    if (p.packed_rgb) {
        // 16x16 tiles of 3-byte pixels, 768 bytes per tile: the eight threads of a row of this wave's 8x8 block hold 24 consecutive
        // bytes = 6 dwords; thread k < 6 of the row assembles dword k from the two pixels it spans (vrt_trace_kernel's store)
        const uint32_t lane = threadIdx.x & 63u, k = lane & 7u;
        const uint32_t first = k + (k >= 3u ? 1u : 0u); // = 4k / 3 for k < 6
        const int src = (int)((lane & ~7u) + first);
        const uint32_t lo = (uint32_t)__shfl((int)rgba, src, 64), hi = (uint32_t)__shfl((int)rgba, src + 1, 64);
        const uint32_t m = k % 3u;
        const uint32_t dword = (m == 0u) ? ((lo & 0xFFFFFFu) | (hi << 24)) : ((m == 1u) ? (((lo >> 8) & 0xFFFFu) | (hi << 16)) : (((lo >> 16) & 0xFFu) | (hi << 8)));
        if (k < 6u)
            reinterpret_cast<uint32_t *>(p.target_rgba8 + (size_t)blockIdx.y * p.batch_target_stride)[(size_t)owned * 192u + in_y * 12u + (in_x >> 3) * 6u + k] = dword;
    }
}

} // namespace vrt
