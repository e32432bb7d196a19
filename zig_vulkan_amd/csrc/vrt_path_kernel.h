// vrt_path_kernel.h — frames with bounces on scenes larger than the caches, round 2: persistent lanes (a ray per lane).  Included by
// vrt_pool_kernel.h / vrt_inst_path.hip.  Round 4's vrt_pool_kernel (a pool of 128 rays per wave) runs where it can; this kernel keeps
// 4^3 bricks, grids whose dimensions are not powers of two and scenes whose occupied cells do not reach the grid's faces.
#pragma once
#include "vrt_trace_kernels.h"

namespace vrt {

#ifndef VRT_DEV_VARIANTS
// (the product build never instantiates the template arguments that reach these loops: declarations for the discarded branches)
VRT_DI void grid_walk_ahead_gfx950(Walk &w, const f3 &inv_dir, AheadRing &a, uint32_t stride_x, uint32_t stride_y, uint32_t stride_z, u32x4 rsrc, AheadWalkRegs &g);
VRT_DI void grid_walk_park_dilated_ahead_gfx950(f3 &side_dist, const f3 &inv_dir, uint32_t &index, uint32_t &cell, uint32_t nm_x, uint32_t nm_y, uint32_t nm_z,
                                                uint32_t &word, u32x4 rsrc, GridParkRegs &g, uint32_t flip, unsigned long long &gone);
VRT_DI void grid_walk_park_dilated64_gfx950(f3 &side_dist, const f3 &inv_dir, uint32_t &index, uint32_t &cell, uint32_t nm_x, uint32_t nm_y, uint32_t nm_z,
                                            unsigned long long &word, u32x4 rsrc, GridParkRegs &g, uint32_t flip, unsigned long long &gone);
VRT_DI void grid_walk_park_dist_gfx950(Walk &w, const f3 &inv_dir, uint32_t &index, uint32_t &cell, uint32_t stride_x, uint32_t stride_y, uint32_t stride_z,
                                       uint32_t &word, u32x4 rsrc, GridParkRegs &g, DistRegs &d);
VRT_DI void skip_empty_block(Walk &w, const RaySetup &s, uint32_t bx, uint32_t by, uint32_t bz, uint32_t &index, uint32_t stride_x, uint32_t stride_y,
                             uint32_t stride_z, bool &more, int &in_axis, float &t_in);
#endif

// ---- frames with bounces: persistent lanes -------------------------------------------------------------------------
// vrt_trace_kernel<SHADE 0> runs the shader's loops in lockstep: the 64 lanes of a wave take sample s together, bounce k
// together, and every GridHit lasts as long as the longest of its 64 walks.  On the path-trace configuration (incoherent
// secondary rays through a sparse field, 16 samples, 3 bounces) that leaves about a fifth of the lanes of an instruction
// busy (367 wave-instructions per ray against ~70 for 64 rays in step; profiles/r02a_cfg4*).  Here a lane is not tied to
// its wave's progress: each lane carries its own (pixel, sample, bounce, ray) and moves through
//     FETCH a pixel -> SAMPLE (camera ray) -> START a ray (slab test, walk set-up) -> WALK (the hand-written park loop,
//     shared by primary, bounce and shadow rays of all lanes) -> DONE (shade: scatter, shadow ray, next bounce) -> END of
//     the path (tone-map, accumulate the sample) -> STORE the pixel -> FETCH ...
// A lane whose ray has left the grid is handed its next ray while its neighbours keep walking: the walk loop returns
// when `path_fin_batch` lanes have finished (or `brick_batch` lanes wait at a brick, or nobody is moving), the transitions
// run for the lanes that need them, and the loop is re-entered with every lane that has a ray.  Pixels come from one
// counter per frame (p.work_counter), 64 consecutive pixels of an 8x8 block at a time while the wave is empty.
// Per lane the sequence of arithmetic operations is exactly ray_color's / main()'s (comp:153-265): the samples of a pixel
// are traced one after the other by the lane that owns the pixel and summed in order, so frames are bit-identical.
enum : int { kLaneFetch = 0, kLaneSample, kLaneStart, kLaneWalk, kLaneDone, kLaneEnd, kLaneStore, kLaneExit };

// FILTER: 512-thread workgroups (eight waves, two per SIMD, share one LDS copy of the block filter); two workgroups per CU:
// 2 x (32 KiB filter + 8 x 4 KiB of staged bricks) = 128 of the CU's 160 KiB, four waves per SIMD.  (640-thread groups for five
// waves per SIMD do not pair up: ten waves leave the SIMDs 3/3/2/2, and 96 registers do not admit a sixth wave.)
// AHEAD (round 3): the walk loop pipelined two trips ahead (grid_walk_ahead_gfx950), on the shader's linear status words.
// DIST (round 3): the walk loop on the L1 distance field of the occupied cells (grid_walk_park_dist_gfx950).
// DIL (round 3): the half-block walk loop on a dilated cell index (all three dimensions powers of two): 1 = with the steps-left
// counters (grid_walk_park_dilated_gfx950: the walk ends at the box of the occupied cells), 2 = without them
// (grid_walk_park_dilated_carry_gfx950: the walk ends at the grid's face; chosen when the box is, or nearly is, the grid);
// 3 = 2 on 4 x 4 x 4-cell words (development); 4 = 2 with the DDA two cells ahead of the test (grid_walk_park_dilated_ahead_gfx950).
template <int B, int MIN_WAVES, bool FILTER, bool HALF = false, bool AHEAD = false, bool DIST = false, int DIL = 0>
__global__ __launch_bounds__(FILTER ? kPathFilterThreads : 256, FILTER ? 4 : MIN_WAVES) void vrt_path_kernel(const TraceParams p) {
    static_assert(!AHEAD || (!HALF && !FILTER), "the two-trips-ahead loop reads the linear status words");
    static_assert(!DIST || (!HALF && !FILTER && !AHEAD), "the distance-field loop has its own status structure");
    static_assert(!DIL || (!HALF && !FILTER && !AHEAD && !DIST), "the dilated-index loop is a walk kind of its own (it reads the half-block words)");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_block_filter[];
    FilterConsts fc{};
    if constexpr (FILTER) {
        // stage the block filter (1 bit per 4x4x4 block of cells: "some cell occupied") once per workgroup; the kernel has no
        // static LDS, so the dynamic region starts at LDS address 0, which the walk loop's ds_read relies on
        const uint32_t nblocks = p.nbx * p.nby * p.nbz;
        const uint32_t nwords = (nblocks + 31u) >> 5;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(p.status_blocks + (size_t)nblocks);
        for (uint32_t i = threadIdx.x; i < nwords; i += blockDim.x) lds_block_filter[i] = src[i];
        __syncthreads();
        const uint32_t lx = 31u - (uint32_t)__builtin_clz(p.grid.dim_x), lz = 31u - (uint32_t)__builtin_clz(p.grid.dim_z);
        fc.wx = lx - 2u;
        fc.shz = lx + 2u;
        fc.mz = (p.grid.dim_z >> 2) - 1u;
        fc.shy = lx + lz + 2u;
        fc.shyb = lx + lz - 4u;
    }
    // brick staging area of this wave (8^3 bricks, p.path_brick_lds): 4 KiB behind the block filter, as an LDS byte address
    [[maybe_unused]] const uint32_t wave_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)lds_block_filter +
                                               (FILTER ? p.path_lds_bytes : 0u) + (threadIdx.x >> 6) * 4096u;
    const PushConstants &pc = p.pcs[blockIdx.y];
    const uint32_t lane = threadIdx.x & 63u;
    const bool sun_enabled = pc.sun.enabled > 0;
    const int spp = pc.cam.samples_per_pixel;
    // Round 4, as vrt_pool_kernel (vrt_pool_kernel.h): where the context holds a sample buffer (TraceParams::pool_samples; the launcher
    // passes it for launches of one frame) the unit of work is one SAMPLE of a pixel — its term of comp:173's sum goes to the buffer and
    // vrt_pool_resolve_kernel finishes the pixel —, so that the frame does not end in a drain of half-finished pixels; and the units come
    // from the counter a chunk per wave at a time either way.
    const bool by_sample = p.pool_samples != nullptr; // (wave-uniform)
    // (a hit's material looked up by the transition that shades it — brick_walk_park_gfx950<..., DEFER> —, not in the development walks
    // whose `word` is not a plain register of the lane)
    constexpr bool kDefer = !AHEAD && DIL != 3;
    const uint32_t uspp = by_sample ? (uint32_t)max(1, spp) : 1u;
    const uint32_t total = p.owned_tiles * (uint32_t)(kTileW * kTileH) * uspp;
    uint32_t *const counter = p.work_counter + blockIdx.y;
    constexpr uint32_t kPathChunk = 256u;
    uint32_t chunk_next = 0u, chunk_end = 0u; // the wave's chunk of units (wave-uniform: every lane goes through the fetch below)
    bool more_chunks = true;
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
    const unsigned long long status_addr = (unsigned long long)p.brick_status;
    u32x4 rsrc;
    rsrc.x = (uint32_t)status_addr;
    rsrc.y = (uint32_t)(status_addr >> 32) | (4u << 16); // stride 4: one record per status word
    rsrc.z = p.status_words;
    rsrc.w = 0x00020000u;

    // the walk loop on half-block words (p.status_halfblocks: derived, 4 x 4 x 2 cells per word; eligible grids only)
    HalfBlockConsts hb;
    u32x4 hb_rsrc;
    constexpr bool halfblocks = HALF; // (a template parameter: two asm blocks with scalar outputs behind a run-time branch do not compile)
    {
        // (computed unconditionally and pinned to SGPRs: they are scalar operands of the hand-written loop)
        auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
        const uint32_t lx = 31u - (uint32_t)__builtin_clz(p.grid.dim_x | 4u), lz = 31u - (uint32_t)__builtin_clz(p.grid.dim_z | 4u);
        hb.nmask = uni(~(3u | (3u << lx) | (1u << (lx + lz))));
        hb.mx = uni((1u << (lx - 2u)) - 1u);
        hb.mzs = uni(((p.grid.dim_z >> 2) - 1u) << (lx - 2u));
        hb.mys = uni(~((1u << (lx + lz - 4u)) - 1u));
        hb.lx = uni(lx);
        hb.lxz = uni(lx + lz);
        const unsigned long long a = (unsigned long long)p.status_halfblocks;
        hb_rsrc.x = uni((uint32_t)a);
        hb_rsrc.y = uni((uint32_t)(a >> 32) | (4u << 16));
        hb_rsrc.z = uni(p.status_words);
        hb_rsrc.w = 0x00020000u;
    }
    // the walk loop on the distance field (p.cell_distance: derived, one byte per cell; a raw buffer, num_records = cells)
    [[maybe_unused]] u32x4 dist_rsrc;
    if constexpr (DIST) {
        auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
        const unsigned long long a = (unsigned long long)p.cell_distance;
        dist_rsrc.x = uni((uint32_t)a);
        dist_rsrc.y = uni((uint32_t)(a >> 32));
        dist_rsrc.z = uni(p.status_cells);
        dist_rsrc.w = 0x00020000u;
    }
    // the status word of a cell, in the layout the walk loop reads (DIST: the cell's distance byte; such a lane's word is an answer
    // for the cell it stands on and it has to ask in its next trip: fresh_word)
    [[maybe_unused]] uint32_t flip = 0u; // DIL, per lane: the field masks of the axes the ray walks down (index ^ flip = the real dilated index)
    auto status_word = [&](uint32_t index) {
        if constexpr (DIST) return (uint32_t)p.cell_distance[index];
        else if constexpr (DIL == 3) return 0u; // (64-bit words: status_word64)
        else if constexpr (DIL) return p.status_halfblocks[(index ^ flip) >> 5];
        else return halfblocks ? p.status_halfblocks[halfblock_word(hb, index)] : p.brick_status[index >> 5];
    };
    [[maybe_unused]] unsigned long long word64 = 0ull; // DIL 3: the lane's 4 x 4 x 4-cell word
    auto status_word64 = [&](uint32_t index) { return reinterpret_cast<const unsigned long long *>(p.status_blocks)[(index ^ flip) >> 6]; };
    [[maybe_unused]] u32x4 blk_rsrc;
    if constexpr (DIL == 3) {
        auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
        const unsigned long long a = (unsigned long long)p.status_blocks;
        blk_rsrc.x = uni((uint32_t)a);
        blk_rsrc.y = uni((uint32_t)(a >> 32) | (8u << 16)); // stride 8: one record per block word
        blk_rsrc.z = uni(p.nbx * p.nby * p.nbz);
        blk_rsrc.w = 0x00020000u;
    }
    [[maybe_unused]] DistRegs dr{0ull, 0};
    [[maybe_unused]] bool fresh_word = false; // DIST: the lane's word was loaded outside the walk loop since the last call

    // the by-cell copy of the occupancy bits and the start-index shortcut (derived structures, TraceParams), wave-uniform
    const bool by_cell = p.cell_occupancy != nullptr;
    const bool start_is_slot = p.start_is_slot != nullptr && __builtin_amdgcn_readfirstlane((int)*p.start_is_slot) != 0;

    // ---- per-lane state ----
    int st = kLaneFetch;
    uint32_t work = 0u;          // pixel: index into this context's tiles, 256 per tile, 8x8 blocks inside
    int sample_i = 0;
    f3 acc = mk3(0, 0, 0);       // sum of the samples' colours (comp:173)
    // the path (RayColor's locals, comp:203-216)
    int loop_count = 0;
    f3 color = mk3(0, 0, 0);
    float cur_dir_y = 0.0f;      // current_ray.direction.y, for BackgroundColor when loop_count ends at 0
    // the ray being walked: the path's current ray (kind 0) or the shadow ray of its last hit (kind 1)
    Ray r = Ray{mk3(0, 0, 0), mk3(0, 0, 1), 1.0f, MAT_NONE};
    int kind = 0;
    bool found = false;
    // kept while the shadow ray is walked: the scattered ray (its origin is the shadow ray's origin, hit.point), the
    // albedo, and whether the material scattered (comp:221-239)
    f3 sc_dir = mk3(0, 0, 1);
    float sc_ir = 1.0f;
    uint32_t sc_ignore = MAT_NONE;
    f3 attenuation = mk3(0, 0, 0);
    bool scattered_ok = false;
    // the walk (grid_hit's locals)
    RaySetup s;
    s.inv_dir = mk3(1, 1, 1);
    s.entry_code = 0;
    s.sx = s.sy = s.sz = 0;
    s.grid_t_min = s.grid_t_max = 0.0f;
    Walk w;
    w.side_dist = mk3(0, 0, 0);
    w.rx = w.ry = w.rz = -1;
    w.t_value = 0.0f;
    int base_x = 0, base_y = 0, base_z = 0;
    uint32_t grid_index = 0u, word = 0u;
    uint32_t stride_x = 0u, stride_y = 0u, stride_z = 0u;
    Hit hit;
    hit.point = hit.normal = mk3(0, 0, 0);
    hit.t = 0.0f;
    hit.index = 0u;
    int hit_axis = 0;
    GridParkRegs g;
    g.alive = 0ull;
    g.out_x = g.out_y = 0ull;
    g.t_out = g.t_in = 0.0f;
    g.code = 3u << 4;
    g.batch = p.path_brick_batch;
    [[maybe_unused]] AheadRing ring{0u, ~0u, ~0u, 0u, 0u, 0u, 0u, 2u, 0.0f, 0.0f, 0.0f}; // (AHEAD: the lane's three cells in flight)
    // FILTER: a walking lane is `ready` once its cell is known to lie in a block that holds occupied cells (it takes trips);
    // otherwise its block is looked up, and jumped over if empty.  `stale`: the lane has jumped since `word` was loaded.
    [[maybe_unused]] bool ready = false, stale = false;

    bool work_left = true; // wave-uniform
#ifdef VRT_DEV_PROFILE
    if (threadIdx.x < 8) vrt_prof[threadIdx.x] = 0ull; // (the brick walk's own phase hooks; static LDS: not with FILTER)
    __syncthreads();
    // development-only (make EXTRA=-DVRT_DEV_PROFILE, tools/path_profile.py): cycles and lane counts per phase, per wave
    unsigned long long pf_t[3] = {0ull, 0ull, 0ull};      // cycles in transitions / walk loop / bricks
    unsigned long long pf_n[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull}; // rounds: transitions, waiting lanes; walk calls, alive lanes at entry,
                                                                                    // alive lanes at exit; brick rounds, parked lanes; hits
#define VRT_PF_T(k, t0) pf_t[k] += __builtin_readcyclecounter() - (t0)
#define VRT_PF_N(k, v) pf_n[k] += (unsigned long long)(v)
#define VRT_PF_NOW() __builtin_readcyclecounter()
#else
#define VRT_PF_T(k, t0)
#define VRT_PF_N(k, v)
#define VRT_PF_NOW() 0ull
#endif
    for (;;) {
        // How many lanes wait for a transition?  Few: leave them waiting and keep the others walking (the divergent
        // code below costs the whole wave its issue slots).
        const unsigned long long walking0 = __builtin_amdgcn_ballot_w64(st == kLaneWalk);
        const unsigned long long waiting = __builtin_amdgcn_ballot_w64(st != kLaneWalk && st != kLaneExit);
        const uint32_t n_walking0 = (uint32_t)__builtin_popcountll(walking0), n_waiting = (uint32_t)__builtin_popcountll(waiting);
        if (n_walking0 == 0u && n_waiting == 0u) break;
        [[maybe_unused]] const unsigned long long pf0 = VRT_PF_NOW();
        if (n_waiting != 0u && (n_walking0 == 0u || n_waiting >= min(p.path_fin_batch, max(1u, n_walking0 >> 1)))) {
            VRT_PF_N(0, 1);
            VRT_PF_N(1, n_waiting);
            // (1) a ray has finished: comp:218-258 from the loop condition's GridHit onwards
            if (st == kLaneDone) {
                bool after_shadow = false;
                if (kind == 0) {
                    if (found) {
                        // (brick_walk_gfx950 records a hit as distance + material + face: comp:433-436 from those)
                        const float t_offset = (g_scale * (1.0f / (float)B)) * 0.05f;
                        hit.normal = axis_normal(s, hit_axis);
                        hit.point = ray_at(r, hit.t) + hit.normal * t_offset;
                        loop_count += 1;
                        Ray scattered = r;
                        bool result = false;
                        if (hit.index & kDeferredHit) {
                            // (round 4, as vrt_pool_kernel: the brick round left comp:337 / :422 / :425 of this hit to this round — the cell is
                            // in `word`, which the finished walk no longer needs)
                            const uint32_t hbrick = p.brick_index[word];
                            const uint32_t hstart = start_is_slot ? hbrick * (uint32_t)(B * B * B) : (p.brick_start_index[hbrick] & 0x7FFFFFFFu);
                            hit.index = p.material_index[hstart + (hit.index & ~kDeferredHit)];
                        }
                        const vrt_material *m = p.materials + hit.index;
                        const uint32_t mtype = m->type;
                        attenuation = mk3(m->albedo_r, m->albedo_g, m->albedo_b);
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
                            st = kLaneStart;
                        } else {
                            color = color + attenuation;
                            // the scattered ray starts where the shadow ray would have: keep the origin in r
                            r.origin = hit.point;
                            after_shadow = true;
                        }
                    } else {
                        cur_dir_y = r.direction.y;
                        st = kLaneEnd; // the while condition failed (comp:218)
                    }
                } else {
                    if (!found) color = color + attenuation * mk3(pc.sun.color[0], pc.sun.color[1], pc.sun.color[2]);
                    after_shadow = true;
                }
                if (after_shadow) {
                    if (!scattered_ok) {
                        st = kLaneEnd; // comp:253-255
                    } else {
                        r.direction = sc_dir; // current_ray = scattered (its origin, hit.point, is r.origin already)
                        r.internal_reflection = sc_ir;
                        r.ignore_type_material = sc_ignore;
                        cur_dir_y = sc_dir.y;
                        kind = 0;
                        st = (loop_count < max_bounce) ? kLaneStart : kLaneEnd;
                    }
                }
            }
            // (2) the path is over: comp:260-264, then the sample loop's accumulation (comp:173)
            if (st == kLaneEnd) {
                if (loop_count == 0) {
                    const f3 sun_color = mk3(pc.sun.color[0], pc.sun.color[1], pc.sun.color[2]);
                    const float t = 0.5f * (cur_dir_y + 1.0f);
                    const f3 bg = fma3(splat3(1.0f - t), splat3(1.0f), mk3(0.5f, 0.7f, 1.0f) * t);
                    color = color + bg * (sun_enabled ? sun_color : splat3(1.0f));
                }
                if (by_sample) {
                    // (the buffer's layout: groups of 64 pixels, sample-major inside a group — vrt_pool_resolve_kernel reads whole lines)
                    const f3 term = color / (color + splat3(1.0f));
                    p.pool_samples[((size_t)(work >> 6) * uspp + (uint32_t)sample_i) * 64u + (work & 63u)] = make_float4(term.x, term.y, term.z, 0.0f);
                    st = kLaneFetch;
                } else {
                    acc = acc + color / (color + splat3(1.0f));
                    sample_i += 1;
                    st = (sample_i < spp) ? kLaneSample : kLaneStore;
                }
            }
            // (3) the pixel is finished: comp:176-177
            if (st == kLaneStore) {
                const uint32_t owned = p.owned_tiles - 1u - (work >> 8);
                const uint32_t tile = p.own_period ? (owned / p.own_count) * p.own_period + p.own_slots[owned % p.own_count] : owned * p.shard_count + p.shard_rank;
                const uint32_t j = work & 255u;
                const uint32_t in_x = ((j >> 6) & 1u) * 8u + (j & 7u), in_y = (j >> 7) * 8u + ((j >> 3) & 7u);
                const uint32_t px = (tile % p.tiles_x) * kTileW + in_x, py = (tile / p.tiles_x) * kTileH + in_y;
                const float fspp = (float)spp;
                const f3 c = mk3(__builtin_sqrtf(acc.x / fspp), __builtin_sqrtf(acc.y / fspp), __builtin_sqrtf(acc.z / fspp));
                const size_t o = (p.shard_count > 1u || p.packed_tiles) ? (size_t)owned * (kTileW * kTileH) + in_y * kTileW + in_x : (size_t)py * p.width + px;
                reinterpret_cast<uint32_t *>(p.target_rgba8 + (size_t)blockIdx.y * p.batch_target_stride)[o] =
                    unorm8(c.x) | (unorm8(c.y) << 8) | (unorm8(c.z) << 16) | (255u << 24);
                if (p.target_rgba32f) reinterpret_cast<float4 *>(p.target_rgba32f)[o] = make_float4(c.x, c.y, c.z, 1.0f);
                st = kLaneFetch;
            }
            // (4) next unit (a pixel, or one sample of a pixel), from the wave's chunk of kPathChunk consecutive units: one atomic per chunk
            {
                const unsigned long long asking = __builtin_amdgcn_ballot_w64(st == kLaneFetch);
                if (asking != 0ull) {
                    if (work_left) {
                        const uint32_t n = (uint32_t)__builtin_popcountll(asking), rank = (uint32_t)__builtin_popcountll(asking & ((1ull << lane) - 1ull));
                        uint32_t unit = 0u;
                        bool got = false;
                        const uint32_t take = min(chunk_end - chunk_next, n);
                        if (st == kLaneFetch && rank < take) unit = chunk_next + rank, got = true;
                        chunk_next += take;
                        if (take < n && more_chunks) {
                            uint32_t first = 0u;
                            if (lane == (uint32_t)__builtin_ctzll(asking)) first = atomicAdd(counter, kPathChunk);
                            first = (uint32_t)__builtin_amdgcn_readlane((int)first, __builtin_ctzll(asking));
                            more_chunks = first < total;
                            chunk_next = more_chunks ? first : 0u;
                            chunk_end = more_chunks ? min(first + kPathChunk, total) : 0u;
                            const uint32_t take2 = min(chunk_end - chunk_next, n - take);
                            if (st == kLaneFetch && rank >= take && rank - take < take2) unit = chunk_next + (rank - take), got = true;
                            chunk_next += take2;
                        }
                        if (got) {
                            work = unit / uspp;
                            sample_i = (int)(unit - work * uspp);
                            acc = mk3(0, 0, 0);
                            st = kLaneSample;
                        }
                        work_left = more_chunks || chunk_next < chunk_end;
                    }
                    if (!work_left && st == kLaneFetch) st = kLaneExit;
                }
            }
            // (5) next sample of the pixel: comp:162-171
            if (st == kLaneSample) {
                const uint32_t owned = p.owned_tiles - 1u - (work >> 8);
                const uint32_t tile = p.own_period ? (owned / p.own_count) * p.own_period + p.own_slots[owned % p.own_count] : owned * p.shard_count + p.shard_rank;
                const uint32_t j = work & 255u;
                const uint32_t in_x = ((j >> 6) & 1u) * 8u + (j & 7u), in_y = (j >> 7) * 8u + ((j >> 3) & 7u);
                const uint32_t px = (tile % p.tiles_x) * kTileW + in_x, py = (tile / p.tiles_x) * kTileH + in_y;
                if (px >= p.width || py >= p.height) {
                    st = kLaneFetch; // outside the image (comp:155-159): nothing to trace, nothing to store
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
                    cur_dir_y = r.direction.y;
                    st = (loop_count < max_bounce) ? kLaneStart : kLaneEnd;
                }
            }
            // (6) a new ray: comp:271-312 (GridHit up to its loop)
            if (st == kLaneStart) {
                found = false;
                st = kLaneDone;
                if (grid_slab(p, r, 0.00001f, t_max, s)) {
                    const float global_t_value = s.grid_t_min + 0.0001f * g_scale; // comp:287
                    const f3 fposition = p.scale_pow2 ? (ray_at(r, global_t_value) - g_min) * p.inv_grid_scale : (ray_at(r, global_t_value) - g_min) / splat3(g_scale);
                    w.side_dist = initial_side_dist(mk3((float)s.sx, (float)s.sy, (float)s.sz), fposition, s.ray_delta());
                    const int px = f2i_clamp(__builtin_floorf(fposition.x));
                    const int py = f2i_clamp(__builtin_floorf(fposition.y));
                    const int pz = f2i_clamp(__builtin_floorf(fposition.z));
                    w.rx = steps_left_box(s.sx, px, lox, hix, zero_budget);
                    w.ry = steps_left_box(s.sy, py, loy, hiy, zero_budget);
                    w.rz = steps_left_box(s.sz, pz, loz, hiz, zero_budget);
                    base_x = walk_base_box(s.sx, px, lox, hix), base_y = walk_base_box(s.sy, py, loy, hiy), base_z = walk_base_box(s.sz, pz, loz, hiz);
                    w.t_value = 0;
                    grid_index = (uint32_t)px + (uint32_t)dx * ((uint32_t)pz + (uint32_t)dz * (uint32_t)py);
                    stride_x = (uint32_t)s.sx, stride_y = (uint32_t)s.sy * (uint32_t)dx * (uint32_t)dz, stride_z = (uint32_t)s.sz * (uint32_t)dx;
                    bool more = (global_t_value <= t_max) && (unsigned)px < (unsigned)dx && (unsigned)py < (unsigned)dy && (unsigned)pz < (unsigned)dz &&
                                (w.rx | w.ry | w.rz) >= 0;
                    int in_axis = 3; // the first cell of the walk was entered through the slab test, not by a step ...
                    float skip_t = 0.0f;
                    // ... unless the ray enters the grid in front of the occupied-cell box and jumps to its near face
                    if (p.cell_bounds && p.skip_to_box)
                        skip_to_box(w, s, (int)((uint32_t)hix - (uint32_t)lox), (int)((uint32_t)hiy - (uint32_t)loy), (int)((uint32_t)hiz - (uint32_t)loz), grid_index,
                                    stride_x, stride_y, stride_z, more, in_axis, skip_t);
                    if constexpr (DIL) {
                        if (more) {
                            // the walk's index in dilated form, from the cell the lane stands on (= base - step * steps left, after the
                            // jump to the box as well); axes walked down are stored mirrored; stride_* become the loop's per-axis
                            // "everything but this axis' field" masks (all ones: the axis is never stepped along)
                            const uint32_t lx = hb.lx, lz = hb.lxz - hb.lx;
                            const uint32_t ly = 31u - (uint32_t)__builtin_clz(p.grid.dim_y);
                            const uint32_t cx = (uint32_t)(base_x - __mul24(s.sx, w.rx)), cy = (uint32_t)(base_y - __mul24(s.sy, w.ry)),
                                           cz = (uint32_t)(base_z - __mul24(s.sz, w.rz));
                            const uint32_t mx = s.sx < 0 ? ((uint32_t)dx - 1u - cx) : cx, my = s.sy < 0 ? ((uint32_t)dy - 1u - cy) : cy,
                                           mz = s.sz < 0 ? ((uint32_t)dz - 1u - cz) : cz;
                            // (DIL 3: 4 x 4 x 4-cell words: two y bits among the low six, and everything above one bit higher)
                            constexpr uint32_t yb = DIL == 3 ? 2u : 1u, lo = 4u + yb;
                            const uint32_t fx = 3u | (((1u << (lx - 2u)) - 1u) << lo), fz = (3u << 2) | (((1u << (lz - 2u)) - 1u) << (lx + lo - 2u)),
                                           fy = (((1u << yb) - 1u) << 4) | (((1u << (ly - yb)) - 1u) << (lx + lz + lo - 4u));
                            grid_index = (mx & 3u) | ((mz & 3u) << 2) | ((my & ((1u << yb) - 1u)) << 4) | ((mx >> 2) << lo) | ((mz >> 2) << (lx + lo - 2u)) |
                                         ((my >> yb) << (lx + lz + lo - 4u));
                            flip = (s.sx < 0 ? fx : 0u) | (s.sy < 0 ? fy : 0u) | (s.sz < 0 ? fz : 0u);
                            stride_x = s.sx != 0 ? ~fx : ~0u, stride_y = s.sy != 0 ? ~fy : ~0u, stride_z = s.sz != 0 ? ~fz : ~0u;
                        }
                    }
                    if (more) {
                        if constexpr (FILTER) {
                            ready = false;
                            stale = true; // (the word is requested when the lane is about to take trips)
                        } else if constexpr (AHEAD) {
                            // prime the ring: q1 = the ray's first cell, q2 = the cell behind one step (comp:345-372), both words asked for
                            ring.q1 = grid_index;
                            ring.w1 = p.brick_status[grid_index >> 5];
                            ring.ts1 = skip_t;
                            int ax = 0;
                            dda_step<true>(w, s.ray_delta(), g_scale, ax, grid_index, stride_x, stride_y, stride_z);
                            ring.ts2 = w.t_value;
                            ring.hist = ((uint32_t)in_axis << 2) | (uint32_t)ax;
                            ring.q2 = (min3i(w.rx, w.ry, w.rz) < 0) ? ~0u : grid_index; // (the step left the box: the sentinel)
                            ring.w2 = (ring.q2 != ~0u) ? p.brick_status[ring.q2 >> 5] : 0u;
                        } else {
                            if constexpr (DIL == 3) word64 = status_word64(grid_index);
                            else word = status_word(grid_index);
                            fresh_word = true;
                        }
                        g.t_out = skip_t;
                        g.code = (uint32_t)in_axis << 4;
                        st = kLaneWalk;
                    }
                }
            }
        }
        VRT_PF_T(0, pf0);
        // (7) every lane that has a ray walks (comp:314-375), until enough of them are done for the next round of transitions
        unsigned long long walking = __builtin_amdgcn_ballot_w64(st == kLaneWalk);
        if (walking == 0ull) continue;
        [[maybe_unused]] const unsigned long long pf1 = VRT_PF_NOW();
        if constexpr (FILTER) {
            // (7a) lanes whose block is not known to hold occupied cells: look the block up (LDS); empty -> jump behind the step
            // that leaves it (no memory access), and again, until enough lanes are ready for trips or the round's budget is spent
            for (uint32_t it = 0; it < p.path_skip_rounds; it++) {
                const bool seeking = (st == kLaneWalk) && !ready;
                if (__builtin_amdgcn_ballot_w64(seeking) == 0ull) break;
                if ((uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(st == kLaneWalk && ready)) >= p.path_ready_batch) break;
                if (seeking) {
                    const uint32_t bi = ((grid_index >> 2) & ((1u << fc.wx) - 1u)) | (((grid_index >> fc.shz) & fc.mz) << fc.wx) | ((grid_index >> fc.shy) << fc.shyb);
                    if ((lds_block_filter[bi >> 5] >> (bi & 31u)) & 1u) {
                        ready = true;
                    } else {
                        bool more = true;
                        int in_axis = 0;
                        float t_in = 0.0f;
                        skip_empty_block(w, s, grid_index & 3u, (grid_index >> (fc.shy - 2u)) & 3u, (grid_index >> (fc.wx + 2u)) & 3u, grid_index, stride_x, stride_y,
                                         stride_z, more, in_axis, t_in);
                        g.t_out = t_in;
                        g.code = (uint32_t)in_axis << 4;
                        stale = true;
                        if (!more) {
                            found = false; // left the box of the occupied cells
                            st = kLaneDone;
                        }
                    }
                }
            }
            walking = __builtin_amdgcn_ballot_w64(st == kLaneWalk && ready);
            if (walking == 0ull) continue;
            if (st == kLaneWalk && ready && stale) {
                word = status_word(grid_index);
                stale = false;
            }
        }
        const uint32_t n_walking = (uint32_t)__builtin_popcountll(walking);
        const uint32_t fin = min(p.path_fin_batch, max(1u, n_walking >> 1));
        g.alive = walking;
        // (FILTER: five trips per call, the way through a block of four cells; then the blocks are looked up again)
        g.min_alive = FILTER ? 65u : (n_walking >= fin ? n_walking - fin + 1u : 1u);
        uint32_t cell; // the occupied cell each parked lane stood on before its last step
        [[maybe_unused]] unsigned long long gone = 0ull; // DIL 2, 3: the parked lanes whose step out of that cell left the grid
        if constexpr (AHEAD) {
            AheadWalkRegs ga;
            ga.alive = g.alive;
            ga.batch = g.batch;
            ga.min_alive = g.min_alive;
            grid_walk_ahead_gfx950(w, s.inv_dir, ring, stride_x, stride_y, stride_z, rsrc, ga);
            g.alive = ga.alive;
            g.parked = ga.parked;
            cell = ring.q0;
        } else if constexpr (DIL == 4) {
            grid_walk_park_dilated_ahead_gfx950(w.side_dist, s.inv_dir, grid_index, cell, stride_x, stride_y, stride_z, word, hb_rsrc, g, flip, gone);
        } else if constexpr (DIL == 3) {
            grid_walk_park_dilated64_gfx950(w.side_dist, s.inv_dir, grid_index, cell, stride_x, stride_y, stride_z, word64, blk_rsrc, g, flip, gone);
        } else if constexpr (DIL == 2) {
            grid_walk_park_dilated_carry_gfx950(w.side_dist, s.inv_dir, grid_index, cell, stride_x, stride_y, stride_z, word, hb_rsrc, g, flip, gone);
        } else if constexpr (DIL == 1) {
            grid_walk_park_dilated_gfx950(w, s.inv_dir, grid_index, cell, stride_x, stride_y, stride_z, word, hb_rsrc, g, flip);
        } else if constexpr (DIST) {
            const unsigned long long fresh = __builtin_amdgcn_ballot_w64(fresh_word);
            dr.pend |= fresh;
            if (fresh_word) dr.k = 0;
            fresh_word = false;
            grid_walk_park_dist_gfx950(w, s.inv_dir, grid_index, cell, stride_x, stride_y, stride_z, word, dist_rsrc, g, dr);
        } else if constexpr (halfblocks) grid_walk_park_halfblocks_gfx950(w, s.inv_dir, grid_index, cell, stride_x, stride_y, stride_z, word, hb_rsrc, g, hb);
        else grid_walk_park_gfx950(w, s.inv_dir, grid_index, cell, stride_x, stride_y, stride_z, word, rsrc, g);
        VRT_PF_T(1, pf1);
        VRT_PF_N(2, 1);
        VRT_PF_N(3, n_walking);
        VRT_PF_N(4, __builtin_popcountll(g.alive));
        [[maybe_unused]] const unsigned long long pf2 = VRT_PF_NOW();
        const bool was_walking = (walking >> lane) & 1ull;
        const bool parked = __builtin_amdgcn_inverse_ballot_w64(g.parked);
        const bool moving = __builtin_amdgcn_inverse_ballot_w64(g.alive);
        if (was_walking && !parked && !moving) {
            found = false; // left the box of the occupied cells
            st = kLaneDone;
        }
        if constexpr (FILTER) {
            if (was_walking) ready = false; // it has moved: its block is looked up again
        }
        if (g.parked != 0ull) {
            VRT_PF_N(5, 1);
            VRT_PF_N(6, __builtin_popcountll(g.parked));
            if (parked) {
                // AHEAD: the lane's DDA state is two steps beyond the occupied cell q0: undo both decrements; q0 was entered through
                // the axis in hist bits 4-5 at distance ts0.  Otherwise the lane has taken one step out of the cell.
                int a = AHEAD ? (int)((ring.hist >> 4) & 3u) : (int)(g.code & 3u);
                const uint32_t out = AHEAD ? ((ring.hist >> 2) & 3u) : ((g.code >> 2) & 3u), out2 = AHEAD ? (ring.hist & 3u) : 3u;
                const float t_into = AHEAD ? ring.ts0 : g.t_in;
                const int rx = w.rx + (out == 0u ? 1 : 0) + (out2 == 0u ? 1 : 0), ry = w.ry + (out == 1u ? 1 : 0) + (out2 == 1u ? 1 : 0),
                          rz = w.rz + (out == 2u ? 1 : 0) + (out2 == 2u ? 1 : 0);
                int cx = base_x - __mul24(s.sx, rx), cy = base_y - __mul24(s.sy, ry), cz = base_z - __mul24(s.sz, rz); // cell position
                if constexpr (DIL >= 2) { // (no counters: the position is the loop's own index, un-mirrored and un-dilated)
                    constexpr uint32_t yb = DIL == 3 ? 2u : 1u, lo = 4u + yb;
                    const uint32_t real = cell ^ flip, lx = hb.lx, lz = hb.lxz - hb.lx;
                    cx = (int)((real & 3u) | ((real >> (lo - 2u)) & (((1u << (lx - 2u)) - 1u) << 2)));
                    cz = (int)(((real >> 2) & 3u) | ((real >> (lx + lo - 4u)) & (((1u << (lz - 2u)) - 1u) << 2)));
                    cy = (int)(((real >> 4) & ((1u << yb) - 1u)) | ((real >> (lx + lz + lo - 4u)) << yb));
                }
                if constexpr (DIL) cell = (uint32_t)cx + (uint32_t)dx * ((uint32_t)cz + (uint32_t)dz * (uint32_t)cy); // (the loop's index is dilated)
                const uint32_t occ_slot = by_cell ? cell : p.brick_index[cell]; // comp:337 (by_cell: only on a solid voxel)
                if constexpr (B == 8) {
                    if (p.path_brick_lds) stage_brick_lds(p, occ_slot, by_cell, wave_lds); // (first: the arithmetic below runs while the brick arrives)
                }
                const f3 brick_min = fma3(mk3((float)cx, (float)cy, (float)cz), splat3(g_scale), g_min);  // comp:331
                const float global_t_value = t_into * g_scale + s.grid_t_min + 0.01f * g_scale;          // comp:347 (deferred) + comp:332
                hit.t = global_t_value;
                bool hit_voxel;
                if constexpr (B == 8) {
                    hit_voxel = p.path_brick_lds ? brick_walk_park_gfx950<B, true, true, kDefer>(p, r, s, g_scale, occ_slot, cell, by_cell, start_is_slot, brick_min, hit, a, hit_axis, wave_lds)
                                                 : brick_walk_park_gfx950<B, false, false, kDefer>(p, r, s, g_scale, occ_slot, cell, by_cell, start_is_slot, brick_min, hit, a, hit_axis);
                } else {
                    hit_voxel = brick_walk_park_gfx950<B, false, false, kDefer>(p, r, s, g_scale, occ_slot, cell, by_cell, start_is_slot, brick_min, hit, a, hit_axis);
                }
                if (hit_voxel) {
                    found = true;
                    st = kLaneDone;
                    if constexpr (kDefer) {
                        if (hit.index & kDeferredHit) word = cell; // (the hit's material is looked up by the transition: where it lies)
                    }
                } else if (!(global_t_value <= t_max) || (DIL >= 2 ? __builtin_amdgcn_inverse_ballot_w64(gone) : (!AHEAD && min3i(w.rx, w.ry, w.rz) < 0))) {
                    found = false; // t became NaN (comp:316), or the step out of this cell left the box
                    st = kLaneDone;
                } else if constexpr (!AHEAD) {
                    if constexpr (DIL == 3) word64 = status_word64(grid_index);
                    else word = status_word(grid_index); // (an A-trip park left the lane's word in the other register set)
                    fresh_word = true;
                }   // (AHEAD: the lane walks on as it is; a step that left the box has put the sentinel into its ring)
            }
        }
        // every lane of the call: the axis of its last step, for its first trip in the next call
        if (!AHEAD && was_walking)
            g.code = parked ? ((g.code >> 2) & 3u) << 4
                            : (__builtin_amdgcn_inverse_ballot_w64(g.out_x) ? 0u : (__builtin_amdgcn_inverse_ballot_w64(g.out_y) ? 1u : 2u)) << 4;
        VRT_PF_T(2, pf2);
    }
#ifdef VRT_DEV_PROFILE
    if (p.wave_timeline && lane == 0u) {
        for (int k = 0; k < 3; k++) atomicAdd(&p.wave_timeline[k], pf_t[k]);
        for (int k = 0; k < 8; k++) atomicAdd(&p.wave_timeline[3 + k], pf_n[k]);
        atomicAdd(&p.wave_timeline[11], 1ull);
    }
    __syncthreads();
    if (p.wave_timeline && threadIdx.x < 8) atomicAdd(&p.wave_timeline[12 + threadIdx.x], vrt_prof[threadIdx.x]);
#endif
#undef VRT_PF_T
#undef VRT_PF_N
#undef VRT_PF_NOW
}

} // namespace vrt
