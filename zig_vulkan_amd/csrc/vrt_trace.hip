// vrt_trace.hip — derived-structure builders, tile schedule, root-side un-swizzle, and the launchers / kernel selection
// called from vrt_api.hip.  The traversal kernels themselves live in vrt_trace_kernels.h and are instantiated by vrt_inst_*.hip.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <climits>
#include "vrt_internal.h"
#include "vrt_kernels.h"

namespace vrt {

// Builds the derived status structures from the uploaded brick_status words (binding 3):
// out[0 .. nblocks)            one uint2 per 4x4x4 block of cells
// out[nblocks ..) as u32 words  filter: bit b set iff block b has any occupied cell
// One thread per block; the filter words are produced by a wave ballot (32 blocks per word).
__global__ __launch_bounds__(256) void vrt_build_status_blocks(const uint32_t *__restrict__ status, uint2 *__restrict__ out, uint32_t dim_x,
                                                               uint32_t dim_y, uint32_t dim_z, uint32_t nbx, uint32_t nby, uint32_t nbz) {
    const uint32_t nblocks = nbx * nby * nbz;
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    uint2 word = make_uint2(0u, 0u);
    if (b < nblocks) {
        const uint32_t bx = b % nbx, bz = (b / nbx) % nbz, by = b / (nbx * nbz);
        for (uint32_t yy = 0; yy < 4u; yy++)
            for (uint32_t zz = 0; zz < 4u; zz++)
                for (uint32_t xx = 0; xx < 4u; xx++) {
                    const uint32_t x = bx * 4u + xx, y = by * 4u + yy, z = bz * 4u + zz;
                    if (x < dim_x && y < dim_y && z < dim_z) {
                        const uint32_t gi = x + dim_x * (z + dim_z * y);
                        const uint32_t bit = (status[gi >> 5] >> (gi & 31u)) & 1u;
                        const uint32_t pos = xx + 4u * zz + 16u * yy;
                        if (pos < 32u) word.x |= bit << pos;
                        else word.y |= bit << (pos - 32u);
                    }
                }
        out[b] = word;
    }
    const unsigned long long nonempty = __ballot((word.x | word.y) != 0u);
    uint32_t *filter = reinterpret_cast<uint32_t *>(out + nblocks);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t base_word = (blockIdx.x * 256u + (threadIdx.x & ~63u)) >> 5; // first filter word of this wave
    const uint32_t nwords = (nblocks + 31u) >> 5;
    if (lane == 0 && base_word < nwords) filter[base_word] = (uint32_t)(nonempty & 0xFFFFFFFFull);
    if (lane == 32 && base_word + 1u < nwords) filter[base_word + 1u] = (uint32_t)(nonempty >> 32);
}

// The status bits ordered by half-blocks of 4 x 4 x 2 cells (TraceParams::status_halfblocks, grid_walk_park_halfblocks_gfx950):
// word (x>>2) + (dim_x/4) * ((z>>2) + (dim_z/4) * (y>>1)), bit (x&3) | (z&3) << 2 | (y&1) << 4.  One thread per word.
__global__ __launch_bounds__(256) void vrt_build_status_halfblocks(const uint32_t *__restrict__ status, uint32_t *__restrict__ out, uint32_t dim_x,
                                                                   uint32_t dim_y, uint32_t dim_z) {
    const uint32_t nx = dim_x >> 2, nz = dim_z >> 2, ny = dim_y >> 1;
    const uint32_t wi = blockIdx.x * 256u + threadIdx.x;
    if (wi >= nx * nz * ny) return;
    const uint32_t bx = wi % nx, bz = (wi / nx) % nz, by = wi / (nx * nz);
    uint32_t word = 0u;
    for (uint32_t k = 0; k < 32u; k++) {
        const uint32_t x = bx * 4u + (k & 3u), z = bz * 4u + ((k >> 2) & 3u), y = by * 2u + (k >> 4);
        const uint32_t gi = x + dim_x * (z + dim_z * y);
        word |= ((status[gi >> 5] >> (gi & 31u)) & 1u) << k;
    }
    out[wi] = word;
}

// The L1 distance field of the occupied cells (TraceParams::cell_distance; grid_walk_park_dist_gfx950): one byte per cell, 0 where
// the status bit is set, else min(255, Manhattan distance in cells to the nearest set bit).  The L1 distance transform separates:
// seed 0 / 255, then along each axis in turn a forward and a backward sweep d = min(d, neighbour + 1) — exact after x, z, y (the
// cap commutes with min and + 1).  One thread per status word for the seed; one thread per line of cells for a sweep (the line of
// thread t along axis a: the t-th combination of the other two coordinates; consecutive threads are consecutive in the fastest
// remaining coordinate).
#ifdef VRT_DEV_VARIANTS // (only vrt_path_kernel<DIST>, a development variant, reads the field)
__global__ __launch_bounds__(256) void vrt_build_distance_seed(const uint32_t *__restrict__ status, uint8_t *__restrict__ out, uint32_t words, uint32_t cells) {
    const uint32_t wi = blockIdx.x * 256u + threadIdx.x;
    if (wi >= words) return;
    const uint32_t bits = status[wi];
    for (uint32_t k = 0; k < 32u; k++) {
        const uint32_t gi = wi * 32u + k;
        if (gi < cells) out[gi] = ((bits >> k) & 1u) ? 0u : 255u;
    }
}
__global__ __launch_bounds__(256) void vrt_build_distance_sweep(uint8_t *__restrict__ d, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, uint32_t axis) {
    // cell index = x + dim_x * (z + dim_z * y)
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    uint32_t n, stride, first;
    if (axis == 0u) { // lines along x: one per (z, y)
        if (t >= dim_z * dim_y) return;
        n = dim_x, stride = 1u, first = t * dim_x;
    } else if (axis == 1u) { // lines along z: one per (x, y)
        if (t >= dim_x * dim_y) return;
        n = dim_z, stride = dim_x, first = (t % dim_x) + dim_x * dim_z * (t / dim_x);
    } else { // lines along y: one per (x, z)
        if (t >= dim_x * dim_z) return;
        n = dim_y, stride = dim_x * dim_z, first = t;
    }
    uint32_t run = 255u;
    for (uint32_t i = 0; i < n; i++) {
        uint8_t *c = d + (size_t)first + (size_t)i * stride;
        run = min(min(run + 1u, 255u), (uint32_t)*c);
        *c = (uint8_t)run;
    }
    run = 255u;
    for (uint32_t i = n; i-- > 0u;) {
        uint8_t *c = d + (size_t)first + (size_t)i * stride;
        run = min(min(run + 1u, 255u), (uint32_t)*c);
        *c = (uint8_t)run;
    }
}
#endif

// One byte per grid cell (TraceParams::status_bytes): 1 where the cell's status bit is set.  One thread per status word.
__global__ __launch_bounds__(256) void vrt_build_status_bytes(const uint32_t *__restrict__ status, uint8_t *__restrict__ out, uint32_t words, uint32_t /*cells*/) {
    const uint32_t wi = blockIdx.x * 256u + threadIdx.x;
    if (wi >= words) return;
    const uint32_t bits = status[wi];
    uint32_t *dst = reinterpret_cast<uint32_t *>(out + (size_t)wi * 32u); // (the allocation holds 32 bytes per status word)
    for (uint32_t k = 0; k < 8u; k++) {
        const uint32_t nib = (bits >> (4u * k)) & 0xFu;
        const uint32_t v = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);
        dst[k] = v;
    }
}

// The occupancy bits of every occupied cell's brick, stored by cell (TraceParams::cell_occupancy): one thread per 8-byte word of
// the copy; words8 = B^3 / 64 words per brick.  Cells whose status bit is clear are never read by the kernels and are left alone.
// Refresh: the cells [scan_lo, scan_hi) are looked at; a cell's words are copied if it lies in [cell_lo, cell_hi) (its status bit or
// brick index was written) or its brick's slot in [slot_lo, slot_hi) (that brick's occupancy bytes were written).
__global__ __launch_bounds__(256) void vrt_build_cell_occupancy(const uint32_t *__restrict__ status, const uint32_t *__restrict__ brick_index,
                                                                const uint2 *__restrict__ occupancy, uint2 *__restrict__ out, uint64_t scan_lo, uint64_t scan_hi,
                                                                uint32_t words8, uint64_t brick_alloc, uint64_t cell_lo, uint64_t cell_hi, uint64_t slot_lo,
                                                                uint64_t slot_hi) {
    const uint64_t w = scan_lo * words8 + (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t cell = w / words8;
    if (cell >= scan_hi) return;
    if (!((status[cell >> 5] >> (cell & 31u)) & 1u)) return;
    const uint32_t slot = brick_index[cell];
    if (slot >= brick_alloc) return; // (malformed scene: the shader would read outside binding 5)
    if (!((cell >= cell_lo && cell < cell_hi) || (slot >= slot_lo && slot < slot_hi))) return;
    out[w] = occupancy[(uint64_t)slot * words8 + (w % words8)];
}


// TraceParams::cell_material: a wave looks at 64 consecutive cells; for every occupied one whose inputs were written — the cell itself
// (status bit / brick index in [cell_lo, cell_hi)), its brick's slot (occupancy bytes / start index in [slot_lo, slot_hi)) or its
// brick's material entries (bytes [mat_lo, mat_hi) of binding 7) — the lanes share the brick's voxels (B^3 / 64 each), find the first
// solid voxel's material and whether every other solid voxel has it too.  0xFF: mixed, no solid voxel, a malformed brick, or the id 255.
template <int B>
__global__ __launch_bounds__(256) void vrt_build_cell_material(const uint32_t *__restrict__ status, const uint32_t *__restrict__ brick_index,
                                                               const uint8_t *__restrict__ occupancy, const uint32_t *__restrict__ start_index,
                                                               const uint8_t *__restrict__ material_index, uint8_t *__restrict__ out, uint32_t cells,
                                                               uint32_t status_words, uint64_t brick_alloc, uint64_t material_bytes, uint64_t cell_lo,
                                                               uint64_t cell_hi, uint64_t slot_lo, uint64_t slot_hi, uint64_t mat_lo, uint64_t mat_hi) {
    constexpr uint32_t kBits = B * B * B, kPerLane = kBits / 64u;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t base = ((uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * 64u;
    if (base >= cells) return;
    const uint32_t w0 = (uint32_t)(base >> 5);
    unsigned long long bits = status[w0] | (w0 + 1u < status_words ? (unsigned long long)status[w0 + 1u] << 32 : 0ull);
    while (bits) { // (uniform over the wave)
        const uint32_t b = (uint32_t)__builtin_ctzll(bits);
        bits &= bits - 1ull;
        const uint64_t cell = base + b;
        if (cell >= cells) break;
        const uint32_t slot = brick_index[cell];
        if (slot >= brick_alloc) continue; // (malformed scene: the shader would read outside bindings 5 / 6)
        const uint64_t start = start_index[slot] & 0x7FFFFFFFu; // comp:422
        const bool written = (cell >= cell_lo && cell < cell_hi) || (slot >= slot_lo && slot < slot_hi) || (start < mat_hi && start + kBits > mat_lo);
        if (!written) continue;
        uint32_t first = 0xFFu;
        bool solid = false, same = true;
        if (start + kBits <= material_bytes) {
            if constexpr (kPerLane == 8u) {
                const uint32_t occ = occupancy[(uint64_t)slot * (kBits / 8u) + lane]; // voxels 8 lane .. 8 lane + 7 (Grid.zig:180-182)
#pragma unroll
                for (uint32_t k = 0; k < 8u; k++) {
                    if (!((occ >> k) & 1u)) continue;
                    const uint32_t id = material_index[start + 8u * lane + k]; // comp:425
                    if (!solid) first = id, solid = true;
                    else same = same && id == first;
                }
            } else {
                solid = ((occupancy[(uint64_t)slot * (kBits / 8u) + (lane >> 3)] >> (lane & 7u)) & 1u) != 0u;
                if (solid) first = material_index[start + lane];
            }
        }
        const unsigned long long any = __builtin_amdgcn_ballot_w64(solid);
        uint32_t id = 0xFFu;
        if (any != 0ull) {
            const uint32_t leader = (uint32_t)__builtin_ctzll(any);
            const uint32_t m = (uint32_t)__shfl((int)first, (int)leader, 64);
            if (__builtin_amdgcn_ballot_w64(solid && (!same || first != m)) == 0ull) id = m;
        }
        if (lane == 0u) out[cell] = (uint8_t)id;
    }
}

// *flag = 1 iff every brick's start index (binding 6) is either unset (0xFFFFFFFF) or slot * bits in its low 31 bits
// (TraceParams::start_is_slot).  The flag is set to 1 before the launch; violators clear it.
__global__ __launch_bounds__(256) void vrt_check_start_is_slot(const uint32_t *__restrict__ start, uint32_t *__restrict__ flag, uint64_t brick_alloc,
                                                               uint32_t bits) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= brick_alloc) return;
    const uint32_t v = start[i];
    if (v != 0xFFFFFFFFu && (uint64_t)(v & 0x7FFFFFFFu) != i * bits) *flag = 0u;
}

// *flag = 1 iff no material record has the type MAT_NONE (TraceParams::materials_plain).  Set to 1 before the launch; violators clear it.
__global__ __launch_bounds__(256) void vrt_check_materials_plain(const vrt_material *__restrict__ materials, uint32_t *__restrict__ flag, uint32_t count) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < count && materials[i].type == 3u /* MAT_NONE */) *flag = 0u;
}

// Bounding box of the occupied grid cells (TraceParams::cell_bounds), from the status bits of binding 3: one thread per
// status word, six atomic maxima over {-x, -y, -z, x, y, z}; bounds[] starts as 0x80808080 (hipMemsetAsync 0x80).
__global__ __launch_bounds__(256) void vrt_build_cell_bounds(const uint32_t *__restrict__ status, int *__restrict__ bounds, uint32_t words,
                                                             uint32_t cells, uint32_t dim_x, uint32_t dim_z) {
    const uint32_t wi = blockIdx.x * 256u + threadIdx.x;
    if (wi >= words) return;
    uint32_t bits = status[wi];
    if (wi == words - 1u && (cells & 31u)) bits &= (1u << (cells & 31u)) - 1u; // bits beyond the last cell mean nothing
    if (bits == 0u) return;
    int m[6] = {INT_MIN, INT_MIN, INT_MIN, INT_MIN, INT_MIN, INT_MIN};
    while (bits) {
        const uint32_t b = (uint32_t)__builtin_ctz(bits);
        bits &= bits - 1u;
        const uint32_t i = wi * 32u + b; // x + dim_x * (z + dim_z * y), comp:318
        const int x = (int)(i % dim_x), z = (int)((i / dim_x) % dim_z), y = (int)(i / (dim_x * dim_z));
        m[0] = max(m[0], -x), m[1] = max(m[1], -y), m[2] = max(m[2], -z);
        m[3] = max(m[3], x), m[4] = max(m[4], y), m[5] = max(m[5], z);
    }
#pragma unroll
    for (int k = 0; k < 6; k++) atomicMax(&bounds[k], m[k]);
}

// Cost-feedback schedule: order[] = owned tile ids in kScheduleBuckets classes of cost (wave-cycles of the most
// recent frame, relative to the maximum), heaviest class first; INSIDE a class the tiles keep the default
// reverse-raster order, so that consecutive workgroups still render neighbouring tiles (a full sort by cost
// measured 6-7 % slower on views without outliers: neighbouring tiles share the lines of the bitmaps they walk).
// With extra > 0 the tiles whose slowest wave would outlast the rest of the frame (longer than 1.25 x the frame's
// wave-cycles spread over all wave slots) are SPLIT: two entries, each renders half of the tile with 32 lanes per
// wave, in a class of their own at the front; at most `extra` tiles, unused spare entries are ~0.  snap[n .. 2n)
// remembers which tiles are split (their measured waves are scaled back up by 1.28 before the test, the measured
// gain of splitting, so that a split tile does not flip back and forth).
// One workgroup.  cost[] ([half][tile][wave], overwritten by every frame) is condensed into snap[] first: frames
// on another stream may still be writing cost[], and every pass below must see the same values or order[] would
// not be a permutation.  The order affects timing only, never pixels.
// Round 4: the halves of split tiles are no longer ONE class in reverse-raster order but kSplitBuckets classes by the tile's slowest
// wave, slowest first — on the reference app's own run nearly every tile that holds terrain is split (3 000 half-tile workgroups for
// 1 024 workgroup slots), and with them unordered the second generation of workgroups held waves as long as the first's longest:
// they started at 100 us and ended at 300 (tools/experiments/timeline_tail.py).
constexpr uint32_t kScheduleBuckets = 8u;
constexpr uint32_t kSplitBuckets = 8u;
constexpr uint32_t kAllBuckets = kScheduleBuckets + kSplitBuckets;
__global__ __launch_bounds__(1024) void vrt_schedule_kernel(const uint32_t *__restrict__ cost, uint32_t *__restrict__ snap,
                                                            const uint32_t *prev_order, uint32_t *order, uint32_t n, uint32_t extra_max,
                                                            uint32_t extra, uint32_t wave_slots) {
    __shared__ uint32_t s_max, s_nsplit, s_longest;
    __shared__ unsigned long long s_total;
    __shared__ uint32_t wave_total[kAllBuckets][16];
    __shared__ uint32_t class_total[kAllBuckets];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    uint32_t *state = snap + n;
    // order[] is stored XCD-major: entry k at (k % 8) * row + k / 8.  extra_max: the spare entries the buffer holds (its layout);
    // extra <= extra_max: how many tiles this sort may split (the frames launch n + extra workgroups)
    const uint32_t row = (n + extra_max + 7u) >> 3;
    if (tid == 0) {
        s_max = 0u;
        s_nsplit = 0u;
        s_longest = 0u;
        s_total = 0ull;
    }
    __syncthreads();
    uint32_t m = 0u, ml = 0u;
    unsigned long long total = 0ull;
    const uint4 *cost4 = reinterpret_cast<const uint4 *>(cost);
#pragma unroll 4
    for (uint32_t i = tid; i < n; i += 1024u) {
        const uint4 w = cost4[i]; // the four waves of the tile, most recent frame
        uint32_t v = w.x + w.y + w.z + w.w;
        uint32_t longest = max(max(w.x, w.y), max(w.z, w.w));
        const uint32_t was_split = extra ? (state[i] & 1u) : 0u;
        if (was_split) {
            const uint4 w2 = cost4[n + i]; // the waves of the second half
            v += w2.x + w2.y + w2.z + w2.w;
            longest = max(longest, max(max(w2.x, w2.y), max(w2.z, w2.w)));
            longest += (longest >> 2) + (longest >> 5); // what the wave would take unsplit
        }
        // (the classes come from the mean of this measurement and the running value: a tile's cycles depend on what
        // runs beside it, i.e. on the order itself, and an order made from one frame's costs alone keeps changing)
        const uint32_t prev = snap[i];
        if (prev) v = (uint32_t)(((unsigned long long)prev + v) >> 1);
        snap[i] = v;
        state[i] = (min(longest, 0x7FFFFFFFu) << 1) | was_split;
        m = max(m, v);
        ml = max(ml, longest);
        total += v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        m = max(m, (uint32_t)__shfl_down(m, off, 64));
        ml = max(ml, (uint32_t)__shfl_down(ml, off, 64));
        total += __shfl_down(total, off, 64);
    }
    if (lane == 0u) {
        atomicMax(&s_max, m);
        atomicMax(&s_longest, ml);
        atomicAdd(&s_total, total);
    }
    __syncthreads(); // (also orders the snap[] / state[] writes before the reads of other threads below)
    const uint32_t mx = s_max;
    if (mx == 0u) { // no measurement yet: keep the current order
        if (order != prev_order)
            for (uint32_t i = tid; i < 8u * row; i += 1024u) order[i] = prev_order[i];
        for (uint32_t i = tid; i < n; i += 1024u) state[i] &= 1u;
        return;
    }
    // A frame whose slowest wave is well below (0.6 x) the time the frame needs anyway (its wave-cycles spread over all
    // wave slots) keeps the plain reverse-raster order: there the launch order cannot shorten anything.  Otherwise: cost
    // classes, and which tiles to split.
    const unsigned long long par = s_total / (wave_slots ? wave_slots : 1u);
    const bool reorder = (unsigned long long)s_longest * 5ull > par * 3ull;
    const unsigned long long threshold = par + (par >> 2);
    const uint32_t top_bar = (uint32_t)(0.85f * (float)s_longest); // (0.7: the headline's V1 +3 %; none: V1 / V2 +2 %)
#pragma unroll 4
    for (uint32_t i = tid; i < n; i += 1024u) {
        // (... and the frame's very longest waves whatever the bar says: a frame whose slowest wave is all of it — the headline's V1 / V2 —
        // ends with that wave)
        uint32_t want = (reorder && extra && ((unsigned long long)(state[i] >> 1) > threshold || (state[i] >> 1) > top_bar)) ? 1u : 0u;
        if (want && atomicAdd(&s_nsplit, 1u) >= extra) want = 0u; // no spare entry left
        state[i] = (state[i] & ~1u) | want; // (bits 1-31: the tile's slowest wave, for the class of a split tile below)
    }
    __syncthreads();
    const uint32_t nsplit = min(s_nsplit, extra);
    const float scale = reorder ? (float)kScheduleBuckets / (float)mx : 0.0f; // (0: one class, i.e. reverse raster)
    // ONE scale for all workgroups when tiles are split: the expected time of the workgroup's slowest wave — the tile's slowest wave,
    // over 1.28 for a half (what splitting gains) — so that a whole tile with a long wave is not launched behind every half
    const float split_scale = (float)kAllBuckets / (float)max(1u, s_longest);
    auto class_of = [&](uint32_t tile, uint32_t st) {
        if (nsplit != 0u) return min(kAllBuckets - 1u, (uint32_t)((float)(st >> 1) * ((st & 1u) ? 0.78125f : 1.0f) * split_scale));
        return min(kScheduleBuckets - 1u, (uint32_t)((float)snap[tile] * scale));
    };
    // thread t owns positions [lo, hi) of the default order (position j = tile n-1-j)
    const uint32_t chunk = (n + 1023u) / 1024u;
    const uint32_t lo = min(n, tid * chunk), hi = min(n, lo + chunk);
    uint32_t cnt[kAllBuckets]; // classes kScheduleBuckets ...: the halves of split tiles
#pragma unroll
    for (uint32_t b = 0; b < kAllBuckets; b++) cnt[b] = 0u;
#pragma unroll 4
    for (uint32_t j = lo; j < hi; j++) {
        const uint32_t tile = n - 1u - j;
        const uint32_t st = state[tile], sp = st & 1u;
        const uint32_t b = class_of(tile, st);
#pragma unroll
        for (uint32_t k = 0; k < kAllBuckets; k++) cnt[k] += (k == b) ? (1u + sp) : 0u;
    }
    // exclusive scan of every class over the threads, heaviest class first: inside the wave by shuffles, over the 16 waves
    // by one thread per class
    uint32_t pos[kAllBuckets];
#pragma unroll
    for (uint32_t b = 0; b < kAllBuckets; b++) {
        uint32_t incl = cnt[b];
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
            if (lane >= (uint32_t)off) incl += up;
        }
        pos[b] = incl - cnt[b]; // exclusive, inside the wave
        if (lane == 63u) wave_total[b][wv] = incl;
    }
    __syncthreads();
    if (tid < kAllBuckets) {
        uint32_t acc = 0u;
        for (uint32_t k = 0; k < 16u; k++) {
            const uint32_t t = wave_total[tid][k];
            wave_total[tid][k] = acc; // -> exclusive over the waves
            acc += t;
        }
        class_total[tid] = acc;
    }
    __syncthreads();
    {
        uint32_t base = 0u;
#pragma unroll
        for (int b = (int)kAllBuckets - 1; b >= 0; b--) {
            pos[b] += base + wave_total[b][wv];
            base += class_total[b];
        }
    }
    for (uint32_t j = lo; j < hi; j++) {
        const uint32_t tile = n - 1u - j;
        const uint32_t st = state[tile], sp = st & 1u;
        const uint32_t b = class_of(tile, st);
        uint32_t at = 0u;
#pragma unroll
        for (uint32_t k = 0; k < kAllBuckets; k++) {
            if (k == b) {
                at = pos[k];
                pos[k] += 1u + sp;
            }
        }
        if (sp) {
            order[(at & 7u) * row + (at >> 3)] = tile | 0x80000000u;
            order[((at + 1u) & 7u) * row + ((at + 1u) >> 3)] = tile | 0xC0000000u;
        } else {
            order[(at & 7u) * row + (at >> 3)] = tile;
        }
    }
    for (uint32_t at = n + nsplit + tid; at < 8u * row; at += 1024u) order[(at & 7u) * row + (at >> 3)] = 0xFFFFFFFFu; // idle workgroups
}

// Root-side un-swizzle of gathered shards (rank-major, tile-major) into a
// row-major frame.  One thread per pixel.
template <typename PIX>
__global__ __launch_bounds__(256) void vrt_assemble_kernel(const PIX *__restrict__ gathered, PIX *__restrict__ frame, uint32_t width,
                                                           uint32_t height, uint32_t tiles_x, uint32_t shard_count,
                                                           uint32_t tiles_per_rank /* tiles between the shards of consecutive ranks */,
                                                           const TileOwnership own, uint32_t frame_src_stride /* pixels between the shards of consecutive frames */) {
    // blockIdx.z: frame of a batch (shards of consecutive frames frame_src_stride pixels apart, frames width*height apart)
    gathered += (size_t)blockIdx.z * frame_src_stride;
    frame += (size_t)blockIdx.z * width * height;
    const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t y = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (x >= width || y >= height) return;
    const uint32_t t = (y / kTileH) * tiles_x + (x / kTileW);
    uint32_t r, i;
    if (own.period) {
        const uint32_t q = t / own.period, j = t % own.period;
        r = own.owner[j];
        i = q * own.count[r] + own.prefix[j];
    } else {
        r = t % shard_count;
        i = t / shard_count;
    }
    const size_t src = ((size_t)r * tiles_per_rank + i) * (kTileW * kTileH) + (y % kTileH) * kTileW + (x % kTileW);
    frame[(size_t)y * width + x] = gathered[src];
}

// The same from RGB shards (3 bytes per pixel, see TraceParams::packed_rgb) into the RGBA8 frame (alpha 255).
__global__ __launch_bounds__(256) void vrt_assemble_rgb_kernel(const uint8_t *__restrict__ gathered, uint32_t *__restrict__ frame, uint32_t width,
                                                               uint32_t height, uint32_t tiles_x, uint32_t shard_count, uint32_t tiles_per_rank,
                                                               const TileOwnership own, uint32_t frame_src_stride_bytes) {
    gathered += (size_t)blockIdx.z * frame_src_stride_bytes;
    frame += (size_t)blockIdx.z * width * height;
    const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t y = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (x >= width || y >= height) return;
    const uint32_t t = (y / kTileH) * tiles_x + (x / kTileW);
    uint32_t r, i;
    if (own.period) {
        const uint32_t q = t / own.period, j = t % own.period;
        r = own.owner[j];
        i = q * own.count[r] + own.prefix[j];
    } else {
        r = t % shard_count;
        i = t / shard_count;
    }
    const uint8_t *src = gathered + (((size_t)r * tiles_per_rank + i) * (kTileW * kTileH) + (y % kTileH) * kTileW + (x % kTileW)) * 3u;
    frame[(size_t)y * width + x] = (uint32_t)src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16) | (255u << 24);
}

// Four pixels per thread (width % 4 == 0): three dwords in, one 16-byte store out.
__global__ __launch_bounds__(256) void vrt_assemble_rgb4_kernel(const uint8_t *__restrict__ gathered, uint32_t *__restrict__ frame, uint32_t width,
                                                                uint32_t height, uint32_t tiles_x, uint32_t shard_count, uint32_t tiles_per_rank,
                                                                const TileOwnership own, uint32_t frame_src_stride_bytes) {
    gathered += (size_t)blockIdx.z * frame_src_stride_bytes;
    frame += (size_t)blockIdx.z * width * height;
    const uint32_t x = (blockIdx.x * 64u + (threadIdx.x & 63u)) * 4u;
    const uint32_t y = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (x >= width || y >= height) return;
    const uint32_t t = (y / kTileH) * tiles_x + (x / kTileW);
    uint32_t r, i;
    if (own.period) {
        const uint32_t q = t / own.period, j = t % own.period;
        r = own.owner[j];
        i = q * own.count[r] + own.prefix[j];
    } else {
        r = t % shard_count;
        i = t / shard_count;
    }
    const uint32_t *src = reinterpret_cast<const uint32_t *>(gathered + (((size_t)r * tiles_per_rank + i) * (kTileW * kTileH) + (y % kTileH) * kTileW + (x % kTileW)) * 3u);
    const uint32_t a = src[0], b = src[1], c = src[2]; // r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
    const uint4 out = make_uint4((a & 0xFFFFFFu) | 0xFF000000u, (a >> 24) | ((b & 0xFFFFu) << 8) | 0xFF000000u,
                                 (b >> 16) | ((c & 0xFFu) << 16) | 0xFF000000u, (c >> 8) | 0xFF000000u);
    *reinterpret_cast<uint4 *>(frame + (size_t)y * width + x) = out;
}

// ---- kernel selection and launchers (called from vrt_api.hip) -----------------------------------------------------------
static const KernelTable *all_tables(int *n) {
    static const KernelTable tables[4] = {inst_trace_b4(), inst_trace_b8(), inst_trace_count(), inst_path()};
    *n = 4;
    return tables;
}
template <typename Pred>
static const KernelEntry *find_entry(Pred pred) {
    int nt = 0;
    const KernelTable *t = all_tables(&nt);
    for (int i = 0; i < nt; i++)
        for (int k = 0; k < t[i].count; k++)
            if (pred(t[i].entries[k])) return &t[i].entries[k];
    return nullptr;
}
const KernelEntry *find_trace_kernel(int b, bool count, int mode, int min_waves, int shade, int block) {
    return find_entry([&](const KernelEntry &e) {
        return !e.path && e.b == b && (e.count != 0) == count && e.mode == mode && e.min_waves == min_waves && e.shade == shade && e.block == block;
    });
}
const KernelEntry *find_path_kernel(int b, int min_waves, bool filter, bool half, bool ahead, bool dist, int dil) {
    return find_entry([&](const KernelEntry &e) {
        return e.path == 1 && e.b == b && e.min_waves == min_waves && (e.filter != 0) == filter && (e.half != 0) == half && (e.ahead != 0) == ahead &&
               (e.dist != 0) == dist && (int)e.dil == dil;
    });
}
const KernelEntry *find_pool_kernel(int b, int min_waves, int slots, int stages) {
    return find_entry([&](const KernelEntry &e) {
        return e.path == 2 && e.b == b && (!min_waves || e.min_waves == min_waves) && (!slots || e.pool_slots == slots) && (!stages || e.pool_stages == stages);
    });
}
const KernelEntry *kernel_entry_of(KernelFn fn) {
    return find_entry([&](const KernelEntry &e) { return e.fn == fn; });
}
int compiled_kernel_count() {
    int nt = 0, n = 0;
    const KernelTable *t = all_tables(&nt);
    for (int i = 0; i < nt; i++) n += t[i].count;
    return n;
}

constexpr int kDefaultMinWaves = 4; // waves per SIMD the register allocator must leave room for

uint32_t resolve_variant(uint32_t variant) {
    return (variant & 0xFFu) == kVariantDefault ? ((variant & ~0xFFu) | (uint32_t)kVariantLinearAlways) : variant;
}

// kernel_variant = mode | (min_waves << 8) | flags.  shade: 0 general, 1 max_bounce <= 1 (no scatter evaluation), 2 the same with
// samples_per_pixel == 1.  Returns nullptr when this build of the library does not hold the kernel asked for (the product
// build compiles the variants the library itself chooses; the others live in the development build, make dev).
KernelFn select_trace_kernel(int brick_dimension, bool counters, uint32_t variant, int shade) {
    variant = resolve_variant(variant);
    const uint32_t vmode = variant & 0xFFu, mw_asked = (variant >> 8) & 0xFFu;
    if (mw_asked != 0u && (mw_asked < 4u || mw_asked > 8u)) return nullptr;
    // (min_waves 5 is vrt_path_kernel's own occupancy: every other kernel reads it as "the library's default")
    const bool to_path = shade == 0 && !counters && vmode == kVariantLinearAlways && !(variant & kVariantLockstepBounce);
    const uint32_t mw = (mw_asked == 5u && !to_path) ? 0u : mw_asked;
    int mode, block = 256;
    switch (vmode) {
        case kVariantLiteral: mode = kStatusLinear; break;
        case kVariantBlocked: mode = kStatusBlocked; break;
        case kVariantBlockedLds: mode = kStatusBlockedLds; break;
        case kVariantLinearWide: mode = kStatusLinearWide; break;
        case kVariantLinearAlways: mode = kStatusLinearAlways; break;
        case kVariantLinearLds: mode = kStatusLinearLds; break;
        case kVariantLinearLds512: mode = kStatusLinearLds; block = 512; break;
        case kVariantLinearAhead: mode = kStatusLinearAhead; break;
        // (byte status: the hand-written loop of frames without bounces; counting builds and the bounce kernels keep the words)
        case kVariantBytes: mode = (counters || shade == 0) ? kStatusLinearAlways : kStatusBytes; break;
        default: return nullptr;
    }
    const KernelEntry *e = nullptr;
    if (shade == 2 && !counters) {
        // the one-sample, no-bounce kernel (the headline's) is held to 72 VGPRs = 7 waves per SIMD (4 registers spilled outside the
        // loops): on frames that keep the GPU full it is 1.5-2 % faster than at its natural 75 VGPRs = 6 waves, on a tail-bound
        // frame (V1x) 4 % slower.  min_waves 4 asks for the natural build, 8 for the 64-VGPR one (development build).
        e = find_trace_kernel(brick_dimension, false, mode, (mw == 0u || mw == 7u) ? 7 : (mw == 8u ? 8 : kDefaultMinWaves), 2, block);
    } else if (shade == 0 && !counters) {
        // frames with bounces: persistent lanes (vrt_path_kernel) unless the lockstep form is asked for (bit 21).
        // (The lockstep bounce kernel takes 114 VGPRs = 4 waves per SIMD.  Its incoherent secondary rays wait on memory, and on a
        // scene that does not stay in the caches more waves pay for the spills: 4K / 2048^3 sparse path trace, ms per frame V1 / V0
        // at 4, 5, 6, 8 waves per SIMD: 83.7 / 283, 71.6 / 238, 66.1 / 219, 60.6 / 201.  On the 512^3 terrain the 8-wave build is
        // 16 % SLOWER than the 4-wave one, so both exist and vrt_create asks for the 8-wave one by the size of bindings 3-5.)
        if (mode == kStatusLinearAlways && !(variant & kVariantLockstepBounce)) {
            // bit 22: behind the LDS block filter (development build); min_waves 5: 96 VGPRs
            const bool filter = (variant & kVariantPathFilter) != 0u;
            e = find_path_kernel(brick_dimension, mw == 5u ? 5 : ((mw >= 6u && !filter) ? (int)mw : kDefaultMinWaves), filter, false);
        } else {
            // (round 6: with the per-lane set-up values formed again from the lane index the kernel fits 96 VGPRs = FIVE waves per SIMD with
            // five registers spilled — 33 until then —: the reference app's run 11.6 -> 12.5 Grays/s; 4 asks for the 4-wave build (development))
            e = find_trace_kernel(brick_dimension, false, mode, (mw == 8u || mw == 6u) ? (int)mw : (mw_asked == 4u ? 4 : 5), 0, block);
            if (!e) e = find_trace_kernel(brick_dimension, false, mode, mw_asked == 4u ? 5 : kDefaultMinWaves, 0, block);
        }
    } else {
        // (the several-samples-per-pixel kernel, shade 1, needs 87 VGPRs left to itself = 5 waves per SIMD; held to 80 it spilled
        // 36 bytes outside the loops and ran at 6: 4K / 1024^3 / 4 rays per pixel 1.215 -> 1.183 ms per frame.  Round 6: its per-lane
        // set-up values are formed again from the lane index where they are needed and one hoisted uniform product is kept scalar: 72
        // VGPRs without a spill = SEVEN waves per SIMD, 5 % faster than six on every view of that workload)
        int mw1 = 7;
#ifdef VRT_DEV_VARIANTS
        if (const char *ev = std::getenv("VRT_DEV_SHADE1_WAVES")) mw1 = std::atoi(ev); // (tuning builds: 5 and 7 for 8^3 bricks)
#endif
        e = find_trace_kernel(brick_dimension, counters, mode, (shade == 1 && !counters) ? mw1 : kDefaultMinWaves, shade, block);
        if (!e && shade == 1 && !counters) e = find_trace_kernel(brick_dimension, counters, mode, 6, shade, block);
    }
    return e ? e->fn : nullptr;
}

// bytes of dynamic LDS the variant needs for this grid
size_t trace_lds_bytes(const TraceParams &p, uint32_t variant) {
    const uint32_t mode = resolve_variant(variant) & 0xFFu;
    if (mode == kVariantBlockedLds) {
        const size_t nwords = ((size_t)p.nbx * p.nby * p.nbz + 31u) >> 5;
        return (nwords * 4u + 15u) & ~(size_t)15u;
    }
    if (mode == kVariantLinearLds || mode == kVariantLinearLds512) {
        size_t bytes = 16; // power of two: the hand-written loop masks byte addresses into the allocation
        while (bytes < (size_t)p.status_words * 4u) bytes <<= 1;
        return bytes;
    }
    return 0;
}

bool is_path_kernel(KernelFn fn) {
    const KernelEntry *e = kernel_entry_of(fn);
    return e && e->path;
}
// the same kernel with the walk loop on half-block words (TraceParams::status_halfblocks); fn itself if it has none
KernelFn path_kernel_halfblock_twin(KernelFn fn) {
    const KernelEntry *e = kernel_entry_of(fn);
    if (!e || e->path != 1 || e->filter || e->half || e->ahead || e->dist || e->dil) return fn;
    const KernelEntry *t = find_path_kernel(e->b, e->min_waves, false, true);
    return t ? t->fn : fn;
}
// the same kernel with the half-block walk loop on a dilated cell index; fn itself if it has none
KernelFn path_kernel_dilated_twin(KernelFn fn, int kind) {
    const KernelEntry *e = kernel_entry_of(fn);
    if (!e || e->path != 1 || e->filter || e->half || e->ahead || e->dist) return fn;
    const KernelEntry *t = find_path_kernel(e->b, e->min_waves, false, false, false, false, kind);
    return t ? t->fn : fn;
}
int path_kernel_dilated_kind(KernelFn fn) {
    const KernelEntry *e = kernel_entry_of(fn);
    return (e && e->path == 1) ? (int)e->dil : 0;
}
// the same kernel with the walk loop on the distance field (TraceParams::cell_distance); fn itself if it has none
KernelFn path_kernel_dist_twin(KernelFn fn) {
    const KernelEntry *e = kernel_entry_of(fn);
    if (!e || e->path != 1 || e->filter || e->half || e->ahead || e->dist || e->dil) return fn;
    const KernelEntry *t = find_path_kernel(e->b, e->min_waves, false, false, false, true);
    return t ? t->fn : fn;
}
// the plain path kernel's twin with the walk loop two trips ahead; fn itself if it has none
KernelFn path_kernel_ahead_twin(KernelFn fn) {
    const KernelEntry *e = kernel_entry_of(fn);
    if (!e || e->path != 1 || e->half || e->filter || e->ahead || e->dist || e->dil) return fn;
    const KernelEntry *t = find_path_kernel(e->b, e->min_waves, false, false, true);
    return t ? t->fn : fn;
}
bool is_path_halfblock_kernel(KernelFn fn) {
    const KernelEntry *e = kernel_entry_of(fn);
    return e && e->path == 1 && e->half;
}
static bool is_path_filter_kernel(KernelFn fn) {
    const KernelEntry *e = kernel_entry_of(fn);
    return e && e->path == 1 && e->filter;
}
const char *kernel_name_of(KernelFn fn) {
    const KernelEntry *e = kernel_entry_of(fn);
    return e ? e->name : "?";
}

hipError_t launch_trace(KernelFn fn, const TraceParams &p, size_t lds_bytes, hipStream_t stream, uint32_t frames) {
    if (p.owned_tiles == 0 || frames == 0) return hipSuccess;
    if (const KernelEntry *pe = kernel_entry_of(fn); pe && pe->path == 2) {
        // a pool of rays per wave (vrt_pool_kernel.h): as many workgroups as the GPU holds, or as the frame has pixels for
        if (!p.work_counter || !p.pool_paths || !p.pool_samples || frames != 1u) return hipErrorInvalidValue;
        hipError_t e = hipMemsetAsync(p.work_counter, 0, kMaxBatchFrames * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
        // (units of work: samples; a frame with fewer of them than the GPU holds paths launches fewer workgroups)
        const uint64_t units = (uint64_t)p.owned_tiles * 256u * (uint64_t)(p.pcs[0].cam.samples_per_pixel > 0 ? p.pcs[0].cam.samples_per_pixel : 1);
        const uint32_t per_group = 4u * (64u + pe->pool_slots), fit = (uint32_t)std::min<uint64_t>((units + per_group - 1u) / per_group, 1u << 30);
        const uint32_t hold = p.pool_cus * pe->min_waves; // (min_waves 256-thread workgroups per CU)
        const uint32_t groups = fit < hold ? (fit ? fit : 1u) : hold;
        VRT_LAUNCH(fn, dim3(groups, 1), dim3(256), pool_group_lds_bytes(pe->pool_slots, pe->pool_stages), stream, p);
        e = hipGetLastError();
        return e != hipSuccess ? e : launch_pool_resolve(p, stream);
    }
    if (is_path_kernel(fn)) {
        // persistent lanes: as many workgroups as the GPU holds a few times over; they take pixels from p.work_counter
        if (!p.work_counter) return hipErrorInvalidValue;
        hipError_t e = hipMemsetAsync(p.work_counter, 0, kMaxBatchFrames * sizeof(uint32_t), stream);
        if (e != hipSuccess) return e;
        const bool filter = is_path_filter_kernel(fn);
        const uint32_t threads = filter ? (uint32_t)kPathFilterThreads : 256u;
        const uint32_t want = filter ? (p.path_groups * 256u + threads - 1u) / threads : p.path_groups;
        const uint32_t groups = p.owned_tiles < want ? p.owned_tiles : want;
        // (samples as units of work — vrt_pool_resolve_kernel behind the kernel — for launches of one frame whose context holds the buffer)
        TraceParams q = p;
        if (frames != 1u) q.pool_samples = nullptr;
        VRT_LAUNCH(fn, dim3(groups, frames), dim3(threads), (filter ? p.path_lds_bytes : 0u) + (p.path_brick_lds ? (threads >> 6) * 4096u : 0u), stream, q);
        e = hipGetLastError();
        return (e != hipSuccess || !q.pool_samples) ? e : launch_pool_resolve(q, stream);
    }
    // The lockstep bounce kernel holds four waves per SIMD, sixteen per CU, and the four waves of a tile end at different times: a
    // 256-thread workgroup waits until four slots of one CU are free — a fifth of the slots stood empty through the body of the reference
    // app's frames (tools/experiments/timeline_tail.py).  As one-wave workgroups every wave that ends is replaced at once: the app's run V0 / V1 / V2 /
    // all-ground -2 / -5.5 / -5 / -9.5 %, same box, two frames in flight -9 %.  The several-samples kernel at six waves per SIMD (4K / 1024^3, two
    // samples): -2 % alone and with two frames in flight.  The one-sample kernels at seven waves per SIMD take it in reverse raster only
    // (two frames in flight, frames of more tiles than the schedule takes): headline -0.7 / -1.8 / -0.8 %, 256^3 -0.3 / -4.4 / -6.2 %
    // with two frames in flight; under the cost schedule, one frame at a time, the headline's V2 lost 11 % (tools/experiments/fif_waves_ab.py).
#ifdef VRT_DEV_PROFILE
    // (the profile build's phase counters are kept per workgroup, eight words each, in a buffer sized for one workgroup per tile)
    const bool profiled = p.wave_timeline != nullptr;
#else
    const bool profiled = false;
#endif
    if (const KernelEntry *te = kernel_entry_of(fn); te && te->path == 0 && (te->shade <= 1 || p.tile_order == 3u) && !te->count && p.wave_groups_bounce && !p.wave_groups &&
                                                       p.block_threads != 512u && !p.split_all && !p.packed_rgb && !profiled) {
        TraceParams q = p;
        q.wave_groups = 1u;
        VRT_LAUNCH(fn, dim3((q.owned_tiles + (q.tile_order == 5u ? q.sched_units : 0u)) * 4u, frames), dim3(64), lds_bytes, stream, q);
        return hipGetLastError();
    }
    // grid.y = the frames of this launch (p.pcs[0 .. frames-1]); workgroups are dispatched x-fastest, so the tiles of
    // frame 0 start first
    if (p.block_threads == 512u) VRT_LAUNCH(fn, dim3((p.owned_tiles + 1u) / 2u, frames), dim3(512), lds_bytes, stream, p);
    else if (p.wave_groups) VRT_LAUNCH(fn, dim3((p.owned_tiles + (p.tile_order == 5u ? p.sched_units : 0u)) * 4u, frames), dim3(64), lds_bytes, stream, p);
    else if (p.tile_order == 3u && p.split_all) VRT_LAUNCH(fn, dim3(p.owned_tiles << p.split_all, frames), dim3(256), lds_bytes, stream, p);
    else VRT_LAUNCH(fn, dim3(p.owned_tiles + (p.tile_order == 5u ? p.sched_units : 0u), frames), dim3(256), lds_bytes, stream, p);
    return hipGetLastError();
}

hipError_t launch_schedule(const uint32_t *cost, uint32_t *snap, const uint32_t *prev_order, uint32_t *order, uint32_t n, uint32_t extra_max,
                           uint32_t extra, uint32_t wave_slots, hipStream_t stream) {
    VRT_LAUNCH(vrt_schedule_kernel, dim3(1), dim3(1024), 0, stream, cost, snap, prev_order, order, n, extra_max, extra < extra_max ? extra : extra_max,
                       wave_slots);
    return hipGetLastError();
}

hipError_t launch_build_status_halfblocks(const TraceParams &p, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, hipStream_t stream) {
    if (!p.status_halfblocks) return hipSuccess;
    const uint32_t words = (dim_x >> 2) * (dim_z >> 2) * (dim_y >> 1);
    VRT_LAUNCH(vrt_build_status_halfblocks, dim3((words + 255u) / 256u), dim3(256), 0, stream, p.brick_status,
                       const_cast<uint32_t *>(p.status_halfblocks), dim_x, dim_y, dim_z);
    return hipGetLastError();
}

hipError_t launch_build_cell_distance(const TraceParams &p, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, hipStream_t stream) {
    if (!p.cell_distance) return hipSuccess;
#ifndef VRT_DEV_VARIANTS
    (void)dim_x, (void)dim_y, (void)dim_z, (void)stream;
    return hipErrorNotSupported; // (no kernel of the product build reads the field: vrt_create never allocates it there)
#else
    uint8_t *d = const_cast<uint8_t *>(p.cell_distance);
    VRT_LAUNCH(vrt_build_distance_seed, dim3((p.status_words + 255u) / 256u), dim3(256), 0, stream, p.brick_status, d, p.status_words, p.status_cells);
    const uint32_t lines[3] = {dim_z * dim_y, dim_x * dim_y, dim_x * dim_z};
    for (uint32_t axis = 0; axis < 3u; axis++)
        VRT_LAUNCH(vrt_build_distance_sweep, dim3((lines[axis] + 255u) / 256u), dim3(256), 0, stream, d, dim_x, dim_y, dim_z, axis);
    return hipGetLastError();
#endif
}

hipError_t launch_build_cell_occupancy(const TraceParams &p, uint32_t brick_dimension, uint64_t brick_alloc, uint64_t cell_lo, uint64_t cell_hi, uint64_t slot_lo,
                                       uint64_t slot_hi, hipStream_t stream) {
    if (!p.cell_occupancy) return hipSuccess;
    const uint32_t words8 = brick_dimension * brick_dimension * brick_dimension / 64u;
    const uint64_t cells = p.status_cells;
    cell_hi = cell_hi < cells ? cell_hi : cells;
    // written brick slots may belong to any cell: every cell is looked at (a read of its status bit and index), the named ones copied;
    // written cells only: those cells
    const bool any_slot = slot_lo < slot_hi;
    const uint64_t scan_lo = any_slot ? 0u : (cell_lo < cell_hi ? cell_lo : 0u), scan_hi = any_slot ? cells : (cell_lo < cell_hi ? cell_hi : 0u);
    if (scan_lo >= scan_hi) return hipSuccess;
    const uint64_t words = (scan_hi - scan_lo) * words8;
    VRT_LAUNCH(vrt_build_cell_occupancy, dim3((uint32_t)((words + 255u) / 256u)), dim3(256), 0, stream, p.brick_status, p.brick_index,
                       reinterpret_cast<const uint2 *>(p.brick_occupancy), reinterpret_cast<uint2 *>(const_cast<uint8_t *>(p.cell_occupancy)), scan_lo, scan_hi,
                       words8, brick_alloc, cell_lo, cell_hi, slot_lo, slot_hi);
    return hipGetLastError();
}

hipError_t launch_build_cell_material(const TraceParams &p, uint32_t brick_dimension, uint64_t brick_alloc, uint64_t cell_lo, uint64_t cell_hi, uint64_t slot_lo,
                                      uint64_t slot_hi, uint64_t mat_lo, uint64_t mat_hi, hipStream_t stream) {
    if (!p.cell_material) return hipSuccess;
    const uint32_t cells = p.status_cells;
    if (cells == 0u || (cell_lo >= cell_hi && slot_lo >= slot_hi && mat_lo >= mat_hi)) return hipSuccess;
    const uint64_t bits = (uint64_t)brick_dimension * brick_dimension * brick_dimension;
    const dim3 grid((cells + 255u) / 256u);
    uint8_t *out = const_cast<uint8_t *>(p.cell_material);
    if (brick_dimension == 8u)
        VRT_LAUNCH(vrt_build_cell_material<8>, grid, dim3(256), 0, stream, p.brick_status, p.brick_index, p.brick_occupancy, p.brick_start_index, p.material_index, out,
                           cells, p.status_words, brick_alloc, brick_alloc * bits, cell_lo, cell_hi, slot_lo, slot_hi, mat_lo, mat_hi);
    else
        VRT_LAUNCH(vrt_build_cell_material<4>, grid, dim3(256), 0, stream, p.brick_status, p.brick_index, p.brick_occupancy, p.brick_start_index, p.material_index, out,
                           cells, p.status_words, brick_alloc, brick_alloc * bits, cell_lo, cell_hi, slot_lo, slot_hi, mat_lo, mat_hi);
    return hipGetLastError();
}

hipError_t launch_check_start_is_slot(const TraceParams &p, uint32_t brick_dimension, uint64_t brick_alloc, hipStream_t stream) {
    if (!p.start_is_slot) return hipSuccess;
    uint32_t *flag = const_cast<uint32_t *>(p.start_is_slot);
    const hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(flag), 1, 1, stream);
    if (e != hipSuccess) return e;
    VRT_LAUNCH(vrt_check_start_is_slot, dim3((uint32_t)((brick_alloc + 255u) / 256u)), dim3(256), 0, stream, p.brick_start_index, flag, brick_alloc,
                       brick_dimension * brick_dimension * brick_dimension);
    return hipGetLastError();
}

hipError_t launch_check_materials_plain(const TraceParams &p, uint32_t count, hipStream_t stream) {
    if (!p.materials_plain) return hipSuccess;
    uint32_t *flag = const_cast<uint32_t *>(p.materials_plain);
    const hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(flag), 1, 1, stream);
    if (e != hipSuccess) return e;
    if (count) VRT_LAUNCH(vrt_check_materials_plain, dim3((count + 255u) / 256u), dim3(256), 0, stream, p.materials, flag, count);
    return hipGetLastError();
}

hipError_t launch_build_cell_bounds(const TraceParams &p, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, hipStream_t stream) {
    if (!p.cell_bounds) return hipSuccess;
    int *bounds = const_cast<int *>(p.cell_bounds);
    hipError_t e = hipMemsetAsync(bounds, 0x80, 6 * sizeof(int), stream);
    if (e != hipSuccess) return e;
    VRT_LAUNCH(vrt_build_cell_bounds, dim3((p.status_words + 255u) / 256u), dim3(256), 0, stream, p.brick_status, bounds, p.status_words,
                       dim_x * dim_y * dim_z, dim_x, dim_z);
    return hipGetLastError();
}

hipError_t launch_build_status_bytes(const TraceParams &p, hipStream_t stream) {
    if (!p.status_bytes) return hipSuccess;
    VRT_LAUNCH(vrt_build_status_bytes, dim3((p.status_words + 255u) / 256u), dim3(256), 0, stream, p.brick_status,
                       const_cast<uint8_t *>(p.status_bytes), p.status_words, p.status_cells);
    return hipGetLastError();
}

hipError_t launch_build_status_blocks(const TraceParams &p, uint32_t dim_x, uint32_t dim_y, uint32_t dim_z, hipStream_t stream) {
    if (!p.status_blocks) return hipSuccess;
    const uint32_t nblocks = p.nbx * p.nby * p.nbz;
    VRT_LAUNCH(vrt_build_status_blocks, dim3((nblocks + 255u) / 256u), dim3(256), 0, stream, p.brick_status,
                       const_cast<uint2 *>(p.status_blocks), dim_x, dim_y, dim_z, p.nbx, p.nby, p.nbz);
    return hipGetLastError();
}

hipError_t launch_assemble_rgb(const void *gathered, void *frame, uint32_t width, uint32_t height, uint32_t tiles_x, uint32_t shard_count,
                               uint32_t tiles_per_rank, const TileOwnership &own, hipStream_t stream, uint32_t frames, uint32_t frame_src_stride_bytes) {
    if (width % 4u == 0u && (reinterpret_cast<uintptr_t>(frame) & 15u) == 0u && (reinterpret_cast<uintptr_t>(gathered) & 3u) == 0u &&
        frame_src_stride_bytes % 4u == 0u) {
        const dim3 grid4((width / 4u + 63u) / 64u, (height + 3u) / 4u, frames);
        VRT_LAUNCH(vrt_assemble_rgb4_kernel, grid4, dim3(256), 0, stream, (const uint8_t *)gathered, (uint32_t *)frame, width, height, tiles_x,
                           shard_count, tiles_per_rank, own, frame_src_stride_bytes);
        return hipGetLastError();
    }
    const dim3 grid((width + 63u) / 64u, (height + 3u) / 4u, frames);
    VRT_LAUNCH(vrt_assemble_rgb_kernel, grid, dim3(256), 0, stream, (const uint8_t *)gathered, (uint32_t *)frame, width, height, tiles_x,
                       shard_count, tiles_per_rank, own, frame_src_stride_bytes);
    return hipGetLastError();
}

hipError_t launch_assemble(const void *gathered, void *frame, uint32_t bytes_per_pixel, uint32_t width, uint32_t height,
                           uint32_t tiles_x, uint32_t shard_count, uint32_t tiles_per_rank, const TileOwnership &own, hipStream_t stream,
                           uint32_t frames, uint32_t frame_src_stride_pixels) {
    const dim3 grid((width + 63u) / 64u, (height + 3u) / 4u, frames);
    if (bytes_per_pixel == 4) {
        VRT_LAUNCH(vrt_assemble_kernel<uint32_t>, grid, dim3(256), 0, stream, (const uint32_t *)gathered, (uint32_t *)frame, width,
                           height, tiles_x, shard_count, tiles_per_rank, own, frame_src_stride_pixels);
    } else if (bytes_per_pixel == 16) {
        VRT_LAUNCH(vrt_assemble_kernel<float4>, grid, dim3(256), 0, stream, (const float4 *)gathered, (float4 *)frame, width, height,
                           tiles_x, shard_count, tiles_per_rank, own, frame_src_stride_pixels);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

} // namespace vrt
