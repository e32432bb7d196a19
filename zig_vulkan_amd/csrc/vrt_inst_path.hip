// vrt_inst_path.hip — vrt_path_kernel (frames with bounces on scenes larger than the caches: persistent lanes).
#include "vrt_pool_kernel.h"
#include "vrt_inst_common.h"

namespace vrt {
namespace {
const KernelEntry kEntries[] = {
    // 5 waves per SIMD (96 VGPRs); HALF: the walk loop on half-block words (grids whose x / z dimensions are powers of two)
    VRT_PATH_ENTRY(4, 5, false, false), VRT_PATH_ENTRY(4, 5, false, true),
    VRT_PATH_ENTRY(8, 5, false, false), VRT_PATH_ENTRY(8, 5, false, true),
    // DIL (round 3): the half-block walk loop on a dilated cell index (grids whose three dimensions are powers of two); 2: without
    // the steps-left counters, the walk ends at the grid's face (scenes whose occupied cells reach the grid's faces)
    VRT_PATH_ENTRY_L(4, 5, false, false, false, false, 1), VRT_PATH_ENTRY_L(8, 5, false, false, false, false, 1),
    VRT_PATH_ENTRY_L(4, 5, false, false, false, false, 2), VRT_PATH_ENTRY_L(8, 5, false, false, false, false, 2),
    // round 4: a pool of rays per wave (vrt_pool_kernel.h): the counter-free dilated-index walk; the four waves of a workgroup share two
    // staging areas for 8^3 bricks (2048^3 path trace, same box: vrt_path_kernel 129.6 ms, a staging area per wave at four waves per SIMD
    // 117.5, two shared at five 112.4).  Round 5: SIX waves per SIMD — the kernel needs 80 VGPRs without a spill by now — with as many
    // ray records in LDS per wave as six workgroups per CU leave room for (LDS is granted by the KiB: 26 624 B each): 48 records of 23
    // dwords 84.8 -> 82.9 ms (56 records and one area: 106; 40 and three: 93; 44 and two: 85.0 — the pool's size is worth more than the
    // areas: 3.4 % per eight records at five waves per SIMD), then 54 records of 21 dwords (the exchange pairs lanes and slots by two
    // cross-lane permutes instead of a scratch row, the walk's code rides with the state): 80.9 ms; 56 would be a seventh KiB too many:
    // five workgroups, 87.8; then 60 records of 19 dwords (GridHit's slab distances are formed again by the brick round instead of
    // carried: + 0.8 % at equal records, 79.8 ms with 60, 79.0 with the phase rule re-tuned) (profiles/r05_pool_sweep.txt).  4^3 bricks (no staging): 64 records at six waves, 13.2 -> 12.0 ms on a 4K /
    // 1024^3 sparse path trace
    VRT_POOL_ENTRY(8, 6, 60, 2),
    VRT_POOL_ENTRY(4, 6, 64, 0),
#ifdef VRT_DEV_VARIANTS
    VRT_POOL_ENTRY(8, 4, 64, 4), VRT_POOL_ENTRY(8, 5, 64, 1), VRT_POOL_ENTRY(8, 6, 56, 1), VRT_POOL_ENTRY(8, 5, 40, 4),
    VRT_POOL_ENTRY(8, 5, 64, 2), VRT_POOL_ENTRY(4, 5, 64, 0), VRT_POOL_ENTRY(8, 6, 56, 2), VRT_POOL_ENTRY(8, 6, 58, 2), VRT_POOL_ENTRY(8, 6, 54, 2), VRT_POOL_ENTRY(8, 6, 40, 3), VRT_POOL_ENTRY(8, 6, 44, 2), VRT_POOL_ENTRY(8, 6, 32, 4), VRT_POOL_ENTRY(8, 7, 48, 2),
#endif
#ifdef VRT_DEV_VARIANTS
    // DIL 4 (round 3): the counter-free dilated-index walk with the DDA two cells ahead of the test (two requests in flight per lane;
    // whoever leaves the loop takes a step back, ~7 % of the trips are walked twice): 128.5 vs 129.8 ms from inside the 2048^3 field,
    // 41.3 vs 40.9 from outside, same box — the round trip of the request is a tenth of the trip, not the half it looked like
    VRT_PATH_ENTRY_L(4, 5, false, false, false, false, 4), VRT_PATH_ENTRY_L(8, 5, false, false, false, false, 4),
    // DIL 3 (round 3): the counter-free dilated-index walk on 4 x 4 x 4-cell words (64 bits): a fifth fewer requests (0.265 per lane-trip
    // against 0.333), register pairs and 64-bit shifts for them: 132.05 vs 129.90 ms on the 2048^3 path trace, same box
    VRT_PATH_ENTRY_L(4, 5, false, false, false, false, 3), VRT_PATH_ENTRY_L(8, 5, false, false, false, false, 3),
    // DIST (round 3): the walk loop on the L1 distance field of the occupied cells, a byte per cell (any grid dimensions): 19 % fewer
    // vector instructions per frame of the 2048^3 path trace than the half-block words, 4.5 x the L2 misses (16 MiB against 2 MiB):
    // 150.5 vs 147.1 ms
    VRT_PATH_ENTRY_D(4, 5, false, false, false, true), VRT_PATH_ENTRY_D(8, 5, false, false, false, true),
    // AHEAD: the walk loop pipelined two trips ahead, on the shader's linear status words (round 3; measured 12 % slower than the
    // one-trip-ahead loop on the same words, 176 vs 157 ms on the 2048^3 path trace: the walk is bound by the L1's rate of scattered
    // requests, not by the latency of the one word a lane has in flight — DESIGN.md §4)
    VRT_PATH_ENTRY_S(4, 5, false, false, true), VRT_PATH_ENTRY_S(8, 5, false, false, true),
    VRT_PATH_ENTRY(4, 4, false, false), VRT_PATH_ENTRY(4, 4, false, true), VRT_PATH_ENTRY(8, 4, false, false), VRT_PATH_ENTRY(8, 4, false, true),
    VRT_PATH_ENTRY(4, 6, false, false), VRT_PATH_ENTRY(8, 6, false, false), VRT_PATH_ENTRY(8, 6, false, true),
    VRT_PATH_ENTRY(8, 7, false, false), VRT_PATH_ENTRY(8, 7, false, true), VRT_PATH_ENTRY(8, 8, false, false), VRT_PATH_ENTRY(8, 8, false, true),
    // the dilated-index walks at other wave counts
    VRT_PATH_ENTRY_L(8, 4, false, false, false, false, 1), VRT_PATH_ENTRY_L(8, 4, false, false, false, false, 2),
    VRT_PATH_ENTRY_L(8, 6, false, false, false, false, 1), VRT_PATH_ENTRY_L(8, 6, false, false, false, false, 2),
    // the block-skipping walk behind the LDS block filter (measured slower: DESIGN.md §4)
    VRT_PATH_ENTRY(4, 4, true, false), VRT_PATH_ENTRY(8, 4, true, false), VRT_PATH_ENTRY(4, 5, true, false), VRT_PATH_ENTRY(8, 5, true, false),
#endif
};
} // namespace
hipError_t launch_pool_resolve(const TraceParams &p, hipStream_t stream) {
    VRT_LAUNCH(vrt_pool_resolve_kernel, dim3(p.owned_tiles, 1), dim3(256), 0, stream, p);
    return hipGetLastError();
}
KernelTable inst_path() { return KernelTable{kEntries, (int)(sizeof kEntries / sizeof kEntries[0])}; }
} // namespace vrt
