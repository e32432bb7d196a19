// vrt_inst_b4.hip — vrt_trace_kernel for 4^3 bricks (product builds; the development variants with -DVRT_DEV_VARIANTS).
#include "vrt_inst_common.h"

namespace vrt {
namespace {
const KernelEntry kEntries[] = {
#ifndef VRT_DEV_VARIANTS
    // one sample, no bounces (the headline's kernel): the shader's words / the byte-per-cell copy, held to 7 waves per SIMD
    VRT_TRACE_ENTRY(4, false, 4, 7, 2, 256), VRT_TRACE_ENTRY(4, false, 7, 7, 2, 256),
    // several samples, no bounces: 7 waves per SIMD (round 6: 72 VGPRs without a spill; 6 until then)
    VRT_TRACE_ENTRY(4, false, 4, 7, 1, 256), VRT_TRACE_ENTRY(4, false, 7, 7, 1, 256),
    // frames with bounces, lockstep: 5 waves (scenes that stay in the caches; round 6: 96 VGPRs, 4 waves / 125 VGPRs until then) and 8
    // (the multi-GPU pipeline on large scenes)
    VRT_TRACE_ENTRY(4, false, 4, 5, 0, 256), VRT_TRACE_ENTRY(4, false, 4, 8, 0, 256),
#else
    VRT_TRACE_ALL_MODES(4, false, 7, 2), VRT_TRACE_ALL_MODES(4, false, 8, 2), VRT_TRACE_ALL_MODES(4, false, 4, 2),
    VRT_TRACE_ALL_MODES(4, false, 6, 1),
    VRT_TRACE_ALL_MODES(4, false, 4, 0), VRT_TRACE_ALL_MODES(4, false, 8, 0),
    VRT_TRACE_ENTRY(4, false, 4, 5, 0, 256), VRT_TRACE_ENTRY(4, false, 4, 6, 0, 256),   // (tuning builds of the lockstep bounce kernel)
    VRT_TRACE_ENTRY(4, false, 4, 7, 1, 256), VRT_TRACE_ENTRY(4, false, 7, 7, 1, 256),   // (the several-samples kernel at the product's 7 waves)
#endif
};
} // namespace
KernelTable inst_trace_b4() { return KernelTable{kEntries, (int)(sizeof kEntries / sizeof kEntries[0])}; }
} // namespace vrt
