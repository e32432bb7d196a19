// vrt_inst_count.hip — the counting builds of vrt_trace_kernel (vrt_config.enable_counters): compiler-generated loops with
// per-lane counters; they fill vrt_get_counters and never produce the frame that is read back (vrt_api.hip, do_dispatch).
#include "vrt_inst_common.h"

namespace vrt {
namespace {
const KernelEntry kEntries[] = {
#ifndef VRT_DEV_VARIANTS
    VRT_TRACE_ENTRY(4, true, 4, 4, 0, 256), VRT_TRACE_ENTRY(4, true, 4, 4, 1, 256), VRT_TRACE_ENTRY(4, true, 4, 4, 2, 256),
    VRT_TRACE_ENTRY(8, true, 4, 4, 0, 256), VRT_TRACE_ENTRY(8, true, 4, 4, 1, 256), VRT_TRACE_ENTRY(8, true, 4, 4, 2, 256),
#else
    VRT_TRACE_ALL_MODES(4, true, 4, 0), VRT_TRACE_ALL_MODES(4, true, 4, 1), VRT_TRACE_ALL_MODES(4, true, 4, 2),
    VRT_TRACE_ALL_MODES(8, true, 4, 0), VRT_TRACE_ALL_MODES(8, true, 4, 1), VRT_TRACE_ALL_MODES(8, true, 4, 2),
#endif
};
} // namespace
KernelTable inst_trace_count() { return KernelTable{kEntries, (int)(sizeof kEntries / sizeof kEntries[0])}; }
} // namespace vrt
