// vrt_inst_common.h — table-entry macros of the instantiation units.  The numeric MODE literal keeps the entry's name
// identical to the demangled symbol (rocprofv3's kernel name).
#pragma once
#include "vrt_trace_kernels.h"

#define VRT_TRACE_ENTRY(B, COUNT, MODE, MW, SHADE, BLOCK)                                                                                   \
    {(vrt::KernelFn)vrt::vrt_trace_kernel<B, COUNT, MODE, MW, SHADE, BLOCK>, "vrt_trace_kernel<" #B ", " #COUNT ", " #MODE ", " #MW ", " #SHADE ", " #BLOCK ">", \
     0, B, COUNT, MODE, MW, SHADE, BLOCK, 0, 0, 0, 0, 0}
#define VRT_PATH_ENTRY_L(B, MW, FILTER, HALF, AHEAD, DIST, DIL)                                                                                \
    {(vrt::KernelFn)vrt::vrt_path_kernel<B, MW, FILTER, HALF, AHEAD, DIST, DIL>,                                                              \
     "vrt_path_kernel<" #B ", " #MW ", " #FILTER ", " #HALF ", " #AHEAD ", " #DIST ", " #DIL ">", 1, B, 0, 4, MW, 0, (FILTER ? 512 : 256), FILTER, HALF, AHEAD, \
     DIST, DIL}
#define VRT_POOL_ENTRY(B, MW, SLOTS, STAGES) {(vrt::KernelFn)vrt::vrt_pool_kernel<B, MW, SLOTS, STAGES>, "vrt_pool_kernel<" #B ", " #MW ", " #SLOTS ", " #STAGES ">", 2, B, 0, 4, MW, 0, 256, 0, 0, 0, 0, 2, SLOTS, STAGES}
#define VRT_PATH_ENTRY_D(B, MW, FILTER, HALF, AHEAD, DIST) VRT_PATH_ENTRY_L(B, MW, FILTER, HALF, AHEAD, DIST, 0)
#define VRT_PATH_ENTRY_S(B, MW, FILTER, HALF, AHEAD) VRT_PATH_ENTRY_D(B, MW, FILTER, HALF, AHEAD, false)
#define VRT_PATH_ENTRY(B, MW, FILTER, HALF) VRT_PATH_ENTRY_S(B, MW, FILTER, HALF, false)
// development build: the eight status modes + the 512-thread LDS variant of one (COUNT, MIN_WAVES, SHADE) combination
#define VRT_TRACE_ALL_MODES(B, COUNT, MW, SHADE)                                                                                              \
    VRT_TRACE_ENTRY(B, COUNT, 0, MW, SHADE, 256), VRT_TRACE_ENTRY(B, COUNT, 1, MW, SHADE, 256), VRT_TRACE_ENTRY(B, COUNT, 2, MW, SHADE, 256),  \
        VRT_TRACE_ENTRY(B, COUNT, 3, MW, SHADE, 256), VRT_TRACE_ENTRY(B, COUNT, 4, MW, SHADE, 256), VRT_TRACE_ENTRY(B, COUNT, 5, MW, SHADE, 256), \
        VRT_TRACE_ENTRY(B, COUNT, 5, MW, SHADE, 512), VRT_TRACE_ENTRY(B, COUNT, 6, MW, SHADE, 256), VRT_TRACE_ENTRY(B, COUNT, 7, MW, SHADE, 256)

static_assert(vrt::kStatusLinearAlways == 4 && vrt::kStatusBytes == 7, "the tables below spell the modes as numbers");
