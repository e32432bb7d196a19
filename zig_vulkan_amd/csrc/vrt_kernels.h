// vrt_kernels.h — the table of compiled traversal kernels: what the instantiation units (vrt_inst_*.hip) export and what
// the selection code in vrt_trace.hip searches.
#pragma once
// Every kernel launch of the library: the launchers report `hipGetLastError()` behind the launch, and that call returns the calling THREAD's
// last error whichever runtime call left it — also one that another library tolerated on this thread just before (RCCL's host code runs on
// the rank's thread between two launches; the test-only RCCL stand-in's hipEventQuery on an event of a destroyed stream left
// "operation not permitted when stream is capturing" behind, and the next frame's launch reported it: round 6).  So the thread's stale
// error is read away first; what the launcher then reports is the launch's own.
#define VRT_LAUNCH(...)                   \
    do {                                  \
        (void)hipGetLastError();          \
        hipLaunchKernelGGL(__VA_ARGS__);  \
    } while (0)
#include "vrt_internal.h"

namespace vrt {

using KernelFn = void (*)(const TraceParams);

// How the brick-level walk of vrt_trace_kernel learns whether a grid cell is occupied (template parameter MODE).
// The product build compiles kStatusLinearAlways and kStatusBytes; the others lost their A/B measurement
// and exist only in the development build (-DVRT_DEV_VARIANTS).
enum StatusMode : int {
    kStatusLinear = 0,       // the shader's own words: bit i%32 of word i/32, cached per lane (comp:318-328)
    kStatusBlocked = 1,      // device-built 4x4x4 block words from global memory, cached per lane
    kStatusBlockedLds = 2,   // same, behind an LDS-resident 1-bit-per-block "non-empty" filter
    kStatusLinearWide = 3,   // linear words for status, 64-bit words for occupancy
    kStatusLinearAlways = 4, // the hand-written loops on the shader's words, one request per trip (counting builds: compiler loops)
    kStatusLinearLds = 5,    // the whole linear status bitmap staged in LDS per workgroup, read on every step
    kStatusLinearAhead = 6,  // as kStatusLinearAlways, software-pipelined in C++: the next cell's word is requested before the current cell is tested
    kStatusBytes = 7         // the hand-written loops on a byte-per-cell copy of the status bits (no shift, no bit-field extract per trip)
};

// One compiled kernel.  `name` is its template-id exactly as rocprofv3 / llvm-cxxfilt print it (vrt_kernel_name()).
struct KernelEntry {
    KernelFn fn;
    const char *name;
    uint8_t path;      // 0: vrt_trace_kernel<B, COUNT, MODE, MIN_WAVES, SHADE, BLOCK>   1: vrt_path_kernel<B, MIN_WAVES, FILTER, HALF, AHEAD, DIST, DIL>
                       // 2: vrt_pool_kernel<B, MIN_WAVES, SLOTS, STAGES> (persistent waves like 1: pixels from TraceParams::work_counter)
                       // 2: vrt_pool_kernel<B, MIN_WAVES> (persistent waves like 1: pixels from TraceParams::work_counter)
    uint8_t b;         // brick dimension
    uint8_t count;     // trace: counting build
    uint8_t mode;      // trace: StatusMode
    uint8_t min_waves; // waves per SIMD the register allocator leaves room for (__launch_bounds__)
    uint8_t shade;     // trace: 0 bounce loop, 1 max_bounce <= 1, 2 ... and one sample per pixel
    uint16_t block;    // trace: threads per workgroup (256; 512 for the two-tiles-per-LDS-copy development variant)
    uint8_t filter;    // path: block-skipping walk behind the LDS block filter (development)
    uint8_t half;      // path: walk loop on half-block words
    uint8_t ahead;     // path: the walk loop pipelined two trips ahead (on the shader's linear words)
    uint8_t dist;      // path: the walk loop on the L1 distance field of the occupied cells (TraceParams::cell_distance; development)
    uint8_t dil;       // path: the half-block walk loop on a dilated cell index (all three grid dimensions powers of two): 1 = the walk
                       // ends at the box of the occupied cells (steps-left counters), 2 = at the grid's face (no counters in the loop)
    uint8_t pool_slots = 0;  // vrt_pool_kernel: ray records in LDS per wave (the wave owns 64 + pool_slots paths)
    uint8_t pool_stages = 0; // vrt_pool_kernel: staging areas for bricks per workgroup (4: one per wave; fewer: shared)
};
struct KernelTable {
    const KernelEntry *entries;
    int count;
};
KernelTable inst_trace_b4();
KernelTable inst_trace_b8();
KernelTable inst_trace_count();
KernelTable inst_path();

const KernelEntry *find_trace_kernel(int b, bool count, int mode, int min_waves, int shade, int block = 256);
const KernelEntry *find_path_kernel(int b, int min_waves, bool filter, bool half, bool ahead = false, bool dist = false, int dil = 0);
const KernelEntry *find_pool_kernel(int b, int min_waves = 0, int slots = 0, int stages = 0); // (0: the first in the table = the library's choice)
const KernelEntry *kernel_entry_of(KernelFn fn);
// vrt_pool_resolve_kernel (vrt_pool_kernel.h) over the pixels of the owned tiles, behind a vrt_pool_kernel on the same stream
hipError_t launch_pool_resolve(const TraceParams &p, hipStream_t stream);
int compiled_kernel_count();

// vrt_pool_kernel (vrt_pool_kernel.h): per wave `slots` ray records of 18 dwords in LDS and a dword of state + walk code per slot (round 5:
// 23 -> 19 dwords per slot — the exchange pairs lanes and slots by two cross-lane permutes instead of a scratch row, the walk's code rides
// with the state — GridHit's slab distances are formed again where they are needed — which is what lets 60 records per wave fit six workgroups per CU);
// per workgroup 16 lock words and `stages` staging areas of 4 KiB for bricks; 64 + slots paths per wave, 16 dwords each in global
// memory (TraceParams::pool_paths, sized for 128 paths per wave)
constexpr uint32_t kPoolRecDwords = 18u;
constexpr uint32_t kPoolStageBytes = 4096u;
constexpr uint32_t pool_group_lds_bytes(uint32_t slots, uint32_t stages) { return 64u + stages * kPoolStageBytes + 4u * (kPoolRecDwords + 1u) * 4u * slots; }
constexpr uint32_t kPoolPaths = 128u;
constexpr uint32_t kPoolPathDwords = 16u;

constexpr int kPathFilterThreads = 512; // vrt_path_kernel<FILTER>: eight waves share one LDS copy of the block filter

} // namespace vrt
