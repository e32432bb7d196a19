// vrt_trace_kernels.h — the traversal kernels (device code, templates).  Included by the instantiation units
// vrt_inst_*.hip (one per brick dimension / kernel family, so that they compile in parallel) — nothing else includes it.
#pragma once
//
// Brickmap traversal for gfx950 (MI355X), wave64.  Replaces the dispatch of assets/shaders/brick_raytracer.comp
// (src/modules/voxel_rt/ComputePipeline.zig:550).  One lane = one pixel, one wave = an 8x8 pixel block (coherent rays), one
// 256-thread workgroup = a 16x16 tile.  Arithmetic follows vrt_math.h's contract operation by operation; what differs from
// the shader is only how memory is touched and how the lanes of a wave are kept busy:
//   * both DDA levels run one hand-scheduled loop (VRT_TRIP_T / VRT_WALK_ASM): 13 vector instructions per trip, the next
//     cell's status requested one trip ahead, lanes that stand on an occupied cell / solid voxel handed back to C++;
//   * the brick-level walk starts at the near face of the bounding box of the occupied cells (skip_to_box, exact) and ends at
//     its far face; grids up to 64^3 cells read a byte-per-cell copy of the status bits, larger ones the shader's words;
//   * frames with bounces on scenes larger than the caches: persistent lanes (vrt_path_kernel), bricks staged in LDS,
//     status bits read by 4 x 4 x 2-cell words;
//   * workgroups are handed tiles in an order that follows their measured cost (vrt_schedule_kernel), round-robin over XCDs.
// Variants that lost their A/B measurement are compiled only with -DVRT_DEV_VARIANTS (make dev): their loops live in
// vrt_trace_kernels_dev.h, included section by section at the places they were measured from.  The persistent-lane kernel of round 2
// is vrt_path_kernel.h, round 4's pool of rays per wave vrt_pool_kernel.h.
#include <hip/hip_runtime.h>
#include "vrt_internal.h"
#include "vrt_kernels.h"
#include "vrt_math.h"

namespace vrt {

constexpr uint32_t MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2, MAT_NONE = 3;

struct Ray {
    f3 origin, direction;
    float internal_reflection;
    uint32_t ignore_type_material;
};

struct Hit {
    f3 point, normal;
    float t;
    uint32_t index;
};

template <bool COUNT>
struct Cnt {
    uint32_t rays = 0, status_loads = 0, bricks_entered = 0, voxel_steps = 0, hits = 0, grid_steps = 0;
    uint32_t wave_grid_iters = 0, wave_brick_walks = 0, wave_voxel_iters = 0;
};
template <>
struct Cnt<false> {};

#define VRT_COUNT(field)   \
    if constexpr (COUNT) { \
        c.field++;         \
    }
// counts once per wave per execution: only the first active lane increments
#define VRT_COUNT_WAVE(field)                                                              \
    if constexpr (COUNT) {                                                                 \
        if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)__ballot(1)) - 1)) c.field++; \
    }

// comp:180-184
VRT_DI Ray create_ray(f3 origin, f3 direction) { return Ray{origin, normalize3(direction), 1.0f, MAT_NONE}; }
// comp:192-195
VRT_DI f3 ray_at(const Ray &r, float t) { return fma3(splat3(t), r.direction, r.origin); }
// comp:267
VRT_DI float safe_inverse(float x) { return (x == 0.0f) ? 1e12f : (1.0f / x); }

// Development-only phase profile (make EXTRA=-DVRT_DEV_PROFILE; tools/experiments/one_tile.py --profile): core-clock
// cycles per phase, summed over the waves of a workgroup in LDS and written to the wave-timeline buffer.
#ifdef VRT_DEV_PROFILE
__shared__ unsigned long long vrt_prof[8];
VRT_DI unsigned long long prof_now() { return __builtin_readcyclecounter(); }
VRT_DI void prof_add(int k, unsigned long long t0) {
    const unsigned long long dt = __builtin_readcyclecounter() - t0;
    const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
    if ((threadIdx.x & 63u) == (uint32_t)(__builtin_ctzll(act))) atomicAdd(&vrt_prof[k], dt);
}
#define VRT_PROF_BEGIN(t) const unsigned long long t = prof_now()
#define VRT_PROF_END(k, t) prof_add(k, t)
#else
#define VRT_PROF_BEGIN(t)
#define VRT_PROF_END(k, t)
#endif

// DDA walker state shared by the two levels.  Instead of the cell position the walker keeps, per
// axis, how many more steps it may take before it leaves the box through the face it is moving
// towards (rem = dim-1-pos for step +1, pos for step -1): stepping decrements one counter and "still
// inside" is min3(rem) >= 0 — two VALU ops and no lane-mask merging on the scalar unit, against
// three compares and their s_and/s_or chain on positions.  An axis with step 0 never moves; its
// counter holds the hang-guard budget instead (see the oracle's header: same definition).  The
// position, needed only when a brick is entered, is base - step*rem.
struct Walk {
    f3 side_dist;
    int rx, ry, rz;
    float t_value;
};

VRT_DI int steps_left(int step, int pos, int dim, int zero_budget) {
    // (pos may be any int when the start lies outside the box; wrap instead of overflowing)
    return step > 0 ? (int)((uint32_t)(dim - 1) - (uint32_t)pos) : (step < 0 ? pos : zero_budget);
}
VRT_DI int walk_base(int step, int pos, int dim) { return step > 0 ? dim - 1 : (step < 0 ? 0 : pos); }
// the same against a box [lo, hi] of cells (pos inside the grid): steps that can be taken before the far face of the box is
// crossed, negative when the position is already beyond it; an axis the ray does not move along gets the hang-guard budget
// while the position is inside the box's range and -1 outside (the ray never reaches the box)
VRT_DI int steps_left_box(int step, int pos, int lo, int hi, int zero_budget) {
    // (as selects: the compiler turns the nested form into two levels of divergent branches per axis)
    const int fwd = (int)((uint32_t)hi - (uint32_t)pos), back = (int)((uint32_t)pos - (uint32_t)lo);
    const int moving = step > 0 ? fwd : back;
    const int still = (fwd | back) >= 0 ? zero_budget : -1; // inside the box's range iff neither difference is negative
    return step != 0 ? moving : still;
}
VRT_DI int walk_base_box(int step, int pos, int lo, int hi) { return step > 0 ? hi : (step < 0 ? lo : pos); }
VRT_DI int min3i(int a, int b, int c) { return min(min(a, b), c); }

// comp:345-372 / comp:440-467: the branchy min-axis step as selects,
//   x<y ? (x<z ? X : Z) : (y<z ? Y : Z)
// also advancing a linear index (cell index x + dim_x*(z + dim_z*y), comp:318, or voxel index,
// comp:412) by the stride of the crossed axis instead of recomputing it with two 32-bit multiplies
// (v_mad_u64_u32 on gfx950, quarter rate); exact in modular u32 arithmetic.  `axis` records the face
// crossed; hit.normal (comp:350,356,364,370) is rebuilt from it only when a voxel is actually hit.
// DEFER_T: leave t_value unscaled (the crossed side distance itself); the brick-level walk only needs
// `t_value = side_dist * scale` (comp:347) when a brick is entered, so it multiplies there.
template <bool DEFER_T>
VRT_DI void dda_step(Walk &w, const f3 &ray_delta, float scale, int &axis, uint32_t &index, uint32_t stride_x, uint32_t stride_y,
                     uint32_t stride_z) {
    const bool x_lt_y = w.side_dist.x < w.side_dist.y;
    const bool x_lt_z = w.side_dist.x < w.side_dist.z;
    const bool y_lt_z = w.side_dist.y < w.side_dist.z;
    // lane masks combined with bitwise operators: one s_and / s_andn2 / s_or each on the scalar unit
    const bool ax = x_lt_y & x_lt_z;
    const bool ay = (!x_lt_y) & y_lt_z;
    const bool axy = ax | ay; // z is crossed when neither x nor y is
    const float sd = ax ? w.side_dist.x : (ay ? w.side_dist.y : w.side_dist.z);
    w.t_value = DEFER_T ? sd : sd * scale;
    const float nx = w.side_dist.x + ray_delta.x;
    const float ny = w.side_dist.y + ray_delta.y;
    const float nz = w.side_dist.z + ray_delta.z;
    w.side_dist.x = ax ? nx : w.side_dist.x;
    w.side_dist.y = ay ? ny : w.side_dist.y;
    w.side_dist.z = axy ? w.side_dist.z : nz;
    w.rx -= ax ? 1 : 0;
    w.ry -= ay ? 1 : 0;
    w.rz -= axy ? 0 : 1;
    index += ax ? stride_x : (ay ? stride_y : stride_z);
    axis = ax ? 0 : (ay ? 1 : 2);
}

// ---- the brick-level walk loop, hand-written for gfx950 -------------------------------------------------
// Measured model (tools/ubench/step_bench.hip): a SIMD issues about one instruction per cycle in total —
// vector, scalar and branch alike — and a wave on its own needs >= 4 cycles per instruction, more across a
// VALU -> SGPR -> SALU -> VALU hand-over.  So the loop is written for the smallest TOTAL instruction count with
// short dependency chains, and everything the compiler adds around an inline-asm step (copies of loop-carried
// lane masks, exit-flag merging on the scalar unit, s_nop padding at the asm boundary) is avoided by keeping the
// whole loop inside one asm block: 30 instructions per trip (18 VALU, 8 SALU, 1 VMEM, 1 waitcnt, 2 branches)
// against 41 for the compiler's loop around the same step.
//
// One trip = take the DDA step out of the current cell (dda_step<true> as selects; comp:345-372 semantics,
// `!(x<y)` is s_andn2 of the x<y mask so NaNs take the shader's branches), request the next cell's status word,
// and only then test the bit of the cell just left, whose word was requested one trip earlier and has had a
// whole step to arrive (s_waitcnt vmcnt(1)).  The steps-left counters are decremented with the crossed-
// axis lane mask as borrow-in; their borrow-OUT is the box-exit test (a counter at 0 is decremented exactly
// when the lane leaves through that face), so leaving lanes are dropped from EXEC with scalar work only.  The
// loop is unrolled (four trips per back edge) with the roles of two register sets alternating (A/B: crossed distance
// and crossed-axis masks of the last and of the previous step), so nothing is copied between trips.  It runs until some lane
// meets an occupied cell or every lane has left the grid; the caller walks the bricks (rare: about once per
// wave per ray) and calls again.  Status words come through a stride-4 buffer resource: a lane outside the grid
// carries an arbitrary index and reads 0 instead of faulting (tools/isa_probe.hip checks this and the
// carry-out-under-partial-EXEC behaviour on the hardware).
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

struct GridWalkRegs {
    unsigned long long alive;        // in/out: lanes still walking the grid
    unsigned long long occ;          // out: lanes whose cell BEFORE their last step is occupied (0: every lane has left)
    unsigned long long out_x, out_y; // in/out: crossed-x / crossed-y lanes of the last step taken
    unsigned long long in_x, in_y;   // out: the same for the step before it (the step INTO the tested cell)
    float t_out, t_in;               // crossed distance of the last step (in/out) and of the one before it (out)
    uint32_t stub;                   // out: 0 the call ended in its first trip, 1/2 in a later trip, 3 every lane left
};

// the next cell's index: the linear cell index advances by the crossed axis' stride ...
#define VRT_STEP_LINEAR(IDX, IDXN, MX, MY)                                \
    "v_cndmask_b32_e64 %[t0], %[stz], %[sty], %[" MY "]\n\t"              \
    "v_cndmask_b32_e64 %[t0], %[t0], %[stx], %[" MX "]\n\t"               \
    "v_add_u32_e32 %[" IDXN "], %[" IDX "], %[t0]\n\t"
// ... or, in the DILATED form (grid_walk_park_dilated_gfx950), the crossed axis' bit field is incremented: stx / sty / stz hold the
// complement of that field's mask (all ones for an axis the ray does not move along: nothing changes), the carry runs through the
// ones filled into the other fields, and the other fields are put back
#define VRT_STEP_DILATED(IDX, IDXN, MX, MY)                               \
    "v_cndmask_b32_e64 %[t0], %[stz], %[sty], %[" MY "]\n\t"              \
    "v_cndmask_b32_e64 %[t0], %[t0], %[stx], %[" MX "]\n\t"               \
    "v_or_b32_e32 %[t1], %[" IDX "], %[t0]\n\t"                           \
    "v_add_u32_e32 %[t1], 1, %[t1]\n\t"                                   \
    "v_bfi_b32 %[" IDXN "], %[t0], %[" IDX "], %[t1]\n\t"
// ... and, where the grid's face is the walk's end (VRT_EXIT_CARRY below), the carry out of bit 31 says that the crossed axis'
// field has overflowed: the carry of a full field runs through the ones of every field above it
#define VRT_STEP_DILATED_CARRY(IDX, IDXN, MX, MY)                         \
    "v_cndmask_b32_e64 %[t0], %[stz], %[sty], %[" MY "]\n\t"              \
    "v_cndmask_b32_e64 %[t0], %[t0], %[stx], %[" MX "]\n\t"               \
    "v_or_b32_e32 %[t1], %[" IDX "], %[t0]\n\t"                           \
    "v_add_co_u32_e64 %[t1], %[ex], 1, %[t1]\n\t"                         \
    "v_bfi_b32 %[" IDXN "], %[t0], %[" IDX "], %[t1]\n\t"
// How a lane learns that its step has left the walk's box (`ex` = those lanes at the end of the trip).  COUNTERS: the steps-left
// counters are decremented with the crossed-axis lane masks as borrow-in; their borrow-out is the test.  CARRY (dilated index,
// VRT_STEP_DILATED_CARRY): the step itself has produced it.
#define VRT_EXIT_COUNTERS_A(MX, MY, MXY)                                  \
    "v_subbrev_co_u32_e64 %[rx], %[ex], 0, %[rx], %[" MX "]\n\t"          \
    "v_subbrev_co_u32_e64 %[ry], %[by], 0, %[ry], %[" MY "]\n\t"          \
    "v_addc_co_u32_e64 %[rz], %[cz], -1, %[rz], %[" MXY "]\n\t"
#define VRT_EXIT_COUNTERS_B                                               \
    "s_or_b64 %[ex], %[ex], %[by]\n\t"                                    \
    "s_orn2_b64 %[ex], %[ex], %[cz]\n\t" /* z: carry-out 0 = borrow */
#define VRT_EXIT_CARRY_A(MX, MY, MXY)
#define VRT_EXIT_CARRY_B
#define VRT_TRIP_T(TS, MX, MY, MXY, IDX, IDXN, WORD, WORDN, LIMIT, LOAD, TEST, OUT) \
    VRT_TRIP_E(VRT_STEP_LINEAR, VRT_EXIT_COUNTERS, TS, MX, MY, MXY, IDX, IDXN, WORD, WORDN, LIMIT, LOAD, TEST, OUT)
// VRT_TRIP_E, the trip.  The crossed distance = the smallest side distance, and the crossed axis from it: the shader's
// x<y ? (x<z ? X : Z) : (y<z ? Y : Z) picks Z whenever z is minimal (ties included), else Y whenever y is, else X — one min3 and two
// equality tests instead of three compares and two selects.  (A walk never holds a NaN side distance: safeInverse keeps 1/dir finite
// and NaN rays fail the slab test.)  The step and the request for the next cell's word depend on the crossed-axis masks only, so
// they come BEFORE the side distance is advanced (round 3: the request leaves nine instructions earlier and the adds run while the
// word asked for a trip ago arrives; 2048^3 path trace 133.9 -> 131.0 ms, 4K / 1024^3 -1.5 %, headline -1 %, same box):
// LOAD##_A issues the request and leaves the trip's lanes in `cz`, LOAD##_B waits for the word asked for a trip ago and restores
// EXEC.  side_dist of the crossed axis += |1/dir| is ONE add under the axis' lane mask as EXEC instead of three adds and three
// selects (a SIMD issues one instruction per clock in total and a vector instruction takes two slots; the all-vector form of
// round 3, 5 S + 16 V per trip, measured the same: DESIGN.md 4 "issue slots").
#define VRT_TRIP_E(STEP, XKIND, TS, MX, MY, MXY, IDX, IDXN, WORD, WORDN, LIMIT, LOAD, TEST, OUT) \
    "v_min3_f32 %[" TS "], %[sdx], %[sdy], %[sdz]\n\t"                    \
    "v_cmp_eq_f32_e64 %[" MXY "], %[sdz], %[" TS "]\n\t"                  \
    "v_cmp_eq_f32_e64 %[" MY "], %[sdy], %[" TS "]\n\t"                   \
    "s_andn2_b64 %[" MY "], %[" MY "], %[" MXY "]\n\t"                    \
    "s_andn2_b64 %[" MXY "], exec, %[" MXY "]\n\t"                        \
    "s_andn2_b64 %[" MX "], %[" MXY "], %[" MY "]\n\t"                    \
    STEP(IDX, IDXN, MX, MY)                                               \
    LOAD##_A(IDX, IDXN, WORD, WORDN)                                      \
    "s_mov_b64 exec, %[" MX "]\n\t"                                       \
    "v_add_f32_e64 %[sdx], %[sdx], |%[ix]|\n\t"                           \
    "s_mov_b64 exec, %[" MY "]\n\t"                                       \
    "v_add_f32_e64 %[sdy], %[sdy], |%[iy]|\n\t"                           \
    "s_andn2_b64 exec, %[cz], %[" MXY "]\n\t"                             \
    "v_add_f32_e64 %[sdz], %[sdz], |%[iz]|\n\t"                           \
    LOAD##_B(IDX, IDXN, WORD, WORDN)                                      \
    TEST(WORD, IDX)                                                       \
    XKIND##_A(MX, MY, MXY)                                                \
    LIMIT(TS, MXY)                                                        \
    XKIND##_B                                                             \
    "s_andn2_b64 exec, exec, %[ex]\n\t"                                   \
    "s_cbranch_vccnz " OUT "\n\t"

#define VRT_TRIP(TS, MX, MY, MXY, IDX, IDXN, WORD, WORDN, LIMIT, LOAD, OUT) VRT_TRIP_T(TS, MX, MY, MXY, IDX, IDXN, WORD, WORDN, LIMIT, LOAD, VRT_TEST_BIT, OUT)
// the cell just left is occupied: bit (index % 32) of its word ...
#define VRT_TEST_BIT(WORD, IDX)                                           \
    "v_bfe_u32 %[t1], %[" WORD "], %[" IDX "], 1\n\t"                     \
    "v_cmp_ne_u32_e32 vcc, 0, %[t1]\n\t"
// ... or, with the status expanded to ONE BYTE PER CELL (TraceParams::status_bytes, derived from binding 3 on every status
// upload), the byte itself: no shift for the address, no bit-field extract for the test — 16 instead of 18 vector instructions
// per trip of a loop that is bound by instruction issue (a wave64 vector instruction takes two of its SIMD's issue slots)
#define VRT_TEST_BYTE(WORD, IDX) "v_cmp_ne_u32_e32 vcc, 0, %[" WORD "]\n\t"

#define VRT_LOAD_BYTE_A(IDX, IDXN, WORD, WORDN)                                      \
    "s_mov_b64 %[cz], exec\n\t"                                          \
    "buffer_load_ubyte %[" WORDN "], %[" IDXN "], %[rsrc], 0 offen\n\t"
#define VRT_LOAD_BYTE_B(IDX, IDXN, WORD, WORDN)                                      \
    "s_mov_b64 exec, %[cz]\n\t"                                          \
    "s_waitcnt vmcnt(1)\n\t"

// Where the bitmap words come from.  Global memory: a stride-4 buffer resource indexed by the word index (an
// index outside the buffer reads 0).  LDS (brick level, grids whose status bitmap fits): the bitmap staged at
// LDS address 0, byte address masked into the power-of-two allocation.  The vector memory pipeline takes
// one wave-wide scattered dword request per ~16 cycles per CU (tools/ubench/step_bench.hip: the trip runs at 63
// cycles per SIMD with the buffer load, 40-44 without); LDS serves the same request several times faster.
#define VRT_LOAD_BUFFER_A(IDX, IDXN, WORD, WORDN)                                    \
    "s_mov_b64 %[cz], exec\n\t"                                          \
    "v_lshrrev_b32_e32 %[t2], 5, %[" IDXN "]\n\t"                         \
    "buffer_load_dword %[" WORDN "], %[t2], %[rsrc], 0 idxen\n\t"
#define VRT_LOAD_BUFFER_B(IDX, IDXN, WORD, WORDN)                                    \
    "s_mov_b64 exec, %[cz]\n\t"                                          \
    "s_waitcnt vmcnt(1)\n\t"
#define VRT_WAIT_BUFFER "s_waitcnt vmcnt(0)\n\t"
#ifdef VRT_DEV_VARIANTS
#define VRT_DEV_SECTION 1
#include "vrt_trace_kernels_dev.h"
#undef VRT_DEV_SECTION
#endif
#define VRT_WAIT_LDS "s_waitcnt lgkmcnt(0)\n\t"

// voxel level only (comp:469): the lane also leaves when the crossed distance, scaled to world units, is not
// <= the distance left inside the grid box (NaN leaves, as `!(t <= max)` does)
#define VRT_NO_LIMIT(TS, MXY)
#define VRT_T_LIMIT(TS, MXY) /* (the x|y mask of this trip is dead by now: its register takes the compare) */ \
    "v_mul_f32_e32 %[t1], %[scale], %[" TS "]\n\t"                        \
    "v_cmp_nle_f32_e64 %[" MXY "], %[t1], %[tmax]\n\t"                    \
    "s_or_b64 %[ex], %[ex], %[" MXY "]\n\t"

// Register sets: an A trip leaves cell idxa (word worda), writes {tsa, mxa, mya} and produces idxb and the
// request for wordb; a B trip the other way round.  A call starts with an A trip.  On exit the B set holds the
// last step and the A set the one before it, idxa/worda the current cell and idxb the cell just left: a call
// that ends in an A trip swaps the sets on its way out.
#define VRT_WALK_ASM(LIMIT, LOAD, TEST, WAITALL)                                                                            \
    "s_mov_b64 %[save], exec\n\t"                                                                         \
    "s_mov_b64 exec, %[alive]\n\t"                                                                        \
    VRT_TRIP_T("tsa", "mxa", "mya", "mxya", "idxa", "idxb", "worda", "wordb", LIMIT, LOAD, TEST, "1f")                  \
    "0:\n\t"                                                                                              \
    VRT_TRIP_T("tsb", "mxb", "myb", "mxyb", "idxb", "idxa", "wordb", "worda", LIMIT, LOAD, TEST, "2f")                  \
    VRT_TRIP_T("tsa", "mxa", "mya", "mxya", "idxa", "idxb", "worda", "wordb", LIMIT, LOAD, TEST, "3f")                  \
    VRT_TRIP_T("tsb", "mxb", "myb", "mxyb", "idxb", "idxa", "wordb", "worda", LIMIT, LOAD, TEST, "2f")                  \
    VRT_TRIP_T("tsa", "mxa", "mya", "mxya", "idxa", "idxb", "worda", "wordb", LIMIT, LOAD, TEST, "3f")                  \
    "s_cbranch_execnz 0b\n\t"                                                                             \
    "s_mov_b32 %[stub], 3\n\t"                                                                            \
    "s_branch 4f\n\t"                                                                                     \
    "1:\n\t"                                                                                              \
    "s_mov_b32 %[stub], 0\n\t"                                                                            \
    "s_branch 4f\n\t"                                                                                     \
    "3:\n\t"                                                                                              \
    "s_mov_b32 %[stub], 2\n\t"                                                                            \
    "4:\n\t"                                                                                              \
    "s_mov_b64 %[occ], vcc\n\t"                                                                           \
    "s_mov_b64 %[alive], exec\n\t"                                                                        \
    "s_mov_b64 exec, %[save]\n\t"                                                                         \
    WAITALL                                                                                               \
    "s_mov_b64 %[ex], %[mxa]\n\t"                                                                         \
    "s_mov_b64 %[mxa], %[mxb]\n\t"                                                                        \
    "s_mov_b64 %[mxb], %[ex]\n\t"                                                                         \
    "s_mov_b64 %[ex], %[mya]\n\t"                                                                         \
    "s_mov_b64 %[mya], %[myb]\n\t"                                                                        \
    "s_mov_b64 %[myb], %[ex]\n\t"                                                                         \
    "v_mov_b32_e32 %[t0], %[tsa]\n\t"                                                                     \
    "v_mov_b32_e32 %[tsa], %[tsb]\n\t"                                                                    \
    "v_mov_b32_e32 %[tsb], %[t0]\n\t"                                                                     \
    "v_mov_b32_e32 %[t0], %[idxa]\n\t"                                                                    \
    "v_mov_b32_e32 %[idxa], %[idxb]\n\t"                                                                  \
    "v_mov_b32_e32 %[idxb], %[t0]\n\t"                                                                    \
    "v_mov_b32_e32 %[worda], %[wordb]\n\t"                                                                \
    "s_branch 6f\n\t"                                                                                     \
    "2:\n\t"                                                                                              \
    "s_mov_b32 %[stub], 1\n\t"                                                                            \
    "s_mov_b64 %[occ], vcc\n\t"                                                                           \
    "s_mov_b64 %[alive], exec\n\t"                                                                        \
    "s_mov_b64 exec, %[save]\n\t"                                                                         \
    WAITALL /* the compiler may move `word`: no load may be in flight outside */                          \
    "6:"

#define VRT_WALK_OUTPUTS                                                                                                                        \
    [sdx] "+v"(w.side_dist.x), [sdy] "+v"(w.side_dist.y), [sdz] "+v"(w.side_dist.z), [rx] "+v"(w.rx), [ry] "+v"(w.ry), [rz] "+v"(w.rz),          \
        [idxa] "+v"(index), [idxb] "=&v"(cell), [worda] "+v"(word), [wordb] "=&v"(wordb), [tsb] "+v"(g.t_out), [tsa] "=&v"(g.t_in),              \
        [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [mxb] "+s"(g.out_x), [myb] "+s"(g.out_y), [alive] "+s"(g.alive), [mxa] "=&s"(g.in_x),    \
        [mya] "=&s"(g.in_y), [mxya] "=&s"(mxya), [mxyb] "=&s"(mxyb), [ex] "=&s"(ex), [by] "=&s"(by), [cz] "=&s"(cz), [save] "=&s"(save),         \
        [occ] "=&s"(g.occ), [stub] "=&s"(g.stub)
#define VRT_WALK_INPUTS \
    [ix] "v"(inv_dir.x), [iy] "v"(inv_dir.y), [iz] "v"(inv_dir.z), [stx] "v"(stride_x), [sty] "v"(stride_y), [stz] "v"(stride_z), [rsrc] "s"(rsrc)

// brick level (comp:314-375): cells of the grid, bits of brick_status
VRT_DI void grid_walk_gfx950(Walk &w, const f3 &inv_dir, uint32_t &index, uint32_t &cell, uint32_t stride_x, uint32_t stride_y, uint32_t stride_z,
                             uint32_t &word, u32x4 rsrc, GridWalkRegs &g) {
    unsigned long long mxya, mxyb, ex, by, cz, save;
    float t0, t1, t2;
    uint32_t wordb;
    asm volatile(VRT_WALK_ASM(VRT_NO_LIMIT, VRT_LOAD_BUFFER, VRT_TEST_BIT, VRT_WAIT_BUFFER) : VRT_WALK_OUTPUTS : VRT_WALK_INPUTS : "vcc", "scc");
}

#ifdef VRT_DEV_VARIANTS
#define VRT_DEV_SECTION 2
#include "vrt_trace_kernels_dev.h"
#undef VRT_DEV_SECTION
#else
// (the status bitmap in LDS, a development variant: declared for the discarded branches of grid_hit)
VRT_DI void grid_walk_lds_gfx950(Walk &w, const f3 &inv_dir, uint32_t &index, uint32_t &cell, uint32_t stride_x, uint32_t stride_y, uint32_t stride_z,
                                 uint32_t &word, uint32_t rsrc, GridWalkRegs &g);
#endif
// brick level on the byte-per-cell copy of the status bits: `index` is the byte offset, `word` the byte of the current cell;
// rsrc: a raw buffer (stride 0) over TraceParams::status_bytes, so that an out-of-grid index reads 0
VRT_DI void grid_walk_bytes_gfx950(Walk &w, const f3 &inv_dir, uint32_t &index, uint32_t &cell, uint32_t stride_x, uint32_t stride_y, uint32_t stride_z,
                                   uint32_t &word, u32x4 rsrc, GridWalkRegs &g) {
    unsigned long long mxya, mxyb, ex, by, cz, save;
    float t0, t1, t2;
    uint32_t wordb;
    asm volatile(VRT_WALK_ASM(VRT_NO_LIMIT, VRT_LOAD_BYTE, VRT_TEST_BYTE, VRT_WAIT_BUFFER) : VRT_WALK_OUTPUTS : VRT_WALK_INPUTS : "vcc", "scc");
}

// voxel level (comp:409-470): voxels of one brick, bits of brick_occupancy addressed by the global bit index
// brick * B^3 + voxel; `scale` and `t_max` as in the loop condition comp:469
VRT_DI void voxel_walk_gfx950(Walk &w, const f3 &inv_dir, uint32_t &index, uint32_t &cell, uint32_t stride_x, uint32_t stride_y, uint32_t stride_z,
                              uint32_t &word, u32x4 rsrc, GridWalkRegs &g, float scale, float t_max) {
    unsigned long long mxya, mxyb, ex, by, cz, save;
    float t0, t1, t2;
    uint32_t wordb;
    asm volatile(VRT_WALK_ASM(VRT_T_LIMIT, VRT_LOAD_BUFFER, VRT_TEST_BIT, VRT_WAIT_BUFFER) : VRT_WALK_OUTPUTS : VRT_WALK_INPUTS, [scale] "s"(scale), [tmax] "v"(t_max) : "vcc", "scc");
}
#undef VRT_WALK_ASM
#undef VRT_WALK_OUTPUTS
#undef VRT_WALK_INPUTS

// ---- the same loop with parking, for frames with bounces ----------------------------------------------------
// Secondary rays are incoherent: the lanes of a wave meet their bricks at different trips, and walking each
// brick at once (the shader's order) runs the long voxel-level code for a few lanes at a time.  Here a lane
// whose trip left an occupied cell behind is PARKED: dropped from EXEC with, per lane, tsa = the distance into
// that cell, tsb = the distance of the step out of it, idxb = the cell, idxa = the cell it now stands on and the
// two crossed axes in `code` (an A trip swaps the two register sets for the parked lanes to get there).  The
// other lanes keep walking until `batch` lanes are parked or nobody is moving; then the call returns and the
// caller walks all parked bricks in ONE execution.  Per lane the sequence of operations is unchanged.  The
// in-axis of a lane's FIRST trip in a call comes from code bits 4-5 (its last step may be many trips old); later
// trips read it from the other set's crossed-axis masks.  On exit the sets are swapped for the still-moving
// lanes as well if the last trip was an A trip, so that set B / idxa / worda are "last step / current cell /
// its status word" for every lane.  (Measured on the 2048^3 path-trace config: 325 -> 220 ms per frame against
// the per-trip state machine this replaces; on primary + shadow frames the plain loop above is 3 % faster.)
struct GridParkRegs {
    unsigned long long alive;        // in: lanes to walk; out: lanes still moving when the call ended
    unsigned long long parked;       // out: lanes that left an occupied cell behind (0: every lane has left)
    unsigned long long out_x, out_y; // in/out: crossed-x / crossed-y lanes of the last trip (moving lanes: their last step)
    float t_out, t_in;               // per lane: crossed distance of the lane's last step (in/out) and of the step before it (out)
    // per lane.  in: bits 4-5 = axis crossed by the lane's last step before this call (3: none, the slab entry).
    // out, parked lanes: bits 0-1 axis INTO the occupied cell, bits 2-3 axis OUT of it
    uint32_t code;
    uint32_t batch;                  // in: the call returns once this many lanes are parked (or nobody is moving)
    uint32_t min_alive = 0;          // in: ... or, at a back edge, once fewer than this many lanes are still moving (0: never)
};

// state of the development walk loops a lane carries between calls (the loops themselves: vrt_trace_kernels_dev.h)
// ---- the park loop pipelined TWO trips ahead (vrt_path_kernel<AHEAD>) ---------------------------------------------------------
// The loops above keep ONE status word in flight per lane: the word of the next cell is requested a trip ahead, and on a scene
// larger than the caches a wave then waits ~670 cycles per trip for it (tools/path_profile.py: 13 400 cycles per call of ~20 trips,
// 27 vector instructions per trip, 5 waves per SIMD).  Here the DDA runs two cells ahead of the test: a lane carries a ring of three
// cells (q0, q1, q2) with their words (w0, w1, w2) and the crossed distances of the steps INTO them (ts0, ts1, ts2); a trip rotates
// the ring, takes the step out of q1 into the new q2, requests q2's word, waits until at most TWO requests are outstanding — i.e.
// for the word of q0, asked for two trips ago — and tests q0.  Per lane the sequence of DDA operations is the shader's (comp:345-372),
// run two steps early.  What that takes:
//   * a lane whose q0 is occupied PARKS as it is: ring and DDA state stay, the caller walks the brick of q0 (entered through the axis
//     in `hist` bits 4-5 at distance ts0; the counters of q0 = the current ones with the two later steps' decrements undone) and, if
//     the brick holds nothing for the ray, the lane simply walks on: its next trip rotates q1 into place.  No roll-back, no re-request;
//   * a step that leaves the box of the occupied cells puts the SENTINEL ~0 into q2, and so does every later step (q1 == ~0): the
//     word index of ~0 lies outside the buffer, its word reads 0, its test fails, and when the sentinel reaches q0 the lane has tested
//     every cell up to the box's face and leaves (`left`).  The cells a lane would "enter" beyond the face are thus never tested
//     (at a face of the grid the next linear index would be a real cell of the next row);
//   * `hist`: two bits per step, the axis of the step into q2 in bits 0-1, into q1 in bits 2-3, into q0 in bits 4-5 (3: q0 was entered
//     without a step — the slab test — at the start of a ray).
// Canonical state between calls, for every lane: [q0 tested, q1 next to be tested, q2 newest], the DDA state (side distances,
// counters) that of q2, w1 / w2 arrived.  A new ray is primed in C++ with one step (path kernel, START).
struct AheadWalkRegs {
    unsigned long long alive;  // in: lanes to walk; out: lanes still moving when the call ended
    unsigned long long parked; // out: lanes whose q0 is occupied
    unsigned long long left;   // out: lanes that have tested every cell up to the face of the box
    uint32_t batch;            // in: the call returns once this many lanes are parked (or nobody is moving)
    uint32_t min_alive;        // in: ... or, at the back edge, once fewer than this many lanes are still moving
};
// The words travel in THREE registers used in turn (a register with a request in flight cannot be moved): trip k of the unrolled
// body (k = 0, 1, 2) requests into W[k] and tests W[(k + 1) % 3].  A lane leaves the loop after some trip k — parked, or still
// moving when the call ends — and `phase` records that k; the caller then puts word(q1) into w1 and word(q2) into w2
// (AheadRing::settle), which is where trip 0 of the next call expects them.  The cells and distances are moved (computed values).
struct AheadRing {
    uint32_t q0, q1, q2, w0, w1, w2, hist, phase;
    float ts0, ts1, ts2;
    VRT_DI void settle() { // after a call: phase k -> word(q1) is in W[(k + 2) % 3], word(q2) in W[k]
        const uint32_t a = phase == 0u ? w2 : (phase == 1u ? w0 : w1), b = phase == 0u ? w0 : (phase == 1u ? w1 : w2);
        w1 = a;
        w2 = b;
        phase = 2u;
    }
};
struct DistRegs {
    unsigned long long pend; // in/out: lanes whose `word` is an answer for their current cell (the others know it to be empty)
    int k;                   // per lane, in/out: trips the lane may take before it asks again
};

#define VRT_PARK(LABEL, PRE, IN_AXIS, OUT_MX, OUT_MY, SWAP, NEXT, EXIT)                                  \
    LABEL ":\n\t"                                                                                         \
    PRE                                                                                                   \
    "s_mov_b64 %[ex], exec\n\t"                                                                           \
    "s_mov_b64 exec, vcc\n\t"                                                                             \
    IN_AXIS                                                                                               \
    "v_cndmask_b32_e64 %[t1], 2, 1, %[" OUT_MY "]\n\t"                                                    \
    "v_cndmask_b32_e64 %[t1], %[t1], 0, %[" OUT_MX "]\n\t"                                                \
    "v_lshl_or_b32 %[code], %[t1], 2, %[t0]\n\t"                                                          \
    SWAP                                                                                                  \
    "s_or_b64 %[parked], %[parked], vcc\n\t"                                                              \
    "s_andn2_b64 exec, %[ex], vcc\n\t"                                                                    \
    "s_bcnt1_i32_b64 %[n], %[parked]\n\t"                                                                 \
    "s_cmp_ge_u32 %[n], %[batch]\n\t"                                                                     \
    "s_cbranch_scc1 " EXIT "\n\t"                                                                         \
    "s_cbranch_execnz " NEXT "\n\t"                                                                       \
    "s_branch " EXIT "\n\t"
#define VRT_IN_FROM_CODE "v_bfe_u32 %[t0], %[code], 4, 2\n\t"
#define VRT_IN_FROM(MX, MY) "v_cndmask_b32_e64 %[t0], 2, 1, %[" MY "]\n\t" "v_cndmask_b32_e64 %[t0], %[t0], 0, %[" MX "]\n\t"
#define VRT_SWAP_SETS "v_swap_b32 %[tsa], %[tsb]\n\t" "v_swap_b32 %[idxa], %[idxb]\n\t"

#define VRT_PARK_WALK_ASM(LIMIT, LOAD, TEST, WAITALL, AT30) VRT_PARK_WALK_ASM_S(VRT_STEP_LINEAR, LIMIT, LOAD, TEST, WAITALL, AT30)
#define VRT_PARK_WALK_ASM_S(STEP, LIMIT, LOAD, TEST, WAITALL, AT30) VRT_PARK_WALK_ASM_T(VRT_TRIP_E, STEP, VRT_EXIT_COUNTERS, "", LIMIT, LOAD, TEST, WAITALL, AT30)
#define VRT_WORD_MOV32 "v_mov_b32_e32 %[worda], %[wordb]\n\t"
#define VRT_PARK_WALK_ASM_T(TRIP, STEP, XKIND, PARKPRE, LIMIT, LOAD, TEST, WAITALL, AT30) \
    VRT_PARK_WALK_ASM_W(TRIP, STEP, XKIND, PARKPRE, LIMIT, LOAD, TEST, WAITALL, VRT_WORD_MOV32 AT30)
// PARKPRE: what a trip's park code does first (VRT_EXIT_CARRY: note the parked lanes whose step out of their cell left the grid)
// (AT30: how worda takes wordb's content when a call ends in an A trip, and whatever else the variant has to swap there)
#define VRT_PARK_WALK_ASM_W(TRIP, STEP, XKIND, PARKPRE, LIMIT, LOAD, TEST, WAITALL, AT30) \
        "s_mov_b64 %[save], exec\n\t" \
        "s_mov_b64 exec, %[alive]\n\t" \
        "s_mov_b64 %[parked], 0\n\t" \
        TRIP(STEP, XKIND, "tsa", "mxa", "mya", "mxya", "idxa", "idxb", "worda", "wordb", LIMIT, LOAD, TEST, "10f") \
        "0:\n\t" \
        TRIP(STEP, XKIND, "tsb", "mxb", "myb", "mxyb", "idxb", "idxa", "wordb", "worda", LIMIT, LOAD, TEST, "11f") \
        "21:\n\t" \
        TRIP(STEP, XKIND, "tsa", "mxa", "mya", "mxya", "idxa", "idxb", "worda", "wordb", LIMIT, LOAD, TEST, "12f") \
        "22:\n\t" \
        TRIP(STEP, XKIND, "tsb", "mxb", "myb", "mxyb", "idxb", "idxa", "wordb", "worda", LIMIT, LOAD, TEST, "13f") \
        "23:\n\t" \
        TRIP(STEP, XKIND, "tsa", "mxa", "mya", "mxya", "idxa", "idxb", "worda", "wordb", LIMIT, LOAD, TEST, "14f") \
        "24:\n\t" \
        "s_cbranch_execz 31f\n\t" \
        /* back edge (after an A trip): keep walking while at least min_alive lanes are moving; fewer -> hand the wave back \
           so that the finished lanes can be given new rays (vrt_path_kernel; min_alive = 0: never) */ \
        "s_bcnt1_i32_b64 %[n], exec\n\t" \
        "s_cmp_ge_u32 %[n], %[minalive]\n\t" \
        "s_cbranch_scc1 0b\n\t" \
        "s_branch 30f\n\t" \
        VRT_PARK("10", PARKPRE, VRT_IN_FROM_CODE, "mxa", "mya", VRT_SWAP_SETS, "0b", "30f") \
        VRT_PARK("11", PARKPRE, VRT_IN_FROM("mxa", "mya"), "mxb", "myb", "", "21b", "31f") \
        VRT_PARK("12", PARKPRE, VRT_IN_FROM("mxb", "myb"), "mxa", "mya", VRT_SWAP_SETS, "22b", "30f") \
        VRT_PARK("13", PARKPRE, VRT_IN_FROM("mxa", "mya"), "mxb", "myb", "", "23b", "31f") \
        VRT_PARK("14", PARKPRE, VRT_IN_FROM("mxb", "myb"), "mxa", "mya", VRT_SWAP_SETS, "24b", "30f") \
        "30:\n\t" /* the last trip was an A trip: swap the sets of the lanes still moving */ \
        "s_mov_b64 %[alive], exec\n\t" \
        WAITALL \
        VRT_SWAP_SETS \
        AT30 \
        "s_mov_b64 %[mxb], %[mxa]\n\t" \
        "s_mov_b64 %[myb], %[mya]\n\t" \
        "s_branch 32f\n\t" \
        "31:\n\t" \
        "s_mov_b64 %[alive], exec\n\t" \
        WAITALL /* the compiler may move `word`: no load may be in flight outside */ \
        "32:\n\t" \
        "s_mov_b64 exec, %[save]"
#define VRT_PARK_WALK_OPERANDS                                                                                                                   \
    [sdx] "+v"(w.side_dist.x), [sdy] "+v"(w.side_dist.y), [sdz] "+v"(w.side_dist.z), [rx] "+v"(w.rx), [ry] "+v"(w.ry), [rz] "+v"(w.rz),          \
        [idxa] "+v"(index), [idxb] "=&v"(cell), [worda] "+v"(word), [wordb] "=&v"(wordb), [tsb] "+v"(g.t_out), [tsa] "=&v"(g.t_in),              \
        [code] "+v"(g.code), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [mxb] "+s"(g.out_x), [myb] "+s"(g.out_y), [alive] "+s"(g.alive),    \
        [mxa] "=&s"(mxa), [mya] "=&s"(mya), [mxya] "=&s"(mxya), [mxyb] "=&s"(mxyb), [ex] "=&s"(ex), [by] "=&s"(by), [cz] "=&s"(cz),              \
        [save] "=&s"(save), [parked] "=&s"(g.parked), [n] "=&s"(n)
#define VRT_PARK_WALK_INPUTS                                                                                                                          \
    [ix] "v"(inv_dir.x), [iy] "v"(inv_dir.y), [iz] "v"(inv_dir.z), [stx] "v"(stride_x), [sty] "v"(stride_y), [stz] "v"(stride_z), [rsrc] "s"(rsrc), \
        [batch] "s"(g.batch), [minalive] "s"(g.min_alive)

VRT_DI void grid_walk_park_gfx950(Walk &w, const f3 &inv_dir, uint32_t &index, uint32_t &cell, uint32_t stride_x, uint32_t stride_y, uint32_t stride_z,
                                  uint32_t &word, u32x4 rsrc, GridParkRegs &g) {
    unsigned long long mxa, mya, mxya, mxyb, ex, by, cz, save;
    float t0, t1, t2;
    uint32_t wordb, n;
    asm volatile(VRT_PARK_WALK_ASM(VRT_NO_LIMIT, VRT_LOAD_BUFFER, VRT_TEST_BIT, VRT_WAIT_BUFFER, "") : VRT_PARK_WALK_OPERANDS : VRT_PARK_WALK_INPUTS : "vcc", "scc");
}

// The voxel level on the same park loop (vrt_path_kernel): voxels of one brick, bits of brick_occupancy by their global bit
// index; `scale`, `t_max` as in the loop condition comp:469.  A lane that has left a SOLID voxel behind is parked; the call
// returns when nobody is moving (batch 64, min_alive 0), so that the material test behind it (three dependent cache misses on a
// scene larger than the caches) runs ONCE for all the lanes of the round instead of once per lane that finds a voxel.
VRT_DI void voxel_walk_park_gfx950(Walk &w, const f3 &inv_dir, uint32_t &index, uint32_t &cell, uint32_t stride_x, uint32_t stride_y, uint32_t stride_z,
                                   uint32_t &word, u32x4 rsrc, GridParkRegs &g, float scale, float t_max) {
    unsigned long long mxa, mya, mxya, mxyb, ex, by, cz, save;
    float t0, t1, t2;
    uint32_t wordb, n;
    asm volatile(VRT_PARK_WALK_ASM(VRT_T_LIMIT, VRT_LOAD_BUFFER, VRT_TEST_BIT, VRT_WAIT_BUFFER, "") : VRT_PARK_WALK_OPERANDS : VRT_PARK_WALK_INPUTS, [scale] "s"(scale), [tmax] "v"(t_max) : "vcc", "scc");
}

// The same with the brick's occupancy bits staged in LDS (8^3 bricks: 64 bytes = 16 words per lane).  On a scene larger than
// the caches the lanes of a round stand in ~25 different bricks; every trip of the loop above asks the L1 for one more word of
// each, the L1 has long dropped the line, and the trip waits for an L2 round trip (a request is only one trip ahead): ~600
// cycles x ~15 trips of the longest lane.  Here the lane's whole brick is fetched ONCE by four global_load_lds_dwordx4 in flight
// together (chunk c of lane l lands at wave base + 1024 c + 16 l: the layout the instruction dictates; tools/ubench/glds_probe.hip)
// and the trips read LDS: word w of the brick at lane base + ((w >> 2) << 10) + ((w & 3) << 2), w = (bit index >> 5) & 15.
#define VRT_LOAD_LDS_BRICK_A(IDX, IDXN, WORD, WORDN)                                 \
    "s_mov_b64 %[cz], exec\n\t"                                         \
    "v_lshrrev_b32_e32 %[t2], 3, %[" IDXN "]\n\t"                         \
    "v_and_b32_e32 %[t0], 48, %[t2]\n\t"                                 \
    "v_and_or_b32 %[t2], %[t2], 12, %[lb]\n\t"                           \
    "v_lshl_add_u32 %[t2], %[t0], 6, %[t2]\n\t"                          \
    "ds_read_b32 %[" WORDN "], %[t2]\n\t"
#define VRT_LOAD_LDS_BRICK_B(IDX, IDXN, WORD, WORDN)                                 \
    "s_mov_b64 exec, %[cz]\n\t"                                         \
    "s_waitcnt lgkmcnt(1)\n\t"
VRT_DI void voxel_walk_park_lds_gfx950(Walk &w, const f3 &inv_dir, uint32_t &index, uint32_t &cell, uint32_t stride_x, uint32_t stride_y, uint32_t stride_z,
                                       uint32_t &word, uint32_t lane_base, GridParkRegs &g, float scale, float t_max) {
    unsigned long long mxa, mya, mxya, mxyb, ex, by, cz, save;
    float t0, t1, t2;
    uint32_t wordb, n;
    const uint32_t rsrc = 0u; // (operand of the shared input list; unused)
    asm volatile(VRT_PARK_WALK_ASM(VRT_T_LIMIT, VRT_LOAD_LDS_BRICK, VRT_TEST_BIT, VRT_WAIT_LDS, "") : VRT_PARK_WALK_OPERANDS : VRT_PARK_WALK_INPUTS, [scale] "s"(scale), [tmax] "v"(t_max), [lb] "v"(lane_base) : "vcc", "scc");
}
#undef VRT_LOAD_LDS_BRICK_A
#undef VRT_LOAD_LDS_BRICK_B
// byte address (LDS) of the word that holds bit `bit_index` of the lane's staged brick
VRT_DI uint32_t brick_lds_address(uint32_t lane_base, uint32_t bit_index) {
    const uint32_t w = (bit_index >> 5) & 15u;
    return lane_base + ((w >> 2) << 10) + ((w & 3u) << 2);
}
// ---- the brick-level park loop on HALF-BLOCK words (vrt_path_kernel on scenes larger than the caches) ---------------------
// Measured on the 2048^3 path trace (tools/experiments/pmc_cfg4.sh): 64.5 G L1 accesses per frame, 0.58 per cycle per CU — a wave-wide
// request of incoherent lanes is one tag look-up per lane, and the L1 handles about one per cycle; 77 % of them are the status
// words of the walk loop, one per lane per trip, although the average L1 miss costs only ~190 cycles.  So the walk is bound by the
// NUMBER of requests.  Here the status bits are read from a derived copy ordered by 4 x 4 x 2 cells (x, z, y) per 32-bit word
// (TraceParams::status_halfblocks, bit (x&3) | (z&3) << 2 | (y&1) << 4): a lane asks for a word only when its step enters
// another half-block — one step in three on average — and keeps the word otherwise.  Same bits tested, same sequence per lane.
// The indices are formed from the linear cell index by bit fields: x and z dimensions powers of two >= 4, y even.
// Every trip issues exactly one request (if no lane changes its half-block, all lanes re-request theirs), so vmcnt(1) still
// means "the word asked for a trip ago has arrived".
struct HalfBlockConsts {
    // cell index = x | z << lx | y << (lx + lz); word index = x >> 2 | (z >> 2) << (lx - 2) | (y >> 1) << (lx + lz - 4)
    //            = ((index >> 2) & mx) | ((index >> 4) & mzs) | ((index >> 5) & mys)
    uint32_t nmask; // ~(3 | 3 << lx | 1 << (lx + lz)): two cell indices in the same half-block agree in these bits
    uint32_t mx;    // (1 << (lx - 2)) - 1
    uint32_t mzs;   // ((dim_z >> 2) - 1) << (lx - 2)
    uint32_t mys;   // ~((1 << (lx + lz - 4)) - 1)
    uint32_t lx;    // log2(dim_x): z & 3 starts here
    uint32_t lxz;   // log2(dim_x) + log2(dim_z): y & 1 is this bit
};
#define VRT_LOAD_HALFBLOCK_A(IDX, IDXN, WORD, WORDN)                       \
    "v_xor_b32_e32 %[t2], %[" IDXN "], %[" IDX "]\n\t"                     \
    "v_and_b32_e32 %[t2], %[nmask], %[t2]\n\t"                             \
    "v_cmp_ne_u32_e64 %[by], 0, %[t2]\n\t"                                 \
    "s_cmp_eq_u64 %[by], 0\n\t"                                            \
    "s_cselect_b64 %[by], exec, %[by]\n\t"                                 \
    "s_and_saveexec_b64 %[cz], %[by]\n\t"                                  \
    "v_lshrrev_b32_e32 %[t2], 2, %[" IDXN "]\n\t"                          \
    "v_and_b32_e32 %[t2], %[mx], %[t2]\n\t"                                \
    "v_lshrrev_b32_e32 %[t0], 4, %[" IDXN "]\n\t"                          \
    "v_and_or_b32 %[t2], %[t0], %[mzs], %[t2]\n\t"                         \
    "v_lshrrev_b32_e32 %[t0], 5, %[" IDXN "]\n\t"                          \
    "v_and_or_b32 %[t2], %[t0], %[mys], %[t2]\n\t"                         \
    "buffer_load_dword %[" WORDN "], %[t2], %[rsrc], 0 idxen\n\t"
#define VRT_LOAD_HALFBLOCK_B(IDX, IDXN, WORD, WORDN)                       \
    "s_andn2_b64 exec, %[cz], %[by]\n\t"                                   \
    "s_waitcnt vmcnt(1)\n\t"                                               \
    "v_mov_b32_e32 %[" WORDN "], %[" WORD "]\n\t"                          \
    "s_mov_b64 exec, %[cz]\n\t"
#define VRT_TEST_HALFBLOCK(WORD, IDX)                                      \
    "v_bfe_u32 %[t1], %[" IDX "], %[lx], 2\n\t"                            \
    "v_and_b32_e32 %[t0], 3, %[" IDX "]\n\t"                               \
    "v_lshl_or_b32 %[t0], %[t1], 2, %[t0]\n\t"                             \
    "v_bfe_u32 %[t1], %[" IDX "], %[lxz], 1\n\t"                           \
    "v_lshl_or_b32 %[t0], %[t1], 4, %[t0]\n\t"                             \
    "v_bfe_u32 %[t1], %[" WORD "], %[t0], 1\n\t"                           \
    "v_cmp_ne_u32_e32 vcc, 0, %[t1]\n\t"
VRT_DI void grid_walk_park_halfblocks_gfx950(Walk &w, const f3 &inv_dir, uint32_t &index, uint32_t &cell, uint32_t stride_x, uint32_t stride_y, uint32_t stride_z,
                                             uint32_t &word, u32x4 rsrc, GridParkRegs &g, const HalfBlockConsts &hb) {
    unsigned long long mxa, mya, mxya, mxyb, ex, by, cz, save;
    float t0, t1, t2;
    uint32_t wordb, n;
    asm volatile(VRT_PARK_WALK_ASM(VRT_NO_LIMIT, VRT_LOAD_HALFBLOCK, VRT_TEST_HALFBLOCK, VRT_WAIT_BUFFER, "")
                 : VRT_PARK_WALK_OPERANDS
                 : VRT_PARK_WALK_INPUTS, [nmask] "s"(hb.nmask), [mx] "s"(hb.mx), [mzs] "s"(hb.mzs), [mys] "s"(hb.mys), [lx] "s"(hb.lx), [lxz] "s"(hb.lxz)
                 : "vcc", "scc");
}
#undef VRT_LOAD_HALFBLOCK_A
#undef VRT_LOAD_HALFBLOCK_B
#undef VRT_TEST_HALFBLOCK
// ---- the half-block park loop on a DILATED cell index (vrt_path_kernel<DIL>, round 3) ------------------------------------------
// The half-block loop above pays 17 of its 29 vector instructions per trip for turning a linear cell index into a half-block word
// index and a bit position.  Here the walk's index IS that pair: bits 0-4 = the cell's place in its half-block
// (x&3 | (z&3) << 2 | (y&1) << 4), the bits above = the half-block word's index (x>>2 | (z>>2) << (lx-2) | (y>>1) << (lx+lz-4)) —
// three bit fields per axis interleaved.  A step increments the crossed axis' field (VRT_STEP_DILATED: the carry runs through ones
// filled into the other fields); so that every step is an increment, an axis the ray walks DOWN is stored mirrored (dim-1-c = ~c
// inside the field: all three dimensions powers of two), and `flip` (per lane: the field masks of those axes) turns the index into
// the real one: word = real >> 5, bit = real & 31 (v_bfe takes it from the low five bits).  Two cells lie in the same half-block
// iff their indices agree above bit 4 — mirrored or not.  22 vector instructions per trip; same words, same requests, same
// sequence of DDA operations per lane.
#define VRT_LOAD_DILATED_A(IDX, IDXN, WORD, WORDN)                          \
    "v_xor_b32_e32 %[t2], %[" IDXN "], %[" IDX "]\n\t"                     \
    "v_cmp_lt_u32_e64 %[by], 31, %[t2]\n\t"                                \
    "s_cmp_eq_u64 %[by], 0\n\t"                                            \
    "s_cselect_b64 %[by], exec, %[by]\n\t"                                 \
    "v_xor_b32_e32 %[r" IDXN "], %[" IDXN "], %[flip]\n\t" /* the real index: kept for the test a trip later */ \
    "s_and_saveexec_b64 %[cz], %[by]\n\t" /* cz: the trip's lanes, until LOADB */ \
    "v_lshrrev_b32_e32 %[t2], 5, %[r" IDXN "]\n\t"                         \
    "buffer_load_dword %[" WORDN "], %[t2], %[rsrc], 0 idxen\n\t"
#define VRT_LOAD_DILATED_B(IDX, IDXN, WORD, WORDN)                          \
    "s_andn2_b64 exec, %[cz], %[by]\n\t"                                   \
    "s_waitcnt vmcnt(1)\n\t"                                               \
    "v_mov_b32_e32 %[" WORDN "], %[" WORD "]\n\t"                          \
    "s_mov_b64 exec, %[cz]\n\t"
#define VRT_TEST_DILATED(WORD, IDX)                                        \
    "v_bfe_u32 %[t1], %[" WORD "], %[r" IDX "], 1\n\t"                     \
    "v_cmp_ne_u32_e32 vcc, 0, %[t1]\n\t"
// nm_x / nm_y / nm_z: per lane, the complement of the axis' field mask (all ones for an axis with ray_step == 0)
VRT_DI void grid_walk_park_dilated_gfx950(Walk &w, const f3 &inv_dir, uint32_t &index, uint32_t &cell, uint32_t nm_x, uint32_t nm_y, uint32_t nm_z,
                                          uint32_t &word, u32x4 rsrc, GridParkRegs &g, uint32_t flip) {
    unsigned long long mxa, mya, mxya, mxyb, ex, by, cz, save;
    float t0, t1, t2;
    uint32_t wordb, n;
    const uint32_t stride_x = nm_x, stride_y = nm_y, stride_z = nm_z; // (the operand list's names)
    uint32_t ridxa, ridxb; // (the real — un-mirrored — indices of the two cells in flight)
    asm volatile("v_xor_b32_e32 %[ridxa], %[idxa], %[flip]\n\t"
                 VRT_PARK_WALK_ASM_T(VRT_TRIP_E, VRT_STEP_DILATED, VRT_EXIT_COUNTERS, "", VRT_NO_LIMIT, VRT_LOAD_DILATED, VRT_TEST_DILATED, VRT_WAIT_BUFFER, "")
                 : VRT_PARK_WALK_OPERANDS, [ridxa] "=&v"(ridxa), [ridxb] "=&v"(ridxb)
                 : VRT_PARK_WALK_INPUTS, [flip] "v"(flip)
                 : "vcc", "scc");
}
// The same loop for a walk that ends at the GRID's face (the box of the occupied cells is, or nearly is, the grid): the steps-left
// counters leave the loop (3 vector + 2 scalar instructions per trip, and six registers that no longer live across it) — a step
// that overflows its field has left the grid, and the increment's carry-out says so (VRT_STEP_DILATED_CARRY).  A parked lane
// whose step out of the occupied cell left the grid is noted in `gone`.  An axis the ray does not move along (mask all ones) "leaves"
// the moment it is selected: with a finite unit direction that never happens before the lane has left along its dominant axis
// (such an axis' side distance starts at 5e11, safe_inverse; the dominant axis' stays below 1.74 x (cells + 1)).
VRT_DI void grid_walk_park_dilated_carry_gfx950(f3 &side_dist, const f3 &inv_dir, uint32_t &index, uint32_t &cell, uint32_t nm_x, uint32_t nm_y, uint32_t nm_z,
                                                uint32_t &word, u32x4 rsrc, GridParkRegs &g, uint32_t flip, unsigned long long &gone) {
    unsigned long long mxa, mya, mxya, mxyb, ex, by, cz, save;
    float t0, t1, t2;
    uint32_t wordb, n;
    const uint32_t stride_x = nm_x, stride_y = nm_y, stride_z = nm_z; // (the operand list's names)
    gone = 0ull;
    uint32_t ridxa, ridxb; // (the real — un-mirrored — indices of the two cells in flight)
    asm volatile("v_xor_b32_e32 %[ridxa], %[idxa], %[flip]\n\t"
                 VRT_PARK_WALK_ASM_T(VRT_TRIP_E, VRT_STEP_DILATED_CARRY, VRT_EXIT_CARRY, "s_and_b64 %[by], %[ex], vcc\n\ts_or_b64 %[gone], %[gone], %[by]\n\t",
                                     VRT_NO_LIMIT, VRT_LOAD_DILATED, VRT_TEST_DILATED, VRT_WAIT_BUFFER, "")
                 : [sdx] "+v"(side_dist.x), [sdy] "+v"(side_dist.y), [sdz] "+v"(side_dist.z), [idxa] "+v"(index), [idxb] "=&v"(cell), [worda] "+v"(word),
                   [wordb] "=&v"(wordb), [tsb] "+v"(g.t_out), [tsa] "=&v"(g.t_in), [code] "+v"(g.code), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
                   [mxb] "+s"(g.out_x), [myb] "+s"(g.out_y), [alive] "+s"(g.alive), [mxa] "=&s"(mxa), [mya] "=&s"(mya), [mxya] "=&s"(mxya), [mxyb] "=&s"(mxyb),
                   [ex] "=&s"(ex), [by] "=&s"(by), [cz] "=&s"(cz), [save] "=&s"(save), [parked] "=&s"(g.parked), [n] "=&s"(n), [gone] "+s"(gone),
                   [ridxa] "=&v"(ridxa), [ridxb] "=&v"(ridxb)
                 : VRT_PARK_WALK_INPUTS, [flip] "v"(flip)
                 : "vcc", "scc");
}
#undef VRT_LOAD_DILATED_A
#undef VRT_LOAD_DILATED_B
#undef VRT_TEST_DILATED
#if defined(VRT_DEV_VARIANTS) || defined(VRT_POOL_AHEAD)
#define VRT_DEV_SECTION 3
#include "vrt_trace_kernels_dev.h"
#undef VRT_DEV_SECTION
#endif
#ifdef VRT_DEV_VARIANTS
#define VRT_DEV_SECTION 4
#include "vrt_trace_kernels_dev.h"
#undef VRT_DEV_SECTION
#endif
// index of the half-block word that holds cell `index`
VRT_DI uint32_t halfblock_word(const HalfBlockConsts &hb, uint32_t index) {
    return ((index >> 2) & hb.mx) | ((index >> 4) & hb.mzs) | ((index >> 5) & hb.mys);
}
#undef VRT_PARK_WALK_ASM
#undef VRT_PARK_WALK_ASM_S
#undef VRT_PARK_WALK_ASM_T
#undef VRT_PARK_WALK_ASM_W
#undef VRT_WORD_MOV32
#undef VRT_PARK_WALK_OPERANDS
#undef VRT_PARK_WALK_INPUTS
#ifdef VRT_DEV_VARIANTS
#define VRT_DEV_SECTION 5
#include "vrt_trace_kernels_dev.h"
#undef VRT_DEV_SECTION
#endif
// ---- block filter (vrt_path_kernel<FILTER>) ---------------------------------------------------------------------------
// The 1-bit-per-4x4x4-cells filter that vrt_build_status_blocks derives from binding 3 ("some cell of the block is occupied"),
// staged in LDS once per workgroup.  The block index is formed from the linear cell index by bit fields, so the grid's x and z
// dimensions must be powers of two >= 4 and y a multiple of 4 (every BASELINE configuration); other grids walk cell by cell.
// (Round 2 first used the filter only to suppress status requests for cells of empty blocks, one trip per cell as before:
// slower, 215 against 204 ms on the 2048^3 path trace.  Now a lane in an empty block JUMPS to the block's exit face,
// skip_empty_block, and only lanes in non-empty blocks take trips.)
struct FilterConsts {
    uint32_t wx;   // log2(dim_x) - 2: width of the x block field
    uint32_t shz;  // log2(dim_x) + 2: the z block field starts here in the cell index
    uint32_t mz;   // (dim_z >> 2) - 1
    uint32_t shy;  // log2(dim_x) + log2(dim_z) + 2
    uint32_t shyb; // log2(dim_x) + log2(dim_z) - 4: where the y block field goes in the block index
};
#undef VRT_PARK
#undef VRT_IN_FROM_CODE
#undef VRT_IN_FROM
#undef VRT_SWAP_SETS
#undef VRT_T_LIMIT
#undef VRT_NO_LIMIT
#undef VRT_TRIP
#undef VRT_TRIP_T
#undef VRT_TEST_BIT
#undef VRT_TEST_BYTE
#undef VRT_LOAD_BYTE_A
#undef VRT_LOAD_BYTE_B

// comp:298 / comp:395
VRT_DI f3 initial_side_dist(f3 fstep, f3 fposition, f3 ray_delta) {
    const f3 intersection_delta = floor3(fposition) - fposition;
    return fma3(fstep, intersection_delta, fstep * 0.5f + splat3(0.5f)) * ray_delta;
}

struct RaySetup {
    f3 inv_dir;   // 1/dir (safeInverse); ray_delta = |1/dir| (comp:295) is formed where it is used: |x| is a free source modifier of the
                  // vector instructions, and three registers fewer stay live across both walks
    VRT_DI f3 ray_delta() const { return abs3(inv_dir); }
    int entry_code; // slab-entry normal (comp:529-531): axis index | sign bits, see axis_normal()
    int sx, sy, sz;
    float grid_t_min, grid_t_max;
};

// comp:522-536 with comp:278 (slab test against the grid box)
VRT_DI bool grid_slab(const TraceParams &p, const Ray &r, float t_min, float t_max, RaySetup &s) {
    const f3 g_min = mk3(p.grid.min_point_base_t[0], p.grid.min_point_base_t[1], p.grid.min_point_base_t[2]);
    const f3 g_max = mk3(p.grid.max_point_scale[0], p.grid.max_point_scale[1], p.grid.max_point_scale[2]);
    const f3 inv = mk3(safe_inverse(r.direction.x), safe_inverse(r.direction.y), safe_inverse(r.direction.z));
    const f3 t_lower = (g_min - r.origin) * inv;
    const f3 t_upper = (g_max - r.origin) * inv;
    const f3 t_mins = mk3(gl_min(t_lower.x, t_upper.x), gl_min(t_lower.y, t_upper.y), gl_min(t_lower.z, t_upper.z));
    const f3 t_maxes = mk3(gl_max(t_lower.x, t_upper.x), gl_max(t_lower.y, t_upper.y), gl_max(t_lower.z, t_upper.z));
    // indexOfMaxComponent, comp:501-503 (ties resolve to 0)
    const bool iy = (t_mins.y > t_mins.x) && (t_mins.y > t_mins.z);
    const bool iz = (t_mins.z > t_mins.x) && (t_mins.z > t_mins.y);
    // iy and iz cannot both hold; index = iy + 2*iz
    const float inv_i = iz ? inv.z : (iy ? inv.y : inv.x);
    const float tmin_i = iz ? t_mins.z : (iy ? t_mins.y : t_mins.x);
    // packed: bits 0-1 axis (0,1,2), bit 2 = negative, bit 3 = zero or NaN.  (Not "the axis now, the sign from inv_dir when
    // needed": a select over the members of RaySetup becomes an indexed load, and the optimiser then moves the whole struct
    // to LDS — 12 KiB per workgroup and +20 % on the headline frame.)
    s.entry_code = (iz ? 2 : (iy ? 1 : 0)) | (inv_i < 0.0f ? 4 : 0) | (!(inv_i < 0.0f) && !(inv_i > 0.0f) ? 8 : 0);
    s.grid_t_min = gl_max(t_min, tmin_i);
    s.grid_t_max = gl_min(t_max, gl_min(gl_min(t_maxes.x, t_maxes.y), t_maxes.z));
    s.inv_dir = inv;
    s.sx = (int)sign1(r.direction.x);
    s.sy = (int)sign1(r.direction.y);
    s.sz = (int)sign1(r.direction.z);
    return s.grid_t_min <= s.grid_t_max;
}

VRT_DI f3 axis_normal(const RaySetup &s, int axis) {
    // normal_axis, comp:304-308: (step < 0) ? 1 : -1 on the crossed axis; axis 3 = slab-entry normal
    const float nx = (s.sx < 0) ? 1.0f : -1.0f, ny = (s.sy < 0) ? 1.0f : -1.0f, nz = (s.sz < 0) ? 1.0f : -1.0f;
    const int ea = s.entry_code & 3;
    const float ev = (s.entry_code & 8) ? 0.0f : ((s.entry_code & 4) ? -1.0f : 1.0f);
    return mk3(axis == 0 ? nx : ((axis == 3 && ea == 0) ? ev : 0.0f), axis == 1 ? ny : ((axis == 3 && ea == 1) ? ev : 0.0f),
               axis == 2 ? nz : ((axis == 3 && ea == 2) ? ev : 0.0f));
}

// bit (index % 32) of a status word.  v_bfe_u32 takes the offset from the low five bits of its operand,
// so no separate `index & 31`; written as asm because the compiler does not drop the mask by itself.
VRT_DI bool status_bit(uint32_t word, uint32_t index) {
    uint32_t r;
    asm("v_bfe_u32 %0, %1, %2, 1" : "=v"(r) : "v"(word), "v"(index));
    return r != 0u;
}

// Word i of the dynamic LDS region.  The traversal kernel declares no static LDS, so its dynamic region
// starts at LDS address 0 (MI355X guide, G17); addressing it as address-space-3 offset 0 saves the
// per-access add of a link-time base the compiler cannot fold.
VRT_DI uint32_t lds_word0(uint32_t i) {
    typedef __attribute__((address_space(3))) const uint32_t lds_u32;
    return reinterpret_cast<lds_u32 *>(0)[i];
}

VRT_DI bool bit64(uint2 w, uint32_t bit) { // bit 0..63 of a 64-bit word held as two dwords
    const uint32_t half = (bit & 32u) ? w.y : w.x;
    return (half >> (bit & 31u)) & 1u;
}

// all three in [0, dim), dim a power of two (4 or 8): the OR of three such values stays below dim, and any negative or larger one lifts it above
VRT_DI bool more_init(int px, int py, int pz, int dim) { return (unsigned)(px | py | pz) < (unsigned)dim; }

// comp:378-471.  Returns true on a (non-ignored) voxel hit and fills `hit`.
// LITERAL: one byte load per voxel step (comp:415); otherwise 64-bit occupancy words:
// brick_occupancy bit v%8 of byte brick*(B^3/8) + v/8 (Grid.zig:180-182) read as little-endian
// 64-bit words — the whole brick for B=4, one y-layer (v = x + 8*(z + 8*y)) for B=8.
template <int B, bool COUNT, bool LITERAL>
VRT_DI bool brick_walk(const TraceParams &p, const Ray &r, const RaySetup &s, float g_scale, uint32_t brick_index, f3 brick_min, Hit &hit,
                       int &axis, Cnt<COUNT> &c) {
    const float brick_voxel_scale = 1.0f / (float)B; // spec const 5, Pipeline.zig:313
    const float voxel_scale = g_scale * brick_voxel_scale;
    const f3 fposition = p.scale_pow2 ? (ray_at(r, hit.t) - brick_min) * p.inv_voxel_scale : (ray_at(r, hit.t) - brick_min) / splat3(voxel_scale);
    Walk w;
    w.side_dist = initial_side_dist(mk3((float)s.sx, (float)s.sy, (float)s.sz), fposition, s.ray_delta());
    const int px = f2i_clamp(__builtin_floorf(fposition.x));
    const int py = f2i_clamp(__builtin_floorf(fposition.y));
    const int pz = f2i_clamp(__builtin_floorf(fposition.z));
    constexpr int kZeroBudget = 3 * B + 8;
    w.rx = steps_left(s.sx, px, B, kZeroBudget);
    w.ry = steps_left(s.sy, py, B, kZeroBudget);
    w.rz = steps_left(s.sz, pz, B, kZeroBudget);
    w.t_value = 0;
    const float local_t_max = s.grid_t_max - hit.t;
    uint32_t voxel_index = (uint32_t)px + (uint32_t)B * ((uint32_t)pz + (uint32_t)B * (uint32_t)py); // comp:412, kept current by dda_step
    const uint32_t stride_x = (uint32_t)s.sx, stride_y = (uint32_t)(s.sy * (B * B)), stride_z = (uint32_t)(s.sz * B);

    uint2 occ = make_uint2(0u, 0u);
    uint32_t occ_layer = ~0u;
    const uint2 *occ_words = reinterpret_cast<const uint2 *>(p.brick_occupancy);
    if constexpr (!LITERAL && B == 4) occ = occ_words[brick_index];
    // B == 8: one 64-bit word per y-layer.  The walk moves through the layers in one direction (sy), so the
    // word of the layer after the current one is requested a layer early: a walk then waits for memory once,
    // not once per layer (these dependent loads are what a wave with many brick walks spends its time on).
    uint2 occ_ahead = make_uint2(0u, 0u);
    [[maybe_unused]] const uint2 *brick_layers = occ_words + (size_t)brick_index * 8u;
    [[maybe_unused]] auto layer_ahead = [&](uint32_t layer) -> uint2 {
        const uint32_t next = layer + (uint32_t)s.sy; // sy == 0: the layer never changes, any valid word will do
        return brick_layers[next < 8u ? next : layer];
    };
    if constexpr (!LITERAL && B == 8) {
        if (more_init(px, py, pz, B)) {
            occ_layer = (uint32_t)py;
            occ = brick_layers[occ_layer];
            occ_ahead = layer_ahead(occ_layer);
        }
    }

    // Single-exit loop (one back-edge condition, no return inside): the structurizer then needs one
    // exec update per iteration instead of a chain of exit-flag merges on the scalar unit.
    bool found = false;
    bool more = (unsigned)px < (unsigned)B && (unsigned)py < (unsigned)B && (unsigned)pz < (unsigned)B && w.t_value <= local_t_max;
    VRT_PROF_BEGIN(tp2);
    while (more) {
        VRT_COUNT(voxel_steps);
        VRT_COUNT_WAVE(wave_voxel_iters);
        bool solid;
        if constexpr (LITERAL) {
            const uint32_t byte = p.brick_occupancy[brick_index * (uint32_t)(B * B * B / 8) + (voxel_index >> 3)];
            solid = (byte >> (voxel_index & 7u)) & 1u;
        } else if constexpr (B == 4) {
            solid = bit64(occ, voxel_index);
        } else {
            const uint32_t layer = voxel_index >> 6; // y
            if (occ_layer != layer) { // crossed into the next layer: its word was requested a layer ago
                occ = occ_ahead;
                occ_layer = layer;
                occ_ahead = layer_ahead(layer);
            }
            solid = bit64(occ, voxel_index & 63u);
        }
        if (solid) {
            VRT_COUNT(hits);
            const uint32_t brick_material_index = p.brick_start_index[brick_index] & 0x7FFFFFFFu; // comp:422
            const uint32_t mi = p.material_index[brick_material_index + voxel_index];
            const vrt_material *m = p.materials + mi;
            const uint32_t mtype = m->type;
            const float mdata = m->type_data;
            const bool ignore_brick = (mtype == r.ignore_type_material) && (r.internal_reflection == mdata); // comp:427
            if (!ignore_brick) {
                hit.index = mi;
                const float t_offset = voxel_scale * 0.05f;
                hit.t += w.t_value - t_offset;
                hit.normal = axis_normal(s, axis);
                hit.point = ray_at(r, hit.t) + hit.normal * t_offset;
                found = true;
            }
        }
        // (after a hit this step is dead work, once per ray; its results are never read)
        dda_step<false>(w, s.ray_delta(), voxel_scale, axis, voxel_index, stride_x, stride_y, stride_z);
        more = !found && min3i(w.rx, w.ry, w.rz) >= 0 && w.t_value <= local_t_max;
    }
    VRT_PROF_END(2, tp2);
    return found;
}

// comp:378-471 on the hand-written voxel loop (voxel_walk_gfx950): same operations per lane as brick_walk.
// The loop returns when some lane has left a solid voxel behind; the material test (comp:422-427) and the hit
// record are done here, and lanes whose voxel is to be ignored walk on.  `axis_in`: the face through which the
// brick was entered (the brick-level walk's crossed axis), used when the very first voxel is the hit.
// BY_CELL (round 4, frames with bounces on scenes that stay in the caches): the brick's bits are read from the by-cell copy
// (TraceParams::cell_occupancy, `cell` = the grid cell) when the context holds one, so that the request for them does not wait for
// brick_index[cell] (comp:337) — which then only the material look-up of a solid voxel needs.  Such frames last as long as their slowest
// wave, and that wave's time is its chain of brick entries: one dependent round trip less per entry.
template <int B, bool EAGER = true, bool BY_CELL = false>
VRT_DI bool brick_walk_gfx950(const TraceParams &p, const Ray &r, const RaySetup &s, float g_scale, uint32_t brick_index, f3 brick_min, Hit &hit,
                              int axis_in, int &hit_axis, uint32_t cell = 0u) {
    const float brick_voxel_scale = 1.0f / (float)B; // spec const 5, Pipeline.zig:313
    const float voxel_scale = g_scale * brick_voxel_scale;
    const f3 fposition = p.scale_pow2 ? (ray_at(r, hit.t) - brick_min) * p.inv_voxel_scale : (ray_at(r, hit.t) - brick_min) / splat3(voxel_scale);
    Walk w;
    w.side_dist = initial_side_dist(mk3((float)s.sx, (float)s.sy, (float)s.sz), fposition, s.ray_delta());
    const int px = f2i_clamp(__builtin_floorf(fposition.x));
    const int py = f2i_clamp(__builtin_floorf(fposition.y));
    const int pz = f2i_clamp(__builtin_floorf(fposition.z));
    constexpr int kZeroBudget = 3 * B + 8;
    w.rx = steps_left(s.sx, px, B, kZeroBudget);
    w.ry = steps_left(s.sy, py, B, kZeroBudget);
    w.rz = steps_left(s.sz, pz, B, kZeroBudget);
    w.t_value = 0;
    const float local_t_max = s.grid_t_max - hit.t;
    // global bit index of the voxel in brick_occupancy: brick * B^3 + voxel index (comp:412-415)
    const bool by_cell = BY_CELL && p.cell_occupancy_lockstep != 0u; // (wave-uniform; vrt_create keeps cells * B^3 below 2^32 for this copy)
    const uint32_t base = (by_cell ? cell : brick_index) * (uint32_t)(B * B * B);
    uint32_t bit_index = base + ((uint32_t)px + (uint32_t)B * ((uint32_t)pz + (uint32_t)B * (uint32_t)py));
    const uint32_t stride_x = (uint32_t)s.sx, stride_y = (uint32_t)(s.sy * (B * B)), stride_z = (uint32_t)(s.sz * B);
    const bool more = more_init(px, py, pz, B) && (0.0f <= local_t_max); // comp:409 with t_value = 0

    const uint8_t *const occupancy = by_cell ? p.cell_occupancy : p.brick_occupancy;
    const unsigned long long occ_addr = (unsigned long long)occupancy;
    u32x4 rsrc;
    rsrc.x = (uint32_t)occ_addr;
    rsrc.y = (uint32_t)(occ_addr >> 32) | (4u << 16);
    rsrc.z = by_cell ? p.status_cells * (uint32_t)(B * B * B / 32) : p.occupancy_words;
    rsrc.w = 0x00020000u;
    uint32_t word = reinterpret_cast<const uint32_t *>(occupancy)[more ? (bit_index >> 5) : 0u];
    // comp:422.  EAGER: requested before the walk, so that a solid voxel's material test starts one dependent load later
    // (scenes that stay in the caches).  Otherwise requested at the first solid voxel: on a scene larger than the caches
    // every request is a 128-byte line from HBM, and four of five brick walks of the path-trace configuration end without
    // a solid voxel (K = 3.2 bricks entered, H = 0.7 hits per ray).
    uint32_t brick_material_index = 0u;
    const bool start_is_slot = p.start_is_slot != nullptr && __builtin_amdgcn_readfirstlane((int)*p.start_is_slot) != 0; // (TraceParams; wave-uniform)
    if constexpr (EAGER) brick_material_index = start_is_slot ? brick_index * (uint32_t)(B * B * B) : (p.brick_start_index[brick_index] & 0x7FFFFFFFu);
    GridWalkRegs g;
    g.alive = __builtin_amdgcn_ballot_w64(more);
    g.out_x = 0ull;
    g.out_y = 0ull;
    g.t_out = 0.0f;
    bool first = true; // wave-uniform
    bool found = false;
    while (g.alive != 0ull) {
        uint32_t solid_bit; // bit index of the voxel each lane stood on before its last step
        VRT_PROF_BEGIN(tp2);
        voxel_walk_gfx950(w, s.inv_dir, bit_index, solid_bit, stride_x, stride_y, stride_z, word, rsrc, g, voxel_scale, local_t_max);
        VRT_PROF_END(2, tp2);
        if (g.occ == 0ull) break; // every lane has left the brick (or the grid box)
        if (__builtin_amdgcn_inverse_ballot_w64(g.occ)) {
            VRT_PROF_BEGIN(tp4);
            const uint32_t voxel_index = solid_bit - base;
            if constexpr (!EAGER) brick_material_index = start_is_slot ? brick_index * (uint32_t)(B * B * B) : (p.brick_start_index[brick_index] & 0x7FFFFFFFu);
            const uint32_t mi = p.material_index[brick_material_index + voxel_index];
            const vrt_material *m = p.materials + mi;
            const uint32_t mtype = m->type;
            const float mdata = m->type_data;
            const bool ignore_brick = (mtype == r.ignore_type_material) && (r.internal_reflection == mdata); // comp:427
            if (!ignore_brick) {
                const int a = (first && g.stub == 0u)
                                  ? axis_in
                                  : (__builtin_amdgcn_inverse_ballot_w64(g.in_x) ? 0 : (__builtin_amdgcn_inverse_ballot_w64(g.in_y) ? 1 : 2));
                // (hit.normal and hit.point are derived from hit_axis and hit.t once the walk is over: six registers
                // less to carry through both loops)
                hit.index = mi;
                const float t_offset = voxel_scale * 0.05f;
                hit.t += g.t_in * voxel_scale - t_offset; // t_value of the step into this voxel (comp:442), 0 for the first
                hit_axis = a;
                found = true;
            }
            VRT_PROF_END(4, tp4);
        }
        asm("s_andn2_b64 %0, %0, %1" : "+s"(g.alive) : "s"(__builtin_amdgcn_ballot_w64(found)) : "scc");
        first = false;
    }
    return found;
}

// comp:378-471 for vrt_path_kernel: the same per-lane operations as brick_walk_gfx950 on the voxel-level PARK loop.  Lanes that
// have left a solid voxel behind wait (parked) until no lane of the round is moving; then the material test (comp:422-427) is made
// for all of them at once — on a scene larger than the caches it is three dependent cache misses (start index -> material id ->
// material), and brick_walk_gfx950 pays them once per lane that finds a voxel (22 % of the wave-cycles of the 2048^3 path trace,
// tools/path_profile.py).  A lane whose voxel is to be ignored (comp:427) walks on in the next pass.
// LDS (8^3 bricks): the lane's brick is staged at LDS byte address wave_lds + 1024 c + 16 lane (c = 0..3) and walked there
// (voxel_walk_park_lds_gfx950).
// occ_slot: where the brick's bits lie, in bricks of the array `occupancy` — the brick's slot in binding 5, or (by_cell) the grid
// cell in the by-cell copy TraceParams::cell_occupancy, in which case the slot (brick_index[cell], comp:337) is looked up only by
// the lanes that have found a solid voxel.  start_is_slot: comp:422's look-up is slot * B^3 for every brick (TraceParams).
// The request for a lane's brick (LDS walk): four 16-byte chunks into the wave's staging area.  vrt_path_kernel issues it as soon as
// it knows the cell (PRESTAGED) — the position arithmetic of the brick entry then runs while the 64 bytes are on their way.
#ifndef VRT_BRICK_LOAD_AUX
#define VRT_BRICK_LOAD_AUX 0 // cache-policy bits of the staging loads (experiment, DESIGN.md appendix: 2 = nt)
#endif
VRT_DI void stage_brick_lds(const TraceParams &p, uint32_t occ_slot, bool by_cell, uint32_t wave_lds) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(by_cell ? p.cell_occupancy : p.brick_occupancy) + (size_t)occ_slot * 16u;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    lds_void *dst = reinterpret_cast<lds_void *>((size_t)__builtin_amdgcn_readfirstlane(wave_lds));
#pragma unroll
    for (int c = 0; c < 4; c++)
        __builtin_amdgcn_global_load_lds((glb_void *)(src + 4 * c), (lds_void *)((__attribute__((address_space(3))) char *)dst + 1024 * c), 16, 0, VRT_BRICK_LOAD_AUX);
}
// DEFER (vrt_pool_kernel, round 4): a lane whose ray ignores no material type any record has (TraceParams::materials_plain, the ray's
// ignore type MAT_NONE) needs none of comp:422-427's three dependent look-ups here — its hit is recorded as the voxel's index in its
// brick with kDeferredHit set in hit.index, and the round of transitions that shades the hit looks the material up (pool_hit_material).
constexpr uint32_t kDeferredHit = 0x80000000u;
template <int B, bool LDS = false, bool PRESTAGED = false, bool DEFER = false>
VRT_DI bool brick_walk_park_gfx950(const TraceParams &p, const Ray &r, const RaySetup &s, float g_scale, uint32_t occ_slot, uint32_t cell, bool by_cell,
                                   bool start_is_slot, f3 brick_min, Hit &hit, int axis_in, int &hit_axis, uint32_t wave_lds = 0u) {
    static_assert(!LDS || B == 8, "the LDS layout is written for 64-byte bricks");
    const float brick_voxel_scale = 1.0f / (float)B; // spec const 5, Pipeline.zig:313
    const float voxel_scale = g_scale * brick_voxel_scale;
    const f3 fposition = p.scale_pow2 ? (ray_at(r, hit.t) - brick_min) * p.inv_voxel_scale : (ray_at(r, hit.t) - brick_min) / splat3(voxel_scale);
    Walk w;
    w.side_dist = initial_side_dist(mk3((float)s.sx, (float)s.sy, (float)s.sz), fposition, s.ray_delta());
    const int px = f2i_clamp(__builtin_floorf(fposition.x));
    const int py = f2i_clamp(__builtin_floorf(fposition.y));
    const int pz = f2i_clamp(__builtin_floorf(fposition.z));
    constexpr int kZeroBudget = 3 * B + 8;
    w.rx = steps_left(s.sx, px, B, kZeroBudget);
    w.ry = steps_left(s.sy, py, B, kZeroBudget);
    w.rz = steps_left(s.sz, pz, B, kZeroBudget);
    w.t_value = 0;
    const float local_t_max = s.grid_t_max - hit.t;
    // (staged in LDS, only the voxel's nine index bits are ever used: no base, and no 32-bit limit on cells * B^3 for the by-cell copy)
    const uint32_t base = LDS ? 0u : occ_slot * (uint32_t)(B * B * B);
    uint32_t bit_index = base + ((uint32_t)px + (uint32_t)B * ((uint32_t)pz + (uint32_t)B * (uint32_t)py));
    const uint32_t stride_x = (uint32_t)s.sx, stride_y = (uint32_t)(s.sy * (B * B)), stride_z = (uint32_t)(s.sz * B);
    const bool more = more_init(px, py, pz, B) && (0.0f <= local_t_max); // comp:409 with t_value = 0

    const uint8_t *const occupancy = by_cell ? p.cell_occupancy : p.brick_occupancy;
    const unsigned long long occ_addr = (unsigned long long)occupancy;
    u32x4 rsrc;
    rsrc.x = (uint32_t)occ_addr;
    rsrc.y = (uint32_t)(occ_addr >> 32) | (4u << 16);
    rsrc.z = by_cell ? p.status_cells * (uint32_t)(B * B * B / 32) : p.occupancy_words; // (by_cell without LDS: vrt_create keeps cells * B^3 below 2^32)
    rsrc.w = 0x00020000u;
    const uint32_t *occ_words = reinterpret_cast<const uint32_t *>(occupancy);
    typedef __attribute__((address_space(3))) const uint32_t lds_u32;
    // (the lane's number is formed here, by an instruction the optimiser may not move: hoisted out of the caller's loop it is one more
    // per-lane register that lives as long as the kernel)
    [[maybe_unused]] uint32_t lane_base;
    if constexpr (LDS) {
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_base));
        lane_base = wave_lds + (lane_base << 4);
    }
    uint32_t word;
    [[maybe_unused]] uint32_t eager_start = 0u;
    VRT_PROF_BEGIN(tp5);
    if constexpr (LDS) {
        if constexpr (!PRESTAGED) stage_brick_lds(p, occ_slot, by_cell, wave_lds);
        // (path_eager_start: the brick's start index travels with the four chunks instead of after the walk)
        if (p.path_eager_start) eager_start = p.brick_start_index[by_cell ? p.brick_index[cell] : occ_slot];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        word = reinterpret_cast<lds_u32 *>(0)[brick_lds_address(lane_base, bit_index) >> 2];
    } else {
        word = occ_words[more ? (bit_index >> 5) : 0u];
    }
    VRT_PROF_END(5, tp5);
    GridParkRegs g;
    g.alive = __builtin_amdgcn_ballot_w64(more);
    g.out_x = 0ull;
    g.out_y = 0ull;
    g.t_out = 0.0f;
    g.t_in = 0.0f;
    g.code = 3u << 4; // the first voxel of the walk was entered through the brick's face, not by a step of this walk
    g.batch = 64u;
    g.min_alive = 0u;
    bool found = false;
    [[maybe_unused]] const bool deferred = DEFER && p.materials_plain != nullptr && __builtin_amdgcn_readfirstlane((int)*p.materials_plain) != 0 &&
                                           r.ignore_type_material == MAT_NONE; // (per lane)
    while (g.alive != 0ull) {
        uint32_t solid_bit; // parked lanes: bit index of the solid voxel they left behind
        VRT_PROF_BEGIN(tp2);
        if constexpr (LDS) voxel_walk_park_lds_gfx950(w, s.inv_dir, bit_index, solid_bit, stride_x, stride_y, stride_z, word, lane_base, g, voxel_scale, local_t_max);
        else voxel_walk_park_gfx950(w, s.inv_dir, bit_index, solid_bit, stride_x, stride_y, stride_z, word, rsrc, g, voxel_scale, local_t_max);
        VRT_PROF_END(2, tp2);
        if (g.parked == 0ull) break; // every lane has left the brick (or the grid box)
        const bool parked = __builtin_amdgcn_inverse_ballot_w64(g.parked);
        bool resume = false;
        VRT_PROF_BEGIN(tp4);
        if (parked && deferred) {
            const uint32_t in = g.code & 3u;
            hit.index = (solid_bit - base) | kDeferredHit;
            const float t_offset = voxel_scale * 0.05f;
            hit.t += g.t_in * voxel_scale - t_offset; // t_value of the step into this voxel (comp:442), 0 for the first
            hit_axis = (in == 3u) ? axis_in : (int)in;
            found = true;
        } else if (parked) {
            const uint32_t voxel_index = solid_bit - base;
            const uint32_t brick_index = by_cell ? p.brick_index[cell] : occ_slot; // comp:337, by the lanes that need it
            const uint32_t brick_material_index = start_is_slot ? brick_index * (uint32_t)(B * B * B)
                                                                : (((LDS && p.path_eager_start) ? eager_start : p.brick_start_index[brick_index]) & 0x7FFFFFFFu); // comp:422
            const uint32_t mi = p.material_index[brick_material_index + voxel_index];
            const vrt_material *m = p.materials + mi;
            const uint32_t mtype = m->type;
            const float mdata = m->type_data;
            const bool ignore_brick = (mtype == r.ignore_type_material) && (r.internal_reflection == mdata); // comp:427
            if (!ignore_brick) {
                const uint32_t in = g.code & 3u;
                hit.index = mi;
                const float t_offset = voxel_scale * 0.05f;
                hit.t += g.t_in * voxel_scale - t_offset; // t_value of the step into this voxel (comp:442), 0 for the first
                hit_axis = (in == 3u) ? axis_in : (int)in;
                found = true;
            } else {
                // walk on: the lane has already taken the step out of the ignored voxel; comp:409 for that step
                resume = min3i(w.rx, w.ry, w.rz) >= 0 && (voxel_scale * g.t_out <= local_t_max);
            }
        }
        VRT_PROF_END(4, tp4);
        // every lane: the axis of its last step, for its first trip in the next call
        g.code = parked ? ((g.code >> 2) & 3u) << 4
                        : (__builtin_amdgcn_inverse_ballot_w64(g.out_x) ? 0u : (__builtin_amdgcn_inverse_ballot_w64(g.out_y) ? 1u : 2u)) << 4;
        if (resume) { // (an A-trip park left the lane's word in the other register set)
            if constexpr (LDS) word = reinterpret_cast<lds_u32 *>(0)[brick_lds_address(lane_base, bit_index) >> 2];
            else word = occ_words[bit_index >> 5];
        }
        asm("s_or_b64 %0, %0, %1" : "+s"(g.alive) : "s"(__builtin_amdgcn_ballot_w64(resume)) : "scc");
    }
    return found;
}

// (enum StatusMode: vrt_kernels.h)

// ---- skip to the box of occupied cells ---------------------------------------------------------------------------
// A ray that enters the grid OUTSIDE the bounding box of the occupied cells along some axis A (a camera above the terrain:
// A = y) walks through cells that are known to be empty until it has crossed A `need` times (need = cells to the box's near
// face).  The shader's walk is a three-way merge of three non-decreasing sequences — the side distances of x, y and z, each
// built by repeated addition of |1/dir| — with ties going to z, then y, then x (comp:345-372).  So the state after the
// need-th crossing of A is known without walking: T = A's side distance after need-1 additions is the distance of that
// crossing; every element of another axis B that precedes T in merge order (c < T, or c == T where B wins the tie) has been
// consumed; B's side distance is its first element that does not.  All additions are the walk's own, in the walk's order per
// axis, so every side distance, counter and index is bit for bit what the walk would hold: ~2 vector instructions per
// skipped step instead of a 13-instruction trip.  A lane whose B counter runs out first has left the box (a miss, as in the
// walk).  Afterwards the lane stands on the first cell inside the box's A range, entered through A at distance T.  One round per
// axis (x, y, z; the order does not matter: each round skips what is left in front of its axis' near face).
VRT_DI float next_below(float t) { // the largest float < t (t finite)
    const uint32_t b = __builtin_bit_cast(uint32_t, t);
    return __builtin_bit_cast(float, t > 0.0f ? b - 1u : (t < 0.0f ? b + 1u : 0x80000001u));
}
// t after `count` further additions of |inv| (per lane): 2 vector + 1 scalar instruction per step, four steps per back edge
#define VRT_SKIP_ADD_STEP                                        \
    "v_add_co_u32_e32 %[n], vcc, -1, %[n]\n\t" /* carry-out: the count was >= 1 */ \
    "s_and_b64 exec, exec, vcc\n\t"                              \
    "v_add_f32_e64 %[t], %[t], |%[d]|\n\t"
VRT_DI void skip_add_gfx950(float &t, float inv, int count) {
    unsigned long long save;
    asm volatile("s_mov_b64 %[save], exec\n\t"
                 "0:\n\t" VRT_SKIP_ADD_STEP VRT_SKIP_ADD_STEP VRT_SKIP_ADD_STEP VRT_SKIP_ADD_STEP
                 "s_cbranch_execnz 0b\n\t"
                 "s_mov_b64 exec, %[save]"
                 : [t] "+v"(t), [n] "+v"(count), [save] "=&s"(save)
                 : [d] "v"(inv)
                 : "vcc", "scc");
}
#undef VRT_SKIP_ADD_STEP
// Consume the elements c, c+|inv|, ... of another axis that are <= lim; returns how many.  The axis has `left` steps before
// the far face of the box: a lane that consumes more has left the box, and what it holds afterwards is never used — so the
// bound is tested once per four steps only (v_cmpx drops a lane from EXEC the moment its element is beyond lim): 3 vector
// instructions per step, one more and the branch per four.
#define VRT_SKIP_MERGE_STEP                                      \
    "v_cmpx_le_f32_e32 vcc, %[c], %[lim]\n\t"                    \
    "v_add_f32_e64 %[c], %[c], |%[d]|\n\t"                       \
    "v_add_u32_e32 %[n], 1, %[n]\n\t"
VRT_DI int skip_merge_gfx950(float &c, float inv, int left, float lim) {
    unsigned long long save;
    int n = 0;
    asm volatile("s_mov_b64 %[save], exec\n\t"
                 "0:\n\t" VRT_SKIP_MERGE_STEP VRT_SKIP_MERGE_STEP VRT_SKIP_MERGE_STEP VRT_SKIP_MERGE_STEP
                 "v_cmpx_ge_i32_e32 vcc, %[r], %[n]\n\t"
                 "s_cbranch_execnz 0b\n\t"
                 "s_mov_b64 exec, %[save]"
                 : [c] "+v"(c), [n] "+v"(n), [save] "=&s"(save)
                 : [d] "v"(inv), [lim] "v"(lim), [r] "v"(left)
                 : "vcc", "scc");
    return n;
}
#undef VRT_SKIP_MERGE_STEP
VRT_DI float &comp3(f3 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
VRT_DI float comp3(const f3 &v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
// The lane crosses axis A (compile time) `need` more times, the last time at distance t (= A's side distance after need-1
// additions): consume what the other two axes hold before t in merge order and move the walk state behind that crossing.
template <int A>
VRT_DI void skip_axis(Walk &w, const RaySetup &s, int need, float t, uint32_t &index, uint32_t stride_x, uint32_t stride_y, uint32_t stride_z, bool &more,
                      int &in_axis, float &t_in) {
    constexpr int B1 = (A + 1) % 3, B2 = (A + 2) % 3; // the other two axes; an axis wins a tie against A iff it comes later in x, y, z
    int *const r[3] = {&w.rx, &w.ry, &w.rz};
    const uint32_t stride[3] = {stride_x, stride_y, stride_z};
    const float t_strict = next_below(t); // c < t  <=>  c <= t_strict
    const int n1 = skip_merge_gfx950(comp3(w.side_dist, B1), comp3(s.inv_dir, B1), *r[B1], B1 > A ? t : t_strict);
    const int n2 = skip_merge_gfx950(comp3(w.side_dist, B2), comp3(s.inv_dir, B2), *r[B2], B2 > A ? t : t_strict);
    comp3(w.side_dist, A) = t + comp3(s.ray_delta(), A);
    *r[A] -= need;
    *r[B1] -= n1;
    *r[B2] -= n2;
    index += (uint32_t)need * stride[A] + (uint32_t)n1 * stride[B1] + (uint32_t)n2 * stride[B2];
    more = (*r[A] | *r[B1] | *r[B2]) >= 0; // a counter below zero: the far face of the box was crossed on the way
    in_axis = A;
    t_in = t;
}
// one round: bring axis A inside the box's range for the lanes that are in front of it
template <int A>
VRT_DI void skip_round(Walk &w, const RaySetup &s, int span, uint32_t &index, uint32_t stride_x, uint32_t stride_y, uint32_t stride_z, bool &more,
                       int &in_axis, float &t_in, bool &skipped) {
    const int step_a = A == 0 ? s.sx : (A == 1 ? s.sy : s.sz);
    const int r_a = A == 0 ? w.rx : (A == 1 ? w.ry : w.rz);
    // crossings still to go before the near face (steps left to the FAR face minus the box's extent); an axis the ray does not
    // move along holds the hang-guard budget instead and never needs any
    const int need = step_a != 0 ? r_a - span : 0;
    const bool want = more && need > 0;
    if (__builtin_amdgcn_ballot_w64(want) == 0ull) return;
    if (want) {
        float t = comp3(w.side_dist, A);
        skip_add_gfx950(t, comp3(s.inv_dir, A), need - 1);
        skip_axis<A>(w, s, need, t, index, stride_x, stride_y, stride_z, more, in_axis, t_in);
        skipped = true;
    }
}
VRT_DI bool skip_to_box(Walk &w, const RaySetup &s, int span_x, int span_y, int span_z, uint32_t &index, uint32_t stride_x, uint32_t stride_y,
                        uint32_t stride_z, bool &more, int &in_axis, float &t_in) {
    bool skipped = false;
    skip_round<0>(w, s, span_x, index, stride_x, stride_y, stride_z, more, in_axis, t_in, skipped);
    skip_round<1>(w, s, span_y, index, stride_x, stride_y, stride_z, more, in_axis, t_in, skipped);
    skip_round<2>(w, s, span_z, index, stride_x, stride_y, stride_z, more, in_axis, t_in, skipped);
    return skipped;
}

#ifdef VRT_DEV_VARIANTS
#define VRT_DEV_SECTION 6
#include "vrt_trace_kernels_dev.h"
#undef VRT_DEV_SECTION
#endif
// A wave-uniform float the optimiser may not move out of a loop (empty asm pinned to an SGPR).
VRT_DI float opaque_uniform(float v) {
    asm volatile("" : "+s"(v));
    return v;
}
VRT_DI f3 opaque_uniform3(const float (&v)[3]) { return mk3(opaque_uniform(v[0]), opaque_uniform(v[1]), opaque_uniform(v[2])); }
// comp:271-376.  t_min = 1e-5, t_max = +inf at every call site (comp:218,247).
// BATCH (used for frames with bounces, whose secondary rays are incoherent): a lane that reaches an
// occupied cell does not walk its brick at once but waits (__ballot) until p.brick_batch lanes are waiting or
// no lane is still moving, so the long voxel-level walk runs for many lanes per execution instead of a
// few (measured on the 2048^3 path-trace config: 3.8 lanes per execution unbatched; a threshold of 4 lanes
// gives +8.5 % there, larger thresholds stall the moving lanes and lose).  The per-lane
// sequence of operations is unchanged; only their interleaving across lanes differs.

// SCALAR_ENTRY: the grid-entry offset 0.0001 * scale (comp:287) formed where it is used (the several-samples kernel; the one-sample
// kernels measured 0.5 % slower with it and keep the compiler's placement)
template <int B, bool COUNT, int MODE, bool BATCH = false, bool SCALAR_ENTRY = false>
VRT_DI bool grid_hit(const TraceParams &p, const uint32_t *lds_filter, const Ray &r, Hit &hit, Cnt<COUNT> &c) {
    const float t_min = 0.00001f;
    const float t_max = __builtin_inff();
    VRT_COUNT(rays);
    VRT_PROF_BEGIN(tp3);
    RaySetup s;
    VRT_PROF_BEGIN(tp6);
    const bool slab_hit = grid_slab(p, r, t_min, t_max, s);
    VRT_PROF_END(6, tp6);
    if (!slab_hit) return false;

    const f3 g_min = mk3(p.grid.min_point_base_t[0], p.grid.min_point_base_t[1], p.grid.min_point_base_t[2]);
    const float g_scale = p.grid.max_point_scale[3];
    const int dx = (int)p.grid.dim_x, dy = (int)p.grid.dim_y, dz = (int)p.grid.dim_z;

    // (opaque: the product of a uniform is formed where it is used, from its scalar register — hoisted out of the sample loop it was the one
    // vector register the several-samples kernel spilled at seven waves per SIMD, round 6)
    float global_t_value = s.grid_t_min + 0.0001f * (SCALAR_ENTRY ? opaque_uniform(g_scale) : g_scale); // comp:287
    const f3 fposition = p.scale_pow2 ? (ray_at(r, global_t_value) - g_min) * p.inv_grid_scale : (ray_at(r, global_t_value) - g_min) / splat3(g_scale);
    Walk w;
    w.side_dist = initial_side_dist(mk3((float)s.sx, (float)s.sy, (float)s.sz), fposition, s.ray_delta());
    const int px = f2i_clamp(__builtin_floorf(fposition.x));
    const int py = f2i_clamp(__builtin_floorf(fposition.y));
    const int pz = f2i_clamp(__builtin_floorf(fposition.z));
    const int zero_budget = dx + dy + dz + 8;
    // The walk ends where the ray leaves the bounding box of the OCCUPIED cells on the far side of an axis (every cell beyond
    // is empty, and the ray cannot come back), not only at the grid's face: same hits, same misses, fewer trips -- sky rays of a
    // camera above the terrain and shadow rays towards the sun stop at the height of the highest brick.  The counting build
    // walks to the grid's face like the shader, so that its counters stay the reference algorithm's — unless asked
    // (count_box, vrt_config.enable_counters = 2) to count what the product kernel itself walks and requests.
    int lox = 0, loy = 0, loz = 0, hix = dx - 1, hiy = dy - 1, hiz = dz - 1;
    if (!COUNT || p.count_box) {
        if (p.cell_bounds) {
            lox = -p.cell_bounds[0], loy = -p.cell_bounds[1], loz = -p.cell_bounds[2];
            hix = p.cell_bounds[3], hiy = p.cell_bounds[4], hiz = p.cell_bounds[5];
        }
    }
    w.rx = steps_left_box(s.sx, px, lox, hix, zero_budget);
    w.ry = steps_left_box(s.sy, py, loy, hiy, zero_budget);
    w.rz = steps_left_box(s.sz, pz, loz, hiz, zero_budget);
    const int base_x = walk_base_box(s.sx, px, lox, hix), base_y = walk_base_box(s.sy, py, loy, hiy), base_z = walk_base_box(s.sz, pz, loz, hiz);
    w.t_value = 0;

    uint32_t word_index = ~0u; // comp:301
    uint32_t word_bits = 0;
    uint32_t block_index = ~0u;
    uint2 block_bits = make_uint2(0u, 0u);
    int axis = 3;
    [[maybe_unused]] int hit_axis = 0; // default kernel: the face of the voxel hit, turned into hit.normal / hit.point after the walk

    // `global_t_value <= t_max` (comp:316) with t_max = +inf only fails for a NaN t, and t only
    // changes when a brick is entered: test it there instead of on every step.
    if (!(global_t_value <= t_max)) return false;
    // comp:318, kept current by dda_step; meaningless (and unused) while the position is outside
    uint32_t grid_index = (uint32_t)px + (uint32_t)dx * ((uint32_t)pz + (uint32_t)dz * (uint32_t)py);
    const uint32_t stride_x = (uint32_t)s.sx, stride_y = (uint32_t)s.sy * (uint32_t)dx * (uint32_t)dz, stride_z = (uint32_t)s.sz * (uint32_t)dx;

    // stop: 0 keep walking, -1 voxel hit, -2 t became NaN; negative values end the loop through the
    // same integer test as the box exit
    int stop = 0;
    bool more = (unsigned)px < (unsigned)dx && (unsigned)py < (unsigned)dy && (unsigned)pz < (unsigned)dz && (w.rx | w.ry | w.rz) >= 0;
    // a ray that enters the grid in front of the box jumps to the box's near face (skip_to_box above); such a lane's first cell
    // was entered by a step through `axis` at crossed distance skip_t, not through the slab test
    [[maybe_unused]] bool skipped = false;
    [[maybe_unused]] float skip_t = 0.0f;
    if ((!COUNT || p.count_box) && p.cell_bounds && p.skip_to_box) {
        VRT_PROF_BEGIN(tp5);
        // (extents in unsigned arithmetic: with no cell occupied the bounds are the 0x80808080 sentinel, whose signed difference overflows;
        // `more` is false for every lane then and the skip does nothing)
        skipped = skip_to_box(w, s, (int)((uint32_t)hix - (uint32_t)lox), (int)((uint32_t)hiy - (uint32_t)loy), (int)((uint32_t)hiz - (uint32_t)loz), grid_index,
                              stride_x, stride_y, stride_z, more, axis, skip_t);
        w.t_value = skip_t;
        VRT_PROF_END(5, tp5);
    }

    auto cell_occupied = [&]() -> bool {
        VRT_COUNT(grid_steps);
        VRT_COUNT_WAVE(wave_grid_iters);
        if constexpr (MODE == kStatusLinearLds) {
            if constexpr (COUNT) {
                const uint32_t wi = grid_index >> 5;
                if (wi != word_index) {
                    word_index = wi;
                    c.status_loads++;
                }
            }
            return status_bit(lds_word0(grid_index >> 5), grid_index); // ds_read_b32 + v_bfe_u32
        } else if constexpr (MODE == kStatusLinearAlways) {
            if constexpr (COUNT) {
                const uint32_t wi = grid_index >> 5;
                if (wi != word_index) {
                    word_index = wi;
                    c.status_loads++;
                }
            }
            return status_bit(p.brick_status[grid_index >> 5], grid_index);
        } else if constexpr (MODE == kStatusLinear || MODE == kStatusLinearWide) {
            const uint32_t wi = grid_index >> 5;
            if (wi != word_index) { // comp:323-326
                word_bits = p.brick_status[wi];
                word_index = wi;
                VRT_COUNT(status_loads);
            }
            return (word_bits >> (grid_index & 31u)) & 1u;
        } else {
            if constexpr (COUNT) { // the algorithmic count follows the reference's word rule
                const uint32_t wi = grid_index >> 5;
                if (wi != word_index) {
                    word_index = wi;
                    c.status_loads++;
                }
            }
            const int cx = base_x - s.sx * w.rx, cy = base_y - s.sy * w.ry, cz = base_z - s.sz * w.rz;
            const uint32_t bi = (uint32_t)(cx >> 2) + p.nbx * ((uint32_t)(cz >> 2) + p.nbz * (uint32_t)(cy >> 2));
            if (bi != block_index) {
                block_index = bi;
                if constexpr (MODE == kStatusBlockedLds) {
                    const uint32_t fw = lds_filter[bi >> 5];
                    block_bits = ((fw >> (bi & 31u)) & 1u) ? p.status_blocks[bi] : make_uint2(0u, 0u);
                } else {
                    block_bits = p.status_blocks[bi];
                }
            }
            return bit64(block_bits, (uint32_t)((cx & 3) | ((cz & 3) << 2) | ((cy & 3) << 4)));
        }
    };
    // walk the brick of the cell reached with steps-left (rx,ry,rz), crossed-distance t and linear index `cell`
    auto enter_brick_at = [&](int rx, int ry, int rz, float t_cross, uint32_t cell, int &brick_axis) {
        const int cx = base_x - __mul24(s.sx, rx), cy = base_y - __mul24(s.sy, ry), cz = base_z - __mul24(s.sz, rz); // cell position
        const f3 brick_min = fma3(mk3((float)cx, (float)cy, (float)cz), splat3(g_scale), g_min);  // comp:331
        global_t_value = t_cross * g_scale + s.grid_t_min + 0.01f * g_scale;                     // comp:347 (deferred) + comp:332
        hit.t = global_t_value;
        const uint32_t brick_index = p.brick_index[cell]; // comp:337
        VRT_COUNT(bricks_entered);
        VRT_COUNT_WAVE(wave_brick_walks);
        bool found;
        if constexpr ((MODE == kStatusLinearAlways || MODE == kStatusLinearLds || MODE == kStatusBytes) && !COUNT) {
            found = brick_walk_gfx950<B, true, BATCH>(p, r, s, g_scale, brick_index, brick_min, hit, brick_axis, hit_axis, cell);
        } else {
            found = brick_walk<B, COUNT, MODE == kStatusLinear || MODE == kStatusLinearLds || MODE == kStatusLinearAhead>(
                p, r, s, g_scale, brick_index, brick_min, hit, brick_axis, c);
        }
        stop = found ? -1 : ((global_t_value <= t_max) ? 0 : -2);
    };
    auto enter_brick = [&]() { enter_brick_at(w.rx, w.ry, w.rz, w.t_value, grid_index, axis); };
    // brick_walk_gfx950 records a hit as distance + material + face; comp:433-436 from those, once the walk is over
    auto finish_hit = [&]() {
        if constexpr ((MODE == kStatusLinearAlways || MODE == kStatusLinearLds || MODE == kStatusBytes) && !COUNT) {
            if (stop == -1) {
                const float t_offset = (g_scale * (1.0f / (float)B)) * 0.05f;
                hit.normal = axis_normal(s, hit_axis);
                hit.point = ray_at(r, hit.t) + hit.normal * t_offset;
            }
        }
    };

    if constexpr (BATCH && MODE == kStatusLinearAlways && !COUNT) {
        // frames with bounces on the default kernel: the hand-written loop with parking (grid_walk_park_gfx950)
        const unsigned long long status_addr = (unsigned long long)p.brick_status;
        u32x4 rsrc;
        rsrc.x = (uint32_t)status_addr;
        rsrc.y = (uint32_t)(status_addr >> 32) | (4u << 16); // stride 4: one record per status word
        rsrc.z = p.status_words;
        rsrc.w = 0x00020000u;
        uint32_t word = p.brick_status[more ? (grid_index >> 5) : 0u];
        GridParkRegs g;
        g.alive = __builtin_amdgcn_ballot_w64(more);
        g.out_x = 0ull;
        g.out_y = 0ull;
        g.t_out = skip_t;
        g.code = (uint32_t)axis << 4; // 3: the first cell of the walk was entered through the slab test, not by a step
        g.batch = p.brick_batch;
        VRT_PROF_END(3, tp3);
        while (g.alive != 0ull) {
            uint32_t cell; // the occupied cell each parked lane stood on before its last step
            VRT_PROF_BEGIN(tp0);
            grid_walk_park_gfx950(w, s.inv_dir, grid_index, cell, stride_x, stride_y, stride_z, word, rsrc, g);
            VRT_PROF_END(0, tp0);
            if (g.parked == 0ull) break; // every lane has left the grid
            const bool parked = __builtin_amdgcn_inverse_ballot_w64(g.parked);
            bool resume = false;
            VRT_PROF_BEGIN(tp1);
            if (parked) {
                int a = (int)(g.code & 3u);
                const uint32_t out = (g.code >> 2) & 3u;
                // the counters as they were on the occupied cell: undo the decrement of the step out of it
                enter_brick_at(w.rx + (out == 0u ? 1 : 0), w.ry + (out == 1u ? 1 : 0), w.rz + (out == 2u ? 1 : 0), g.t_in, cell, a);
                resume = (stop == 0) && min3i(w.rx, w.ry, w.rz) >= 0;
            }
            VRT_PROF_END(1, tp1);
            // every lane: the axis of its last step, for its first trip in the next call
            g.code = parked ? ((g.code >> 2) & 3u) << 4
                            : (__builtin_amdgcn_inverse_ballot_w64(g.out_x) ? 0u : (__builtin_amdgcn_inverse_ballot_w64(g.out_y) ? 1u : 2u)) << 4;
            if (resume) word = p.brick_status[grid_index >> 5]; // (an A-trip park left the lane's word in the other register set)
            // (as asm: the compiler would do this on the vector unit and could not hand the result back to an SGPR operand)
            asm("s_or_b64 %0, %0, %1" : "+s"(g.alive) : "s"(__builtin_amdgcn_ballot_w64(resume)) : "scc");
        }
        finish_hit();
        return stop == -1;
    } else if constexpr (BATCH) {
        // lane state: 0 at a cell (test it), 1 waiting to walk a brick, 2 finished, 3 take the DDA step
        int state = more ? 0 : 2;
        while (__any(state != 2)) {
            if (state == 0) state = cell_occupied() ? 1 : 3;
            const unsigned long long waiting = __ballot(state == 1);
            if (waiting != 0ull) {
                const unsigned long long moving = __ballot(state == 3);
                if (moving == 0ull || (uint32_t)__popcll(waiting) >= p.brick_batch) {
                    if (state == 1) {
                        enter_brick();
                        state = (stop != 0) ? 2 : 3;
                    }
                }
            }
            if (state == 3) {
                dda_step<true>(w, s.ray_delta(), g_scale, axis, grid_index, stride_x, stride_y, stride_z);
                state = (min3i(w.rx, w.ry, w.rz) >= 0) ? 0 : 2;
            }
        }
        finish_hit();
        return stop == -1;
    } else if constexpr (MODE == kStatusLinearAhead) {
        // Software-pipelined walk.  The DDA step does not depend on the cell test, so it is taken first and
        // the status word of the NEXT cell is requested right away; the current cell is tested while that load
        // is in flight (its latency otherwise sits between every two steps).  If the current cell is occupied
        // its brick is walked with the pre-step state, rebuilt from the post-step state and the crossed axis.
        uint32_t word = more ? p.brick_status[grid_index >> 5] : 0u;
        while (more) {
            VRT_COUNT(grid_steps);
            VRT_COUNT_WAVE(wave_grid_iters);
            if constexpr (COUNT) {
                const uint32_t wi = grid_index >> 5;
                if (wi != word_index) {
                    word_index = wi;
                    c.status_loads++;
                }
            }
            const bool occupied = status_bit(word, grid_index);
            const float t_here = w.t_value; // crossed distance of the step INTO the current cell
            int axis_here = axis;
            dda_step<true>(w, s.ray_delta(), g_scale, axis, grid_index, stride_x, stride_y, stride_z);
            const bool inside = min3i(w.rx, w.ry, w.rz) >= 0;
            word = inside ? p.brick_status[grid_index >> 5] : 0u;
            if (occupied) {
                const int a = axis; // the step just taken left the current cell through this axis
                enter_brick_at(w.rx + (a == 0 ? 1 : 0), w.ry + (a == 1 ? 1 : 0), w.rz + (a == 2 ? 1 : 0), t_here,
                               grid_index - (a == 0 ? stride_x : (a == 1 ? stride_y : stride_z)), axis_here);
            }
            more = ((inside ? 0 : -1) | stop) >= 0;
        }
        return stop == -1;
    } else if constexpr ((MODE == kStatusLinearAlways || MODE == kStatusLinearLds || MODE == kStatusBytes) && !COUNT) {
        // The shipped default: grid_walk_gfx950 runs trips until some lane stands on an occupied cell (or all
        // lanes have left); the bricks are walked here, with the state from BEFORE the lane's last step rebuilt
        // from the post-step state and the crossed-axis lane masks, and the walk is resumed.
        const unsigned long long status_addr = (MODE == kStatusBytes) ? (unsigned long long)p.status_bytes : (unsigned long long)p.brick_status;
        u32x4 rsrc;
        rsrc.x = (uint32_t)status_addr;
        // words: stride 4, one record per status word.  bytes: a raw buffer (stride 0), num_records = bytes = cells
        rsrc.y = (uint32_t)(status_addr >> 32) | ((MODE == kStatusBytes) ? 0u : (4u << 16));
        rsrc.z = (MODE == kStatusBytes) ? p.status_cells : p.status_words;
        rsrc.w = 0x00020000u;
        // LDS variant: byte-address mask of the power-of-two LDS allocation holding the bitmap (trace_lds_bytes)
        [[maybe_unused]] const uint32_t lds_mask = (0xFFFFFFFFu >> __builtin_clz(p.status_words * 4u - 1u)) & ~3u;
        uint32_t word;
        if constexpr (MODE == kStatusLinearLds) word = lds_word0(more ? (grid_index >> 5) : 0u);
        else if constexpr (MODE == kStatusBytes) word = p.status_bytes[more ? grid_index : 0u];
        else word = p.brick_status[more ? (grid_index >> 5) : 0u];
        GridWalkRegs g;
        g.alive = __builtin_amdgcn_ballot_w64(more);
        g.out_x = __builtin_amdgcn_ballot_w64(skipped && axis == 0);
        g.out_y = __builtin_amdgcn_ballot_w64(skipped && axis == 1);
        g.t_out = skip_t;
        bool first = true; // wave-uniform
        VRT_PROF_END(3, tp3);
        while (g.alive != 0ull) {
            uint32_t cell; // the cell each lane stood on before its last step
            VRT_PROF_BEGIN(tp0);
            if constexpr (MODE == kStatusLinearLds) grid_walk_lds_gfx950(w, s.inv_dir, grid_index, cell, stride_x, stride_y, stride_z, word, lds_mask, g);
            else if constexpr (MODE == kStatusBytes) grid_walk_bytes_gfx950(w, s.inv_dir, grid_index, cell, stride_x, stride_y, stride_z, word, rsrc, g);
            else grid_walk_gfx950(w, s.inv_dir, grid_index, cell, stride_x, stride_y, stride_z, word, rsrc, g);
            VRT_PROF_END(0, tp0);
            if (g.occ == 0ull) break; // every lane has left the grid
            if (__builtin_amdgcn_inverse_ballot_w64(g.occ)) {
                // axis 3: the first cell of the walk was entered through the slab test, not by a step
                int a = (first && g.stub == 0u && !skipped) ? 3
                                                : (__builtin_amdgcn_inverse_ballot_w64(g.in_x) ? 0 : (__builtin_amdgcn_inverse_ballot_w64(g.in_y) ? 1 : 2));
                const bool out_x = __builtin_amdgcn_inverse_ballot_w64(g.out_x), out_y = __builtin_amdgcn_inverse_ballot_w64(g.out_y);
                VRT_PROF_BEGIN(tp1);
                enter_brick_at(w.rx + (out_x ? 1 : 0), w.ry + (out_y ? 1 : 0), w.rz + ((out_x | out_y) ? 0 : 1), g.t_in, cell, a);
                VRT_PROF_END(1, tp1);
            }
            // (as asm: the compiler would do this on the vector unit and could not hand the result back to an SGPR operand)
            asm("s_andn2_b64 %0, %0, %1" : "+s"(g.alive) : "s"(__builtin_amdgcn_ballot_w64(stop != 0)) : "scc");
            first = false;
        }
        finish_hit();
        return stop == -1;
    } else {
        while (more) { // single-exit loop, see brick_walk
            if (cell_occupied()) enter_brick();
            dda_step<true>(w, s.ray_delta(), g_scale, axis, grid_index, stride_x, stride_y, stride_z);
            more = (min3i(w.rx, w.ry, w.rz) | stop) >= 0;
        }
        return stop == -1;
    }
}

// ---- scatter functions (comp:539-596) -------------------------------------
VRT_DI bool scatter_lambertian(const Hit &hit, Ray &scattered) {
    const f3 rv = rand_vec3_range(hit.point.x + hit.point.z, hit.point.y + hit.point.z, -0.4f, 0.4f);
    scattered = create_ray(hit.point, normalize3(hit.normal + rv));
    return true;
}
VRT_DI bool scatter_metal(float fuzz, const Ray &r_in, const Hit &hit, Ray &scattered) {
    const f3 reflected = reflect3(r_in.direction, hit.normal);
    const f3 rv = rand_vec3_range(hit.point.x + hit.point.z, hit.point.y + hit.point.z, -fuzz, fuzz);
    scattered = create_ray(hit.point, reflected + rv);
    return dot3(scattered.direction, hit.normal) > 0;
}
VRT_DI bool transmission_direction(float n1, float n2, f3 ray_dir, f3 normal, f3 &refrac_dir) {
    const float eta = n1 / n2;
    const float c1 = -dot3(ray_dir, normal);
    const float w = eta * c1;
    const float c2m = (w - eta) * (w + eta);
    if (c2m < -1.0f) return false;
    refrac_dir = fma3(splat3(eta), ray_dir, normal * (w - __builtin_sqrtf(1.0f + c2m)));
    return true;
}
VRT_DI bool scatter_dielectric(float ir, const Ray &r_in, const Hit &hit, Ray &scattered) {
    const f3 rv = rand_vec3_range(hit.point.x + hit.point.z, hit.point.y + hit.point.z, -0.05f, 0.05f);
    const f3 normal = normalize3(hit.normal + rv);
    f3 direction = mk3(0, 0, 0);
    const bool should_refract = transmission_direction(ir, r_in.internal_reflection, r_in.direction, normal, direction);
    if (should_refract && rand_3(hit.point) > 0.5f) {
        scattered = create_ray(hit.point, direction);
        scattered.ignore_type_material = MAT_DIELECTRIC;
        scattered.internal_reflection = ir;
    } else {
        direction = reflect3(r_in.direction, normal);
        scattered = create_ray(hit.point, direction);
    }
    return true;
}

// comp:203-265 when push_constant.max_bounce <= 1 (Camera.Config.max_bounce = 0: "only primary
// ray", Camera.zig:74).  The bounce loop then runs at most once, so the scatter functions — whose only
// products are the next ray and the continue flag — have no observable effect and are not evaluated.
template <int B, bool COUNT, int MODE, bool SCALAR_ENTRY = false>
VRT_DI f3 ray_color_single(const TraceParams &p, const PushConstants &pc, const uint32_t *lds_filter, const Ray &ray, Cnt<COUNT> &c) {
    const bool sun_enabled = pc.sun.enabled > 0;
    const f3 sun_color = mk3(pc.sun.color[0], pc.sun.color[1], pc.sun.color[2]);
    f3 color = mk3(0, 0, 0);
    int loop_count = 0;
    Hit hit;
    if (pc.cam.max_bounce > 0 && grid_hit<B, COUNT, MODE, false, SCALAR_ENTRY>(p, lds_filter, ray, hit, c)) {
        // (the material record is read AFTER the shadow ray: only its index stays live across the second walk — three registers
        // fewer at the kernel's point of highest pressure; the record sits in the scalar / L1 cache)
        const uint32_t material = hit.index;
        bool lit = true;
        if (sun_enabled) {
            const f3 sun_position = mk3(pc.sun.position[0], pc.sun.position[1], pc.sun.position[2]);
            const f3 rv = rand_vec3_range(ray.direction.x + ray.direction.z, ray.direction.y + ray.direction.z, -pc.sun.radius,
                                          pc.sun.radius);
            const Ray shadow_ray = create_ray(hit.point, (sun_position + rv) - hit.point);
            Hit shadow_hit;
            lit = !grid_hit<B, COUNT, MODE, false, SCALAR_ENTRY>(p, lds_filter, shadow_ray, shadow_hit, c);
        }
        const vrt_material *m = p.materials + material;
        const uint32_t mtype = m->type;
        const f3 attenuation = mk3(m->albedo_r, m->albedo_g, m->albedo_b);
        loop_count = (mtype <= MAT_DIELECTRIC) ? 1 : 0; // unknown type: loop_count -= 1 (comp:235-238)
        if (sun_enabled) {
            if (lit) color = color + attenuation * sun_color;
        } else {
            color = color + attenuation;
        }
    }
    if (loop_count == 0) {
        const float t = 0.5f * (ray.direction.y + 1.0f);
        const f3 bg = fma3(splat3(1.0f - t), splat3(1.0f), mk3(0.5f, 0.7f, 1.0f) * t);
        color = color + bg * (sun_enabled ? sun_color : splat3(1.0f));
    }
    return color / (color + splat3(1.0f));
}

// comp:203-265
template <int B, bool COUNT, int MODE>
VRT_DI f3 ray_color(const TraceParams &p, const PushConstants &pc, const uint32_t *lds_filter, Ray current_ray, Cnt<COUNT> &c) {
    const bool sun_enabled = pc.sun.enabled > 0;
    const f3 sun_color = mk3(pc.sun.color[0], pc.sun.color[1], pc.sun.color[2]);
    const f3 sun_position = mk3(pc.sun.position[0], pc.sun.position[1], pc.sun.position[2]);
    const int max_bounce = pc.cam.max_bounce;
    Hit hit;
    hit.point = mk3(0, 0, 0);
    hit.normal = mk3(0, 0, 0);
    hit.t = 0;
    hit.index = 0;
    int loop_count = 0;
    f3 color = mk3(0, 0, 0);

    while (loop_count < max_bounce && grid_hit<B, COUNT, MODE, true, !COUNT>(p, lds_filter, current_ray, hit, c)) {
        loop_count += 1;
        Ray scattered = current_ray;
        bool result = false;
        const vrt_material *m = p.materials + hit.index;
        const uint32_t mtype = m->type;
        const f3 attenuation = mk3(m->albedo_r, m->albedo_g, m->albedo_b);
        const float mdata = m->type_data;
        switch (mtype) {
            case MAT_LAMBERTIAN: result = scatter_lambertian(hit, scattered); break;
            case MAT_METAL: result = scatter_metal(mdata, current_ray, hit, scattered); break;
            case MAT_DIELECTRIC: result = scatter_dielectric(mdata, current_ray, hit, scattered); break;
            default:
                loop_count -= 1;
                result = false;
                break;
        }
        if (sun_enabled) {
            const f3 rv = rand_vec3_range(current_ray.direction.x + current_ray.direction.z,
                                          current_ray.direction.y + current_ray.direction.z, -pc.sun.radius, pc.sun.radius);
            const f3 sun_sample_position = sun_position + rv;
            const f3 shadow_ray_dir = sun_sample_position - hit.point;
            // CreateShadowRay, comp:186-190: sun_enabled > 0 here, so the ignore type is MAT_NONE
            Ray shadow_ray = create_ray(hit.point, shadow_ray_dir);
            Hit shadow_hit;
            if (!grid_hit<B, COUNT, MODE, true, !COUNT>(p, lds_filter, shadow_ray, shadow_hit, c)) {
                color = color + attenuation * sun_color;
            }
        } else {
            color = color + attenuation;
        }
        if (!result) break;
        current_ray = scattered;
    }
    if (loop_count == 0) {
        // BackgroundColor, comp:197-201
        const float t = 0.5f * (current_ray.direction.y + 1.0f);
        const f3 bg = fma3(splat3(1.0f - t), splat3(1.0f), mk3(0.5f, 0.7f, 1.0f) * t);
        color = color + bg * (sun_enabled ? sun_color : splat3(1.0f));
    }
    return color / (color + splat3(1.0f));
}

VRT_DI uint32_t unorm8(float c) {
    c = (c > 0.0f) ? c : 0.0f; // NaN -> 0
    c = (c > 1.0f) ? 1.0f : c;
    return (uint32_t)__builtin_rintf(c * 255.0f);
}

// Workgroup -> image tile.  Block b executes on XCD b % 8; give XCD k the k-th
// contiguous slice of this context's tile list.
VRT_DI uint32_t xcd_slice_index(uint32_t b, uint32_t n) {
    const uint32_t q = n >> 3, rem = n & 7u;
    const uint32_t xcd = b & 7u, i = b >> 3;
    return (xcd < rem) ? xcd * (q + 1u) + i : rem * (q + 1u) + (xcd - rem) * q + i;
}

// comp:153-178

// This lane's index in its wave from the hardware, by an instruction the optimiser may neither hoist nor merge with an earlier
// copy: what is derived from it (the pixel's place in its tile, its coordinates, its address) is formed again where it is needed
// instead of staying in registers across the traversal.  (round 6: the several-samples kernel at six waves per SIMD kept eleven such
// values in scratch — the pixel's coordinates as floats, the jitter's operands, the target offset.)
VRT_DI uint32_t fresh_lane() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// SHADE: 0 general bounce loop; 1 max_bounce <= 1 (ray_color_single); 2 the same with one sample per
// pixel (no accumulator kept live across the traversal)
template <int B, bool COUNT, int MODE, int MIN_WAVES, int SHADE, int BLOCK = 256>
__global__ __launch_bounds__(BLOCK, MIN_WAVES) void vrt_trace_kernel(const TraceParams p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_filter[];
    if constexpr (MODE == kStatusLinearLds) {
        // stage the brick-status bitmap (binding 3) in LDS: 16 bytes per lane per trip
        const uint32_t nvec = (p.status_words + 3u) >> 2;
        const uint4 *src = reinterpret_cast<const uint4 *>(p.brick_status);
        uint4 *dst = reinterpret_cast<uint4 *>(lds_filter);
        for (uint32_t i = threadIdx.x; i < nvec; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
    }
    if constexpr (MODE == kStatusBlockedLds) {
        // stage the block filter (1 bit per 4x4x4 block of cells) once per workgroup
        const uint32_t nwords = (p.nbx * p.nby * p.nbz + 31u) >> 5;
        const uint32_t *src = reinterpret_cast<const uint32_t *>(p.status_blocks + (size_t)p.nbx * p.nby * p.nbz);
        for (uint32_t i = threadIdx.x; i < nwords; i += blockDim.x) lds_filter[i] = src[i];
        __syncthreads();
    }
    // One wave = one 8x8 pixel block.  wave_groups: the workgroup IS one wave (64 threads), so the
    // hardware dispatcher hands 8x8 blocks to whichever SIMD frees a slot (dynamic load balance at
    // wave granularity); otherwise a 256-thread workgroup covers one 16x16 tile with four waves.
    // BLOCK 512: two 16x16 tiles per workgroup share one LDS copy of the status bitmap
    // p.split_all = s (small frames, reverse raster only): 2^s workgroups per tile, each renders 8 >> s rows of every 8x8 block on
    // 64 >> s lanes per wave — a frame with fewer waves than the GPU has SIMDs lasts as long as its slowest wave, and a wave walks the
    // bricks its lanes meet one after the other: fewer lanes, a shorter chain
    const uint32_t unit = p.wave_groups ? (blockIdx.x >> 2) : (BLOCK == 512 ? blockIdx.x * 2u + (threadIdx.x >> 8) : (blockIdx.x >> p.split_all));
    // (SHADE 1: the wave's number in a scalar register — with fresh_lane() below nothing then keeps threadIdx.x alive)
    const uint32_t wave = p.wave_groups ? (blockIdx.x & 3u)
                                        : ((SHADE <= 1 && !COUNT) ? ((uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) & 3u) : ((threadIdx.x >> 6) & 3u));
    if (BLOCK == 512 && unit >= p.owned_tiles) return; // odd tile count: the last workgroup's second half is idle
    uint32_t owned;
    uint32_t split = 0u, half = 0u; // split: this workgroup renders one half of the tile (rows 4*half .. 4*half+3 of each 8x8 block)
    if (p.tile_order == 5u) {
        // cost-feedback schedule: tiles sorted by the time they took last frame, heaviest first, so the
        // kernel's tail is made of cheap tiles (longest-processing-time-first list scheduling).  Launch
        // order also spreads consecutive tiles over XCDs (block b runs on XCD b % 8).
        // (stored XCD-major: workgroup b runs on XCD b % 8, and the eight XCDs read disjoint lines of the list)
        // An entry with bit 31 renders HALF of its tile (bit 30 says which) on 32 lanes per wave: the schedule kernel
        // splits the tiles whose slowest wave would otherwise outlast the rest of the frame (the lanes of such a wave
        // walk their bricks one after the other; half the lanes, fewer separate walks: 88 -> 70 us for the slowest
        // tile of view V1).  The list has p.sched_extra spare entries for the second halves; unused ones are ~0.
        const uint32_t entry = p.tile_schedule[(unit & 7u) * ((p.owned_tiles + p.sched_extra + 7u) >> 3) + (unit >> 3)];
        if (entry == 0xFFFFFFFFu) return; // a spare entry (uniform over the workgroup)
        owned = entry & 0x3FFFFFFFu;
        split = entry >> 31;
        half = (entry >> 30) & 1u;
    } else if (p.tile_order == 6u) {
        // raster order; consecutive tiles go to consecutive XCDs: every XCD samples the whole image
        // (a contiguous band per XCD measured 22-28 % slower on the headline frame: sky bands idle)
        owned = unit;
    } else if (p.tile_order == 3u) {
        // reverse raster, consecutive tiles on consecutive XCDs: every XCD samples the whole image, and
        // the rows that usually hold the ground (long rays) start first so that sky tiles fill the tail
        owned = p.owned_tiles - 1u - unit;
        split = p.split_all;
        half = blockIdx.x & ((1u << p.split_all) - 1u);
    } else if (p.tile_order == 4u) {
        // strided permutation: consecutive launches sample the whole image (stride coprime to the count)
        owned = (uint32_t)(((unsigned long long)unit * p.tile_stride) % p.owned_tiles);
    } else if (p.tile_order == 2u) {
        // XCD k gets a band of tile columns: slice index runs column-major over the tile grid
        const uint32_t cm = xcd_slice_index(unit, p.owned_tiles);
        const uint32_t col = cm / p.tiles_y, row = cm % p.tiles_y;
        owned = row * p.tiles_x + col;
    } else {
        owned = xcd_slice_index(unit, p.owned_tiles);
    }
    const uint32_t tile = p.own_period ? (owned / p.own_count) * p.own_period + p.own_slots[owned % p.own_count] : owned * p.shard_count + p.shard_rank;
    const uint32_t tile_x = tile % p.tiles_x, tile_y = tile / p.tiles_x;
    // lane -> pixel: wave w of the tile covers the 8x8 quadrant (w&1, w>>1)
    const uint32_t lane = (SHADE <= 1 && !COUNT) ? fresh_lane() : (threadIdx.x & 63u);
    const PushConstants &pc = p.pcs[blockIdx.y]; // frame blockIdx.y of this launch (kernarg segment, scalar loads)
    // A half-tile workgroup of a frame with TWO samples per pixel gives its idle lanes the second sample (round 4): lanes 0-31 trace
    // sample 0 of the wave's 32 pixels, lanes 32-63 sample 1 of the same pixels, and lane l adds lane l + 32's colour to its own —
    // (0 + s0) + s1, the sample loop's own sum (comp:173) — before the tone-map.  The slowest waves of a bounce frame (the frame lasts
    // as long as they do: the reference app's run, tools/experiments/timeline.py) then have half the GridHits to go through one after the other.
    // (a quarter- or eighth-tile workgroup: 16 or 8 pixels per wave, twice as many lanes at work)
    const bool dual = SHADE != 2 && !COUNT && split != 0u && !p.packed_rgb && pc.cam.samples_per_pixel == 2; // (uniform over the workgroup)
    const uint32_t pixels = 64u >> split; // pixels of its 8x8 block this wave renders
    const uint32_t rows = 8u >> split; // rows of its 8x8 block this wave renders (split: 4 or 2), from row `half * rows`
    // lane -> place in the tile (everything but the lane is uniform over the wave)
    auto place = [&](uint32_t ln, uint32_t &ix, uint32_t &iy) {
        const uint32_t pl = dual ? (ln & (pixels - 1u)) : ln;
        ix = (wave & 1u) * 8u + (pl & 7u);
        iy = (wave >> 1) * 8u + half * rows + ((pl >> 3) & (rows - 1u));
    };
    uint32_t in_x, in_y;
    place(lane, in_x, in_y);
    const uint32_t px = tile_x * kTileW + in_x;
    const uint32_t py = tile_y * kTileH + in_y;

#ifdef VRT_DEV_PROFILE
    if (threadIdx.x < 8) vrt_prof[threadIdx.x] = 0ull;
    __syncthreads();
#endif
    VRT_PROF_BEGIN(tp7);
    const unsigned long long t_begin = (p.tile_order == 5u) ? __builtin_readcyclecounter() : 0ull;
    const unsigned long long wall_begin = p.wave_timeline ? wall_clock64() : 0ull;
    Cnt<COUNT> c;
    const bool inside = (px < p.width) && (py < p.height) && lane < (dual ? 2u * pixels : pixels); // comp:155-159
    uint32_t rgba = 0u; // this lane's pixel (0 outside the image), also needed after the branch by the RGB shard store
    if (inside) {
        f3 color = mk3(0, 0, 0);
        const int spp = (SHADE == 2) ? 1 : pc.cam.samples_per_pixel;
        const float x0 = (float)px, y0 = (float)py;
        // CameraGetRay operands, comp:474-477
        const f3 horizontal = mk3(pc.cam.horizontal[0], pc.cam.horizontal[1], pc.cam.horizontal[2]);
        const f3 vertical = mk3(pc.cam.vertical[0], pc.cam.vertical[1], pc.cam.vertical[2]);
        const f3 llc = mk3(pc.cam.lower_left_corner[0], pc.cam.lower_left_corner[1], pc.cam.lower_left_corner[2]);
        const f3 origin = mk3(pc.cam.origin[0], pc.cam.origin[1], pc.cam.origin[2]);
        if constexpr (SHADE == 2) {
            // sample 0 is un-jittered: hash12(0) = 0 (comp:167-170)
            const float u = (x0 + 0.0f) / (float)(pc.cam.image_width - 1u);
            const float v = (y0 + 0.0f) / (float)(pc.cam.image_height - 1u);
            const f3 ray_dir = fma3(horizontal, splat3(u), llc) + fma3(splat3(v), vertical, -origin);
            color = mk3(0, 0, 0) + ray_color_single<B, COUNT, MODE>(p, pc, lds_filter, create_ray(origin, ray_dir), c);
        } else {
            const int sample_begin = dual ? (int)(lane >= pixels ? 1u : 0u) : 0, sample_end = dual ? sample_begin + 1 : spp;
            for (int sample_i = sample_begin; sample_i < sample_end; sample_i++) {
                // Re-derive the camera vectors from their SGPRs in every trip: a VALU op takes a single scalar
                // operand, so the compiler copies them to VGPRs — hoisted out of this loop, twelve copies would
                // stay live across the whole traversal and cost a wave per SIMD.
                const f3 horizontal = opaque_uniform3(pc.cam.horizontal);
                const f3 vertical = opaque_uniform3(pc.cam.vertical);
                const f3 llc = opaque_uniform3(pc.cam.lower_left_corner);
                const f3 origin = opaque_uniform3(pc.cam.origin);
                const float flag = (sample_i > 0) ? 1.0f : 0.0f;
                float x = x0, y = y0;
                if constexpr ((SHADE <= 1 && !COUNT)) { // (the pixel's coordinates formed again from the lane in every trip: not kept across the traversal)
                    uint32_t sx, sy;
                    place(fresh_lane(), sx, sy);
                    x = (float)(tile_x * kTileW + sx), y = (float)(tile_y * kTileH + sy);
                }
                const float noise_x = hash_12_jitter(x + (float)sample_i, y, flag);
                const float u = (x + noise_x) / (float)(pc.cam.image_width - 1u);
                const float noise_y = hash_12_jitter(x, y + (float)sample_i, flag);
                const float v = (y + noise_y) / (float)(pc.cam.image_height - 1u);
                const f3 ray_dir = fma3(horizontal, splat3(u), llc) + fma3(splat3(v), vertical, -origin);
                if constexpr (SHADE == 1) color = color + ray_color_single<B, COUNT, MODE, !COUNT>(p, pc, lds_filter, create_ray(origin, ray_dir), c);
                else color = color + ray_color<B, COUNT, MODE>(p, pc, lds_filter, create_ray(origin, ray_dir), c);
            }
        }
        bool writer = true;
        if constexpr (SHADE != 2 && !COUNT) {
            if (dual) { // (both lanes of a pixel are inside the image or neither is: the source lane is active)
                const uint32_t lane = (SHADE <= 1 && !COUNT) ? fresh_lane() : (threadIdx.x & 63u);
                const int from = (int)((lane + pixels) & 63u);
                const f3 second = mk3(__shfl(color.x, from, 64), __shfl(color.y, from, 64), __shfl(color.z, from, 64));
                color = color + second;
                writer = lane < pixels;
            }
        }
        const float fspp = (float)spp;
        color = mk3(__builtin_sqrtf(color.x / fspp), __builtin_sqrtf(color.y / fspp), __builtin_sqrtf(color.z / fspp));

        size_t o;
        uint32_t ox = in_x, oy = in_y;
        if constexpr ((SHADE <= 1 && !COUNT)) place(fresh_lane(), ox, oy);
        if (p.shard_count > 1u || p.packed_tiles) {
            o = (size_t)owned * (kTileW * kTileH) + oy * kTileW + ox; // packed tile-major shard
        } else {
            o = (size_t)(tile_y * kTileH + oy) * p.width + (tile_x * kTileW + ox); // row-major frame
        }
        rgba = unorm8(color.x) | (unorm8(color.y) << 8) | (unorm8(color.z) << 16) | (255u << 24);
        if (writer) {
            if (!p.packed_rgb) reinterpret_cast<uint32_t *>(p.target_rgba8 + (size_t)blockIdx.y * p.batch_target_stride)[o] = rgba;
            if (p.target_rgba32f) {
                reinterpret_cast<float4 *>(p.target_rgba32f)[o] = make_float4(color.x, color.y, color.z, 1.0f);
            }
        }
    }
    if (p.packed_rgb) {
        // RGB shard (multi-GPU pipeline): 16x16 tiles of 3-byte pixels, 768 bytes per tile.  The eight lanes of a row of
        // this wave's 8x8 block hold 24 consecutive bytes = 6 dwords; lane k < 6 of the row assembles dword k from the
        // two pixels it spans (bytes 4k .. 4k+3; pixel = byte / 3) and stores it.
        const uint32_t lane = (SHADE <= 1 && !COUNT) ? fresh_lane() : (threadIdx.x & 63u);
        uint32_t in_x, in_y;
        place(lane, in_x, in_y);
        const uint32_t k = lane & 7u;
        const uint32_t first = k + (k >= 3u ? 1u : 0u); // = 4k / 3 for k < 6
        const int src = (int)((lane & ~7u) + first);
        const uint32_t lo = (uint32_t)__shfl((int)rgba, src, 64), hi = (uint32_t)__shfl((int)rgba, src + 1, 64);
        const uint32_t m = k % 3u;
        const uint32_t dword = (m == 0u) ? ((lo & 0xFFFFFFu) | (hi << 24)) : ((m == 1u) ? (((lo >> 8) & 0xFFFFu) | (hi << 16)) : (((lo >> 16) & 0xFFu) | (hi << 8)));
        if (k < 6u && lane < (64u >> split)) // (a half-tile workgroup: the idle half of the wave holds no pixels)
            reinterpret_cast<uint32_t *>(p.target_rgba8 + (size_t)blockIdx.y * p.batch_target_stride)[(size_t)owned * 192u + in_y * 12u + (in_x >> 3) * 6u + k] = dword;
    }
    const uint32_t lane_end = (SHADE <= 1 && !COUNT) ? fresh_lane() : lane;
#ifdef VRT_DEV_PROFILE
    VRT_PROF_END(7, tp7);
    __syncthreads();
    if (p.wave_timeline && threadIdx.x < 8) p.wave_timeline[(size_t)blockIdx.x * 8 + threadIdx.x] = vrt_prof[threadIdx.x];
#else
    if (p.wave_timeline && lane_end == 0) {
        const size_t w_id = (size_t)blockIdx.x * (blockDim.x >> 6) + (((SHADE <= 1 && !COUNT) && BLOCK != 512) ? (p.wave_groups ? 0u : wave) : (threadIdx.x >> 6));
        p.wave_timeline[2 * w_id] = wall_begin;
        p.wave_timeline[2 * w_id + 1] = wall_clock64();
    }
#endif
    if (p.tile_order == 5u) {
        // the wave's cycles, one plain store per wave into its own slot: the tile's cost for the next schedule.  (An
        // atomic add per wave into one word per tile measured 4.5 % of the kernel: the wave's slot is held until the
        // atomic is acknowledged.)
        const unsigned long long dt = __builtin_readcyclecounter() - t_begin;
        if (lane_end == 0) p.tile_cost[((size_t)half * p.owned_tiles + owned) * 4u + wave] = (uint32_t)(dt >> 6);
    }
    if constexpr (COUNT) {
        // wave-level reduction, then one atomic per wave per counter
        unsigned long long v[9] = {c.rays, c.status_loads, c.bricks_entered, c.voxel_steps, c.hits, c.grid_steps,
                                   c.wave_grid_iters, c.wave_brick_walks, c.wave_voxel_iters};
#pragma unroll
        for (int k = 0; k < 9; k++) {
            unsigned long long s = v[k];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            v[k] = s;
        }
        if (lane_end == 0) {
            atomicAdd(&p.counters->rays, v[0]);
            atomicAdd(&p.counters->status_loads, v[1]);
            atomicAdd(&p.counters->bricks_entered, v[2]);
            atomicAdd(&p.counters->voxel_steps, v[3]);
            atomicAdd(&p.counters->hits, v[4]);
            atomicAdd(&p.counters->grid_steps, v[5]);
            atomicAdd(&p.counters->wave_grid_iters, v[6]);
            atomicAdd(&p.counters->wave_brick_walks, v[7]);
            atomicAdd(&p.counters->wave_voxel_iters, v[8]);
        }
    }
}

} // namespace vrt
