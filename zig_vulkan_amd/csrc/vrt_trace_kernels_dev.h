// vrt_trace_kernels_dev.h — walk loops that lost their A/B measurement: kept for `make dev` (-DVRT_DEV_VARIANTS), where the
// parity tests still run them, and out of the product header.  Not a header of its own: vrt_trace_kernels.h includes it once per
// section, at the place in the file where the section's macros (VRT_WALK_ASM, VRT_PARK_WALK_ASM_T, ...) are defined.
#ifndef VRT_DEV_SECTION
#error "included by vrt_trace_kernels.h only"
#endif
#if VRT_DEV_SECTION == 1
#define VRT_LOAD_LDS_A(IDX, IDXN, WORD, WORDN)                                       \
    "s_mov_b64 %[cz], exec\n\t"                                          \
    "v_lshrrev_b32_e32 %[t2], 3, %[" IDXN "]\n\t"                         \
    "v_and_b32_e32 %[t2], %[rsrc], %[t2]\n\t"                             \
    "ds_read_b32 %[" WORDN "], %[t2]\n\t"
#define VRT_LOAD_LDS_B(IDX, IDXN, WORD, WORDN)                                       \
    "s_mov_b64 exec, %[cz]\n\t"                                          \
    "s_waitcnt lgkmcnt(1)\n\t"
#endif
#if VRT_DEV_SECTION == 2
// brick level with the status bitmap in LDS; `rsrc` is the byte-address mask (allocation size - 4)
VRT_DI void grid_walk_lds_gfx950(Walk &w, const f3 &inv_dir, uint32_t &index, uint32_t &cell, uint32_t stride_x, uint32_t stride_y, uint32_t stride_z,
                                 uint32_t &word, uint32_t rsrc, GridWalkRegs &g) {
    unsigned long long mxya, mxyb, ex, by, cz, save;
    float t0, t1, t2;
    uint32_t wordb;
    asm volatile(VRT_WALK_ASM(VRT_NO_LIMIT, VRT_LOAD_LDS, VRT_TEST_BIT, VRT_WAIT_LDS) : VRT_WALK_OUTPUTS : VRT_WALK_INPUTS : "vcc", "scc");
}

#endif
#if VRT_DEV_SECTION == 3
// ---- the counter-free dilated loop with the DDA TWO cells ahead of the test (vrt_path_kernel<..., DIL 4>, round 3) -----------------
// The trip of the loops above lasts as long as the round trip of its one request plus the instructions between a word's arrival and
// the next request: the word of the cell entered in trip k is tested in trip k + 1.  Here it is tested in trip k + 2 —
// two requests in flight per lane — and the round trip leaves the chain.  Three register sets (x, y, z: cell, word, crossed
// distance, crossed-axis masks, carry, keep mask) are used in turn, so nothing rotates inside the loop:
//   trip k (sets K = k % 3, N = next, J = previous):  step c_k -> c_k+1 (idxN), request word(c_k+1) -> wN by the lanes that enter
//   another half-block (the others, kpN, take it from wK one trip later: a register with a request in flight cannot be read),
//   advance the side distance; wait until at most two requests are outstanding (word(c_k-1) has arrived); wK <- wJ for kpK;
//   test c_k-1 in wJ; lanes whose step k - 1 left the grid (cyJ) have now had their last cell tested and leave.
// Between calls, and for parked lanes, the state is the ONE-ahead loops' (the caller cannot tell the difference): whoever leaves
// the loop after trip k — parked on c_k-1, or still moving when the call ends — takes step k back (the crossed axis' side distance
// := the crossed distance of trip k, which IS its old value; the cell := c_k; the carry of that step is forgotten), and a call
// starts with a trip that tests nothing.  One trip per lane per call and per brick entered is walked twice (about 7 % of the trips
// of the 2048^3 path trace); per lane the sequence of DDA operations is unchanged.
#define VRT_A2_HEAD(K, N)                                                                                   \
    "v_min3_f32 %[ts" K "], %[sdx], %[sdy], %[sdz]\n\t"                                                     \
    "v_cmp_eq_f32_e64 %[mt], %[sdz], %[ts" K "]\n\t"                                                        \
    "v_cmp_eq_f32_e64 %[mY" K "], %[sdy], %[ts" K "]\n\t"                                                   \
    "s_andn2_b64 %[mY" K "], %[mY" K "], %[mt]\n\t"                                                         \
    "s_andn2_b64 %[mt], exec, %[mt]\n\t"                                                                    \
    "s_andn2_b64 %[mX" K "], %[mt], %[mY" K "]\n\t"                                                         \
    "v_cndmask_b32_e64 %[t0], %[stz], %[sty], %[mY" K "]\n\t"                                               \
    "v_cndmask_b32_e64 %[t0], %[t0], %[stx], %[mX" K "]\n\t"                                                \
    "v_or_b32_e32 %[t1], %[idx" K "], %[t0]\n\t"                                                            \
    "v_add_co_u32_e64 %[t1], %[cy" K "], 1, %[t1]\n\t"                                                      \
    "v_bfi_b32 %[idx" N "], %[t0], %[idx" K "], %[t1]\n\t"                                                  \
    "v_xor_b32_e32 %[t2], %[idx" N "], %[idx" K "]\n\t"                                                     \
    "v_cmp_lt_u32_e64 %[by], 31, %[t2]\n\t"                                                                 \
    "s_cmp_eq_u64 %[by], 0\n\t"                                                                             \
    "s_cselect_b64 %[by], exec, %[by]\n\t"                                                                  \
    "s_and_saveexec_b64 %[cz], %[by]\n\t"                                                                   \
    "v_xor_b32_e32 %[t2], %[idx" N "], %[flip]\n\t"                                                         \
    "v_lshrrev_b32_e32 %[t2], 5, %[t2]\n\t"                                                                 \
    "buffer_load_dword %[w" N "], %[t2], %[rsrc], 0 idxen\n\t"                                              \
    "s_andn2_b64 %[kp" N "], %[cz], %[by]\n\t"                                                              \
    "s_mov_b64 exec, %[mX" K "]\n\t"                                                                        \
    "v_add_f32_e64 %[sdx], %[sdx], |%[ix]|\n\t"                                                             \
    "s_mov_b64 exec, %[mY" K "]\n\t"                                                                        \
    "v_add_f32_e64 %[sdy], %[sdy], |%[iy]|\n\t"                                                             \
    "s_andn2_b64 exec, %[cz], %[mt]\n\t"                                                                    \
    "v_add_f32_e64 %[sdz], %[sdz], |%[iz]|\n\t"
#define VRT_A2_TAIL(K, J, PARK)                                                                             \
    "s_and_b64 exec, %[cz], %[kp" K "]\n\t"                                                                 \
    "s_waitcnt vmcnt(2)\n\t"                                                                                \
    "v_mov_b32_e32 %[w" K "], %[w" J "]\n\t"                                                                \
    "s_mov_b64 exec, %[cz]\n\t"                                                                             \
    "v_xor_b32_e32 %[t1], %[idx" J "], %[flip]\n\t"                                                         \
    "v_bfe_u32 %[t1], %[w" J "], %[t1], 1\n\t"                                                              \
    "v_cmp_ne_u32_e32 vcc, 0, %[t1]\n\t"                                                                    \
    "s_andn2_b64 exec, exec, %[cy" J "]\n\t"                                                                \
    "s_cbranch_vccnz " PARK "\n\t"
// take step K back for the lanes in MASK (a scalar pair; clobbers by, EXEC)
#define VRT_A2_UNSTEP(K, MASK)                                                                              \
    "s_and_b64 exec, " MASK ", %[mX" K "]\n\t"                                                              \
    "v_mov_b32_e32 %[sdx], %[ts" K "]\n\t"                                                                  \
    "s_and_b64 exec, " MASK ", %[mY" K "]\n\t"                                                              \
    "v_mov_b32_e32 %[sdy], %[ts" K "]\n\t"                                                                  \
    "s_or_b64 %[by], %[mX" K "], %[mY" K "]\n\t"                                                            \
    "s_andn2_b64 exec, " MASK ", %[by]\n\t"                                                                 \
    "v_mov_b32_e32 %[sdz], %[ts" K "]\n\t"
// the lanes in vcc have their cell c_k-1 occupied: the one-ahead loops' parked state (see GridParkRegs), step k taken back
#define VRT_A2_PARK(LABEL, K, J, INAXIS, TSIN, MOVIDX, NEXT, EXIT)                                          \
    LABEL ":\n\t"                                                                                           \
    "s_and_b64 %[by], %[cy" J "], vcc\n\t"                                                                  \
    "s_or_b64 %[gone], %[gone], %[by]\n\t"                                                                  \
    "s_mov_b64 %[ex], exec\n\t"                                                                             \
    "s_mov_b64 exec, vcc\n\t"                                                                               \
    INAXIS                                                                                                  \
    "v_cndmask_b32_e64 %[t1], 2, 1, %[mY" J "]\n\t"                                                         \
    "v_cndmask_b32_e64 %[t1], %[t1], 0, %[mX" J "]\n\t"                                                     \
    "v_lshl_or_b32 %[code], %[t1], 2, %[t0]\n\t"                                                            \
    "v_mov_b32_e32 %[cell], %[idx" J "]\n\t"                                                                \
    MOVIDX                                                                                                  \
    "v_mov_b32_e32 %[tin], " TSIN "\n\t"                                                                    \
    "v_mov_b32_e32 %[tout], %[ts" J "]\n\t"                                                                 \
    VRT_A2_UNSTEP(K, "vcc")                                                                                 \
    "s_or_b64 %[parked], %[parked], vcc\n\t"                                                                \
    "s_andn2_b64 exec, %[ex], vcc\n\t"                                                                      \
    "s_bcnt1_i32_b64 %[n], %[parked]\n\t"                                                                   \
    "s_cmp_ge_u32 %[n], %[batch]\n\t"                                                                       \
    "s_cbranch_scc1 " EXIT "\n\t"                                                                           \
    "s_cbranch_execnz " NEXT "\n\t"                                                                         \
    "s_branch " EXIT "\n\t"
#define VRT_A2_IN_CODE "v_bfe_u32 %[t0], %[code], 4, 2\n\t"
#define VRT_A2_IN_SET(I) "v_cndmask_b32_e64 %[t0], 2, 1, %[mY" I "]\n\t" "v_cndmask_b32_e64 %[t0], %[t0], 0, %[mX" I "]\n\t"
// the call ends after trip K with the lanes in EXEC still moving: step k back, their cell, its word and their last step where the caller looks for them
#define VRT_A2_EXIT(LABEL, K, J, MOVIDX)                                                                    \
    LABEL ":\n\t"                                                                                           \
    "s_mov_b64 %[alive], exec\n\t"                                                                          \
    "s_waitcnt vmcnt(0)\n\t"                                                                                \
    VRT_A2_UNSTEP(K, "%[alive]")                                                                            \
    "s_mov_b64 exec, %[alive]\n\t"                                                                          \
    MOVIDX                                                                                                  \
    "v_mov_b32_e32 %[tout], %[ts" J "]\n\t"                                                                 \
    "s_mov_b64 %[mxb], %[mX" J "]\n\t"                                                                      \
    "s_mov_b64 %[myb], %[mY" J "]\n\t"                                                                      \
    "s_branch 99f\n\t"
VRT_DI void grid_walk_park_dilated_ahead_gfx950(f3 &side_dist, const f3 &inv_dir, uint32_t &index, uint32_t &cell, uint32_t nm_x, uint32_t nm_y, uint32_t nm_z,
                                                uint32_t &word, u32x4 rsrc, GridParkRegs &g, uint32_t flip, unsigned long long &gone) {
    unsigned long long mXx, mYx, mXy, mYy, mXz, mYz, cyx, cyy, cyz, kpx, kpy, kpz, mt, ex, by, cz, save;
    float tsx, tsy, tsz, t0, t1, t2;
    uint32_t idxy, idxz, wy, wz, n;
    gone = 0ull;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "s_mov_b64 exec, %[alive]\n\t"
        "s_mov_b64 %[parked], 0\n\t"
        "s_mov_b64 %[kpx], 0\n\t"
        /* trip 0: nothing to test yet */
        VRT_A2_HEAD("x", "y")
        "s_mov_b64 exec, %[cz]\n\t"
        /* trip 1: tests c_0 (its word came with the call; it was entered by the lane's last step before the call: code, tout) */
        VRT_A2_HEAD("y", "z")
        VRT_A2_TAIL("y", "x", "10f")
        "0:\n\t"
        VRT_A2_HEAD("z", "x")
        VRT_A2_TAIL("z", "y", "11f")
        "21:\n\t"
        VRT_A2_HEAD("x", "y")
        VRT_A2_TAIL("x", "z", "12f")
        "22:\n\t"
        VRT_A2_HEAD("y", "z")
        VRT_A2_TAIL("y", "x", "13f")
        "23:\n\t"
        "s_cbranch_execz 31f\n\t"
        "s_bcnt1_i32_b64 %[n], exec\n\t"
        "s_cmp_ge_u32 %[n], %[minalive]\n\t"
        "s_cbranch_scc1 0b\n\t"
        "s_branch 31f\n\t"
        VRT_A2_PARK("10", "y", "x", VRT_A2_IN_CODE, "%[tout]", "v_mov_b32_e32 %[idxx], %[idxy]\n\t", "0b", "31f")
        VRT_A2_PARK("11", "z", "y", VRT_A2_IN_SET("x"), "%[tsx]", "v_mov_b32_e32 %[idxx], %[idxz]\n\t", "21b", "32f")
        VRT_A2_PARK("12", "x", "z", VRT_A2_IN_SET("y"), "%[tsy]", "", "22b", "30f")
        VRT_A2_PARK("13", "y", "x", VRT_A2_IN_SET("z"), "%[tsz]", "v_mov_b32_e32 %[idxx], %[idxy]\n\t", "23b", "31f")
        VRT_A2_EXIT("30", "x", "z", "")
        VRT_A2_EXIT("31", "y", "x", "v_mov_b32_e32 %[idxx], %[idxy]\n\t" "v_mov_b32_e32 %[wx], %[wy]\n\t")
        VRT_A2_EXIT("32", "z", "y", "v_mov_b32_e32 %[idxx], %[idxz]\n\t" "v_mov_b32_e32 %[wx], %[wz]\n\t")
        "99:\n\t"
        "s_mov_b64 exec, %[save]"
        : [sdx] "+v"(side_dist.x), [sdy] "+v"(side_dist.y), [sdz] "+v"(side_dist.z), [idxx] "+v"(index), [idxy] "=&v"(idxy), [idxz] "=&v"(idxz),
          [cell] "=&v"(cell), [wx] "+v"(word), [wy] "=&v"(wy), [wz] "=&v"(wz), [tsx] "=&v"(tsx), [tsy] "=&v"(tsy), [tsz] "=&v"(tsz), [tout] "+v"(g.t_out),
          [tin] "=&v"(g.t_in), [code] "+v"(g.code), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [mxb] "+s"(g.out_x), [myb] "+s"(g.out_y),
          [alive] "+s"(g.alive), [mXx] "=&s"(mXx), [mYx] "=&s"(mYx), [mXy] "=&s"(mXy), [mYy] "=&s"(mYy), [mXz] "=&s"(mXz), [mYz] "=&s"(mYz), [cyx] "=&s"(cyx),
          [cyy] "=&s"(cyy), [cyz] "=&s"(cyz), [kpx] "=&s"(kpx), [kpy] "=&s"(kpy), [kpz] "=&s"(kpz), [mt] "=&s"(mt), [ex] "=&s"(ex), [by] "=&s"(by), [cz] "=&s"(cz),
          [save] "=&s"(save), [parked] "=&s"(g.parked), [n] "=&s"(n), [gone] "+s"(gone)
        : [ix] "v"(inv_dir.x), [iy] "v"(inv_dir.y), [iz] "v"(inv_dir.z), [stx] "v"(nm_x), [sty] "v"(nm_y), [stz] "v"(nm_z), [rsrc] "s"(rsrc), [batch] "s"(g.batch),
          [minalive] "s"(g.min_alive), [flip] "v"(flip)
        : "vcc", "scc");
}
#undef VRT_A2_HEAD
#undef VRT_A2_TAIL
#undef VRT_A2_UNSTEP
#undef VRT_A2_PARK
#undef VRT_A2_IN_CODE
#undef VRT_A2_IN_SET
#undef VRT_A2_EXIT
// The counter-free dilated loop on 4 x 4 x 4-CELL words (vrt_path_kernel<..., DIL 3>): the 64-bit words of TraceParams::status_blocks,
// index bits 0-5 = the cell's place in its block (x&3 | (z&3) << 2 | (y&3) << 4), the bits above = the block's number.  A lane asks
// when its step enters another block: 0.265 times per trip in the 2048^3 sparse field against 0.333 for half-blocks
// (tools/experiments/request_replay.py).  The word is a register pair; the test shifts the cell's bit into bit 63 (v_lshlrev_b64 by ~index, of
// which the instruction reads the low six bits) and compares with 0.
#define VRT_LOAD_DILATED64_A(IDX, IDXN, WORD, WORDN)                       \
    "v_xor_b32_e32 %[t2], %[" IDXN "], %[" IDX "]\n\t"                     \
    "v_cmp_lt_u32_e64 %[by], 63, %[t2]\n\t"                                \
    "s_cmp_eq_u64 %[by], 0\n\t"                                            \
    "s_cselect_b64 %[by], exec, %[by]\n\t"                                 \
    "s_and_saveexec_b64 %[cz], %[by]\n\t"                                  \
    "v_xor_b32_e32 %[t2], %[" IDXN "], %[flip]\n\t"                        \
    "v_lshrrev_b32_e32 %[t2], 6, %[t2]\n\t"                                \
    "buffer_load_dwordx2 %[" WORDN "], %[t2], %[rsrc], 0 idxen\n\t"
#define VRT_LOAD_DILATED64_B(IDX, IDXN, WORD, WORDN)                       \
    "s_andn2_b64 exec, %[cz], %[by]\n\t"                                   \
    "s_waitcnt vmcnt(1)\n\t"                                               \
    "v_mov_b64_e32 %[" WORDN "], %[" WORD "]\n\t"                          \
    "s_mov_b64 exec, %[cz]\n\t"
#define VRT_TEST_DILATED64(WORD, IDX)                                      \
    "v_xnor_b32_e32 %[t1], %[" IDX "], %[flip]\n\t"                        \
    "v_lshlrev_b64 %[tp], %[t1], %[" WORD "]\n\t"                          \
    "v_cmp_gt_i64_e64 vcc, 0, %[tp]\n\t"
VRT_DI void grid_walk_park_dilated64_gfx950(f3 &side_dist, const f3 &inv_dir, uint32_t &index, uint32_t &cell, uint32_t nm_x, uint32_t nm_y, uint32_t nm_z,
                                            unsigned long long &word, u32x4 rsrc, GridParkRegs &g, uint32_t flip, unsigned long long &gone) {
    unsigned long long mxa, mya, mxya, mxyb, ex, by, cz, save;
    typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
    u32x2 wordb, tp, wa = __builtin_bit_cast(u32x2, word);
    float t0, t1, t2;
    uint32_t n;
    const uint32_t stride_x = nm_x, stride_y = nm_y, stride_z = nm_z; // (the operand list's names)
    gone = 0ull;
    asm volatile(VRT_PARK_WALK_ASM_W(VRT_TRIP_E, VRT_STEP_DILATED_CARRY, VRT_EXIT_CARRY, "s_and_b64 %[by], %[ex], vcc\n\ts_or_b64 %[gone], %[gone], %[by]\n\t",
                                     VRT_NO_LIMIT, VRT_LOAD_DILATED64, VRT_TEST_DILATED64, VRT_WAIT_BUFFER, "v_mov_b64_e32 %[worda], %[wordb]\n\t")
                 : [sdx] "+v"(side_dist.x), [sdy] "+v"(side_dist.y), [sdz] "+v"(side_dist.z), [idxa] "+v"(index), [idxb] "=&v"(cell), [worda] "+v"(wa),
                   [wordb] "=&v"(wordb), [tp] "=&v"(tp), [tsb] "+v"(g.t_out), [tsa] "=&v"(g.t_in), [code] "+v"(g.code), [t0] "=&v"(t0), [t1] "=&v"(t1),
                   [t2] "=&v"(t2), [mxb] "+s"(g.out_x), [myb] "+s"(g.out_y), [alive] "+s"(g.alive), [mxa] "=&s"(mxa), [mya] "=&s"(mya), [mxya] "=&s"(mxya),
                   [mxyb] "=&s"(mxyb), [ex] "=&s"(ex), [by] "=&s"(by), [cz] "=&s"(cz), [save] "=&s"(save), [parked] "=&s"(g.parked), [n] "=&s"(n), [gone] "+s"(gone)
                 : VRT_PARK_WALK_INPUTS, [flip] "v"(flip)
                 : "vcc", "scc");
    word = __builtin_bit_cast(unsigned long long, wa);
}
#undef VRT_LOAD_DILATED64_A
#undef VRT_LOAD_DILATED64_B
#undef VRT_TEST_DILATED64
#endif
#if VRT_DEV_SECTION == 4
// ---- the brick-level park loop on a DISTANCE FIELD (vrt_path_kernel<DIST>, round 3) ---------------------------------------------
// A DDA trip moves one cell along one axis, so n trips reach exactly the cells within L1 (Manhattan) distance n of where they
// started.  TraceParams::cell_distance holds, per cell, its L1 distance in cells to the nearest occupied cell (0 = occupied, capped
// at 255; derived from binding 3 at every status upload).  A lane that has read d > 0 at cell P knows the next d - 1 cells of ANY walk
// to be empty: it takes those trips without asking — same DDA operations, same order, fewer tests — and asks again for the cell
// d trips behind P.  Per lane: `k` = trips it may still take before it has to ask (<= 0: ask in this trip), and the usual one-trip
// pipeline: the byte of the cell entered is requested in the trip that enters it and tested in the next, after that trip's step
// (a lane that asks keeps asking every trip until an answer > 1 arrives: the answer for P is only there when P + 1 has been asked).
// Against the half-block words (27 vector instructions per trip, a request per lane every third trip) a trip is 17 vector
// instructions and, in the 2048^3 sparse field, a lane asks about twice per d cells.  The index is the plain linear cell index =
// the byte offset: any grid dimensions.  If no lane of the wave has to ask, all do (a trip always issues one request, so that
// vmcnt(1) keeps its meaning); `nd<word>` = the lanes whose <word> register holds an answer.
#define VRT_LOAD_DIST_A(IDX, IDXN, WORD, WORDN)                                    \
    "v_cmp_gt_i32_e64 %[nd" WORDN "], 1, %[k]\n\t"                                 \
    "v_add_u32_e32 %[k], -1, %[k]\n\t"                                             \
    "s_cmp_eq_u64 %[nd" WORDN "], 0\n\t"                                           \
    "s_cselect_b64 %[nd" WORDN "], exec, %[nd" WORDN "]\n\t"                       \
    "s_and_saveexec_b64 %[cz], %[nd" WORDN "]\n\t"                                 \
    "buffer_load_ubyte %[" WORDN "], %[" IDXN "], %[rsrc], 0 offen\n\t"
#define VRT_LOAD_DIST_B(IDX, IDXN, WORD, WORDN)                                    \
    "s_mov_b64 exec, %[cz]\n\t"                                                    \
    "s_waitcnt vmcnt(1)\n\t"
#define VRT_TEST_DIST(WORD, IDX)                                                   \
    "v_cmp_eq_u32_e32 vcc, 0, %[" WORD "]\n\t"                                     \
    "v_add_u32_e32 %[t1], -2, %[" WORD "]\n\t"                                     \
    "s_and_b64 vcc, vcc, %[nd" WORD "]\n\t"         /* occupied: an answer, and it is 0 */ \
    "v_cndmask_b32_e64 %[k], %[k], %[t1], %[nd" WORD "]\n\t"
VRT_DI void grid_walk_park_dist_gfx950(Walk &w, const f3 &inv_dir, uint32_t &index, uint32_t &cell, uint32_t stride_x, uint32_t stride_y, uint32_t stride_z,
                                       uint32_t &word, u32x4 rsrc, GridParkRegs &g, DistRegs &d) {
    unsigned long long mxa, mya, mxya, mxyb, ex, by, cz, save, ndwordb;
    float t0, t1, t2;
    uint32_t wordb, n;
    asm volatile(VRT_PARK_WALK_ASM(VRT_NO_LIMIT, VRT_LOAD_DIST, VRT_TEST_DIST, VRT_WAIT_BUFFER, "s_mov_b64 %[ndworda], %[ndwordb]\n\t")
                 : VRT_PARK_WALK_OPERANDS, [ndworda] "+s"(d.pend), [ndwordb] "=&s"(ndwordb), [k] "+v"(d.k)
                 : VRT_PARK_WALK_INPUTS
                 : "vcc", "scc");
}
#undef VRT_LOAD_DIST_A
#undef VRT_LOAD_DIST_B
#undef VRT_TEST_DIST
#endif
#if VRT_DEV_SECTION == 5
#define VRT_AHEAD_TRIP(WL, WT, K, NEXT)                                                        \
        /* rotate the cells and the crossed distances */                                        \
        "v_mov_b32_e32 %[q0], %[q1]\n\t"                                                        \
        "v_mov_b32_e32 %[q1], %[q2]\n\t"                                                        \
        "v_mov_b32_e32 %[ts0], %[ts1]\n\t"                                                      \
        "v_mov_b32_e32 %[ts1], %[ts2]\n\t"                                                      \
        /* the step out of q1 (comp:345-372), as in VRT_TRIP_T */                               \
        "v_min3_f32 %[ts2], %[sdx], %[sdy], %[sdz]\n\t"                                         \
        "v_cmp_eq_f32_e64 %[mxy], %[sdz], %[ts2]\n\t"                                           \
        "v_cmp_eq_f32_e64 %[my], %[sdy], %[ts2]\n\t"                                            \
        "s_andn2_b64 %[my], %[my], %[mxy]\n\t"                                                  \
        "s_andn2_b64 %[mxy], exec, %[mxy]\n\t"                                                  \
        "s_andn2_b64 %[mx], %[mxy], %[my]\n\t"                                                  \
        "s_mov_b64 %[ex], exec\n\t"                                                             \
        "s_mov_b64 exec, %[mx]\n\t"                                                             \
        "v_add_f32_e64 %[sdx], %[sdx], |%[ix]|\n\t"                                             \
        "s_mov_b64 exec, %[my]\n\t"                                                             \
        "v_add_f32_e64 %[sdy], %[sdy], |%[iy]|\n\t"                                             \
        "s_andn2_b64 exec, %[ex], %[mxy]\n\t"                                                   \
        "v_add_f32_e64 %[sdz], %[sdz], |%[iz]|\n\t"                                             \
        "s_mov_b64 exec, %[ex]\n\t"                                                             \
        "v_cndmask_b32_e64 %[t1], 2, 1, %[my]\n\t"                                              \
        "v_cndmask_b32_e64 %[t1], %[t1], 0, %[mx]\n\t"                                          \
        "v_lshl_or_b32 %[hist], %[hist], 2, %[t1]\n\t"                                          \
        "v_cndmask_b32_e64 %[t0], %[stz], %[sty], %[my]\n\t"                                    \
        "v_cndmask_b32_e64 %[t0], %[t0], %[stx], %[mx]\n\t"                                     \
        "v_add_u32_e32 %[t2], %[q1], %[t0]\n\t"                                                 \
        "v_subbrev_co_u32_e64 %[rx], %[ex], 0, %[rx], %[mx]\n\t"                                \
        "v_subbrev_co_u32_e64 %[ry], %[by], 0, %[ry], %[my]\n\t"                                \
        "v_addc_co_u32_e64 %[rz], %[cz], -1, %[rz], %[mxy]\n\t"                                 \
        "s_or_b64 %[ex], %[ex], %[by]\n\t"                                                      \
        "s_orn2_b64 %[ex], %[ex], %[cz]\n\t"    /* z: carry-out 0 = borrow; the step left the box ... */ \
        "v_cmp_eq_u32_e64 %[by], -1, %[q1]\n\t" /* ... or an earlier one did */                  \
        "s_or_b64 %[ex], %[ex], %[by]\n\t"                                                      \
        "v_cndmask_b32_e64 %[q2], %[t2], -1, %[ex]\n\t"                                         \
        /* q2's word; the word of q0, asked for two trips ago, has arrived when at most two requests are outstanding */ \
        "v_lshrrev_b32_e32 %[t0], 5, %[q2]\n\t"                                                 \
        "buffer_load_dword %[" WL "], %[t0], %[rsrc], 0 idxen\n\t"                              \
        "s_waitcnt vmcnt(2)\n\t"                                                                \
        "v_bfe_u32 %[t1], %[" WT "], %[q0], 1\n\t"                                              \
        "v_cmp_ne_u32_e32 vcc, 0, %[t1]\n\t"                                                    \
        "v_cmp_eq_u32_e64 %[by], -1, %[q0]\n\t" /* the sentinel has arrived: every cell up to the face has been tested */ \
        "s_or_b64 %[left], %[left], %[by]\n\t"                                                  \
        "s_andn2_b64 exec, exec, %[by]\n\t"                                                     \
        "s_cbranch_vccz " NEXT "f\n\t"                                                          \
        "s_mov_b64 %[ex], exec\n\t"                                                             \
        "s_mov_b64 exec, vcc\n\t"                                                               \
        "v_mov_b32_e32 %[phase], " K "\n\t"                                                     \
        "s_andn2_b64 exec, %[ex], vcc\n\t"                                                      \
        "s_or_b64 %[parked], %[parked], vcc\n\t"                                                \
        "s_bcnt1_i32_b64 %[n], %[parked]\n\t"                                                   \
        "s_cmp_ge_u32 %[n], %[batch]\n\t"                                                       \
        "s_cbranch_scc1 9" K "f\n\t"                                                            \
        NEXT ":\n\t"
VRT_DI void grid_walk_ahead_gfx950(Walk &w, const f3 &inv_dir, AheadRing &a, uint32_t stride_x, uint32_t stride_y, uint32_t stride_z, u32x4 rsrc,
                                   AheadWalkRegs &g) {
    unsigned long long mx, my, mxy, ex, by, cz, save;
    float t0, t1, t2;
    uint32_t n;
    asm volatile(
        "s_mov_b64 %[save], exec\n\t"
        "s_mov_b64 exec, %[alive]\n\t"
        "s_mov_b64 %[parked], 0\n\t"
        "s_mov_b64 %[left], 0\n\t"
        "0:\n\t"
        VRT_AHEAD_TRIP("w0", "w1", "0", "20")
        "s_cbranch_execz 90f\n\t"
        VRT_AHEAD_TRIP("w1", "w2", "1", "21")
        "s_cbranch_execz 91f\n\t"
        VRT_AHEAD_TRIP("w2", "w0", "2", "22")
        "s_cbranch_execz 92f\n\t"
        "s_bcnt1_i32_b64 %[n], exec\n\t"
        "s_cmp_ge_u32 %[n], %[minalive]\n\t"
        "s_cbranch_scc1 0b\n\t"
        "92:\n\t"
        "v_mov_b32_e32 %[phase], 2\n\t"
        "s_branch 99f\n\t"
        "90:\n\t"
        "v_mov_b32_e32 %[phase], 0\n\t"
        "s_branch 99f\n\t"
        "91:\n\t"
        "v_mov_b32_e32 %[phase], 1\n\t"
        "99:\n\t"
        "s_mov_b64 %[alive], exec\n\t"
        "s_mov_b64 exec, %[save]\n\t"
        "s_waitcnt vmcnt(0)"
        : [sdx] "+v"(w.side_dist.x), [sdy] "+v"(w.side_dist.y), [sdz] "+v"(w.side_dist.z), [rx] "+v"(w.rx), [ry] "+v"(w.ry), [rz] "+v"(w.rz),
          [q0] "+v"(a.q0), [q1] "+v"(a.q1), [q2] "+v"(a.q2), [w0] "+v"(a.w0), [w1] "+v"(a.w1), [w2] "+v"(a.w2), [ts0] "+v"(a.ts0), [ts1] "+v"(a.ts1),
          [ts2] "+v"(a.ts2), [hist] "+v"(a.hist), [phase] "+v"(a.phase), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [mx] "=&s"(mx), [my] "=&s"(my),
          [mxy] "=&s"(mxy), [ex] "=&s"(ex), [by] "=&s"(by), [cz] "=&s"(cz), [save] "=&s"(save), [parked] "=&s"(g.parked), [left] "=&s"(g.left),
          [alive] "+s"(g.alive), [n] "=&s"(n)
        : [ix] "v"(inv_dir.x), [iy] "v"(inv_dir.y), [iz] "v"(inv_dir.z), [stx] "v"(stride_x), [sty] "v"(stride_y), [stz] "v"(stride_z), [rsrc] "s"(rsrc),
          [batch] "s"(g.batch), [minalive] "s"(g.min_alive)
        : "vcc", "scc", "memory");
    a.settle();
}
#undef VRT_AHEAD_TRIP
#endif
#if VRT_DEV_SECTION == 6
// The lane stands in a 4x4x4 block of cells that holds no occupied cell (at cell (bx, by, bz) of it): jump behind the step
// that leaves the block.  Each axis has its exit crossing (the e-th from now, e = cells to the block's face in the ray's
// direction + 1, at the side distance after e-1 additions); the one that comes first in merge order (smallest distance; z
// before y before x among equals, as the walk picks) is the step that leaves the block, and everything before it is consumed
// exactly as skip_to_box does.  ~100 vector instructions and no memory access for what would be one to ten trips.
VRT_DI void skip_empty_block(Walk &w, const RaySetup &s, uint32_t bx, uint32_t by, uint32_t bz, uint32_t &index, uint32_t stride_x, uint32_t stride_y,
                             uint32_t stride_z, bool &more, int &in_axis, float &t_in) {
    const int ex = s.sx > 0 ? 4 - (int)bx : (int)bx + 1, ey = s.sy > 0 ? 4 - (int)by : (int)by + 1, ez = s.sz > 0 ? 4 - (int)bz : (int)bz + 1;
    auto exit_distance = [](float sd, float d, int e, int step) {
        const float a1 = sd + d, a2 = a1 + d, a3 = a2 + d;
        const float t = e == 1 ? sd : (e == 2 ? a1 : (e == 3 ? a2 : a3));
        return step != 0 ? t : __builtin_inff(); // an axis the ray does not move along is never crossed
    };
    const float tx = exit_distance(w.side_dist.x, s.ray_delta().x, ex, s.sx);
    const float ty = exit_distance(w.side_dist.y, s.ray_delta().y, ey, s.sy);
    const float tz = exit_distance(w.side_dist.z, s.ray_delta().z, ez, s.sz);
    const bool az = tz <= tx && tz <= ty, ay = !az && ty <= tx, ax = !az && !ay;
    const float t = ax ? tx : (ay ? ty : tz);
    const float t_strict = next_below(t); // c < t  <=>  c <= t_strict
    // The other two axes: their elements that precede t in merge order (x loses every tie, z wins every tie, y wins against x
    // only) — at most three each, their own exit crossing comes later.  Straight-line: no loop, no lane mask juggling.
    const float never = -__builtin_inff();
    auto consume = [](float &c, float d, float lim, int &n) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const bool before = c <= lim;
            c = before ? c + d : c;
            n += before ? 1 : 0;
        }
    };
    float cx = w.side_dist.x, cy = w.side_dist.y, cz = w.side_dist.z;
    int nx = 0, ny = 0, nz = 0;
    consume(cx, s.ray_delta().x, ax ? never : t_strict, nx);
    consume(cy, s.ray_delta().y, ay ? never : (ax ? t : t_strict), ny);
    consume(cz, s.ray_delta().z, az ? never : t, nz);
    w.side_dist.x = ax ? t + s.ray_delta().x : cx;
    w.side_dist.y = ay ? t + s.ray_delta().y : cy;
    w.side_dist.z = az ? t + s.ray_delta().z : cz;
    nx = ax ? ex : nx;
    ny = ay ? ey : ny;
    nz = az ? ez : nz;
    w.rx -= nx;
    w.ry -= ny;
    w.rz -= nz;
    index += (uint32_t)nx * stride_x + (uint32_t)ny * stride_y + (uint32_t)nz * stride_z;
    more = (w.rx | w.ry | w.rz) >= 0; // a counter below zero: the far face of the box of occupied cells was crossed on the way
    in_axis = ax ? 0 : (ay ? 1 : 2);
    t_in = t;
}

#endif
