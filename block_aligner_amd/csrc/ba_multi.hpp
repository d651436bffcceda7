// block_aligner_amd — four pairs per wavefront while the block is 128 cells, inside ONE persistent kernel.
//
// The per-pair kernel (ba_driver.hpp) spends ~37 vector instructions on a 128-cell column of which 9 are the 64-lane prefix
// scan and pays ~55 instructions of driver shell per 8-column step, all for one pair. k_multi hosts FOUR pairs per wave, one
// per 16-lane DPP row ("slot"), with EIGHT cells per lane: lane l of a slot owns cells 8l .. 8l+7 of the block's vector axis as
// four packed 2 x i16 registers. The gap scan along the column (avx2.rs:297-338 + scan_block.rs:1144-1150) then is an in-lane
// chain over the lane's four registers followed by ONE 16-lane DPP scan (4 steps) that serves all four pairs at once; the
// orthogonal border shifts by exactly one lane per step (8 entries), i.e. by a DPP row shift instead of an LDS round trip; and
// the driver decisions of scan_block.rs:332-558 are evaluated for the four slots together with vector compares and selects
// (what is wave-uniform in the per-pair kernel is row-uniform here, replicated over the slot's lanes in VGPRs).
//
// A slot takes only PLAIN shift steps at 128 cells. Whatever else a pair needs -- its first block, a grow, the steps at larger
// block sizes, a shrink, X-drop termination, the end of the matrix, the early column break -- is done by the same wave in
// "solo" mode: the pair's state moves to LDS / scalars and Aligner::run (ba_driver.hpp: all 64 lanes on that one pair, the code
// of the per-pair kernel) takes it until the pair is finished or its next step is again a plain shift step at 128 cells, while
// the other three slots wait in their registers. No lane is idle in either mode. A step whose outcome calls for more than a
// shift is rolled back in the slot and repeated solo.
//
// The location of a step's maximum (X-drop: scan_block.rs:370-404) is not computed in a slot: only the LAST improving step's
// location reaches the result, and that step is also the checkpoint (scan_block.rs:406-427). A slot keeps the state BEFORE its
// last improving step; whenever the pair goes solo that step is repeated once (Aligner::run, MultiIO::ck_pre), which yields the
// location and the borders the reference's checkpoint holds.
//
// Trace words and rectangle records are those of the per-pair kernel (word (col >> 2) * 64 + (row >> 1) of a 128-row
// rectangle holds cells (row, row + 1) x 4 columns; with 8 cells per lane these are a lane's four registers, one 16-byte store).
#pragma once
#include "ba_quad.hpp"

namespace ba {
#ifndef MQ_SOLO_PRIO
#define MQ_SOLO_PRIO 2   // wave priority while a pair is in solo mode (the traceback waves run at BA_TB_PRIO = 3)
#endif

template <int N>
__device__ __forceinline__ int row_bcast(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x150 + N, 0xf, 0xf, false); }   // row_newbcast:N
// lane l <- src[l + 1] inside the row; lane 15 keeps `last`
__device__ __forceinline__ int row_shl1_keep(int last, int src) { return __builtin_amdgcn_update_dpp(last, src, 0x101, 0xf, 0xf, false); }
__device__ __forceinline__ int splat_hi(int x) { const s16x2 t = as_s(x); return as_i(s16x2{t.y, t.y}); }
__device__ __forceinline__ int splat_lo(int x) { const s16x2 t = as_s(x); return as_i(s16x2{t.x, t.x}); }

// mask bit set ? a : b, as one v_cndmask_b32 (the lane mask in a scalar register pair)
__device__ __forceinline__ int sel_mask(unsigned long long m, int a, int b) {
    int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}

// ---- slots of SL = 16 lanes (128 cells, four pairs per wave) or SL = 32 lanes (256 cells, two pairs per wave: round 6). The 16-lane forms are DPP
// row operations; a 32-lane slot spans two DPP rows, so its shifts are whole-wave shifts with the slot's own edge lane patched, and its broadcasts
// go through the LDS crossbar (ds_bpermute: three per step). l = the lane inside its slot.
// (SL = 64, round 6: one slot of 512 cells -- the whole wave: the edges are the wave's, the broadcasts scalar reads of lane 0 / 63)
template <int SL> __device__ __forceinline__ int slot_first(int v) {   // the slot's first lane's value, to all its lanes
    if constexpr (SL == 16) return row_bcast<0>(v);
    else if constexpr (SL == 64) return __builtin_amdgcn_readfirstlane(v);
    else return __builtin_amdgcn_ds_bpermute((lane_id() & 32) << 2, v);
}
template <int SL> __device__ __forceinline__ int slot_last(int v) {    // the slot's last lane's value, to all its lanes
    if constexpr (SL == 16) return row_bcast<15>(v);
    else if constexpr (SL == 64) return __builtin_amdgcn_readlane(v, 63);
    else return __builtin_amdgcn_ds_bpermute(((lane_id() & 32) + 31) << 2, v);
}
template <int SL> __device__ __forceinline__ int slot_shr1_z(int v, int l) {   // lane l <- l - 1 inside the slot; its first lane <- 0
    if constexpr (SL == 16) return row_shr1_z(v);
    else if constexpr (SL == 64) return wave_shr1_z(v);
    else { const int t = wave_shr1_z(v); return l == 0 ? 0 : t; }
}
template <int SL> __device__ __forceinline__ int slot_shl1_keep(int last, int src, int l) {   // lane l <- src[l + 1] inside the slot; its last lane keeps `last`
    if constexpr (SL == 16) return row_shl1_keep(last, src);
    else { const int t = __builtin_amdgcn_update_dpp(last, src, 0x130, 0xf, 0xf, false); return l == SL - 1 ? last : t; }   // (wave_shl:1; lane 63 has no source: keeps `last`)
}

struct MultiConsts {
    int G[4];             // {(2k+1) g, (2k+2) g}: what a gap that enters the lane above its first cell has lost on reaching register k
    int laneKG, lanem1KG; // l * 8g; (l - 1) * 8g, except row lane 0 which holds -32768 (no lane above: a candidate that never wins)
    int w0;               // {0, W}: the floor of the carry that enters the lane from above, per half. Per cell the column's R is also at least
                          // max(zero-shift-in artefact of the reference's in-vector scan, the MIN = 0 carry above the column) =: V(cell); for the lane's
                          // first seven cells V = G[k] in every lane (artefact ((cell & 7) + 1) g >= carry (cell + 1) g), i.e. max(cs + G, G) =
                          // max(cs, 0) + G (saturating adds of non-positive constants compose); the eighth cell's V differs per lane -- 8g in lane 0,
                          // 12g in the other even lanes, (8l + 8) g in the odd ones -- and is written V = W + 8g: ONE max per column (cs against
                          // {0, W}) replaces the four per-register maxima with V
};

// (one definition for the kernels and k_lane_kat) l: the lane inside its slot's sixteen
__device__ __forceinline__ MultiConsts make_multi_consts(int l, int gx) {
    MultiConsts mc;
    mc.laneKG = l * 8 * gx; mc.lanem1KG = l ? (l - 1) * 8 * gx : -32768;
#pragma unroll
    for (int k = 0; k < 4; k++) mc.G[k] = pk(max(-32768, (2 * k + 1) * gx), max(-32768, (2 * k + 2) * gx));
    mc.w0 = pk(0, l == 0 ? 0 : ((l & 1) ? max(-32768, 8 * l * gx) : 4 * gx));   // (see MultiConsts)
    return mc;
}
// R of the lane above's last cell, from the lane's own last R (high half of its fourth register): one 16-lane scan on values re-based by l * 8g
// serves the wave's four slots (no clamp: see fast_scan), floored at {0, W} (MultiConsts::w0)
template <int SL = 16>
__device__ __forceinline__ int multi_carry(int r3, const MultiConsts& mc, int l = 0) {
    if constexpr (SL == 16) {
        const int pm = wave_prefix_max16((int)as_s(r3).y - mc.laneKG);
        return vmax(scan8_splat_lo(add_row_shr1(pm, mc.lanem1KG)), mc.w0);
    } else {   // 32 lanes: the scan restarts at lane 32 (wave_prefix_max32); the shift by one lane crosses the slots' edge, whose lane takes the filler instead
        const int pm = SL == 64 ? wave_prefix_max((int)as_s(r3).y - mc.laneKG) : wave_prefix_max32((int)as_s(r3).y - mc.laneKG);
        const int cin = add_shr1(pm, mc.lanem1KG);
        return vmax(scan8_splat_lo(l == 0 ? -32768 : cin), mc.w0);
    }
}

struct MultiOut { int mx, act_max8, pas_max8, corner_new; };

// One 8-column shift step for the four slots of a wave, 8 cells per lane (scan_block.rs:147-246 with place_block 1083-1228 and
// the border moves 1003-1061 folded in). (Ad, Ac): the border pair along the step's vector axis, (Pd, Pr) the orthogonal pair.
// tout: this lane's eight trace words of the step.
// SPM (round 5): the special alignment modes a slot takes, as in small_rect (ba_small.hpp) -- 1 = LOCAL_START (every cell's D is at least the relative
// zero rz2; TRACE: a zero mask of one bit per cell, two words per lane -- cells 0 .. 3 / 4 .. 7, byte = cell, bit = column --, stored by the caller behind
// the rectangle's 128 trace words), 2 = FREE_QUERY_START_GAPS (fqs_row0: row 0 of a right step starts from the relative zero in every column).
template <int KIND, bool TRACE, int SPM = 0, int SL = 16>
__device__ __forceinline__ void multi_rect(const char* table, const FillConsts& fc, const MultiConsts& mc, int l, int (&Ad)[4], int (&Ac)[4],
                                           int (&Pd)[4], int (&Pr)[4], uint2 vb, uint32_t cb_lo, uint32_t cb_hi, int corner, int off_add,
                                           uint32_t* __restrict__ tout, bool store, MultiOut& o, int rz2 = 0, bool fqs_row0 = false, int* zout = nullptr) {
    uint32_t zacc[2] = {0, 0};   // LOCAL_START + TRACE: "D differs from the relative zero", inverted at the end
    const int offa = splat(off_add);
    int d[4], c[4], pd[4], pr[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {   // just_offset (scan_block.rs:1003-1012)
        d[k] = adds(Ad[k], offa); c[k] = adds(Ac[k], offa); pd[k] = adds(Pd[k], offa); pr[k] = adds(Pr[k], offa);
    }
    o.corner_new = slot_first<SL>(pd[3]) >> 16;   // D_corner for a following orthogonal step: the orthogonal border's entry 7, re-based
    ScoreKey<KIND> key[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t w = k < 2 ? vb.x : vb.y;
        key[k] = make_key<KIND>((int)((w >> (16 * (k & 1))) & 0xffu), (int)((w >> (16 * (k & 1) + 8)) & 0xffu));
    }
    uint32_t kb[4] = {0, 0, 0, 0}, cbs_lo = 0, cbs_hi = 0;   // NUC: see add_byte (ba_device.hpp)
    if constexpr (KIND == KIND_NUC) {
        const uint32_t tb = (uint32_t)(uintptr_t)table, k01 = nuc_keys2(vb.x), k23 = nuc_keys2(vb.y);
        kb[0] = add_word(k01, tb, 0); kb[1] = add_word(k01, tb, 1); kb[2] = add_word(k23, tb, 0); kb[3] = add_word(k23, tb, 1);
        cbs_lo = nuc_col_offsets(cb_lo); cbs_hi = nuc_col_offsets(cb_hi);
    }
    int dmax[4] = {0, 0, 0, 0}, tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int nvD[4] = {0, 0, 0, 0}, nvR[4] = {0, 0, 0, 0};   // the last cells of the 8 new columns: the orthogonal border's new entries (row lane 15)
    int holdD = 0, holdR = 0;
#pragma unroll
    for (int j = 0; j < STEP; j++) {
        const int cb = (int)(((j < 4 ? cb_lo : cb_hi) >> (8 * (j & 3))) & 0xffu);
        int sc[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if constexpr (KIND == KIND_NUC) sc[k] = lds_read_i32(add_byte(j < 4 ? cbs_lo : cbs_hi, kb[k], j));
            else sc[k] = fetch_score<KIND>(table, key[k], cb);
        }
        // D00: the previous column shifted down one cell (scan_block.rs:1125); only column 0 has a cell above the block
        int prev = slot_shr1_z<SL>(d[3], l);
        if (j == 0) prev = l == 0 ? (int)((uint32_t)corner << 16) : prev;
        int d00[4];
        d00[0] = __builtin_amdgcn_alignbit(d[0], prev, 16);
#pragma unroll
        for (int k = 1; k < 4; k++) d00[k] = __builtin_amdgcn_alignbit(d[k], d[k - 1], 16);
        int d11[4], copen[4], cn[4], x[4], r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            d11[k] = adds(d00[k], sc[k]);
            if (SPM == 2 && k == 0) d11[0] = fqs_row0 ? (int)(((uint32_t)d11[0] & 0xffff0000u) | ((uint32_t)rz2 & 0xffffu)) : d11[0];   // row 0 of a right step: a free start in every column (scan_block.rs:1130-1132)
            if (SPM == 1) d11[k] = vmax(d11[k], rz2);   // a local alignment may start anywhere (scan_block.rs:1134-1136)
            copen[k] = adds(d[k], fc.go2);
            cn[k] = vmax(adds(c[k], fc.ge2), copen[k]);
            d11[k] = vmax(d11[k], cn[k]);
            x[k] = adds(d11[k], fc.ome2);                                  // D11_open
            r[k] = scan8_inreg(x[k], fc.ge2);                              // inside the register
        }
        // R11: the chain over the lane's registers, then one 16-lane scan on values re-based by l * 8g
        scan8_chain(r, mc.G[0]);
        const int cs = multi_carry<SL>(r[3], mc, l);    // R of the lane above's last cell (no clamp: see fast_scan), floored: see MultiConsts::w0
        int dn[4];
        uint32_t sC[4], sR[4], sCo[4], sRo[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            r[k] = scan8_apply(r[k], cs, mc.G[k], k == 3);
            dn[k] = vmax(d11[k], r[k]);
            if (TRACE) {   // the cell's four flags as sign bits of saturating differences (see fast_rect); packed below, two registers at a time
                sC[k] = (uint32_t)subs(cn[k], dn[k]); sR[k] = (uint32_t)subs(r[k], dn[k]); sCo[k] = (uint32_t)subs(copen[k], cn[k]); sRo[k] = (uint32_t)subs(x[k], r[k]);
            }
            dmax[k] = vmax(dmax[k], dn[k]);
            d[k] = dn[k]; c[k] = cn[k];
        }
        if (TRACE && SPM == 1) {   // zero mask (scan_block.rs:1184-1187): D >= the relative zero, so "differs" is the sign of rz - D
#pragma unroll
            for (int p2 = 0; p2 < 2; p2++) {
                const uint32_t pZ = (uint32_t)__builtin_amdgcn_perm(subs(rz2, dn[2 * p2 + 1]), subs(rz2, dn[2 * p2]), 0x0b0a0908);
                zacc[p2] |= pZ & (0x01010101u << j);
            }
        }
        if (TRACE) {
            // Trace words of a slot's rectangle: 4 consecutive cells (the two registers of a pair) x 2 columns, a byte per cell, the even
            // column in its low nibble; a lane's eight words -- its 8 cells x the step's 8 columns -- are contiguous in memory (word
            // lane * 8 + (column >> 1) * 2 + register pair), so that the traceback finds everything around a path cell in one cache line.
            // The sign bytes of two registers' differences are gathered by one v_perm each, so that the three bit-field inserts that build a
            // nibble {nRo, nCo, nR, nC} (see fast_rect) serve four cells instead of two.
#pragma unroll
            for (int p2 = 0; p2 < 2; p2++) {
                // (v_perm selectors 8 .. 11 replicate a sign bit over the byte: four clean 0x00 / 0xff masks, merged by three bit-field inserts
                // into the nibble {nRo, nCo, nR, nC} in BOTH halves of every byte -- round 5: 7.5 instead of 11 instructions per register pair)
                const uint32_t pC = (uint32_t)__builtin_amdgcn_perm((int)sC[2 * p2 + 1], (int)sC[2 * p2], 0x0b0a0908), pR = (uint32_t)__builtin_amdgcn_perm((int)sR[2 * p2 + 1], (int)sR[2 * p2], 0x0b0a0908);
                const uint32_t pCo = (uint32_t)__builtin_amdgcn_perm((int)sCo[2 * p2 + 1], (int)sCo[2 * p2], 0x0b0a0908), pRo = (uint32_t)__builtin_amdgcn_perm((int)sRo[2 * p2 + 1], (int)sRo[2 * p2], 0x0b0a0908);
                const uint32_t lo2 = bfi(0x55555555u, pC, pR), hi2 = bfi(0x55555555u, pCo, pRo);
                const uint32_t nib = bfi(0x33333333u, lo2, hi2);
                if (j & 1) tacc[2 * (j >> 1) + p2] = (int)bfi(0xF0F0F0F0u, nib, (uint32_t)tacc[2 * (j >> 1) + p2]);
                else tacc[2 * (j >> 1) + p2] = (int)nib;   // (the high nibbles are the odd column's, written next)
            }
        }
        // the last cell of the column feeds the orthogonal border (scan_block.rs:1213-1214): two columns to a register
        if (j & 1) { nvD[j >> 1] = __builtin_amdgcn_perm(dn[3], holdD, 0x07060302); nvR[j >> 1] = __builtin_amdgcn_perm(r[3], holdR, 0x07060302); }
        else { holdD = dn[3]; holdR = r[3]; }
    }
    if (TRACE) {
#ifndef MQ_X_NOTRSTORE
        if (store) { *(int4*)tout = int4{tacc[0], tacc[1], tacc[2], tacc[3]}; *(int4*)(tout + 4) = int4{tacc[4], tacc[5], tacc[6], tacc[7]}; }
#else
        asm volatile("" :: "v"(tacc[0]), "v"(tacc[1]), "v"(tacc[2]), "v"(tacc[3]), "v"(tacc[4]), "v"(tacc[5]), "v"(tacc[6]), "v"(tacc[7]));
#endif
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {   // shift_and_offset (scan_block.rs:1040-1061): 8 entries = one lane
        Pd[k] = slot_shl1_keep<SL>(nvD[k], pd[k], l); Pr[k] = slot_shl1_keep<SL>(nvR[k], pr[k], l);
        Ad[k] = d[k]; Ac[k] = c[k];
    }
    {   // max of the first 8 entries of both borders (row lane 0), to every lane of the slot (scan_block.rs:1020-1022)
        const int ma = vmax(vmax(d[0], d[1]), vmax(d[2], d[3])), mb = vmax(vmax(Pd[0], Pd[1]), vmax(Pd[2], Pd[3]));
        const s16x2 sa = as_s(ma), sb = as_s(mb);
        const int xa = vmax(ma, as_i(s16x2{sa.y, sa.x})), xb = vmax(mb, as_i(s16x2{sb.y, sb.x}));
        const s16x2 rr = as_s(slot_first<SL>(__builtin_amdgcn_perm(xb, xa, 0x05040100)));
        o.act_max8 = rr.x; o.pas_max8 = rr.y;
    }
    const int mm = vmax(vmax(dmax[0], dmax[1]), vmax(dmax[2], dmax[3]));
    const int m32 = max(mm & 0xffff, (int)((uint32_t)mm >> 16));   // halves are >= 0 (D_max starts at MIN = 0)
    o.mx = slot_last<SL>(SL == 16 ? wave_prefix_max16(m32) : (SL == 64 ? wave_prefix_max(m32) : wave_prefix_max32(m32)));
    if constexpr (TRACE && SPM == 1) { zout[0] = (int)~zacc[0]; zout[1] = (int)~zacc[1]; }
}

// Per wave and slot, in an L2-resident arena (BatchParams::big):
//   two buffers that alternate between "the state before the step in flight" (written before every step: a step that is rolled
//   back, or that a slot may not take, leaves the slot's registers behind as garbage and the pair's state here) and "the
//   checkpoint" (the state before the last improving step: on an improving step the two simply change roles):
//           A_d, A_c, P_d, P_r (16 lanes x 16 bytes each, relative to the direction in the scalars) + 16 scalars:
//           [0] 1 = state before a step / 0 = checkpoint as the reference keeps it (after the step), [1] i, [2] j, [3] offset,
//           [4] trace words, [5] rectangles (flag 1: before the step), [6] direction, [7] off_add, [8] corner
//   and a record of 32 scalars: the slot's whole driver state while the wave is in solo mode (the slots' registers are live only
//   inside the loop of shift steps: around a solo episode every slot is stored / loaded, so that the solo code -- the whole
//   per-pair driver, which needs every register -- and the step loop are allocated independently of each other)
// (sizes: MQ_BUF_BYTES / MQ_SLOT_BYTES / MQ_WAVE_BYTES in ba_params.h)
enum { MR_PAIR = 0, MR_BEST_I, MR_BEST_J, MR_CELLS_LO, MR_CELLS_HI, MR_BUDGET, MR_STATUS, MR_TSLOT, MR_SI, MR_SJ, MR_DIR, MR_PREV_DIR, MR_OFF, MR_OFF_MAX,
       MR_BEST_MAX, MR_Y_DROP, MR_X_ITER, MR_D_CORNER, MR_NSTEPS, MR_TRACE_TOP, MR_NBLOCKS, MR_SEL, MR_WORDS };
__device__ __forceinline__ int mq_load(const char* p) { return __hip_atomic_load((const int*)p, BA_RLX_AGENT); }   // past the L1: the wave reads back its own stores

#ifndef MQ_PARTIAL_MB
// Slots of this many cells and more go on stepping when the wave cannot refill its empty ones, instead of finishing the live pairs one at a time in solo mode
// (round 6). A 256-cell slot's plain steps in the per-pair driver are its generic rectangles and its shell -- the loop of steps with one of two slots live is
// faster (22 kbp pairs at 256..4096, same box: 600 pairs 54.6 -> 34.9 ms, per-pair kernel 45.4; 2500 pairs 55.7 -> 37.2, per-pair 77.4; 13 kbp at 256..2048: 30 k
// pairs 133.3 -> 125.7); a 128-cell slot's are the register path, which beats a quarter-full loop of steps (config 3, 100 k pairs 157.7 -> 161.9 ms, 12.5 k 30.1 ->
// 31.9: tools/dev/partial_ab.sh, partial_ab2.sh).
#define MQ_PARTIAL_MB 256
#endif
#ifndef MQ_WAVES_EU
#define MQ_WAVES_EU 4   // (waves per SIMD the kernel is compiled for. Tried, round 5: 2 = 256 registers, no spills -- config 3 163.6 -> 205.1 ms, 25 k pairs 55.6 -> 64.1: two waves do not fill the vector unit)
#endif
// WPW / EU (round 6): waves per workgroup and waves per SIMD the instantiation is compiled and launched for. The batch geometry is the host's choice
// (ba_host.cpp batch_build): eight-wave workgroups at four waves per SIMD for batches of many rounds; four-wave workgroups -- one wave per SIMD,
// EU of them per CU -- at three or two waves per SIMD (168 / 256 registers) for batches whose pairs fill the slots of that many waves about once.
template <int PMAX, int KIND, bool TRACE, bool XDROP, int SPM = 0, int MBP = 128, int WPW = WAVES_PER_WG, int EU = MQ_WAVES_EU>
__global__ void __launch_bounds__(WPW * 64, (PMAX >= 16 ? 2 : EU)) k_multi(const BatchParams bp) {
    // a slot rectangle's words on the trace stack (LOCAL_START: the zero mask takes 32 words behind the 128 trace words; the stack advances as the
    // per-pair kernel's does -- a mask word per trace word, Aligner::add_block); the walkers' records and mode bits (the special modes: room for the
    // zero-mask bits, the early stops of scan_block.rs:1597-1611)
    // MBP (round 6): the slots' block size -- 128 cells (four slots of 16 lanes: MQ_B) or 256 (two slots of 32 lanes, the pairs percent_len starts at 256 cells:
    // reads above 12.8 kbp, lib.rs:109-111). The same code: a slot's lanes, its buffers and its rectangles scale with it.
    constexpr uint32_t MB = (uint32_t)MBP;
    constexpr int SL = MBP / 8, NSLOT = 64 / SL;
    constexpr uint32_t ALLM = (1u << NSLOT) - 1u;                                                     // every slot live
    constexpr uint32_t ARR = 2u * MB, BUFL = 4u * ARR, BUFA = BUFL + 64u, SLOTA = 2u * BUFA + 128u;   // bytes: a border array of a slot, a state buffer in LDS / in the arena, a slot in the arena
    static_assert((MBP == 128 || MBP == 256 || MBP == 512) && NSLOT * SLOTA <= MQ_WAVE_BYTES && NSLOT * 2u * BUFL <= MQ_LDS_SCALARS && (MBP == 128 || SPM == 0), "k_multi slot geometry");
    static_assert(MBP != 128 || (BUFA == MQ_BUF_BYTES && SLOTA == MQ_SLOT_BYTES), "ba_params.h");
    constexpr uint32_t MQ_TW = (STEP * MB / 8) * (SPM == 1 ? 2u : 1u);
    constexpr int TB_LB = SPM ? (int)TB_LANE_BYTES_LOC : (int)TB_LANE_BYTES_L2;
    constexpr uint32_t TB_MASK = SPM ? ~0u : (uint32_t)F_CIGAR_EQ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id(), l = lane & (SL - 1), g = lane / SL;
    const int wave = uni((int)threadIdx.x >> 6);
    {   // workgroup-shared scoring table, as in k_align
        char* tab = smem;
        if (KIND == KIND_NUC) {
            nuc_table_fill(tab, bp.matrix, (int)threadIdx.x, WPW * 64);   // (layout: ba_device.hpp nuc_key_off)
        } else {
            const int nbytes = KIND == KIND_AA ? 27 * 32 : 2;
            for (int k = (int)threadIdx.x; k < nbytes; k += WPW * 64) tab[k] = (char)bp.matrix[k];
        }
    }
    __syncthreads();
    constexpr uint32_t LCLS = (uint32_t)PMAX * 128u;
    constexpr uint32_t ab = lds_array_bytes_h(LCLS);
    char* base = smem + lds_table_bytes_h(KIND) + (uint32_t)wave * mq_wave_bytes_h(LCLS);
    WaveLds L;
    L.D_col = (short*)(base + 0 * ab); L.C_col = (short*)(base + 1 * ab);
    L.D_row = (short*)(base + 2 * ab); L.R_row = (short*)(base + 3 * ab);
    L.misc = (short*)(base + 4 * ab);
    L.table = smem;
    const int gx = bp.gap_extend;
    const uint32_t stride = bp.tb_stride;
    const bool batch_traceback = TRACE && stride > 0;
#if defined(BA_TIMING) || defined(BA_ENDHIST)
    if (bp.prof && blockIdx.x == 0 && wave == 1 && is_lane(0)) atomicMax(bp.prof + 43, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
#ifdef BA_TIMING
    unsigned long long t_solo = 0, t_quad = 0, t_wait = 0, n_solo = 0, n_quad = 0;
#endif
    if (batch_traceback && wave == 0 && blockIdx.x % stride == 0) {
        if (!(bp.flags & 0x800u))   // (development switch: as if the traceback waves were never resident)
        traceback_consumer<TB_LB, 8, 3>(bp, TB_MASK, (unsigned char*)base, 64u, true, 1);   // (8 cells per call: the window of a slot's rectangle is 16 rows x 8 columns; 4: -0.8 %, 12: -2 %. Three records fetched ahead: +0.8 % over two)   // (records in this wave's own region)
#if defined(BA_TIMING) || defined(BA_ENDHIST)
        if (bp.prof && is_lane(0)) atomicMax(bp.prof + 42, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
        return;
    }
    const uint32_t cons_before = batch_traceback ? (blockIdx.x + stride - 1) / stride + (blockIdx.x % stride == 0 ? 1u : 0u) : 0u;
    const uint32_t fill_wave = blockIdx.x * WPW + (uint32_t)wave - cons_before;
    const uint32_t max_size = bp.max_size;
    const bool keep_pre = XDROP || MB < max_size;   // a slot keeps the state before its last improving step
    char* const wave_mem = (char*)bp.big + (uint64_t)fill_wave * MQ_WAVE_BYTES;

    // (values derived again after the per-pair driver instead of being kept across it: laundered so that they are not hoisted)
    auto coldp_big = [&]() { const __attribute__((address_space(4))) BatchParams* p = (const __attribute__((address_space(4))) BatchParams*)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(p)); return p->big; };
    auto fill_wave_of = [&]() { uint32_t b = blockIdx.x, w = (uint32_t)wave; asm volatile("" : "+s"(b), "+s"(w)); const uint32_t st = ((const __attribute__((address_space(4))) BatchParams*)__builtin_amdgcn_kernarg_segment_ptr())->tb_stride;
                                const uint32_t cb = (TRACE && st > 0) ? (b + st - 1) / st + (b % st == 0 ? 1u : 0u) : 0u; return b * WPW + w - cb; };
    uint32_t live_m = 0, pend_m = 0;   // wave-uniform: bit s = slot s holds a pair / its pair has to go through solo mode
    uint32_t w_next = 0, w_end = 0;
    bool more = true, drain = false, fin_counted = false;
    // The end of the batch (round 4): a wave whose slots are filled for the last time (the pool is closed for them) takes its live slots solo,
    // one after the other -- and offers the ones it is not working on to waves that have nothing left (mq_donate[its id], bit s; a slot's
    // state in the arena is a complete resumable record). `don` (bits 8 / 9 of the wave's own word): bit 0 this wave has been counted in mq_donate[waves] (it will offer nothing
    // more), bit 1 its word holds offers; donor_id: the wave whose slot `solo` is (a slot taken over), else this wave.
    // (neither is kept in a register across the loop of steps: the id is derived again, the bits live in the wave's own word, bits 8 / 9)
#define don_final (bp.mq_donate + gridDim.x * WPW)   /* fill waves that will offer nothing more */
#define don_offers (don_final + 32)                            /* (its own cache line) offers outstanding */
    // a slot's memory in the arena: `id` = blockIdx * WPW + wave of the wave that owns it
    auto slot_mem_of = [&](uint32_t id, uint32_t s_i) {
        const uint32_t db = id / WPW;
        const uint32_t dcb = batch_traceback ? (db + stride - 1) / stride + (db % stride == 0 ? 1u : 0u) : 0u;
        return (char*)bp.big + (uint64_t)(id - dcb) * MQ_WAVE_BYTES + s_i * SLOTA;
    };
    // one pass over the waves' words (mq_donate[waves + 32]: offers outstanding, a hint that saves the pass): take one offered slot
    auto claim_offer = [&](uint32_t my_id, uint32_t& id_out, int& s_out) {
        uint32_t avail = 0;
        if (is_lane(0)) avail = __hip_atomic_load(don_offers, BA_RLX_AGENT);
        if ((int)uni((int)avail) <= 0) return false;
        const uint32_t n_ids = gridDim.x * WPW;
        for (uint32_t k0 = 0; k0 < n_ids; k0 += 64) {
            const uint32_t idx = ((my_id & ~63u) + k0 + (uint32_t)lane) % n_ids;   // (every word at least once, starting near this wave's own; the number of waves need not be a multiple of 64)
            const uint32_t v = __hip_atomic_load(bp.mq_donate + idx, BA_RLX_AGENT) & ALLM;
            unsigned long long m = __ballot(v != 0u);
            while (m) {
                const int src = __builtin_ctzll(m);
                const uint32_t vv = (uint32_t)__builtin_amdgcn_readlane((int)v, src), id = (uint32_t)__builtin_amdgcn_readlane((int)idx, src);
                const int s_i = __builtin_ctz(vv);
                uint32_t old = 0;
                if (is_lane(0)) {
                    old = __hip_atomic_fetch_and(bp.mq_donate + id, ~(1u << s_i), BA_RLX_AGENT);
                    if ((old >> s_i) & 1u) __hip_atomic_fetch_sub(don_offers, 1u, BA_RLX_AGENT);
                }
                old = (uint32_t)uni((int)old);
                if ((old >> s_i) & 1u) { id_out = id; s_out = s_i; return true; }
                m &= m - 1;   // (taken meanwhile; the word's other offers are seen by the next pass)
            }
        }
        return false;
    };
    // an empty slot of this wave takes a slot that another wave offers (its state in the arena is a complete resumable record: a 2.3 KB copy)
    auto try_refill = [&](uint32_t my_id) {
        uint32_t id = 0; int s_o = 0;
        if ((bp.flags & 0x1000u) || !claim_offer(my_id, id, s_o)) return false;   // (0x1000 / 0x2000: development switches)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int e = __builtin_ctz(~live_m & ALLM);
        const char* src = slot_mem_of(id, (uint32_t)s_o);
        char* dst = wave_mem + (uint32_t)e * SLOTA;
#pragma unroll
        for (int k = 0; k < (int)(SLOTA / 256u); k++) *(int*)(dst + 256 * k + 4 * lane) = mq_load(src + 256 * k + 4 * lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the copy is in the L2 before this wave reads it back)
        live_m |= 1u << e;
        return true;
    };
    auto my_id_of = [&]() { uint32_t b = blockIdx.x, w = (uint32_t)wave; asm volatile("" : "+s"(b), "+s"(w)); return b * WPW + w; };

    for (;;) {
        // ================= solo mode: one pair at a time on all 64 lanes, the slots' state in memory. Whose turn? a slot whose step
        // was rolled back, else a new pair for an idle slot, else -- the batch is running out and fewer than four slots are live: a
        // step would idle lanes -- a live slot, to its end.
        for (;;) {
            int solo = -1; bool fresh = false, to_end = false;
            uint32_t new_pair = 0, donor_id = ~0u;
            if (pend_m) solo = __builtin_ctz(pend_m);
            else if (more && (drain ? live_m == 0u : live_m != ALLM)) {
                if (w_next == w_end) {
                    uint32_t v = 0;
                    if (is_lane(0)) v = atomicAdd(bp.work_counter, bp.work_chunk);
                    w_next = (uint32_t)uni((int)v);
                    if (w_next >= bp.n) { more = false; w_next = w_end = 0; continue; }
                    w_end = min(w_next + bp.work_chunk, bp.n);
                }
                new_pair = w_next++; solo = __builtin_ctz(~live_m & ALLM); fresh = true;
                to_end = drain;                                      // (the batch is running out: this pair is not for a slot)
                if (new_pair + bp.mq_drain >= bp.n) drain = true;    // from the next pair on
            } else if (live_m && live_m != ALLM && MB < (uint32_t)(MQ_PARTIAL_MB)) {
                // (tried in round 5: a wave with three live slots goes on stepping instead -- three slots take steps about as efficiently as a pair in
                // solo mode --: no difference, 167.6 against 167.0 ms)
                solo = __builtin_ctz(live_m); to_end = true;
                if (TRACE && bp.mq_donate) {   // (the score-only kernels are compiled without the end-of-batch code: see DESIGN.md)
                    const uint32_t my_id = my_id_of();
                    uint32_t don = 0;
                    if (is_lane(0)) don = __hip_atomic_load(bp.mq_donate + my_id, BA_RLX_AGENT);
                    don = (uint32_t)uni((int)don) >> 8;
                    if (!(don & 1u)) {   // (here: the batch has run out, or this wave takes the pairs that are left one at a time -- either way the batch fills none of its slots again)
                        // an empty slot is first filled with a slot that another wave offers: the wave stays at four pairs per step
                        if (try_refill(my_id)) continue;
                        // nothing on offer: this wave's own slots are, except the one it works on now (solo, to its end)
                        const uint32_t others = live_m & ~(1u << solo);
                        if (others) {
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the slots' records and trace words, before the offer
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        }
                        if (is_lane(0)) {
                            __hip_atomic_store(bp.mq_donate + my_id, others | 0x100u | (others ? 0x200u : 0u), BA_RLX_AGENT);
                            if (others) __hip_atomic_fetch_add(don_offers, (uint32_t)__builtin_popcount(others), BA_RLX_AGENT);
                        }
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the offer before the count: a wave that sees every wave counted has seen every offer)
                        if (is_lane(0)) __hip_atomic_fetch_add(don_final, 1u, BA_RLX_AGENT);
                    } else if (don & 2u) {   // its own offer back, unless somebody has taken it
                        uint32_t old = 0;
                        if (is_lane(0)) {
                            old = __hip_atomic_fetch_and(bp.mq_donate + my_id, ~(1u << solo), BA_RLX_AGENT);
                            if ((old >> solo) & 1u) __hip_atomic_fetch_sub(don_offers, 1u, BA_RLX_AGENT);
                        }
                        old = (uint32_t)uni((int)old);
                        if (!((old >> solo) & 1u)) { live_m &= ~(1u << solo); continue; }
                    }
                }
            } else if (TRACE && bp.mq_donate && !live_m && !more) {
                // nothing left here: take over a slot that another wave offers and run it solo to its end; leave when no wave can offer any more
                const uint32_t my_id = my_id_of();
                if (is_lane(0) && !(__hip_atomic_fetch_or(bp.mq_donate + my_id, 0x100u, BA_RLX_AGENT) & 0x100u)) __hip_atomic_fetch_add(don_final, 1u, BA_RLX_AGENT);
                const uint32_t n_fill = gridDim.x * WPW - (batch_traceback ? (gridDim.x + stride - 1) / stride : 0u);
                bool got = false;
                uint32_t nap = 1;
                for (;;) {
                    uint32_t nf = 0;
                    if (is_lane(0)) nf = __hip_atomic_load(don_final, BA_RLX_AGENT);
                    nf = (uint32_t)uni((int)nf);
                    got = !(bp.flags & 0x2000u) && claim_offer(my_id, donor_id, solo);
                    if (got || nf >= n_fill) break;
                    for (uint32_t z = 0; z < nap; z++) __builtin_amdgcn_s_sleep(127);   // (thousands of waves poll two words: 3 .. 50 us apart)
                    nap = nap < 16u ? nap * 2u : 16u;
                }
                if (!got) { donor_id = ~0u; break; }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                to_end = true;
            }
            else {
                // (a wave of wide slots that goes on stepping with an empty slot the batch cannot fill any more: it will offer nothing -- counted as the waves
                // that finish their pairs solo are, or the idle waves would not turn into the helpers whose walks this wave's trace slots wait for. Without
                // it a per-pair-ranges batch took 25 s: every wait ended by its time-out.)
                if constexpr (TRACE && MB >= (uint32_t)(MQ_PARTIAL_MB)) {
                    if (bp.mq_donate && live_m && live_m != ALLM && !more && !fin_counted) {
                        fin_counted = true;
                        const uint32_t my_id = my_id_of();
                        if (is_lane(0) && !(__hip_atomic_fetch_or(bp.mq_donate + my_id, 0x100u, BA_RLX_AGENT) & 0x100u)) __hip_atomic_fetch_add(don_final, 1u, BA_RLX_AGENT);
                    }
                }
                break;
            }

            BA_TSTAMP(ts_a);
            const FillConsts fc = make_fill_consts(lane, bp.gap_open, gx);   // two cells per lane (as k_align)
            Aligner<PMAX, KIND, TRACE, XDROP, SPM != 0, true, MBP> al(bp, L, fc);
            PairState st{};
            uint32_t s_pair, s_slot;
            char* smem_s = wave_mem + (uint32_t)solo * SLOTA;
            if (donor_id != ~0u) smem_s = slot_mem_of(donor_id, (uint32_t)solo);   // (a slot taken over: the other wave's region of the arena)
            char* const rec = smem_s + 2 * BUFA;
            al.ckpt = bp.ckpt + (uint64_t)(fill_wave + bp.ckpt_wave0) * 8 * bp.max_size;   // (before import_slot: a 256-cell slot's checkpoint goes there)
            if (fresh) {
                s_pair = new_pair;
                // a free trace slot of this wave (a slot is busy from the pair's start until its traceback is done)
                s_slot = fill_wave * bp.slots_per_wave;
                if (batch_traceback) {
                    BA_TSTAMP(tw_a);
                    uint32_t seen = 0, idle_n = 0;
                    for (;;) {   // (never gives up: see Aligner::acquire_slot)
                        uint32_t f = 0, head = 0;
                        if ((uint32_t)lane < bp.slots_per_wave) f = __hip_atomic_load(bp.slot_free + s_slot + lane, BA_RLX_AGENT);
                        if (is_lane(0)) head = __hip_atomic_load(bp.tb_ctrl + 32, BA_RLX_AGENT);
                        const unsigned long long fm = __ballot(f != 0);
                        if (fm) {
                            s_slot += (uint32_t)__builtin_ctzll(fm);
                            if (is_lane(0)) __hip_atomic_store(bp.slot_free + s_slot, 0u, BA_RLX_AGENT);
                            break;
                        }
                        head = (uint32_t)uni((int)head);
                        if (head != seen) { seen = head; idle_n = 0; }
                        // nobody has taken a traceback for a few milliseconds: walk one here (see traceback_help_one; the wave's LDS region
                        // is free: the slots' buffers are in the arena while the wave is in solo mode)
                        else if ((++idle_n & 2047u) == 0 && traceback_help_one<TB_LB>(bp, TB_MASK, (unsigned char*)base)) idle_n = 0;
                        __builtin_amdgcn_s_sleep(64);
                    }
#ifdef BA_TIMING
                    t_wait += __builtin_amdgcn_s_memtime() - tw_a;
#endif
                } else if (TRACE) {
                    // (no hand-off: the pair's traceback is walked by this wave before the slot is reused; the four slots of the wave
                    // still need a trace slot each while their pairs are live)
                    s_slot += (uint32_t)solo;
                }
            } else {
                const int rv = lane < MR_WORDS ? mq_load(rec + 4 * lane) : 0;   // the slot's record: lane k holds word k
#define BA_W(v, k) __builtin_amdgcn_readlane(v, k)
                const uint32_t s_sel = (uint32_t)BA_W(rv, MR_SEL);
                const char* live_b = smem_s + (s_sel ^ 1u) * BUFA; const char* ck_b = smem_s + s_sel * BUFA;
                const int cv = lane < 16 ? mq_load(ck_b + BUFL + 4 * lane) : 0;   // the checkpoint's scalars
                s_pair = (uint32_t)BA_W(rv, MR_PAIR); s_slot = (uint32_t)BA_W(rv, MR_TSLOT);
                st.si = (uint32_t)BA_W(rv, MR_SI); st.sj = (uint32_t)BA_W(rv, MR_SJ); st.dir = BA_W(rv, MR_DIR); st.prev_dir = BA_W(rv, MR_PREV_DIR);
                st.off = BA_W(rv, MR_OFF); st.off_max = BA_W(rv, MR_OFF_MAX); st.best_max = BA_W(rv, MR_BEST_MAX);
                st.y_drop_iter = (uint32_t)BA_W(rv, MR_Y_DROP); st.x_drop_iter = BA_W(rv, MR_X_ITER); st.D_corner = BA_W(rv, MR_D_CORNER);
                st.trace_top = (uint32_t)BA_W(rv, MR_TRACE_TOP); st.nblocks = (uint32_t)BA_W(rv, MR_NBLOCKS);
                const uint32_t ns = (uint32_t)BA_W(rv, MR_NSTEPS);
                st.best_i = (uint32_t)BA_W(rv, MR_BEST_I); st.best_j = (uint32_t)BA_W(rv, MR_BEST_J);
                st.cells = ((unsigned long long)(uint32_t)BA_W(rv, MR_CELLS_LO) | ((unsigned long long)(uint32_t)BA_W(rv, MR_CELLS_HI) << 32)) + (unsigned long long)ns * (STEP * MB);
                const uint32_t bud = (uint32_t)BA_W(rv, MR_BUDGET);
                st.step_budget = bud > ns ? bud - ns : 1u;
                st.status = (uint32_t)BA_W(rv, MR_STATUS);
                const int flag = BA_W(cv, 0);
                const bool ck_pre = flag != 0;
                st.ck_i = (uint32_t)BA_W(cv, 1); st.ck_j = (uint32_t)BA_W(cv, 2); st.ck_off = BA_W(cv, 3);
                st.ck_tt = (uint32_t)BA_W(cv, 4) + (flag ? MQ_TW : 0u); st.ck_nb = (uint32_t)BA_W(cv, 5) + (flag ? 1u : 0u);
                const int ck_dir = BA_W(cv, 6), ck_offadd = BA_W(cv, 7), ck_corner = BA_W(cv, 8);
#undef BA_W
                {   // the borders: lane l's 16 bytes of each array (lanes 0 .. 15), into the canonical order D_col, C_col, D_row, R_row
                    int reg[16], ckr[16];
                    const bool lr = st.dir == DIR_RIGHT, cr = ck_dir == DIR_RIGHT;
                    // (lanes 0 .. SL - 1 hold the slot's cells: here l = lane for them)
                    const uint32_t oAd = (lr ? 0u : 2 * ARR) + l * 16, oAc = (lr ? ARR : 3 * ARR) + l * 16, oPd = (lr ? 2 * ARR : 0u) + l * 16, oPr = (lr ? 3 * ARR : ARR) + l * 16;
                    const uint32_t cAd = (cr ? 0u : 2 * ARR) + l * 16, cAc = (cr ? ARR : 3 * ARR) + l * 16, cPd = (cr ? 2 * ARR : 0u) + l * 16, cPr = (cr ? 3 * ARR : ARR) + l * 16;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        reg[k] = mq_load(live_b + oAd + 4 * k); reg[4 + k] = mq_load(live_b + oAc + 4 * k); reg[8 + k] = mq_load(live_b + oPd + 4 * k); reg[12 + k] = mq_load(live_b + oPr + 4 * k);
                        ckr[k] = mq_load(ck_b + cAd + 4 * k); ckr[4 + k] = mq_load(ck_b + cAc + 4 * k); ckr[8 + k] = mq_load(ck_b + cPd + 4 * k); ckr[12 + k] = mq_load(ck_b + cPr + 4 * k);
                    }
                    al.import_slot(reg, ckr, s_pair, ck_pre || MB < max_size, ck_pre, ck_dir, ck_offadd, ck_corner, st.ck_i, st.ck_j, st.best_i, st.best_j, st.ck_off);
                }
            }
            al.trace = bp.trace_arena + (uint64_t)s_slot * bp.trace_stride;
            al.blocks = bp.blocks + (uint64_t)s_slot * bp.blocks_stride;
            // The per-pair driver needs every scalar register: what the wave keeps across it is parked in the lanes of one VGPR
            // (the compiler would otherwise spill and reload these values inside the driver's inner loops).
            int keepv = 0;
            park<0>(keepv, (int)live_m); park<1>(keepv, (int)pend_m); park<2>(keepv, (int)w_next); park<3>(keepv, (int)w_end);
            park<4>(keepv, (more ? 1 : 0) | (fresh ? 2 : 0) | (to_end ? 4 : 0) | (drain ? 8 : 0)); park<5>(keepv, solo); park<6>(keepv, (int)s_pair); park<7>(keepv, (int)s_slot);
            BA_TSTAMP(tr_a);
            // (a pair in solo mode keeps the wave's other three slots waiting: its dependent chain goes first among the SIMD's waves -- config 3: 193.2 -> 185.2 ms)
            __builtin_amdgcn_s_setprio(MQ_SOLO_PRIO);
            st = al.run(s_pair, s_slot, batch_traceback, nullptr, fresh ? MM_FRESH : MM_RESUME, st, !to_end, !fresh && !to_end);
            __builtin_amdgcn_s_setprio(0);
            live_m = (uint32_t)unpark<0>(keepv); pend_m = (uint32_t)unpark<1>(keepv); w_next = (uint32_t)unpark<2>(keepv); w_end = (uint32_t)unpark<3>(keepv);
            { const int fl = unpark<4>(keepv); more = fl & 1; fresh = fl & 2; to_end = fl & 4; drain = fl & 8; }
            solo = unpark<5>(keepv); s_pair = (uint32_t)unpark<6>(keepv); s_slot = (uint32_t)unpark<7>(keepv);
            char* const smem_s2 = (char*)coldp_big() + (uint64_t)fill_wave_of() * MQ_WAVE_BYTES + (uint32_t)solo * SLOTA;
            char* const rec2 = smem_s2 + 2 * BUFA;
#ifdef BA_TIMING
            {   // development: the per-pair driver's phase timers (as k_align), + 45 = run(), 56 / 57 / 58 = run() of a new pair / a pair back from its slot / a pair taken to its end
                const unsigned long long tr_b = __builtin_amdgcn_s_memtime();
                al.prof[45] += tr_b - tr_a; al.prof[46] += fresh ? 1 : 0;
                if (bp.prof && is_lane(0)) {
                    for (int k = 0; k < 48; k++) if (k != 17 && !(k >= 20 && k < 32) && !(k >= 40 && k < 44)) atomicAdd(bp.prof + k, al.prof[k]);
                    atomicAdd(bp.prof + (fresh ? 56 : (to_end ? 58 : 57)), tr_b - tr_a);
                }
            }
#endif
            pend_m &= ~(1u << solo);
            if (st.exited) {   // the pair's next step is a plain shift step at MB cells: into the slot (its memory)
                const bool rgt = st.dir == DIR_RIGHT;
                if (lane < SL) {
                    const int l8 = 8 * l;
                    const int4 a_d = *(const int4*)((rgt ? L.D_col : L.D_row) + l8), a_c = *(const int4*)((rgt ? L.C_col : L.R_row) + l8);
                    const int4 p_d = *(const int4*)((rgt ? L.D_row : L.D_col) + l8), p_r = *(const int4*)((rgt ? L.R_row : L.C_col) + l8);
                    char* b1 = smem_s2 + BUFA;   // the state at the top of the loop: buffer 1 (sel = 0)
                    *(int4*)(b1 + l * 16) = a_d; *(int4*)(b1 + ARR + l * 16) = a_c; *(int4*)(b1 + 2 * ARR + l * 16) = p_d; *(int4*)(b1 + 3 * ARR + l * 16) = p_r;
                    if (MB < max_size) {   // the checkpoint as the reference keeps it (after its step): buffer 0
                        char* b0 = smem_s2;
                        *(int4*)(b0 + l * 16) = *(const int4*)(L.D_col + MB + l8); *(int4*)(b0 + ARR + l * 16) = *(const int4*)(L.C_col + MB + l8);
                        *(int4*)(b0 + 2 * ARR + l * 16) = *(const int4*)(L.D_row + MB + l8); *(int4*)(b0 + 3 * ARR + l * 16) = *(const int4*)(L.R_row + MB + l8);
                    }
                }
                if (is_lane(0)) {
                    char* b0 = smem_s2;
                    *(int4*)(b0 + BUFL) = int4{0, (int)st.ck_i, (int)st.ck_j, st.ck_off}; *(int4*)(b0 + BUFL + 16) = int4{(int)st.ck_tt, (int)st.ck_nb, DIR_RIGHT, 0};
                    *(int*)(b0 + BUFL + 32) = 0;
                    int* rc = (int*)rec2;
                    rc[MR_PAIR] = (int)s_pair; rc[MR_BEST_I] = (int)st.best_i; rc[MR_BEST_J] = (int)st.best_j; rc[MR_CELLS_LO] = (int)(uint32_t)st.cells; rc[MR_CELLS_HI] = (int)(uint32_t)(st.cells >> 32);
                    rc[MR_BUDGET] = (int)st.step_budget; rc[MR_STATUS] = (int)st.status; rc[MR_TSLOT] = (int)s_slot; rc[MR_SI] = (int)st.si; rc[MR_SJ] = (int)st.sj; rc[MR_DIR] = st.dir;
                    rc[MR_PREV_DIR] = st.prev_dir; rc[MR_OFF] = st.off; rc[MR_OFF_MAX] = st.off_max; rc[MR_BEST_MAX] = st.best_max; rc[MR_Y_DROP] = (int)st.y_drop_iter;
                    rc[MR_X_ITER] = st.x_drop_iter; rc[MR_D_CORNER] = st.D_corner; rc[MR_NSTEPS] = 0; rc[MR_TRACE_TOP] = (int)st.trace_top; rc[MR_NBLOCKS] = (int)st.nblocks; rc[MR_SEL] = 0;
                }
                lds_sync();
                live_m |= 1u << solo;
            } else live_m &= ~(1u << solo);
#ifdef BA_TIMING
            t_solo += __builtin_amdgcn_s_memtime() - ts_a; n_solo++;
#endif
        }
        if (!live_m) break;

        // ================= the slots: registers from memory, shift steps until a slot needs solo mode, registers to memory
        {
            BA_TSTAMP(tq_a);
            char* const slot_mem = wave_mem + (uint32_t)g * SLOTA;   // this lane's slot
            const int x_drop = bp.x_drop;
            FillConsts fq;   // only the three gap constants (wave-uniform)
            fq.go2 = splat(bp.gap_open); fq.ge2 = splat(gx); fq.ome2 = splat(clamp16(bp.gap_open - gx));
            MultiConsts mc = make_multi_consts(l, gx);  // eight cells per lane
            const uint64_t tcap64 = bp.trace_stride, bcap64 = bp.blocks_stride;
            const uint32_t tcap = (uint32_t)(tcap64 < 0x7fffffffull ? tcap64 : 0x7fffffffull), bcap = (uint32_t)(bcap64 < 0x7fffffffull ? bcap64 : 0x7fffffffull);
            // ---- slot state (row-uniform, replicated over the slot's lanes)
            const bool live = (live_m >> g) & 1u;   // (all four, except when the batch holds fewer pairs than the wave has slots)
            const char* rec = slot_mem + 2 * BUFA;
#define BA_R(k) (live ? mq_load(rec + 4 * (k)) : 0)
            const uint32_t pair = (uint32_t)BA_R(MR_PAIR), tslot = (uint32_t)BA_R(MR_TSLOT);
            uint32_t si = (uint32_t)BA_R(MR_SI), sj = (uint32_t)BA_R(MR_SJ), y_drop = (uint32_t)BA_R(MR_Y_DROP);
            uint32_t trace_top = (uint32_t)BA_R(MR_TRACE_TOP), nblocks = (uint32_t)BA_R(MR_NBLOCKS), sel = (uint32_t)BA_R(MR_SEL);
            int dir = BA_R(MR_DIR), prev_dir = BA_R(MR_PREV_DIR), off = BA_R(MR_OFF), off_max = BA_R(MR_OFF_MAX), best_max = BA_R(MR_BEST_MAX);
            int x_iter = BA_R(MR_X_ITER), D_corner = BA_R(MR_D_CORNER);
#undef BA_R
            int A_d[4], A_c[4], P_d[4], P_r[4];   // borders: (A) along the vector axis of the step at `dir`, (P) orthogonal
            {
                const char* b = slot_mem + (sel ^ 1u) * BUFA + l * 16;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    A_d[k] = live ? mq_load(b + 4 * k) : 0; A_c[k] = live ? mq_load(b + ARR + 4 * k) : 0;
                    P_d[k] = live ? mq_load(b + 2 * ARR + 4 * k) : 0; P_r[k] = live ? mq_load(b + 3 * ARR + 4 * k) : 0;
                }
            }
            // While the slots run, the two buffers of every slot live in this wave's LDS region (the solo borders' space): the checkpoint
            // (buffer `sel`) comes from the arena now and goes back after the loop; the other one is rewritten before every step.
            char* const lbuf = base + (uint32_t)g * (2u * BUFL);                 // this slot's buffers: + which * BUFL
            int* const lsc = (int*)(base + MQ_LDS_SCALARS) + (uint32_t)g * MQ_LSC_INTS;   // their scalars: + which * 9; + 20: eight constants of the slot's pair
            lds_sync();
            // (score-only kernels have the registers: there the values stay where they were -- 119.8 against 125.5 ms at config 3 without traceback)
            const unsigned long long qa_r = TRACE ? 0ull : (unsigned long long)(bp.pool + (live ? bp.q_off[pair] : 0ull)), ra_r = TRACE ? 0ull : (unsigned long long)(bp.pool + (live ? bp.r_off[pair] : 0ull));
            const uint32_t qlen_r = (!TRACE && live) ? bp.q_len[pair] : 0u, rlen_r = (!TRACE && live) ? bp.r_len[pair] : 0u;
            if (TRACE && l == 0) {
                // what a step needs of its pair only at its start -- the two sequence images and their lengths, the trace slot -- is read from
                // LDS in every step instead of living in seven registers through the columns
                const unsigned long long qa = (unsigned long long)(bp.pool + (live ? bp.q_off[pair] : 0ull)), ra = (unsigned long long)(bp.pool + (live ? bp.r_off[pair] : 0ull));
                *(int4*)(lsc + 20) = int4{(int)(uint32_t)qa, (int)(uint32_t)(qa >> 32), (int)(uint32_t)ra, (int)(uint32_t)(ra >> 32)};
                *(int4*)(lsc + 24) = int4{live ? (int)bp.q_len[pair] : 0, live ? (int)bp.r_len[pair] : 0, (int)tslot, 0};
            }
            if (live) {
                const char* cb = slot_mem + sel * BUFA;
                char* d = lbuf + sel * BUFL + l * 16;
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    const char* sp = cb + a * ARR + l * 16;
                    *(int4*)(d + a * ARR) = int4{mq_load(sp), mq_load(sp + 4), mq_load(sp + 8), mq_load(sp + 12)};
                }
                if (l < 9) lsc[sel * 9u + l] = mq_load(cb + BUFL + 4 * l);
            }
            lds_sync();
            // sequence bytes of the next step, fetched one step ahead for both possible directions
            uint2 pf_qv = {0, 0}, pf_rv = {0, 0}, pf_qc = {0, 0}, pf_rc = {0, 0}; bool pf_ok = false;
            bool leave = false;
            // the slot's registers and the scalars of the step at the top into the buffer `which` (see above)
            auto stage = [&](uint32_t which, uint32_t s_i, uint32_t s_j, int s_off, uint32_t s_tt, uint32_t s_nb, int s_dir, int s_offadd, int s_corner) {
                char* b = lbuf + which * BUFL + l * 16;
                *(int4*)(b) = int4{A_d[0], A_d[1], A_d[2], A_d[3]}; *(int4*)(b + ARR) = int4{A_c[0], A_c[1], A_c[2], A_c[3]};
                *(int4*)(b + 2 * ARR) = int4{P_d[0], P_d[1], P_d[2], P_d[3]}; *(int4*)(b + 3 * ARR) = int4{P_r[0], P_r[1], P_r[2], P_r[3]};
                if (l == 0) {
                    int* sc = lsc + which * 9u;
                    sc[0] = 1; sc[1] = (int)s_i; sc[2] = (int)s_j; sc[3] = s_off; sc[4] = (int)s_tt; sc[5] = (int)s_nb; sc[6] = s_dir; sc[7] = s_offadd; sc[8] = s_corner;
                }
            };
            do {
                // ---- the step every live slot is about to take (scan_block.rs:147-246)
                const bool right = dir == DIR_RIGHT;
                int4 cA = {0, 0, 0, 0}, cB = {0, 0, 0, 0};
                if (TRACE) { cA = *(const int4*)(lsc + 20); cB = *(const int4*)(lsc + 24); }
                const uint8_t* const qp = TRACE ? (const uint8_t*)((unsigned long long)(uint32_t)cA.x | ((unsigned long long)(uint32_t)cA.y << 32)) : (const uint8_t*)qa_r;
                const uint8_t* const rp = TRACE ? (const uint8_t*)((unsigned long long)(uint32_t)cA.z | ((unsigned long long)(uint32_t)cA.w << 32)) : (const uint8_t*)ra_r;
                const uint32_t qlen = TRACE ? (uint32_t)cB.x : qlen_r, rlen = TRACE ? (uint32_t)cB.y : rlen_r;
                const uint32_t ri = right ? si : sj, rj = (right ? sj : si) + (MB - STEP);
                const uint32_t lenV = right ? qlen : rlen, lenC = right ? rlen : qlen;
                const bool q_out = si + MB > qlen, r_out = sj + MB > rlen;
                // a step that could break early at the matrix edge (never with X-drop) or that the trace slot has no room for is not a slot's
                bool elig = XDROP || ri + MB <= lenV || rj + STEP <= lenC;
                if (TRACE) elig = elig && nblocks < bcap && trace_top + MQ_TW + 64 <= tcap;
                leave = live && !elig;
                const bool run = live && !leave;
                const int off_n = off_max;
                const int off_add = sat16(off - off_n);
                const int corner = (prev_dir != dir && prev_dir != DIR_GROW) ? sat16(D_corner + off_add) : 0;
                // the state before the step: for a slot that leaves (its registers do not survive the step) and for the checkpoint
#ifndef MQ_X_NOSTAGE
                if (live) stage(sel ^ 1u, si, sj, off_n, trace_top, nblocks, dir, off_add, corner);
#endif
                // (the lane's number inside its row: from the hardware's lane count in every step, not from a kernel-wide value -- the allocator
                // kept that one, widened to 64 bits, in scratch memory and reloaded it before the prefetch and before the trace stores, each
                // time behind every outstanding memory operation)
                // (should the allocator keep the column code's per-lane constants in scratch memory after all: reloaded here, before the prefetch
                // is issued, a reload waits for nothing but the previous step's stores)
                asm volatile("" : "+v"(mc.laneKG), "+v"(mc.lanem1KG), "+v"(mc.w0));
                uint32_t l8;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l8));
                l8 = (l8 & (uint32_t)(SL - 1)) * 8u;
                uint2 vb, cbv;
                {
                    const uint8_t* Vp = right ? qp : rp; const uint8_t* Cp = right ? rp : qp;
                    vb = right ? pf_qv : pf_rv; cbv = right ? pf_rc : pf_qc;
                    if (__any(run && !pf_ok)) {   // a slot that has just taken its pair (back)
                        if (run && !pf_ok) {
                            const uint32_t* vp = (const uint32_t*)(Vp + (ri + l8));   // (images are 4-byte aligned, positions multiples of 8)
                            vb.x = vp[0]; vb.y = vp[1];
                            const uint32_t* cp = (const uint32_t*)(Cp + rj);
                            cbv.x = cp[0]; cbv.y = cp[1];
                        }
                    }
                    asm volatile("" : "+v"(vb.x), "+v"(vb.y));   // (consume the old prefetch before the next one is issued: the memory counter is in-order)
                    if (run) {                                    // for the step after this one, whichever way it goes
                        const uint32_t* a = (const uint32_t*)(qp + (si + l8)); const uint32_t* b = (const uint32_t*)(rp + (sj + l8));
                        const uint32_t* cq = (const uint32_t*)(qp + si + MB); const uint32_t* cr = (const uint32_t*)(rp + sj + MB);
                        pf_qv.x = a[0]; pf_qv.y = a[1]; pf_rv.x = b[0]; pf_rv.y = b[1];
                        pf_qc.x = cq[0]; pf_qc.y = cq[1]; pf_rc.x = cr[0]; pf_rc.y = cr[1];
                        pf_ok = true;
                    }
                }
                uint32_t* tw = nullptr;
                if (TRACE) {
                    // (the slot's base address is derived again in every step -- two multiply-adds -- instead of living in a register pair across
                    // the loop: with the pair the allocator spilled it and reloaded it before the stores, behind every outstanding memory operation)
                    uint32_t ts = (uint32_t)cB.z; asm volatile("" : "+v"(ts));
                    tw = bp.trace_arena + (uint64_t)ts * bp.trace_stride + (trace_top + l8);
                    if (run && l == 0) {   // add_block(i, j, width, height, right) in matrix orientation (scan_block.rs:154,204)
                        BlockRec br;
                        br.i = (right ? ri : rj) | 0x80000000u;   // (bit 31: words of 4 cells x 2 columns, see multi_rect)
                        br.j = right ? rj : ri; br.h = (uint16_t)(right ? MB : STEP); br.w = (uint16_t)(right ? STEP : MB);
                        br.trace_base = trace_top | (right ? 0x80000000u : 0u);
                        bp.blocks[(uint64_t)(uint32_t)cB.z * bp.blocks_stride + nblocks] = br;
                    }
                }
                MultiOut o;
                int zw[2] = {0, 0};
                multi_rect<KIND, TRACE, SPM, SL>(smem, fq, mc, l, A_d, A_c, P_d, P_r, vb, cbv.x, cbv.y, corner, off_add, tw, run, o, SPM ? splat(sat16(ZERO - off_n)) : 0,
                                             SPM == 2 && right && ri == 0u && l == 0, zw);
                if (TRACE && SPM == 1 && run) *(int2*)(tw - 8 * l + 128 + 2 * l) = int2{zw[0], zw[1]};   // the rectangle's zero mask behind its 128 trace words

                // ---- what does the step call for? (scan_block.rs:332-558; nothing is committed yet)
                const int right_max = right ? o.act_max8 : o.pas_max8, down_max = right ? o.pas_max8 : o.act_max8;
                const int new_off_max = off_n + o.mx - ZERO;
                const bool improve = new_off_max > best_max;
                const uint32_t new_y = improve ? 0u : y_drop + 1;
                bool stop = q_out && r_out;                                                                   // end of the matrix
                if (XDROP) stop = stop || (!improve && new_off_max < best_max - x_drop && x_iter >= 1);      // X-drop termination
                stop = stop || (!q_out && !r_out && 2 * MB <= max_size && new_y > MB / STEP - 1);        // grow
                const bool commit = run && !stop;
                leave = leave || (run && stop);   // rolled back: the pair's state is what was staged before this step
                {
                    // Straight-line selects, no branch (round 4): as nested conditionals the compiler copied the sixteen border registers three
                    // times per step (before the commit branch, inside it, and for the exchange: ~64 moves of 1156 instructions).
                    const bool imp = commit && improve;
                    if (keep_pre) sel ^= imp ? 1u : 0u;   // the state staged before this step is the checkpoint now (and yields the location of the maximum)
                    best_max = imp ? new_off_max : best_max;
                    off = commit ? off_n : off; off_max = commit ? new_off_max : off_max; y_drop = commit ? new_y : y_drop;
                    prev_dir = commit ? dir : prev_dir; D_corner = commit ? o.corner_new : D_corner;
                    if (TRACE) { trace_top += commit ? MQ_TW : 0u; nblocks += commit ? 1u : 0u; }
                    if (XDROP) x_iter = commit ? ((new_off_max < best_max - x_drop) ? x_iter + 1 : 0) : x_iter;
                    const bool go_down = r_out || (!q_out && down_max > right_max);   // forced at the matrix edge, else greedy (ties -> right)
                    si += (commit && go_down) ? (uint32_t)STEP : 0u; sj += (commit && !go_down) ? (uint32_t)STEP : 0u;
                    const int ndir = go_down ? DIR_DOWN : DIR_RIGHT;
                    // the borders change roles with the direction (whatever a slot that does not commit holds is not read again: its state is
                    // what was staged before the step)
                    // (sixteen v_cndmask on one lane mask -- round 5: left to itself the compiler made two exec-masked regions of ~30 moves)
                    const unsigned long long swap_m = __ballot(ndir != dir);
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int td = A_d[k], tc = A_c[k];
                        A_d[k] = sel_mask(swap_m, P_d[k], td); A_c[k] = sel_mask(swap_m, P_r[k], tc); P_d[k] = sel_mask(swap_m, td, P_d[k]); P_r[k] = sel_mask(swap_m, tc, P_r[k]);
                    }
                    dir = commit ? ndir : dir;
                }
#ifdef BA_TIMING
                n_quad++;
#endif
            } while (!__any(leave));
            // ---- every slot's state at the top of the loop to memory: the registers of the slots that took the step (the others'
            // was staged before it)
            if (live && !leave) stage(sel ^ 1u, si, sj, 0, trace_top, nblocks, dir, 0, 0);
            lds_sync();
            if (live) {   // both buffers back to the arena (solo mode needs the LDS region, and reads a pair's state from the arena)
#pragma unroll
                for (int w = 0; w < 2; w++) {
                    const char* sp = lbuf + w * BUFL + l * 16;
                    char* d = slot_mem + w * BUFA + l * 16;
#pragma unroll
                    for (int a = 0; a < 4; a++) *(int4*)(d + a * ARR) = *(const int4*)(sp + a * ARR);
                    if (l < 9) *(int*)(slot_mem + w * BUFA + BUFL + 4 * l) = lsc[w * 9 + l];
                }
            }
            lds_sync();
            if (live && l == 0) {
                int* rc = (int*)(slot_mem + 2 * BUFA);
                // (the steps taken here are not counted in a register: every one moved the block by STEP in one direction)
                const uint32_t nsteps = (uint32_t)mq_load((const char*)(rc + MR_NSTEPS)) + ((si + sj) - (uint32_t)mq_load((const char*)(rc + MR_SI)) - (uint32_t)mq_load((const char*)(rc + MR_SJ))) / (uint32_t)STEP;
                rc[MR_SI] = (int)si; rc[MR_SJ] = (int)sj; rc[MR_DIR] = dir; rc[MR_PREV_DIR] = prev_dir; rc[MR_OFF] = off; rc[MR_OFF_MAX] = off_max; rc[MR_BEST_MAX] = best_max;
                rc[MR_Y_DROP] = (int)y_drop; rc[MR_X_ITER] = x_iter; rc[MR_D_CORNER] = D_corner; rc[MR_NSTEPS] = (int)nsteps; rc[MR_TRACE_TOP] = (int)trace_top;
                rc[MR_NBLOCKS] = (int)nblocks; rc[MR_SEL] = (int)sel;
            }
            const unsigned long long lm = __ballot(leave && l == 0);
            pend_m = 0;
#pragma unroll
            for (int sl_i = 0; sl_i < NSLOT; sl_i++) pend_m |= (uint32_t)((lm >> (sl_i * SL)) & 1ull) << sl_i;
#ifdef BA_TIMING
            t_quad += __builtin_amdgcn_s_memtime() - tq_a;
#endif
        }
    }
#ifdef BA_TIMING
    if (bp.prof && is_lane(0)) {
        atomicMax(bp.prof + 40, (unsigned long long)__builtin_amdgcn_s_memrealtime());
        atomicAdd(bp.prof + 50, t_solo); atomicAdd(bp.prof + 51, t_quad); atomicAdd(bp.prof + 52, t_wait); atomicAdd(bp.prof + 53, n_solo); atomicAdd(bp.prof + 54, n_quad); atomicAdd(bp.prof + 55, 1ull);
    }
#endif
#ifdef BA_ENDHIST
    if (bp.prof && is_lane(0)) {   // development (tools/dev/ragged_end.py): when did this fill wave run out of pairs? 4 ms buckets after the launch's start
        const unsigned long long t0 = __hip_atomic_load(bp.prof + 43, BA_RLX_AGENT), t = __builtin_amdgcn_s_memrealtime();
        const unsigned long long k = t0 && t > t0 ? (t - t0) / 400000ull : 0ull;
        atomicAdd(bp.prof + 64 + (k < 63ull ? k : 63ull), 1ull);
        atomicMax(bp.prof + 40, t);
    }
#endif
    // a fill wave joins the traceback side with one lane once the batch has no pairs left for it (see k_align)
    if (batch_traceback) {
        lds_sync();
#ifndef MQ_HELPER_LANES
#define MQ_HELPER_LANES 0u   // 0: the emptied wave walks one path at a time with all its lanes (walk_wave); k > 0: k paths, one per lane (1 -> 4 lanes: +1.5 % at config 3; 4 lanes -> the whole wave: +3 %)
#endif
        if constexpr (SPM != 0) traceback_consumer<TB_LB>(bp, TB_MASK, (unsigned char*)base, 4u, false);   // (the whole-wave walk takes no mode bits: four lanes' walks instead)
        else {
#if MQ_HELPER_LANES == 0
        traceback_helper_wave<true>(bp, (uint32_t*)base, 2048u);
#else
        traceback_consumer<(int)TB_LANE_BYTES_L2>(bp, (uint32_t)F_CIGAR_EQ, (unsigned char*)base, MQ_HELPER_LANES, false);
#endif
        }
#if defined(BA_TIMING) || defined(BA_ENDHIST)
        if (bp.prof && is_lane(0)) atomicMax(bp.prof + 41, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
    }
}

#undef don_final
#undef don_offers
}  // namespace ba
