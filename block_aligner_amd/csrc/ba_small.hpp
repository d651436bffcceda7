// block_aligner_amd — sixteen pairs per wavefront while the block is 32 cells, inside ONE persistent kernel.
//
// Batches that start at the reference's usual minimum block size (examples/uc_bench.rs:85-100, pssm_bench.rs:94-100: 32..=256;
// nanopore_bench.rs at 1 kbp) used to run through k_quad (ba_quad.hpp: four pairs per wave, two cells per lane) and a queue to
// the per-pair kernel for everything that is not a plain shift step. k_small joins the two ideas of ba_multi.hpp and ba_quad.hpp:
//   * slots of FOUR lanes x EIGHT cells (multi_rect's column code: the gap scan is an in-lane chain over the lane's four packed
//     registers followed by a two-step DPP scan inside the quad, border shifts are DPP quad permutes), sixteen pairs per wave;
//   * a slot takes a pair from the batch by itself, runs its first block (scan_block.rs:260-305 with prev_size = 0) as four right
//     steps of 8 columns over zeroed borders, then plain shift steps at 32 cells, and finishes a global alignment whose last step
//     has reached the end of both sequences (scan_block.rs:560-570, 1216-1224) -- all inside the loop of steps;
//   * whatever else a pair needs -- a grow, the steps at larger block sizes, a shrink, X-drop termination, the early column break
//     -- is done by the SAME wave in solo mode: the slots' state goes to memory, Aligner::run (ba_driver.hpp: all 64 lanes on that
//     one pair, the code of the per-pair kernel) takes the pair until it is finished or its next step is again a plain shift step
//     at 32 cells. No queue, no second launch, no PairCont round trip.
// As in k_multi a slot keeps no location of its steps' maxima: the state BEFORE the last improving step is the checkpoint (two LDS
// buffers per slot that change roles), and the solo driver repeats that step once (Aligner::import_slot). A pair whose last
// improving step is still its first block when it leaves the slot is simply run again from the start by the solo driver.
// TRACE batches are pair-slot batches (ba_params.h): trace words of a slot's rectangle are 4 cells x 2 columns per word (bit 31 of
// BlockRec::i), the first block is four records of 32 x 8 cells; k_walk (its L2 form) walks all paths after the fill.
#pragma once
#include "ba_multi.hpp"

namespace ba {

constexpr int SM_B = 32, SM_LW = 4, SM_NS = 16;
enum { MR_FLAGS = MR_WORDS, MR_BOOT, MR_BMX, SM_MR_WORDS };   // flags: bit 0 = run the pair from its start in solo mode, bit 1 = ... and to its end
static_assert(SM_MR_WORDS <= 32 && SM_B == (int)SM_B_HOST && SM_NS == (int)SM_SLOTS, "k_small record");

template <int N>
__device__ __forceinline__ int quad_bcast(int v) { return __builtin_amdgcn_update_dpp(v, v, N | (N << 2) | (N << 4) | (N << 6), 0xf, 0xf, false); }
// inclusive prefix max inside every quad: quad_perm [0,0,1,2] then [0,1,0,1]
__device__ __forceinline__ int quad_prefix_max(int v) {
    asm volatile(
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[0,0,1,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}
// max over the quad, in every lane: [1,0,3,2] then [2,3,0,1]
__device__ __forceinline__ int quad_all_max(int v) {
    asm volatile(
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}
// lane l <- src[l - 1] inside the quad (quad lane 0 keeps its own value: the caller replaces it)
__device__ __forceinline__ int quad_shr1(int src) { return __builtin_amdgcn_update_dpp(src, src, 0x90, 0xf, 0xf, false); }
// lane l <- l == 3 ? last[l] : src[l + 1] inside the quad, for the eight registers of the orthogonal border pair at once (one mask set-up)
__device__ __forceinline__ void quad_shl1_keep8(int (&d)[4], int (&r)[4], const int (&sd)[4], const int (&sr)[4], const int (&ld)[4], const int (&lr)[4], unsigned long long last_mask) {
    asm volatile(
        "s_mov_b64 vcc, %16\n\ts_nop 1\n\t"
        "v_cndmask_b32_dpp %0, %8, %17, vcc quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %1, %9, %18, vcc quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %2, %10, %19, vcc quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %3, %11, %20, vcc quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %4, %12, %21, vcc quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %5, %13, %22, vcc quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %6, %14, %23, vcc quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_dpp %7, %15, %24, vcc quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
        : "v"(sd[0]), "v"(sd[1]), "v"(sd[2]), "v"(sd[3]), "v"(sr[0]), "v"(sr[1]), "v"(sr[2]), "v"(sr[3]), "s"(last_mask),
          "v"(ld[0]), "v"(ld[1]), "v"(ld[2]), "v"(ld[3]), "v"(lr[0]), "v"(lr[1]), "v"(lr[2]), "v"(lr[3])
        : "vcc");
}

struct SmallConsts {
    int G[4];            // {(2k+1) g, (2k+2) g}: what a gap that enters the lane above its first cell has lost on reaching register k (wave-uniform)
    int laneKG;          // l * 8g
    int g8;              // 8g (wave-uniform): (l - 1) * 8g = laneKG - g8
    // Per cell the column's R is at least V = max(zero-shift-in artefact of the reference's in-vector scan (avx2.rs:315-338), the MIN = 0 carry above
    // the column). For a lane's first seven cells that is G[k] in every lane -- cell c of a 16-cell vector sees a virtual zero at distance
    // (c & 7) + 1, never farther than the carry from above the column --; only the eighth cell (vector cell 7: the artefact's 12g, 15: its absence)
    // differs from lane to lane. Since max(cs + G, G) = max(cs, 0) + G (saturating adds of non-positive constants compose), the carry from the
    // lane above is floored ONCE per column -- at {0, W}, V(eighth cell) = W + 8g -- instead of a max with V per register (see MultiConsts::w0).
    int w0;
};
// (one definition for the kernel and k_lane_kat) l: the lane inside its slot's four
__device__ __forceinline__ SmallConsts make_small_consts(int l, int gx) {
    SmallConsts mc;
    mc.laneKG = l * 8 * gx; mc.g8 = 8 * gx;
#pragma unroll
    for (int k = 0; k < 4; k++) mc.G[k] = pk(max(-32768, (2 * k + 1) * gx), max(-32768, (2 * k + 2) * gx));
    mc.w0 = pk(0, l == 0 ? 0 : ((l & 1) ? max(-32768, 8 * l * gx) : 4 * gx));   // (see SmallConsts)
    return mc;
}
// R of the lane above's last cell: a scan over the quad's four lanes on values re-based by l * 8g; quad lane 0 has no lane above (a candidate that
// never wins); floored at {0, W} (SmallConsts::w0)
__device__ __forceinline__ int small_carry(int r3, const SmallConsts& mc, int l) {
    const int pm = quad_prefix_max((int)as_s(r3).y - mc.laneKG);
    int cin = __builtin_amdgcn_update_dpp(pm, pm, 0x90, 0xf, 0xf, false) + mc.laneKG - mc.g8;
    cin = l == 0 ? -32768 : cin;
    return vmax(scan8_splat_lo(cin), mc.w0);
}

// Sequence-to-profile steps (place_block_profile_*, scan_block.rs:612-783; ba_quad.hpp QuadProfile for two cells per lane). A slot's step runs along
// the query ("right": one profile position per column, gap costs uniform in a column) or along the profile ("down": one query residue per column, gap
// costs per row with the roles of C and R exchanged, scan_block.rs:671-682), and the sixteen slots of a wave differ. The loads have ONE form for both:
// eight rows of the transposed table aa_pos[residue][position], 16 bytes each, at position `pos` -- right: the rows of the lane's eight residues at the
// step's first column (S[m] = residue m x 8 columns), down: the rows of the step's eight column residues at the lane's first position (S[j] = column j x
// 8 cells) -- and three 16-byte vectors of per-position costs at `pos` (right: the 8 columns', down: the lane's 8 cells').
struct SmallProfile {
    int S[8][4];                    // scores (see above), packed pairs
    int G1[4], G2[4], G3[4];        // right: gap_open_C + extend / gap_open_R / gap_close_C of the 8 columns; down: gap_open_R + extend / gap_open_C / gap_close_C of the lane's 8 cells
    int selE, selO;                 // cost selectors of even / odd columns (v_perm, first source = the column-packed register): right = splat its even / odd half, down = the lane's own pair
    int selXE, selXO, selY;         // the close cost's: C11_end = C11 + gap_close_C in right steps only, R11_end = R11 + gap_close_R in down steps only (0x0c0c0c0c = zero)
    unsigned long long rmask;       // lanes of slots that take a right step
    int twords[8];                  // out (TRACE): the lane's eight trace words of the step -- stored by the caller behind the next step's loads (the memory
                                    // counter is in order: loads issued behind the stores would wait for them)
};

// One 8-column shift step for the sixteen slots of a wave (multi_rect for quads). first_cell: this lane holds cell (0, 0) of a pair's
// first block (scan_block.rs:1130-1132). fin_any / fin_col / dsel: a global alignment's last step is among the slots: D of column fin_col.
// The orthogonal border pair is not live across the columns: its values of before the step are read back from the buffer the step's state
// was staged in (pstage: this lane's 16 bytes of P_d, P_r 64 bytes behind) when the step shifts it -- eight registers fewer in the columns,
// which is what keeps the loop of steps free of scratch reloads (a reload waits for every memory operation in flight, the step's trace
// stores included: with 9 of them in the loop the TRACE kernels ran 1.6 x slower than the score-only ones).
// SPM (round 5): the special alignment modes a slot takes -- 1 = LOCAL_START (every cell's D is at least the relative zero, scan_block.rs:1134-1136; TRACE: a
// zero mask of one bit per cell behind the rectangle's trace words -- two words per lane: cells 0 .. 3 / 4 .. 7, byte = cell, bit = column), 2 =
// FREE_QUERY_START_GAPS (row 0 of a right step starts from the relative zero in every column, scan_block.rs:1130-1132). rz2 = the relative zero, both halves.
template <int KIND, bool TRACE, bool FIN, int SPM = 0>
__device__ __forceinline__ void small_rect(const char* table, const FillConsts& fc, const SmallConsts& mc, int l, int (&Ad)[4], int (&Ac)[4],
                                           int (&Pd)[4], int (&Pr)[4], const char* pstage, uint2 vb, uint32_t cb_lo, uint32_t cb_hi, int corner, int off_add, bool first_cell,
                                           uint32_t* __restrict__ tout, bool fin_any, uint32_t fin_col, int (&dsel)[4], MultiOut& o, SmallProfile* sp = nullptr,
                                           int rz2 = 0, bool fqs_row0 = false, int* zout = nullptr) {
    constexpr bool PROF = KIND == KIND_PROFILE;
    uint32_t zacc[2] = {0, 0};   // LOCAL_START + TRACE: "D differs from the relative zero", inverted at the end
    const int offa = splat(off_add);
    int d[4], c[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {   // just_offset (scan_block.rs:1003-1012)
        d[k] = adds(Ad[k], offa); c[k] = adds(Ac[k], offa);
    }
    ScoreKey<KIND> key[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t w = k < 2 ? vb.x : vb.y;
        key[k] = make_key<KIND>((int)((w >> (16 * (k & 1))) & 0xffu), (int)((w >> (16 * (k & 1) + 8)) & 0xffu));
    }
    int Yk[4] = {0, 0, 0, 0};   // profiles, down steps: gap_close_R of the lane's cells (zero in right steps)
    int tdef[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // profiles: the step's trace words are stored by the caller, behind the loads of the next step (sp->twords)
    if constexpr (PROF) {
#pragma unroll
        for (int k = 0; k < 4; k++) Yk[k] = __builtin_amdgcn_perm(sp->G3[k], sp->G3[k], sp->selY);
    }
    uint32_t kb[4] = {0, 0, 0, 0}, cbs_lo = 0, cbs_hi = 0;   // NUC: see add_byte (ba_device.hpp)
    if constexpr (KIND == KIND_NUC) {
        const uint32_t tb = (uint32_t)(uintptr_t)table, k01 = nuc_keys2(vb.x), k23 = nuc_keys2(vb.y);
        kb[0] = add_word(k01, tb, 0); kb[1] = add_word(k01, tb, 1); kb[2] = add_word(k23, tb, 0); kb[3] = add_word(k23, tb, 1);
        cbs_lo = nuc_col_offsets(cb_lo); cbs_hi = nuc_col_offsets(cb_hi);
    }
    int dmax[4] = {0, 0, 0, 0}, tacc[2] = {0, 0};
    int nvD[4] = {0, 0, 0, 0}, nvR[4] = {0, 0, 0, 0};   // the last cells of the 8 new columns: the orthogonal border's new entries (quad lane 3)
    int holdD = 0, holdR = 0;
#pragma unroll
    for (int j = 0; j < STEP; j++) {
        const int cb = (int)(((j < 4 ? cb_lo : cb_hi) >> (8 * (j & 3))) & 0xffu);
        int sc[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if constexpr (KIND == KIND_NUC) sc[k] = lds_read_i32(add_byte(j < 4 ? cbs_lo : cbs_hi, kb[k], j));
            else if constexpr (PROF) {   // right: cells 2k, 2k + 1 of column j out of their two rows; down: the column's row, cells 2k, 2k + 1
                const int t = __builtin_amdgcn_perm(sp->S[2 * k + 1][j >> 1], sp->S[2 * k][j >> 1], (j & 1) ? 0x07060302 : 0x05040100);
                sc[k] = sel_mask(sp->rmask, t, sp->S[j][k]);
            }
            else sc[k] = fetch_score<KIND>(table, key[k], cb);
        }
        int Xj = 0;   // profiles, right steps: the column's gap_close_C (zero in down steps)
        if constexpr (PROF) Xj = __builtin_amdgcn_perm(sp->G3[j >> 1], sp->G3[j >> 1], (j & 1) ? sp->selXO : sp->selXE);
        // D00: the previous column shifted down one cell (scan_block.rs:1125); only column 0 has a cell above the block
        int prev = quad_shr1(d[3]);
        prev = l == 0 ? (j == 0 ? (int)((uint32_t)corner << 16) : 0) : prev;
        int d00[4];
        d00[0] = __builtin_amdgcn_alignbit(d[0], prev, 16);
#pragma unroll
        for (int k = 1; k < 4; k++) d00[k] = __builtin_amdgcn_alignbit(d[k], d[k - 1], 16);
        int d11[4], copen[4], cn[4], cend[4], x[4], r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            d11[k] = adds(d00[k], sc[k]);
            if (j == 0 && k == 0) d11[0] = first_cell ? (int)(((uint32_t)d11[0] & 0xffff0000u) | (uint32_t)ZERO) : d11[0];   // cell (0,0) starts from the relative zero
            if (SPM == 2 && k == 0) d11[0] = fqs_row0 ? (int)(((uint32_t)d11[0] & 0xffff0000u) | ((uint32_t)rz2 & 0xffffu)) : d11[0];   // row 0 of a right step: a free start in every column
            if (SPM == 1) d11[k] = vmax(d11[k], rz2);   // a local alignment may start anywhere
            int goC = fc.go2, goR = fc.ome2;
            if constexpr (PROF) {   // position-specific costs: the column's (right) or the cell's (down), scan_block.rs:658-676
                const int gsel = (j & 1) ? sp->selO : sp->selE;
                goC = __builtin_amdgcn_perm(sp->G1[j >> 1], sp->G1[k], gsel); goR = __builtin_amdgcn_perm(sp->G2[j >> 1], sp->G2[k], gsel);
            }
            copen[k] = adds(d[k], goC);
            cn[k] = vmax(adds(c[k], fc.ge2), copen[k]);
            cend[k] = PROF ? adds(cn[k], Xj) : cn[k];                      // C11_end (scan_block.rs:694)
            d11[k] = vmax(d11[k], cend[k]);
            x[k] = adds(d11[k], goR);                                      // D11_open
            r[k] = scan8_inreg(x[k], fc.ge2);                              // inside the register
        }
        // R11: the chain over the lane's registers, then a scan over the quad's four lanes on values re-based by l * 8g
        scan8_chain(r, mc.G[0]);
        const int cs = small_carry(r[3], mc, l);   // R of the lane above's last cell, floored: see SmallConsts::w0
        int dn[4];
#pragma unroll
        for (int p2 = 0; p2 < 2; p2++) {   // two registers at a time: their trace flags are packed before the next two are touched (fewer values alive)
            uint32_t sC[2], sR[2], sCo[2], sRo[2];
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                const int k = 2 * p2 + kk;
                r[k] = scan8_apply(r[k], cs, mc.G[k], k == 3);
                const int rend = PROF ? adds(r[k], Yk[k]) : r[k];          // R11_end (scan_block.rs:704)
                dn[k] = vmax(d11[k], rend);
                if (TRACE) {   // the cell's four flags as sign bits of saturating differences (see fast_rect)
                    sC[kk] = (uint32_t)subs(cend[k], dn[k]); sR[kk] = (uint32_t)subs(rend, dn[k]); sCo[kk] = (uint32_t)subs(copen[k], cn[k]); sRo[kk] = (uint32_t)subs(x[k], r[k]);
                }
                dmax[k] = vmax(dmax[k], dn[k]);
                d[k] = dn[k]; c[k] = cn[k];
            }
            if (TRACE && SPM == 1) {   // zero mask (scan_block.rs:1184-1187): D >= the relative zero, so "differs" is the sign of rz - D
                const uint32_t pZ = (uint32_t)__builtin_amdgcn_perm(subs(rz2, dn[2 * p2 + 1]), subs(rz2, dn[2 * p2]), 0x0b0a0908);
                zacc[p2] |= pZ & (0x01010101u << j);
            }
            if (TRACE) {   // trace words: see multi_rect (4 consecutive cells x 2 columns per word, a lane's eight words contiguous)
                const uint32_t pC = (uint32_t)__builtin_amdgcn_perm((int)sC[1], (int)sC[0], 0x0b0a0908), pR = (uint32_t)__builtin_amdgcn_perm((int)sR[1], (int)sR[0], 0x0b0a0908);
                const uint32_t pCo = (uint32_t)__builtin_amdgcn_perm((int)sCo[1], (int)sCo[0], 0x0b0a0908), pRo = (uint32_t)__builtin_amdgcn_perm((int)sRo[1], (int)sRo[0], 0x0b0a0908);
                const uint32_t lo2 = bfi(0x55555555u, pC, pR), hi2 = bfi(0x55555555u, pCo, pRo);   // (sign-replicating selectors: clean byte masks, see multi_rect)
                const uint32_t nib = bfi(0x33333333u, lo2, hi2);        // the nibble in both halves of every byte
                if (j & 1) tacc[p2] = (int)bfi(0xF0F0F0F0u, nib, (uint32_t)tacc[p2]);
                else tacc[p2] = (int)nib;
            }
        }
        if (FIN) {   // (selects in every step, not under a wave-uniform branch on fin_any: the branch cut the step into a basic block per column,
                     // which the scheduler cannot interleave -- 40 more instructions per step buy one block of eight columns)
#pragma unroll
            for (int k = 0; k < 4; k++) dsel[k] = fin_col == (uint32_t)j ? dn[k] : dsel[k];
        }
        if constexpr (TRACE && PROF) { if (j & 1) { tdef[2 * (j >> 1)] = tacc[0]; tdef[2 * (j >> 1) + 1] = tacc[1]; } }
        // (two column pairs' four words per store: two accumulators live across two more columns, half the store instructions -- round 5, same box: 200 k
        // 1 kbp pairs with traceback 14.47 -> 14.12 ms, 400 k protein pairs 9.74 -> 9.71; unpredicated: a slot without a step writes to the wave's sink)
        else if (TRACE && (j & 3) == 1) { tdef[0] = tacc[0]; tdef[1] = tacc[1]; }
        else if (TRACE && (j & 3) == 3) *(int4*)(tout + 4 * (j >> 2)) = int4{tdef[0], tdef[1], tacc[0], tacc[1]};   // (a column pair's two words: the accumulators do not live on; unpredicated: a slot without a step writes to the wave's sink)
        // the last cell of the column feeds the orthogonal border (scan_block.rs:1213-1214): two columns to a register
        if (j & 1) { nvD[j >> 1] = __builtin_amdgcn_perm(dn[3], holdD, 0x07060302); nvR[j >> 1] = __builtin_amdgcn_perm(r[3], holdR, 0x07060302); }
        else { holdD = dn[3]; holdR = r[3]; }
#ifdef SM_COLUMN_BARRIER
        __builtin_amdgcn_sched_barrier(0);   // (columns are not interleaved by the scheduler: shorter live ranges)
#endif
    }
    // shift_and_offset (scan_block.rs:1040-1061): 8 entries = one lane; the border as it was before the step comes back from its staged copy
    int pd[4], pr[4];
    {
        const int4 sd = *(const int4*)pstage, sr = *(const int4*)(pstage + 64);
        pd[0] = adds(sd.x, offa); pd[1] = adds(sd.y, offa); pd[2] = adds(sd.z, offa); pd[3] = adds(sd.w, offa);
        pr[0] = adds(sr.x, offa); pr[1] = adds(sr.y, offa); pr[2] = adds(sr.z, offa); pr[3] = adds(sr.w, offa);
    }
    o.corner_new = quad_bcast<0>(pd[3]) >> 16;   // D_corner for a following orthogonal step: the orthogonal border's entry 7, re-based
    quad_shl1_keep8(Pd, Pr, pd, pr, nvD, nvR, 0x8888888888888888ull);   // (Pd, Pr: outputs -- the border pair after the step)
#pragma unroll
    for (int k = 0; k < 4; k++) { Ad[k] = d[k]; Ac[k] = c[k]; }
    {   // max of the first 8 entries of both borders (quad lane 0), to every lane of the slot (scan_block.rs:1020-1022)
        const int ma = vmax(vmax(d[0], d[1]), vmax(d[2], d[3])), mb = vmax(vmax(Pd[0], Pd[1]), vmax(Pd[2], Pd[3]));
        const s16x2 sa = as_s(ma), sb = as_s(mb);
        const int xa = vmax(ma, as_i(s16x2{sa.y, sa.x})), xb = vmax(mb, as_i(s16x2{sb.y, sb.x}));
        const s16x2 rr = as_s(quad_bcast<0>(__builtin_amdgcn_perm(xb, xa, 0x05040100)));
        o.act_max8 = rr.x; o.pas_max8 = rr.y;
    }
    const int mm = vmax(vmax(dmax[0], dmax[1]), vmax(dmax[2], dmax[3]));
    const int m32 = max(mm & 0xffff, (int)((uint32_t)mm >> 16));   // halves are >= 0 (D_max starts at MIN = 0)
    o.mx = quad_all_max(m32);
    if constexpr (TRACE && PROF) {
#pragma unroll
        for (int k = 0; k < 8; k++) sp->twords[k] = tdef[k];
    }
    if constexpr (TRACE && SPM == 1) { zout[0] = (int)~zacc[0]; zout[1] = (int)~zacc[1]; }
}

template <int PMAX, int KIND, bool TRACE, bool XDROP, int SPM = 0>
#ifndef SM_WAVES_EU
#define SM_WAVES_EU 4   // (waves per SIMD the kernel is compiled for; 2 -- 256 registers -- was tried for the traced kernels: see DESIGN.md)
#endif
// (sequence-to-profile steps hold 32 score and 12 cost registers across their columns: that instantiation is compiled for two waves per SIMD)
__global__ void __launch_bounds__(WAVES_PER_WG * 64) __attribute__((amdgpu_waves_per_eu(KIND == KIND_PROFILE ? 2 : SM_WAVES_EU, KIND == KIND_PROFILE ? 2 : SM_WAVES_EU))) k_small(const BatchParams bp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id(), l = lane & (SM_LW - 1), g = lane >> 2;
    const int wave = uni((int)threadIdx.x >> 6);
    {   // workgroup-shared scoring table, as in k_align
        char* tab = smem;
        if (KIND == KIND_NUC) {
            nuc_table_fill(tab, bp.matrix, (int)threadIdx.x, WAVES_PER_WG * 64);   // (layout: ba_device.hpp nuc_key_off)
        } else {
            const int nbytes = KIND == KIND_AA ? 27 * 32 : (KIND == KIND_BYTES ? 2 : 0);   // PROFILE: scores live in the pair's image
            for (int k = (int)threadIdx.x; k < nbytes; k += WAVES_PER_WG * 64) tab[k] = (char)bp.matrix[k];
        }
    }
    __syncthreads();
    // (the launch of the batch's longest pairs, beside the main one: one wave per SIMD takes pairs -- four solo drivers to a SIMD would slow
    // each other down, and these chains are what the launch waits for)
    if (bp.cq_side && wave >= 4) return;
    constexpr uint32_t LCLS = (uint32_t)PMAX * 128u;
    constexpr uint32_t ab = lds_array_bytes_h(LCLS);
    char* base = smem + lds_table_bytes_h(KIND) + (uint32_t)wave * sm_wave_bytes_h(LCLS, KIND);
    WaveLds L;
    L.D_col = (short*)(base + 0 * ab); L.C_col = (short*)(base + 1 * ab);
    L.D_row = (short*)(base + 2 * ab); L.R_row = (short*)(base + 3 * ab);
    L.misc = (short*)(base + 4 * ab);
    L.table = smem;
    const int gx = bp.gap_extend;
    const uint32_t fill_wave = blockIdx.x * WAVES_PER_WG + (uint32_t)wave;
    const uint32_t max_size = bp.max_size;
    // a slot rectangle's words on the trace stack. LOCAL_START: the zero mask takes 8 words behind the 32 trace words, but the stack advances as the
    // per-pair kernel's does (a mask word per trace word, Aligner::add_block), so that a pair's trace words count its cells the same way on every path
    constexpr uint32_t SM_TW = (STEP * SM_B / 8) * (SPM == 1 ? 2u : 1u);
    const bool keep_pre = XDROP || (uint32_t)SM_B < max_size;   // a slot keeps the state before its last improving step
    char* const wave_mem = (char*)bp.big + (uint64_t)fill_wave * SM_WAVE_BYTES;

    // (values derived again after the per-pair driver instead of being kept across it: laundered so that they are not hoisted)
    auto coldp_big = [&]() { const __attribute__((address_space(4))) BatchParams* p = (const __attribute__((address_space(4))) BatchParams*)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(p)); return p->big; };
    auto fill_wave_of = [&]() { uint32_t b = blockIdx.x, w = (uint32_t)wave; asm volatile("" : "+s"(b), "+s"(w)); return b * WAVES_PER_WG + w; };
    uint32_t live_m = 0, pend_m = 0;   // wave-uniform: bit s = slot s holds a pair / its pair has to go through solo mode
    uint32_t w_next = 0, w_end = 0;
    bool more = true;
    // The batch's longest pairs (the first sm_excl_n of the batch order: ba_host.cpp) are not for slots: sixteen pairs to a wave make each
    // of them sixteen times as long in flight -- and every solo episode of a neighbour stops the other fifteen --, so a pair many times the
    // average length would end the launch alone (400 k protein pairs 22 .. 8881: 10.7 ms with the 77 longest, 5.3 without). Each wave takes
    // at most a few of them, one at a time, before it starts its slots, and runs them from start to end on all its lanes.
    const uint32_t excl_n = bp.sm_excl_n;
    bool excl_more = excl_n > bp.sm_excl_first;

    for (;;) {
        if (excl_more && !live_m && !pend_m) {
            uint32_t e = 0;
            if (is_lane(0)) e = atomicAdd(bp.work_counter + 2, 1u);
            e = (uint32_t)uni((int)e) + bp.sm_excl_first;
            if (e < excl_n) {
                if (is_lane(0)) { int* rc = (int*)(wave_mem + 2 * SM_BUF_BYTES); rc[MR_PAIR] = (int)e; rc[MR_FLAGS] = 3; }
                pend_m = 1u;
            } else excl_more = false;
        }
        // ================= solo mode: the pairs whose step was rolled back (or that a slot cannot take), one at a time on all 64 lanes
        while (pend_m) {
            int solo = __builtin_ctz(pend_m);
            const FillConsts fc = make_fill_consts(lane, bp.gap_open, gx);   // two cells per lane (as k_align)
            Aligner<PMAX, KIND, TRACE, XDROP, SPM != 0, true, SM_B> al(bp, L, fc);
            PairState st{};
            char* const smem_s = wave_mem + (uint32_t)solo * SM_SLOT_BYTES;
            char* const rec = smem_s + 2 * SM_BUF_BYTES;
            const int rv = lane < SM_MR_WORDS ? mq_load(rec + 4 * lane) : 0;   // the slot's record: lane k holds word k
#define BA_W(v, k) __builtin_amdgcn_readlane(v, k)
            uint32_t s_pair = (uint32_t)BA_W(rv, MR_PAIR);
            bool fresh = (BA_W(rv, MR_FLAGS) & 1) != 0;
            const bool to_end = (BA_W(rv, MR_FLAGS) & 2) != 0;
            if (!fresh) {
                const uint32_t s_sel = (uint32_t)BA_W(rv, MR_SEL);
                const char* live_b = smem_s + (s_sel ^ 1u) * SM_BUF_BYTES; const char* ck_b = smem_s + s_sel * SM_BUF_BYTES;
                const int cv = lane < 8 ? mq_load(ck_b + 256 + 4 * lane) : 0;   // the checkpoint's scalars
                const int flag = BA_W(cv, 0) & 0xff;
                if (flag == 2 && keep_pre) fresh = true;   // the last improving step is still the first block: once more from the start
                else {
                    st.si = (uint32_t)BA_W(rv, MR_SI); st.sj = (uint32_t)BA_W(rv, MR_SJ); st.dir = BA_W(rv, MR_DIR); st.prev_dir = BA_W(rv, MR_PREV_DIR);
                    st.off = BA_W(rv, MR_OFF); st.off_max = BA_W(rv, MR_OFF_MAX); st.best_max = BA_W(rv, MR_BEST_MAX);
                    st.y_drop_iter = (uint32_t)BA_W(rv, MR_Y_DROP); st.x_drop_iter = BA_W(rv, MR_X_ITER); st.D_corner = BA_W(rv, MR_D_CORNER);
                    st.trace_top = (uint32_t)BA_W(rv, MR_TRACE_TOP); st.nblocks = (uint32_t)BA_W(rv, MR_NBLOCKS);
                    const uint32_t ns = (uint32_t)BA_W(rv, MR_NSTEPS);
                    st.best_i = (uint32_t)BA_W(rv, MR_BEST_I); st.best_j = (uint32_t)BA_W(rv, MR_BEST_J);
                    st.cells = ((unsigned long long)(uint32_t)BA_W(rv, MR_CELLS_LO) | ((unsigned long long)(uint32_t)BA_W(rv, MR_CELLS_HI) << 32)) + (unsigned long long)ns * (STEP * SM_B);
                    const uint32_t bud = (uint32_t)BA_W(rv, MR_BUDGET);
                    st.step_budget = bud > ns ? bud - ns : 1u;
                    st.status = (uint32_t)BA_W(rv, MR_STATUS);
                    const bool ck_pre = flag != 0;
                    st.ck_i = (uint32_t)BA_W(cv, 1); st.ck_j = (uint32_t)BA_W(cv, 2); st.ck_off = BA_W(cv, 3);
                    st.ck_tt = (uint32_t)BA_W(cv, 4) + (flag ? SM_TW : 0u);
                    st.ck_nb = (uint32_t)BA_W(cv, 5) - (TRACE ? (uint32_t)bp.blocks_off[s_pair] : 0u) + (flag ? 1u : 0u);   // (staged as an absolute record position)
                    const int ck_dir = (BA_W(cv, 0) >> 8) & 0xff, ck_offadd = BA_W(cv, 6), ck_corner = BA_W(cv, 7);
                    {   // the borders: lane l's 16 bytes of each array (lanes 0 .. 3), into the canonical order D_col, C_col, D_row, R_row
                        int reg[16], ckr[16];
                        const bool lr = st.dir == DIR_RIGHT, cr = ck_dir == DIR_RIGHT;
                        const uint32_t oAd = (lr ? 0u : 128u) + l * 16, oAc = (lr ? 64u : 192u) + l * 16, oPd = (lr ? 128u : 0u) + l * 16, oPr = (lr ? 192u : 64u) + l * 16;
                        const uint32_t cAd = (cr ? 0u : 128u) + l * 16, cAc = (cr ? 64u : 192u) + l * 16, cPd = (cr ? 128u : 0u) + l * 16, cPr = (cr ? 192u : 64u) + l * 16;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            reg[k] = mq_load(live_b + oAd + 4 * k); reg[4 + k] = mq_load(live_b + oAc + 4 * k); reg[8 + k] = mq_load(live_b + oPd + 4 * k); reg[12 + k] = mq_load(live_b + oPr + 4 * k);
                            ckr[k] = mq_load(ck_b + cAd + 4 * k); ckr[4 + k] = mq_load(ck_b + cAc + 4 * k); ckr[8 + k] = mq_load(ck_b + cPd + 4 * k); ckr[12 + k] = mq_load(ck_b + cPr + 4 * k);
                        }
                        al.import_slot(reg, ckr, s_pair, ck_pre || (uint32_t)SM_B < max_size, ck_pre, ck_dir, ck_offadd, ck_corner, st.ck_i, st.ck_j, st.best_i, st.best_j, st.ck_off);
                    }
                }
            }
#undef BA_W
            if (TRACE) { al.trace = bp.trace_arena + bp.trace_off[s_pair]; al.blocks = bp.blocks + bp.blocks_off[s_pair]; }
            else { al.trace = bp.trace_arena; al.blocks = bp.blocks; }
            al.ckpt = bp.ckpt + (uint64_t)(fill_wave + bp.ckpt_wave0) * 8 * bp.max_size;
            // The per-pair driver needs every scalar register: what the wave keeps across it is parked in the lanes of one VGPR
            int keepv = 0;
            park<0>(keepv, (int)live_m); park<1>(keepv, (int)pend_m); park<2>(keepv, (int)w_next); park<3>(keepv, (int)w_end);
            park<4>(keepv, (more ? 1 : 0) | (excl_more ? 2 : 0)); park<5>(keepv, solo); park<6>(keepv, (int)s_pair);
            // (a pair in solo mode keeps the wave's other fifteen slots waiting: its dependent chain goes first among the SIMD's waves)
#ifndef SM_EXCL_PRIO
#define SM_EXCL_PRIO MQ_SOLO_PRIO   // (priority of the pairs run from start to end -- the batch's longest, whose chain ends the launch; 3: see DESIGN.md)
#endif
            if (to_end) __builtin_amdgcn_s_setprio(SM_EXCL_PRIO); else __builtin_amdgcn_s_setprio(MQ_SOLO_PRIO);
            // (a pair run from start to end here -- one of the batch's longest -- walks its path at once, with the whole wave: the batch's longest
            // walks overlap with the fill. Tried instead: the first waves to run out of work take these walks from a counter -- they all run out
            // at about the same time, so the walks only lengthen the launch: protein set with traceback 11.1 -> 13.3 ms)
            st = al.run(s_pair, s_pair, false, nullptr, fresh ? MM_FRESH : MM_RESUME, st, !to_end, !fresh, to_end);   // (walk_now: tried -- these walks left to k_walk_l2's whole-wave walkers: C4 with traceback 11.23 -> 11.07 ms, within the spread)
            __builtin_amdgcn_s_setprio(0);
            live_m = (uint32_t)unpark<0>(keepv); pend_m = (uint32_t)unpark<1>(keepv); w_next = (uint32_t)unpark<2>(keepv); w_end = (uint32_t)unpark<3>(keepv);
            more = unpark<4>(keepv) & 1; excl_more = unpark<4>(keepv) & 2; solo = unpark<5>(keepv); s_pair = (uint32_t)unpark<6>(keepv);
            char* const smem_s2 = (char*)coldp_big() + (uint64_t)fill_wave_of() * SM_WAVE_BYTES + (uint32_t)solo * SM_SLOT_BYTES;
            char* const rec2 = smem_s2 + 2 * SM_BUF_BYTES;
            pend_m &= ~(1u << solo);
            if (st.exited) {   // the pair's next step is a plain shift step at 32 cells: into the slot (its memory)
                const bool rgt = st.dir == DIR_RIGHT;
                if (lane < SM_LW) {
                    const int l8 = 8 * l;
                    const int4 a_d = *(const int4*)((rgt ? L.D_col : L.D_row) + l8), a_c = *(const int4*)((rgt ? L.C_col : L.R_row) + l8);
                    const int4 p_d = *(const int4*)((rgt ? L.D_row : L.D_col) + l8), p_r = *(const int4*)((rgt ? L.R_row : L.C_col) + l8);
                    char* b1 = smem_s2 + SM_BUF_BYTES;   // the state at the top of the loop: buffer 1 (sel = 0)
                    *(int4*)(b1 + l * 16) = a_d; *(int4*)(b1 + 64 + l * 16) = a_c; *(int4*)(b1 + 128 + l * 16) = p_d; *(int4*)(b1 + 192 + l * 16) = p_r;
                    if ((uint32_t)SM_B < max_size) {   // the checkpoint as the reference keeps it (after its step): buffer 0
                        char* b0 = smem_s2;
                        *(int4*)(b0 + l * 16) = *(const int4*)(L.D_col + SM_B + l8); *(int4*)(b0 + 64 + l * 16) = *(const int4*)(L.C_col + SM_B + l8);
                        *(int4*)(b0 + 128 + l * 16) = *(const int4*)(L.D_row + SM_B + l8); *(int4*)(b0 + 192 + l * 16) = *(const int4*)(L.R_row + SM_B + l8);
                    }
                }
                if (is_lane(0)) {
                    char* b0 = smem_s2;
                    *(int4*)(b0 + 256) = int4{0 | (DIR_RIGHT << 8), (int)st.ck_i, (int)st.ck_j, st.ck_off}; *(int4*)(b0 + 272) = int4{(int)st.ck_tt, (int)(st.ck_nb + (TRACE ? (uint32_t)bp.blocks_off[s_pair] : 0u)), 0, 0};
                    int* rc = (int*)rec2;
                    rc[MR_PAIR] = (int)s_pair; rc[MR_BEST_I] = (int)st.best_i; rc[MR_BEST_J] = (int)st.best_j; rc[MR_CELLS_LO] = (int)(uint32_t)st.cells; rc[MR_CELLS_HI] = (int)(uint32_t)(st.cells >> 32);
                    rc[MR_BUDGET] = (int)st.step_budget; rc[MR_STATUS] = (int)st.status; rc[MR_TSLOT] = (int)s_pair; rc[MR_SI] = (int)st.si; rc[MR_SJ] = (int)st.sj; rc[MR_DIR] = st.dir;
                    rc[MR_PREV_DIR] = st.prev_dir; rc[MR_OFF] = st.off; rc[MR_OFF_MAX] = st.off_max; rc[MR_BEST_MAX] = st.best_max; rc[MR_Y_DROP] = (int)st.y_drop_iter;
                    rc[MR_X_ITER] = st.x_drop_iter; rc[MR_D_CORNER] = st.D_corner; rc[MR_NSTEPS] = 0; rc[MR_TRACE_TOP] = (int)st.trace_top; rc[MR_NBLOCKS] = (int)st.nblocks; rc[MR_SEL] = 0;
                    rc[MR_FLAGS] = 0; rc[MR_BOOT] = 0; rc[MR_BMX] = 0;
                }
                lds_sync();
                live_m |= 1u << solo;
            } else live_m &= ~(1u << solo);
        }
        if (excl_more) continue;
        if (!live_m && !more) break;

        // ================= the slots: registers from memory, steps until a slot needs solo mode (or the batch is done), registers to memory
        {
            char* const slot_mem = wave_mem + (uint32_t)g * SM_SLOT_BYTES;   // this lane's slot
            const int x_drop = bp.x_drop;
            const uint32_t total = bp.n;
            FillConsts fq;   // only the three gap constants (wave-uniform)
            fq.go2 = splat(bp.gap_open); fq.ge2 = splat(gx); fq.ome2 = splat(clamp16(bp.gap_open - gx));
            SmallConsts mc = make_small_consts(l, gx);  // eight cells per lane
            // ---- slot state (row-uniform, replicated over the slot's lanes)
            const bool live0 = (live_m >> g) & 1u;
            const char* rec = slot_mem + 2 * SM_BUF_BYTES;
#define BA_R(k) (live0 ? mq_load(rec + 4 * (k)) : 0)
            uint32_t pair = live0 ? (uint32_t)mq_load(rec + 4 * MR_PAIR) : ~0u;
            // (few registers: every one that lives across the columns of a step counts -- see small_rect. The small counters share one:
            // bits = boot sub-steps left [2:0] | "from the start, solo" [3] | prev_dir [5:4] | x_drop_iter [7:6]; ymix = the first block's
            // maximum so far while boot > 0, y_drop_iter after it; the steps taken since the record was written follow from si + sj)
            uint32_t si = (uint32_t)BA_R(MR_SI), sj = (uint32_t)BA_R(MR_SJ);
            uint32_t trace_top = (uint32_t)BA_R(MR_TRACE_TOP), sel = (uint32_t)BA_R(MR_SEL);
            uint32_t bits = ((uint32_t)BA_R(MR_BOOT) & 7u) | (((uint32_t)BA_R(MR_FLAGS) & 1u) << 3) | (((uint32_t)BA_R(MR_PREV_DIR) & 3u) << 4) | (((uint32_t)BA_R(MR_X_ITER) & 3u) << 6);
            int ymix = (bits & 7u) ? BA_R(MR_BMX) : BA_R(MR_Y_DROP);
            int dir = BA_R(MR_DIR), off = BA_R(MR_OFF), off_max = BA_R(MR_OFF_MAX), best_max = BA_R(MR_BEST_MAX);
            int D_corner = BA_R(MR_D_CORNER);
            uint32_t qlen = live0 ? bp.q_len[pair] : 0u, rlen = live0 ? bp.r_len[pair] : 0u;
            const uint8_t* qp = bp.pool + (live0 ? bp.q_off[pair] : 0ull); const uint8_t* rp = bp.pool + (live0 ? bp.r_off[pair] : 0ull);
            // TRACE: the pair's trace region (units of 16 words: regions are cut at multiples of 16), the next rectangle record (absolute), and
            // the number of steps region and record list still have room for
            uint32_t tr16 = 0, bpos = 0; int room = 0;
            auto room_of = [](uint32_t tcap, uint32_t bcap, uint32_t tt, uint32_t nb) -> int {
                const uint32_t a = tcap >= tt + 64u ? (tcap - 64u - tt) / SM_TW : 0u, b = bcap > nb ? bcap - nb : 0u;
                return (int)min(min(a, b), 0x7fffffffu);
            };
            if (TRACE && live0) {
                const uint64_t t0 = bp.trace_off[pair], b0 = bp.blocks_off[pair];
                const uint32_t nb = (uint32_t)BA_R(MR_NBLOCKS);
                tr16 = (uint32_t)(t0 >> 4); bpos = (uint32_t)b0 + nb;
                room = room_of((uint32_t)min(bp.trace_off[pair + 1] - t0, (uint64_t)0x7fffffffu), (uint32_t)min(bp.blocks_off[pair + 1] - b0, (uint64_t)0x7fffffffu), trace_top, nb);
            }
#undef BA_R
            // While the slots run, the two buffers of every slot live in this wave's LDS region (the solo borders' space): the checkpoint
            // (buffer `sel`) comes from the arena now and goes back after the loop. Of the other one -- the state at the top of the step in
            // flight -- the border pair along the step's vector axis (A) is in registers and staged before every step; the orthogonal pair (P)
            // lives ONLY there: a step reads it when it shifts it and writes the new one behind its driver decisions.
            char* const lbuf = base + (uint32_t)g * 512u;                        // this slot's buffers: + which * 256
            int* const lsc = (int*)(base + SM_LDS_SCALARS) + (uint32_t)g * 16u;   // their scalars: + which * 8
            int A_d[4], A_c[4];
            lds_sync();
            {
                const char* b = slot_mem + (sel ^ 1u) * SM_BUF_BYTES + l * 16;
#pragma unroll
                for (int k = 0; k < 4; k++) { A_d[k] = live0 ? mq_load(b + 4 * k) : 0; A_c[k] = live0 ? mq_load(b + 64 + 4 * k) : 0; }
                if (live0) {
                    char* d = lbuf + (sel ^ 1u) * 256u + l * 16;
                    *(int4*)(d + 128) = int4{mq_load(b + 128), mq_load(b + 132), mq_load(b + 136), mq_load(b + 140)};
                    *(int4*)(d + 192) = int4{mq_load(b + 192), mq_load(b + 196), mq_load(b + 200), mq_load(b + 204)};
                }
            }
            if (live0) {
                const char* cb = slot_mem + sel * SM_BUF_BYTES;
                char* d = lbuf + sel * 256u + l * 16;
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    const char* sp = cb + a * 64 + l * 16;
                    *(int4*)(d + a * 64) = int4{mq_load(sp), mq_load(sp + 4), mq_load(sp + 8), mq_load(sp + 12)};
                }
                lsc[sel * 8u + l] = mq_load(cb + 256 + 4 * l); lsc[sel * 8u + 4 + l] = mq_load(cb + 272 + 4 * l);
            }
            lds_sync();
            // sequence bytes of the next step, fetched one step ahead for both possible directions
            uint2 pf_qv = {0, 0}, pf_rv = {0, 0}, pf_qc = {0, 0}, pf_rc = {0, 0}; bool pf_ok = false;
            bool leave = false;
            // Sequence-to-profile slots: the step's operands (SmallProfile) are loaded one step ahead -- at the end of the step before, as soon as its
            // decisions have fixed where the next one goes, in front of that step's trace stores -- out of two 16-byte windows of the query that are
            // themselves fetched a step earlier (win_v: the lane's residues q[win_si + 8l ..], win_c: the column residues q[win_si + 24 ..]; the next
            // step starts at win_si or win_si + 8, so either half of a window serves it). Two dependent round trips per step otherwise: with two waves
            // per SIMD they were 64 % of the fill's time (80 k PSSM pairs: 2.2 ms for 0.76 ms of vector instructions).
            SmallProfile spf{};
            bool sp_ok = false;
            uint2 sp_vb = {0, 0};   // the lane's eight residues of the step the operands were loaded for
            uint4 win_v = {0, 0, 0, 0}, win_c = {0, 0, 0, 0}; uint32_t win_si = 0;
            auto load_profile_ops = [&](SmallProfile& sp_, bool right_, uint32_t ri_, uint32_t rj_, uint2 vb_, uint2 cb_, const uint8_t* rp_, uint32_t rlen_) {
                // rp = the pair's AAProfile image (ba_params.h). One form of loads for both directions (see SmallProfile): eight rows of aa_pos and
                // three cost vectors at `pos` -- the step's first column (right) or the lane's first position (down)
                const uint32_t P = profile_positions(rlen_, max_size);
                const short* aa_pos = (const short*)(rp_ + (uint64_t)P * 32);
                const short* goCp = aa_pos + (uint64_t)P * 32; const short* clCp = goCp + P; const short* goRp = clCp + P;
                const uint32_t pos = right_ ? rj_ : ri_ + 8u * (uint32_t)l;
                const uint32_t b_lo = right_ ? vb_.x : cb_.x, b_hi = right_ ? vb_.y : cb_.y;   // the lane's eight residues / the step's eight column residues
                // (right steps: NOT the rows of the lane's eight residues out of aa_pos -- 32 cache lines per slot and step for 512 bytes of use: 80 k PSSM
                // pairs moved 10 GB per launch and ran at the memory's pace, 2.2 ms for 0.76 ms of vector instructions -- but the eight columns' rows of
                // pos_aa[position][residue], 256 contiguous bytes per SLOT, 64 per lane, which stage_right_rows looks the residues up in through LDS)
                const signed char* pos_rows = (const signed char*)rp_ + (uint64_t)rj_ * 32 + 64u * (uint32_t)l;
                uint4 row[8];
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const uint32_t res = ((m < 4 ? b_lo : b_hi) >> (8 * (m & 3))) & 31u;
                    const void* src = right_ ? (const void*)(pos_rows + 16 * (m & 3)) : (const void*)(aa_pos + (uint64_t)res * P + pos);
                    __builtin_memcpy(&row[m], src, 16);
                }
                uint4 g1, g2, g3;
                __builtin_memcpy(&g1, (right_ ? goCp : goRp) + pos, 16); __builtin_memcpy(&g2, (right_ ? goRp : goCp) + pos, 16); __builtin_memcpy(&g3, clCp + pos, 16);
#pragma unroll
                for (int m = 0; m < 8; m++) { sp_.S[m][0] = (int)row[m].x; sp_.S[m][1] = (int)row[m].y; sp_.S[m][2] = (int)row[m].z; sp_.S[m][3] = (int)row[m].w; }
                sp_.G1[0] = adds((int)g1.x, fq.ge2); sp_.G1[1] = adds((int)g1.y, fq.ge2); sp_.G1[2] = adds((int)g1.z, fq.ge2); sp_.G1[3] = adds((int)g1.w, fq.ge2);
                sp_.G2[0] = (int)g2.x; sp_.G2[1] = (int)g2.y; sp_.G2[2] = (int)g2.z; sp_.G2[3] = (int)g2.w;
                sp_.G3[0] = (int)g3.x; sp_.G3[1] = (int)g3.y; sp_.G3[2] = (int)g3.z; sp_.G3[3] = (int)g3.w;
            };
            // the slot's A registers and the scalars of the step at the top into the buffer `which` (see ba_multi.hpp; rectangle records as
            // absolute positions here)
            auto stageA = [&](uint32_t which, int s_flag, uint32_t s_i, uint32_t s_j, int s_off, uint32_t s_tt, uint32_t s_nb, int s_dir, int s_offadd, int s_corner) {
                char* b = lbuf + which * 256u + l * 16;
                *(int4*)(b) = int4{A_d[0], A_d[1], A_d[2], A_d[3]}; *(int4*)(b + 64) = int4{A_c[0], A_c[1], A_c[2], A_c[3]};
                if (l == 0) {
                    int* sc = lsc + which * 8u;
                    *(int4*)sc = int4{s_flag | (s_dir << 8), (int)s_i, (int)s_j, s_off}; *(int4*)(sc + 4) = int4{(int)s_tt, (int)s_nb, s_offadd, s_corner};
                }
            };
            constexpr uint32_t NBOOT = SM_B / STEP;
            for (;;) {
                // ---- idle slots take the next pairs of the batch; a pair starts here, with its first block (boot sub-steps, as in k_quad).
                // Pairs shorter than a block in either dimension are the solo driver's from the start.
                while (more && __any(pair == ~0u)) {
                    if (w_next == w_end) {
                        uint32_t v = 0;
                        if (is_lane(0)) v = atomicAdd(bp.work_counter, bp.work_chunk);
                        w_next = (uint32_t)uni((int)v) + excl_n;   // (the first excl_n pairs of the batch order are taken one to a wave, above)
                        if (w_next >= total) { more = false; w_next = w_end = 0; break; }
                        w_end = min(w_next + bp.work_chunk, total);
                    }
                    const bool idle = pair == ~0u;
                    const unsigned long long im = __ballot(idle && l == 0);   // the idle slots, in slot order, take consecutive positions
                    const uint32_t rank = (uint32_t)__popcll(im & ((1ull << (g * SM_LW)) - 1ull));
                    const uint32_t take = min((uint32_t)__popcll(im), w_end - w_next);
                    const uint32_t idx = w_next + rank;
                    w_next += take;
                    if (idle && rank < take) {
                        pair = idx;
                        qlen = bp.q_len[idx]; rlen = bp.r_len[idx];
                        bits = NBOOT | ((qlen < (uint32_t)SM_B || rlen < (uint32_t)SM_B) ? 8u : 0u) | ((uint32_t)DIR_GROW << 4);
                        // the state Block::align starts from (scan_block.rs:123-146), seen as four right steps of 8 columns that end
                        // with the block at (0, 0): borders MIN = 0, no offset yet
                        si = 0; sj = (uint32_t)-(SM_B - STEP); dir = DIR_RIGHT; off = 0; off_max = 0; best_max = 0;
                        ymix = 0; D_corner = 0; trace_top = 0; sel = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) { A_d[k] = 0; A_c[k] = 0; }
                        *(int4*)(lbuf + 256u + 128u + l * 16) = int4{0, 0, 0, 0}; *(int4*)(lbuf + 256u + 192u + l * 16) = int4{0, 0, 0, 0};   // P of buffer 1 (sel = 0)
                        if (l == 0) { *(int4*)lsc = int4{0, 0, 0, 0}; *(int4*)(lsc + 4) = int4{0, 0, 0, 0}; }   // buffer 0's scalars: "no checkpoint" (flag 0) until an improving step writes them -- a slot that leaves before (trace room exhausted at max_size == 32) must not hand the solo driver what the previous tenant left
                        qp = bp.pool + bp.q_off[idx]; rp = bp.pool + bp.r_off[idx];
                        if (TRACE) {
                            const uint64_t t0 = bp.trace_off[idx], b0 = bp.blocks_off[idx];
                            tr16 = (uint32_t)(t0 >> 4); bpos = (uint32_t)b0;
                            room = room_of((uint32_t)min(bp.trace_off[idx + 1] - t0, (uint64_t)0x7fffffffu), (uint32_t)min(bp.blocks_off[idx + 1] - b0, (uint64_t)0x7fffffffu), 0u, 0u);
                        }
                        if (l == 0) {   // what the record holds besides the loop's state
                            int* rc = (int*)(slot_mem + 2 * SM_BUF_BYTES);
                            rc[MR_PAIR] = (int)idx; rc[MR_BEST_I] = 0; rc[MR_BEST_J] = 0; rc[MR_CELLS_LO] = 0; rc[MR_CELLS_HI] = 0;
                            rc[MR_BUDGET] = (int)(64u * ((qlen + rlen) / STEP + 64u)); rc[MR_STATUS] = 0; rc[MR_TSLOT] = (int)idx;
                            rc[MR_SI] = 0; rc[MR_SJ] = -(SM_B - STEP); rc[MR_NSTEPS] = 0;   // (the position the steps are counted from)
                        }
                        pf_ok = false; sp_ok = false;
                    }
                }
                const bool live = pair != ~0u;
                if (!__any(live)) break;

                // ---- the step every live slot is about to take (scan_block.rs:147-246)
                const bool right = dir == DIR_RIGHT;
                const uint32_t boot = bits & 7u; const int prev_dir = (int)((bits >> 4) & 3u), x_iter = (int)((bits >> 6) & 3u);
                const uint32_t ri = right ? si : sj, rj = (right ? sj : si) + (SM_B - STEP);
                const uint32_t lenV = right ? qlen : rlen, lenC = right ? rlen : qlen;
                const bool q_out = si + SM_B > qlen, r_out = sj + SM_B > rlen;
                // a step that could break early at the matrix edge (vectors past the end of their sequence and a column at or past the end of
                // the other; never with X-drop) is not a slot's -- except the step that ends a global alignment (`fin` below)
                bool elig = XDROP || ri + SM_B <= lenV || rj + STEP <= lenC || (boot == 0 && q_out && r_out);
                if (TRACE) elig = elig && room > 0;   // (a step that would not fit is the solo driver's to report)
                leave = live && (!elig || (bits & 8u));
                if (leave && boot) bits |= 8u;   // (a first block cut short: once more from the start)
                const bool run = live && !leave;
                const int off_n = off_max;
                const int off_add = sat16(off - off_n);
                const int corner = (prev_dir != dir && prev_dir != DIR_GROW) ? sat16(D_corner + off_add) : 0;
                // the state before the step: for a slot that leaves (its registers do not survive the step) and for the checkpoint
                if (live) stageA(sel ^ 1u, boot ? 2 : 1, si, sj, off_n, trace_top, bpos, dir, off_add, corner);
                uint2 vb, cbv;
                {
                    const uint8_t* Vp = right ? qp : rp; const uint8_t* Cp = right ? rp : qp;
                    vb = right ? pf_qv : pf_rv; cbv = right ? pf_rc : pf_qc;
                    if (__any(run && !pf_ok)) {   // a slot that has just taken its pair (back)
                        if (run && !pf_ok) {
                            const uint32_t* vp = (const uint32_t*)(Vp + ri + 8 * l);   // (images are 4-byte aligned, positions multiples of 8)
                            vb.x = vp[0]; vb.y = vp[1];
                            const uint32_t* cp = (const uint32_t*)(Cp + rj);
                            cbv.x = cp[0]; cbv.y = cp[1];
                        }
                    }
#ifdef SM_PREFETCH_EARLY
                    asm volatile("" : "+v"(vb.x), "+v"(vb.y));   // (consume the old prefetch before the next one is issued: the memory counter is in-order)
                    // sequence bytes of the step after this one, whichever way it goes: in flight across the columns, so that the wait for them at
                    // the top of the next step does not wait for this step's trace stores (issued behind them: the memory counter is in order)
                    if (run) {
                        const uint32_t* a = (const uint32_t*)(qp + si + 8 * l); const uint32_t* b = (const uint32_t*)(rp + (boot > 1 ? 0u : sj) + 8 * l);   // (first block: sj is not a position yet)
                        const uint32_t* cq = (const uint32_t*)(qp + si + SM_B); const uint32_t* cr = (const uint32_t*)(rp + (uint32_t)(sj + SM_B));   // (32-bit sum: sj is "negative" during the first block)
                        pf_qv.x = a[0]; pf_qv.y = a[1]; pf_rv.x = b[0]; pf_rv.y = b[1];
                        pf_qc.x = cq[0]; pf_qc.y = cq[1]; pf_rc.x = cr[0]; pf_rc.y = cr[1];
                        pf_ok = true;
                    }
#endif
                }
                uint32_t* tw = nullptr;
                if (TRACE) {
                    // (a slot without a step writes to the wave's sink: the first 64 x 16 words of the arena's sink area, ba_host.cpp; selected by
                    // 32-bit pieces: no 64-bit pointer pair lives across the step)
#ifdef SM_X_SINK   // (development: every trace store to the sink -- what do the stores' addresses cost?)
                    const uint32_t t16 = fill_wave * 64u + (uint32_t)lane, tt = 0u;
#else
                    const uint32_t t16 = run ? tr16 : fill_wave * 64u + (uint32_t)lane, tt = run ? trace_top : 0u;
#endif
                    tw = bp.trace_arena + ((uint64_t)t16 << 4) + tt + 8 * l;
#ifndef SM_X_NOREC
                    if (run && l == 0) {   // add_block(i, j, width, height, right) in matrix orientation (scan_block.rs:154,204,284); the first block: four records
                        BlockRec br;
                        br.i = (right ? ri : rj) | 0x80000000u;   // (bit 31: words of 4 cells x 2 columns, see multi_rect)
                        br.j = right ? rj : ri; br.h = (uint16_t)(right ? SM_B : STEP); br.w = (uint16_t)(right ? STEP : SM_B);
                        br.trace_base = trace_top | (right ? 0x80000000u : 0u);
                        bp.blocks[bpos] = br;
                    }
#endif
                }
                if constexpr (KIND == KIND_PROFILE) {
                    // The step's operands were loaded at the end of the previous step (behind its decisions, see below) -- unless the slot has just taken its
                    // pair (back): then here, with the latency in the open.
                    if (__any(run && !sp_ok)) {
                        if (run && !sp_ok) {
                            load_profile_ops(spf, right, ri, rj, vb, cbv, rp, rlen);
                            sp_vb = vb;
                            const uint32_t* wv = (const uint32_t*)(qp + si + 8 * l); const uint32_t* wc = (const uint32_t*)(qp + si + (SM_B - STEP));
                            win_v = uint4{wv[0], wv[1], wv[2], wv[3]}; win_c = uint4{wc[0], wc[1], wc[2], wc[3]}; win_si = si;
                            sp_ok = true;
                        }
                    }
                    {   // right steps: the eight columns' rows (this lane's 64 of the slot's 256 bytes in S[0 .. 3]) through LDS, then S[m] = residue m x 8 columns
                        signed char* stg = (signed char*)base + sm_wave_bytes_h(LCLS) + (uint32_t)g * 256u;
                        const bool rr = run && right;
                        if (rr) {
#pragma unroll
                            for (int m = 0; m < 4; m++) *(int4*)(stg + 64 * l + 16 * m) = int4{spf.S[m][0], spf.S[m][1], spf.S[m][2], spf.S[m][3]};
                        }
                        lds_sync();
                        if (rr) {
#pragma unroll
                            for (int m = 0; m < 8; m++) {
                                const uint32_t res = ((m < 4 ? sp_vb.x : sp_vb.y) >> (8 * (m & 3))) & 31u;
#pragma unroll
                                for (int c = 0; c < 4; c++) spf.S[m][c] = pk((int)stg[(2 * c) * 32 + res], (int)stg[(2 * c + 1) * 32 + res]);
                            }
                        }
                        lds_sync();
                    }
                    spf.selE = right ? 0x05040504 : 0x03020100; spf.selO = right ? 0x07060706 : 0x03020100;
                    spf.selXE = right ? 0x05040504 : 0x0c0c0c0c; spf.selXO = right ? 0x07060706 : 0x0c0c0c0c; spf.selY = right ? 0x0c0c0c0c : 0x03020100;
                    spf.rmask = __ballot(right);
                }
                const bool fin = !XDROP && run && boot == 0 && q_out && r_out;   // the last step of a global alignment
                const bool fin_any = !XDROP && __any(fin);
                const uint32_t fin_col = lenC - rj;   // its last computed column (0 .. 7: columns rj .. lenC)
                int dsel[4] = {0, 0, 0, 0};
                int Pn_d[4], Pn_r[4];   // the orthogonal border pair after the step
                MultiOut o;
                int zw[2] = {0, 0};
                small_rect<KIND, TRACE, !XDROP, SPM>(smem, fq, mc, l, A_d, A_c, Pn_d, Pn_r, lbuf + (sel ^ 1u) * 256u + 128u + l * 16, vb, cbv.x, cbv.y, corner, off_add,
                                                     run && boot == NBOOT && l == 0 && SPM != 1, tw, fin_any, fin ? fin_col : 8u, dsel, o, KIND == KIND_PROFILE ? &spf : nullptr,
                                                     SPM ? splat(sat16(ZERO - off_n)) : 0, SPM == 2 && right && ri == 0u && l == 0, zw);
                if (TRACE && SPM == 1) *(int2*)(tw - 8 * l + 32 + 2 * l) = int2{zw[0], zw[1]};   // the rectangle's zero mask behind its 32 trace words (idle slots: the sink)
#ifndef SM_PREFETCH_EARLY
                // sequence bytes of the step after this one, whichever way it goes: issued behind the columns (the eight registers are not live
                // across them; consecutive steps read consecutive bytes, mostly out of the L1). The alternative -- issued at the top, in flight
                // across the columns (-DSM_PREFETCH_EARLY) -- costs ~60 more instructions per step for the same time (same-box A/B: C2 -3.5 %).
                // (profiles: the query's bytes come out of the windows, see load_profile_ops)
                if (KIND != KIND_PROFILE && run) {
                    const uint32_t* a = (const uint32_t*)(qp + si + 8 * l); const uint32_t* b = (const uint32_t*)(rp + (boot > 1 ? 0u : sj) + 8 * l);
                    const uint32_t* cq = (const uint32_t*)(qp + si + SM_B); const uint32_t* cr = (const uint32_t*)(rp + (uint32_t)(sj + SM_B));
                    pf_qv.x = a[0]; pf_qv.y = a[1]; pf_rv.x = b[0]; pf_rv.y = b[1];
                    pf_qc.x = cq[0]; pf_qc.y = cq[1]; pf_rc.x = cr[0]; pf_rc.y = cr[1];
                    pf_ok = true;
                }
#endif
                if (boot) {   // the first block's maximum so far
                    ymix = boot == NBOOT ? o.mx : max(ymix, o.mx);
                    o.mx = ymix;
                }
                const bool bsub = run && boot > 1, blast = run && boot == 1;

                // ---- what does the step call for? (scan_block.rs:332-558; nothing is committed yet)
                const int right_max = right ? o.act_max8 : o.pas_max8, down_max = right ? o.pas_max8 : o.act_max8;
                const int new_off_max = off_n + o.mx - ZERO;
                const bool improve = new_off_max > best_max;
                const uint32_t new_y = improve ? 0u : (uint32_t)ymix + 1;   // (during the first block ymix is not a count: its last sub-step goes on only if it improves)
                bool stop = q_out && r_out;                                                                   // end of the matrix
                if (XDROP) stop = stop || (!improve && new_off_max < best_max - x_drop && x_iter >= 1);      // X-drop termination
                stop = stop || (!q_out && !r_out && 2 * (uint32_t)SM_B <= max_size && new_y > SM_B / STEP - 1);   // grow
                // the first block: no decisions between its sub-steps; after the last one anything but a plain shift step (an empty improvement --
                // the driver would grow at once --, the end of the matrix) sends the pair to the solo driver, from its start
                const bool again = blast && (stop || !improve);
                const bool commit = run && !stop && !bsub && !again;
                if (!XDROP && fin_any) {
                    // The last step of a global alignment: its columns stop at the end of the column sequence (scan_block.rs:1216-1224; the trace
                    // index still advances over the whole rectangle), the score is the vector border's entry at the end of the vector sequence
                    // after the last computed column (scan_block.rs:560-570). The pair is complete: results are written here.
                    const uint32_t idx = lenV - ri;         // 0 .. 31: both ends are inside the block
                    const uint32_t kreg = (idx >> 1) & 3u;
                    const int dv = kreg == 0 ? dsel[0] : (kreg == 1 ? dsel[1] : (kreg == 2 ? dsel[2] : dsel[3]));
                    const int v = __builtin_amdgcn_ds_bpermute((int)(((uint32_t)lane & ~3u) + (idx >> 3)) << 2, dv);
                    const int sc16 = (idx & 1) ? v >> 16 : (int)(short)v;
                    if (fin && l == 0) {
                        const char* rcp = slot_mem + 2 * SM_BUF_BYTES;
                        const unsigned long long cells0 = (unsigned long long)(uint32_t)mq_load(rcp + 4 * MR_CELLS_LO) | ((unsigned long long)(uint32_t)mq_load(rcp + 4 * MR_CELLS_HI) << 32);
                        const uint32_t nb1 = TRACE ? bpos - (uint32_t)bp.blocks_off[pair] + 1u : 0u;
                        bp.score[pair] = off_n + sc16 - ZERO; bp.query_idx[pair] = qlen; bp.reference_idx[pair] = rlen;
                        const uint32_t nsteps = (uint32_t)mq_load(rcp + 4 * MR_NSTEPS) + ((si + sj) - ((uint32_t)mq_load(rcp + 4 * MR_SI) + (uint32_t)mq_load(rcp + 4 * MR_SJ))) / STEP;
                        if (bp.cells) bp.cells[pair] = cells0 + (unsigned long long)nsteps * (STEP * SM_B) + (unsigned long long)(fin_col + 1u) * SM_B;
                        if (bp.status) bp.status[pair] = (uint32_t)mq_load(rcp + 4 * MR_STATUS);
                        if (bp.nblocks_out) bp.nblocks_out[pair] = nb1;
                        if (bp.trace_words_out) bp.trace_words_out[pair] = TRACE ? trace_top + SM_TW : 0;
                        if (bp.slot_out) bp.slot_out[pair] = pair;
                        if (TRACE) bp.slot_info[pair] = SlotInfo{pair, nb1, qlen, rlen};
                    }
                    if (fin) pair = ~0u;
                }
                if (again) bits |= 8u;
                leave = leave || again || (run && stop && !bsub && !blast && !fin);   // rolled back: the pair's state is what was staged before this step
                if (bsub) {
                    sj += STEP; bits -= 1u;
                    if (TRACE) { trace_top += SM_TW; bpos++; room--; }
                }
                bool swap = false;
                if (commit) {
                    if (improve) {
                        if (keep_pre) sel ^= 1u;   // the state staged before this step is the checkpoint now (and yields the location of the maximum)
                        best_max = new_off_max;
                    }
                    off = off_n; off_max = new_off_max; ymix = (int)new_y; D_corner = blast ? 0 : o.corner_new;
                    if (TRACE) { trace_top += SM_TW; bpos++; room--; }
                    const uint32_t nx = (XDROP && off_max < best_max - x_drop) ? (uint32_t)x_iter + 1u : 0u;
                    bits = ((uint32_t)(blast ? DIR_GROW : dir) << 4) | (nx << 6);   // boot = 0, no restart flag
                    const bool go_down = r_out || (!q_out && down_max > right_max);   // forced at the matrix edge, else greedy (ties -> right)
                    si += go_down ? (uint32_t)STEP : 0u; sj += go_down ? 0u : (uint32_t)STEP;
                    const int ndir = go_down ? DIR_DOWN : DIR_RIGHT;
                    swap = ndir != dir;   // the borders change roles with the direction
                    dir = ndir;
                }
                if (commit || bsub) {   // the orthogonal pair of the next step: to its place in the buffer the next step is staged in
                    int4 npd, npr;
                    npd.x = swap ? A_d[0] : Pn_d[0]; npd.y = swap ? A_d[1] : Pn_d[1]; npd.z = swap ? A_d[2] : Pn_d[2]; npd.w = swap ? A_d[3] : Pn_d[3];
                    npr.x = swap ? A_c[0] : Pn_r[0]; npr.y = swap ? A_c[1] : Pn_r[1]; npr.z = swap ? A_c[2] : Pn_r[2]; npr.w = swap ? A_c[3] : Pn_r[3];
#pragma unroll
                    for (int k = 0; k < 4; k++) { A_d[k] = swap ? Pn_d[k] : A_d[k]; A_c[k] = swap ? Pn_r[k] : A_c[k]; }
                    char* b = lbuf + (sel ^ 1u) * 256u + l * 16;
                    *(int4*)(b + 128) = npd; *(int4*)(b + 192) = npr;
                }
                if constexpr (KIND == KIND_PROFILE) {
                    // ---- the next step's operands, as soon as it is known where it goes (see load_profile_ops); then this step's trace words
                    const bool goes_on = (commit || bsub) && pair != ~0u;
                    if (goes_on) {
                        const bool nright = dir == DIR_RIGHT;
                        const bool hi = si != win_si;   // (the next step starts at win_si or 8 further down)
                        uint2 nvb, ncb;
                        nvb.x = hi ? win_v.z : win_v.x; nvb.y = hi ? win_v.w : win_v.y; ncb.x = hi ? win_c.z : win_c.x; ncb.y = hi ? win_c.w : win_c.y;
                        load_profile_ops(spf, nright, nright ? si : sj, (nright ? sj : si) + (SM_B - STEP), nvb, ncb, rp, rlen);
                        sp_vb = nvb;
                        const uint32_t* wv = (const uint32_t*)(qp + si + 8 * l); const uint32_t* wc = (const uint32_t*)(qp + si + (SM_B - STEP));
                        win_v = uint4{wv[0], wv[1], wv[2], wv[3]}; win_c = uint4{wc[0], wc[1], wc[2], wc[3]}; win_si = si;
                    }
                    sp_ok = goes_on; pf_ok = goes_on;
                    if (TRACE) { *(int4*)tw = int4{spf.twords[0], spf.twords[1], spf.twords[2], spf.twords[3]}; *(int4*)(tw + 4) = int4{spf.twords[4], spf.twords[5], spf.twords[6], spf.twords[7]}; }
                }
                if (__any(leave)) break;
            }
            // ---- every slot's state at the top of the loop to memory: the A registers of the slots that took the step (the others' were
            // staged before it; P is in the buffer already)
            const bool live = pair != ~0u;
            if (live && !leave) stageA(sel ^ 1u, (bits & 7u) ? 2 : 1, si, sj, 0, trace_top, bpos, dir, 0, 0);
            lds_sync();
            if (live) {   // both buffers back to the arena (solo mode needs the LDS region, and reads a pair's state from the arena)
#pragma unroll
                for (int w = 0; w < 2; w++) {
                    const char* sp = lbuf + w * 256 + l * 16;
                    char* d = slot_mem + w * SM_BUF_BYTES + l * 16;
#pragma unroll
                    for (int a = 0; a < 4; a++) *(int4*)(d + a * 64) = *(const int4*)(sp + a * 64);
                    *(int*)(slot_mem + w * SM_BUF_BYTES + 256 + 4 * l) = lsc[w * 8 + l]; *(int*)(slot_mem + w * SM_BUF_BYTES + 272 + 4 * l) = lsc[w * 8 + 4 + l];
                }
            }
            lds_sync();
            if (live && l == 0) {
                int* rc = (int*)(slot_mem + 2 * SM_BUF_BYTES);
                const char* rcp = (const char*)rc;
                rc[MR_NSTEPS] = (int)((uint32_t)mq_load(rcp + 4 * MR_NSTEPS) + ((si + sj) - ((uint32_t)mq_load(rcp + 4 * MR_SI) + (uint32_t)mq_load(rcp + 4 * MR_SJ))) / STEP);
                rc[MR_SI] = (int)si; rc[MR_SJ] = (int)sj; rc[MR_DIR] = dir; rc[MR_PREV_DIR] = (int)((bits >> 4) & 3u); rc[MR_OFF] = off; rc[MR_OFF_MAX] = off_max; rc[MR_BEST_MAX] = best_max;
                rc[MR_Y_DROP] = (bits & 7u) ? 0 : ymix; rc[MR_X_ITER] = (int)((bits >> 6) & 3u); rc[MR_D_CORNER] = D_corner; rc[MR_TRACE_TOP] = (int)trace_top;
                rc[MR_NBLOCKS] = TRACE ? (int)(bpos - (uint32_t)bp.blocks_off[pair]) : 0; rc[MR_SEL] = (int)sel; rc[MR_FLAGS] = (int)((bits >> 3) & 1u); rc[MR_BOOT] = (int)(bits & 7u);
                rc[MR_BMX] = (bits & 7u) ? ymix : 0;
            }
            const unsigned long long lm = __ballot(leave && l == 0), vm = __ballot(live && l == 0);
            pend_m = 0; live_m = 0;
#pragma unroll
            for (int s = 0; s < SM_NS; s++) { pend_m |= (uint32_t)((lm >> (s * SM_LW)) & 1ull) << s; live_m |= (uint32_t)((vm >> (s * SM_LW)) & 1ull) << s; }
        }
    }
}

}  // namespace ba
