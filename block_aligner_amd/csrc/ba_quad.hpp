// block_aligner_amd — four pairs per wavefront while the block is 32 cells.
//
// A 32-cell block fills 16 of a wave's 64 lanes (two cells per lane), so the per-pair kernels (ba_driver.hpp) spend three
// quarters of every vector instruction on nothing when a batch starts at the reference's usual minimum block size
// (examples/uc_bench.rs:85-100, examples/pssm_bench.rs:94-100: 32..=256). k_quad runs FOUR pairs per wave, one per 16-lane
// DPP row ("slot"): row_shr DPP stays inside a row, so the column code of fast_rect carries over lane for lane; what is
// wave-uniform there (offsets, direction, column bytes, the whole driver state) is row-uniform here and lives in VGPRs,
// replicated across the slot's 16 lanes, and the driver decisions of scan_block.rs:332-558 are evaluated for all four slots at
// once with vector compares and selects. A slot executes a pair's first block and then only PLAIN shift steps at 32 cells:
// whatever else the pair needs (a grow, termination, the early column break at the matrix edge) is done by the per-pair kernel
// afterwards: a pair leaves as a PairCont record (ba_params.h) holding the driver's state at the top of its loop; a step whose
// outcome calls for anything but another shift is rolled back and the pair leaves with its pre-step state.
// All four kinds (sequence-to-profile steps: QuadProfile below). TRACE batches are pair-slot batches (ba_params.h): a pair's
// trace words and rectangle records go to the pair's own region, which the per-pair kernel continues and k_walk reads after
// the last fill kernel.
#pragma once
#include "ba_driver.hpp"

namespace ba {

constexpr int QUAD_B = 32;             // block size of a slot
constexpr int QUAD_SLOT_BYTES = 512;   // LDS per slot: D scratch at 0, R scratch at +128, the two sinks at +256 / +384
constexpr int QUAD_PR = 64;            // D -> R scratch, in entries

// lane SRC (0..15) of every 16-lane slot to all lanes of the slot: ds_swizzle, bit-mask mode (and 0x10, or SRC)
template <int SRC>
__device__ __forceinline__ int slot_bcast(int v) { return __builtin_amdgcn_ds_swizzle(v, 0x10 | (SRC << 5)); }
__device__ __forceinline__ int row_shr1_z(int src) { return __builtin_amdgcn_update_dpp(0, src, 0x111, 0xf, 0xf, true); }   // lane l <- l-1 inside the row; row lane 0 <- 0
__device__ __forceinline__ int add_row_shr1(int a, int b) {   // a[l-1] + b[l] inside a row (row lane 0: 0 + b)
    int t;
    asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(t) : "v"(a), "v"(b));
    return t;
}

// ---- eight cells per lane (k_multi, k_small, place_rect8): the pieces of a column's R11 (avx2.rs:297-338 per 16-cell vector + the carry from
// vector to vector, scan_block.rs:1144-1150), named so that the kernels and the device-side known-answer test of the lane primitives
// (k_lane_kat, ba_kernels.hip) run the same code. x = D11_open (scan_block.rs:1143) of the lane's cells 2k, 2k + 1 in register k.
__device__ __forceinline__ int scan8_splat_lo(int x) { const s16x2 t = as_s(x); return as_i(s16x2{t.x, t.x}); }
__device__ __forceinline__ int scan8_splat_hi(int x) { const s16x2 t = as_s(x); return as_i(s16x2{t.y, t.y}); }
// inside a register: the second cell against the first's gap
__device__ __forceinline__ int scan8_inreg(int x, int ge2) { return vmax(x, scan8_splat_lo(adds(x, ge2))); }
// the chain over the lane's four registers; G0 = {g, 2g}
__device__ __forceinline__ void scan8_chain(int (&r)[4], int G0) {
#pragma unroll
    for (int k = 1; k < 4; k++) r[k] = vmax(r[k], adds(scan8_splat_hi(r[k - 1]), G0));
}
// what entered the lane from above (cs: see multi_carry / small_carry), into register k; last = the lane's fourth register, whose second cell has its
// own floor in the high half of cs
__device__ __forceinline__ int scan8_apply(int rk, int cs, int Gk, bool last) { return vmax(rk, adds(last ? cs : scan8_splat_lo(cs), Gk)); }
__device__ __forceinline__ int sat16(int x) { return x < -32768 ? -32768 : (x > 32767 ? 32767 : x); }

struct QuadOut { int mx, row, col, act_max8, pas_max8, corner_new; };

// Sequence-to-profile steps (place_block_profile_*, scan_block.rs:612-783). A slot's step runs along the query ("right": one
// profile position per column, costs uniform in a column) or along the profile ("down": one query residue per column, costs per
// row), and the four slots of a wave differ: every per-column operand is picked by v_perm with a per-lane selector, from the
// column-packed registers (right) or from the lane's own pair (down).
struct QuadProfile {
    int A[4], B[4];                 // right: the 8 columns' scores of this lane's residues a / b (aa_pos rows); down: columns 2m / 2m + 1
    int selE, selO;                 // score selectors for even / odd columns
    int pgoC[4], pclC[4], pgoR[4];  // right: gap_open_C + extend, gap_close_C, gap_open_R of the 8 columns (packed); down: unused
    int vgoC, vgoR, vclR;           // down: gap_open_R + extend, gap_open_C, gap_close_C of this lane's two rows (swapped roles,
                                    // scan_block.rs:671-682); right: unused (vclR = 0)
    int gselE, gselO;               // cost selectors: right = splat the even / odd half of the packed register, down = the lane's pair
};

// One 8-column shift step for the four slots of a wave (fast_rect with row-uniform operands in VGPRs).
template <int KIND, bool TRACE, bool XDROP>
__device__ __forceinline__ void quad_rect(const char* table, const FillConsts& fc, int l, int& Ad, int& Ac, int& Pd, int& Pr, short* Pl, short* sink,
                                          int vec_a, int vec_b, uint32_t cb_lo, uint32_t cb_hi, int corner, int off_add, int loc_thr,
                                          uint32_t* __restrict__ tout, bool store, bool first_cell, QuadOut& o, int (&dcol)[STEP], const QuadProfile* pq = nullptr) {
    constexpr bool PROF = KIND == KIND_PROFILE;
    const int offa = splat(off_add);
    int d = adds(Ad, offa), c = adds(Ac, offa);
    const int pd = adds(Pd, offa), pr = adds(Pr, offa);
    lds_fence();
    *(int*)(Pl + 2 * l) = pd; *(int*)(Pl + QUAD_PR + 2 * l) = pr;
    o.corner_new = slot_bcast<3>(pd) >> 16;
    const ScoreKey<KIND> key = make_key<KIND>(vec_a, vec_b);
    int dmax = 0, tacc = 0;
    short* last_base = l == 15 ? Pl + QUAD_B : sink;
#pragma unroll
    for (int j = 0; j < STEP; j++) {
        int sc, goC = fc.go2, goR = fc.ome2, clC = 0;
        if constexpr (PROF) {
            const int m = j >> 1;
            sc = __builtin_amdgcn_perm(pq->B[m], pq->A[m], (j & 1) ? pq->selO : pq->selE);
            const int gsel = (j & 1) ? pq->gselO : pq->gselE;
            goC = __builtin_amdgcn_perm(pq->pgoC[m], pq->vgoC, gsel);
            goR = __builtin_amdgcn_perm(pq->pgoR[m], pq->vgoR, gsel);
            clC = __builtin_amdgcn_perm(pq->pclC[m], 0, gsel);
        } else {
            const int cb = (int)(((j < 4 ? cb_lo : cb_hi) >> (8 * (j & 3))) & 0xffu);
            sc = fetch_score<KIND>(table, key, cb);
        }
        int prev = row_shr1_z(d);
        if (j == 0) prev = l == 0 ? (int)((uint32_t)corner << 16) : prev;
        const int d00 = __builtin_amdgcn_alignbit(d, prev, 16);
        int d11 = adds(d00, sc);
        if (j == 0) d11 = first_cell ? (int)(((uint32_t)d11 & 0xffff0000u) | (uint32_t)ZERO) : d11;   // cell (0,0) starts from the relative zero (scan_block.rs:1130-1132)
        const int copen = adds(d, goC);
        const int cn = vmax(adds(c, fc.ge2), copen);
        const int cend = PROF ? adds(cn, clC) : cn;          // C11_end (scan_block.rs:697-701)
        d11 = vmax(d11, cend);
        const int x = adds(d11, goR);
        const s16x2 t2 = as_s(adds(x, fc.ge2));
        int r = vmax(x, as_i(s16x2{t2.x, t2.x}));
        const int pm = wave_prefix_max16((int)as_s(r).y - fc.laneKG);
        const s16x2 cs = as_s(add_row_shr1(pm, fc.lanem1KG));
        r = vmax(vmax(r, adds(as_i(s16x2{cs.x, cs.x}), fc.g12)), fc.vconst_top);
        const int rend = PROF ? adds(r, pq->vclR) : r;       // R11_end (scan_block.rs:712-716)
        const int dn = vmax(d11, rend);
        if (TRACE) {   // the cell's four flags as sign bits of saturating differences, one nibble per column (see fast_rect)
            const uint32_t sC = (uint32_t)subs(cend, dn), sR = (uint32_t)subs(rend, dn), sCo = (uint32_t)subs(copen, cn), sRo = (uint32_t)subs(x, r);
            const uint32_t hi2 = bfi(0x80008000u, sRo, sCo >> 1), lo2 = bfi(0x80008000u, sR, sC >> 1);
            const uint32_t nib = bfi(0xC000C000u, hi2, lo2 >> 2);
            tacc = (int)(((uint32_t)tacc >> 4) | (nib & 0xF000F000u));
            if ((j & 3) == 3) {
                if (store) __hip_atomic_store(tout + (j >> 2) * (QUAD_B / 2) + l, (uint32_t)tacc, BA_RLX_AGENT);   // (through the L2: see k_quad's wt())
                tacc = 0;
            }
        }
        dmax = vmax(dmax, dn);
        dcol[j] = dn;
        d = dn; c = cn;
        last_base[j] = (short)(d >> 16); last_base[QUAD_PR + j] = (short)(r >> 16);
    }
    lds_fence();
    Pd = *(const int*)(Pl + 2 * l + STEP); Pr = *(const int*)(Pl + QUAD_PR + 2 * l + STEP);
    Ad = d; Ac = c;
    {   // max of the first 8 entries of both borders (lanes 0..3 of the slot), to every lane of the slot
        const s16x2 sa = as_s(d), sb = as_s(Pd);
        const int xa = vmax(d, as_i(s16x2{sa.y, sa.x})), xb = vmax(Pd, as_i(s16x2{sb.y, sb.x}));
        int m = __builtin_amdgcn_perm(xb, xa, 0x05040100);
        m = vmax(m, __builtin_amdgcn_update_dpp(m, m, 0xB1, 0xf, 0xf, false));
        m = vmax(m, __builtin_amdgcn_update_dpp(m, m, 0x4E, 0xf, 0xf, false));
        const s16x2 rr = as_s(slot_bcast<0>(m));
        o.act_max8 = rr.x; o.pas_max8 = rr.y;
    }
    const int m32 = max(dmax & 0xffff, (int)((uint32_t)dmax >> 16));   // halves are >= 0 (D_max starts at MIN = 0)
    const int M = slot_bcast<15>(wave_prefix_max16(m32));
    o.mx = M; o.row = 0; o.col = 0;
    if (XDROP && __any(M > loc_thr)) {   // (some slot raises its best score: the location matters)
        // location of the maximum inside each slot: among the cells equal to it, smallest row % 16, then largest column, then
        // largest row (avx2.rs:271-274 + scan_block.rs:1198-1200). jl1 = 1 + the last column in which this lane's cell holds M.
        const int Ms = splat(M);
        int jl1 = 0;
#pragma unroll
        for (int j = 0; j < STEP; j++) jl1 = vmaxu(jl1, pk_mul(eq01(dcol[j], Ms, fc.ones), splat(j + 1)));
        int best = 0;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int jl = h ? (jl1 >> 16) & 0xffff : jl1 & 0xffff;
            const int row = 2 * l + h;
            const int k = ((15 - (row & 15)) << 11) | (jl << 7) | row;
            best = max(best, jl ? k : 0);
        }
        best = slot_bcast<15>(wave_prefix_max16(best));
        const int jl = (best >> 7) & 15;
        o.col = jl ? jl - 1 : 0;
        o.row = jl ? best & 127 : 0;   // no cell equals the maximum (it is the initial MIN): lane 0 / column 0 / vector 0
    }
}

template <int KIND, bool TRACE, bool XDROP>
__global__ void __launch_bounds__(WAVES_PER_WG * 64, 4) k_quad(const BatchParams bp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id(), l = lane & 15, g = lane >> 4;
    const int wave = (int)threadIdx.x >> 6;
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
    char* sl = smem + lds_table_bytes_h(KIND) + (uint32_t)(wave * 4 + g) * QUAD_SLOT_BYTES;
    short* Pl = (short*)sl;
    short* sink = (short*)(sl + 256) + l;
    FillConsts fc;
    fc = make_fill_consts(l, bp.gap_open, bp.gap_extend);
    const uint32_t total = bp.n;
    const uint32_t max_size = bp.max_size;
    const int x_drop = bp.x_drop;
    if (lane == 0) __hip_atomic_store(bp.cq_ctrl + 20, 1u, BA_RLX_AGENT);   // on the device: the launch beside this one may take queue tickets (k_align)

    // ---- slot state (row-uniform, replicated over the slot's lanes)
    uint32_t pair = ~0u, si = 0, sj = 0, qlen = 0, rlen = 0, y_drop = 0, best_i = 0, best_j = 0, ck_i = 0, ck_j = 0, budget = 0, nsteps = 0;
    int dir = DIR_RIGHT, prev_dir = DIR_GROW, off = 0, off_max = 0, best_max = 0, x_iter = 0, D_corner = 0, ck_off = 0;
    unsigned long long cells0 = 0;
    const uint8_t* qp = bp.pool; const uint8_t* rp = bp.pool;
    int Dcol = 0, Ccol = 0, Drow = 0, Rrow = 0, ck0 = 0, ck1 = 0, ck2 = 0, ck3 = 0;
    // sequence bytes of the next step, fetched one step ahead for both possible directions (a pair's steps are a dependent
    // chain: without this every step of the batch's longest pair waits a memory round trip)
    int pf_qv = 0, pf_rv = 0; uint2 pf_qc = {0, 0}, pf_rc = {0, 0}; bool pf_ok = false;
    // TRACE: the pair's trace stack (its own region of the arenas) and the stack heights of the checkpoint
    uint32_t trace_top = 0, nblocks = 0, ck_tt = 0, ck_nb = 0, tcap = 0, bcap = 0;
    uint32_t* tr = bp.trace_arena; BlockRec* bl = bp.blocks;
    // The first block (scan_block.rs:260-305 with prev_size = 0: one 32 x 32 rectangle) runs as four right steps of 8 columns
    // over zeroed borders -- the same recurrences, the row border filling up as it shifts --: boot = sub-steps left; no driver
    // decision is taken between them, their maxima combine (bmx, brow, bcol) and the last one is followed by the driver step of a grow.
    int boot = 0, bmx = 0, brow = 0, bcol = 0;
    uint32_t w_next = 0, w_end = 0;   // this wave's share of the work counter (wave-uniform)
    bool more = true;

    // What a pair's next kernel reads while this one is still running (records, trace words, rectangle records) is written with
    // agent-scope stores: they go through this XCD's L2 to memory, so that publishing a record needs no release fence -- which
    // would write back every dirty line of the L2 -- only the wait for the stores' completion.
    auto wt = [](uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, BA_RLX_AGENT); };
    // a pair for the per-pair kernel: its queue entry (1 + 2 * pair + from-scratch bit), ba_params.h
    auto enqueue = [&](uint32_t entry) {
        const uint32_t pos = __hip_atomic_fetch_add(bp.cq_ctrl, 1u, BA_RLX_AGENT);
        __hip_atomic_store(bp.cq_queue + pos, entry, BA_RLX_AGENT);
    };
    for (;;) {
        // ---- idle slots take the next pairs of the batch. A pair starts here, with its first block (boot, below); pairs shorter
        // than a block in either dimension are the per-pair kernel's from the start (flag 2: no record, run it from scratch)
        bool idle = pair == ~0u;
        while (more && __any(idle)) {
            if (w_next == w_end) {
                uint32_t v = 0;
                if (lane == 0) v = atomicAdd(bp.work_counter, bp.work_chunk);
                w_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
                if (w_next >= total) { more = false; break; }
                w_end = min(w_next + bp.work_chunk, total);
            }
            // the idle slots, in slot order, take consecutive positions
            const unsigned long long im = __ballot(idle && l == 0);
            const uint32_t rank = (uint32_t)__popcll(im & ((1ull << (g * 16)) - 1ull));
            const uint32_t idx = w_next + rank;
            const uint32_t take = min((uint32_t)__popcll(im), w_end - w_next);
            w_next += take;
            if (idle && rank < take) {
                qlen = bp.q_len[idx]; rlen = bp.r_len[idx];
                if (qlen < (uint32_t)QUAD_B || rlen < (uint32_t)QUAD_B) { if (l == 0) { bp.cont_out_flag[idx] = 2u; enqueue(2u * idx + 2u); } }
                else {
                    pair = idx;
                    // the state Block::align starts from (scan_block.rs:123-146), seen as four right steps of 8 columns that
                    // end with the block at (0, 0): borders MIN = 0, no offset yet
                    si = 0; sj = (uint32_t)-(QUAD_B - STEP); dir = DIR_RIGHT; prev_dir = DIR_GROW; off = 0; off_max = 0; best_max = 0;
                    y_drop = 0; x_iter = 0; D_corner = 0; best_i = 0; best_j = 0; ck_i = 0; ck_j = 0; ck_off = 0; cells0 = 0; nsteps = 0;
                    budget = 64u * ((qlen + rlen) / STEP + 64u);   // the per-pair kernel's watchdog
                    Dcol = 0; Ccol = 0; Drow = 0; Rrow = 0; ck0 = 0; ck1 = 0; ck2 = 0; ck3 = 0;
                    boot = QUAD_B / STEP; bmx = 0; brow = 0; bcol = 0;
                    qp = bp.pool + bp.q_off[pair]; rp = bp.pool + bp.r_off[pair];
                    if (TRACE) {
                        trace_top = 0; nblocks = 0; ck_tt = 0; ck_nb = 0;
                        const uint64_t t0 = bp.trace_off[pair], b0 = bp.blocks_off[pair];
                        tr = bp.trace_arena + t0; bl = bp.blocks + b0;
                        tcap = (uint32_t)min(bp.trace_off[pair + 1] - t0, (uint64_t)0x7fffffffu); bcap = (uint32_t)min(bp.blocks_off[pair + 1] - b0, (uint64_t)0x7fffffffu);
                    }
                    pf_ok = false;
                }
            }
            idle = pair == ~0u;
        }
        if (__all(idle)) break;

        // ---- the step every live slot is about to take
        const bool right = dir == DIR_RIGHT;
        const uint32_t ri = right ? si : sj, rj = (right ? sj : si) + (QUAD_B - STEP);
        const uint32_t lenV = right ? qlen : rlen, lenC = right ? rlen : qlen;
        // a step that could break early at the matrix edge (vectors past the end of their sequence and a column at or past
        // the end of the other; never with X-drop) is not ours: leave before it
        // -- except the step that ends a global alignment (both at once: the block has reached the last row and the last column):
        // that one is finished here, see `fin` below
        const bool q_out = si + QUAD_B > qlen, r_out = sj + QUAD_B > rlen;
        bool elig = XDROP || ri + QUAD_B <= lenV || rj + STEP <= lenC || (boot == 0 && q_out && r_out);
        if (TRACE) elig = elig && nblocks < bcap && trace_top + (STEP * QUAD_B / 8) + 64 <= tcap;   // (a step that would not fit is the per-pair kernel's to report)
        bool leave = !idle && (!elig || budget <= 1);
        const bool run = !idle && !leave;
        const int off_n = off_max;
        const int off_add = sat16(off - off_n);
        const int corner = (prev_dir != dir && prev_dir != DIR_GROW) ? sat16(D_corner + off_add) : 0;
        int vc = 0; uint32_t cb_lo = 0, cb_hi = 0;
        QuadProfile pq{};
        if constexpr (KIND == KIND_PROFILE) {
            // rp = the pair's AAProfile image (ba_params.h): scores from the transposed table aa_pos[residue][position], the
            // per-position gap costs behind it. Right: 8 positions (columns) of this lane's two residues; down: this lane's two
            // positions (rows) for each of the 8 column residues.
            if (run) {
                const uint32_t P = profile_positions(rlen, max_size);
                const short* aa_pos = (const short*)(rp + (uint64_t)P * 32);
                const short* goCp = aa_pos + (uint64_t)P * 32; const short* clCp = goCp + P; const short* goRp = clCp + P;
                if (right) {
                    const int v2 = (int)*(const unsigned short*)(qp + ri + 2 * l);
                    uint4 ga, gb, g1, g2, g3;
                    __builtin_memcpy(&ga, aa_pos + (uint64_t)(v2 & 31) * P + rj, 16); __builtin_memcpy(&gb, aa_pos + (uint64_t)((v2 >> 8) & 31) * P + rj, 16);
                    __builtin_memcpy(&g1, goCp + rj, 16); __builtin_memcpy(&g2, clCp + rj, 16); __builtin_memcpy(&g3, goRp + rj, 16);
                    pq.A[0] = (int)ga.x; pq.A[1] = (int)ga.y; pq.A[2] = (int)ga.z; pq.A[3] = (int)ga.w;
                    pq.B[0] = (int)gb.x; pq.B[1] = (int)gb.y; pq.B[2] = (int)gb.z; pq.B[3] = (int)gb.w;
                    pq.pgoC[0] = adds((int)g1.x, fc.ge2); pq.pgoC[1] = adds((int)g1.y, fc.ge2); pq.pgoC[2] = adds((int)g1.z, fc.ge2); pq.pgoC[3] = adds((int)g1.w, fc.ge2);
                    pq.pclC[0] = (int)g2.x; pq.pclC[1] = (int)g2.y; pq.pclC[2] = (int)g2.z; pq.pclC[3] = (int)g2.w;
                    pq.pgoR[0] = (int)g3.x; pq.pgoR[1] = (int)g3.y; pq.pgoR[2] = (int)g3.z; pq.pgoR[3] = (int)g3.w;
                    pq.selE = 0x05040100; pq.selO = 0x07060302; pq.gselE = 0x05040504; pq.gselO = 0x07060706;
                } else {
                    const uint2 cb = *(const uint2*)(qp + rj);
                    int gs[STEP];
#pragma unroll
                    for (int k = 0; k < STEP; k++) {
                        const uint32_t cbyte = ((k < 4 ? cb.x : cb.y) >> (8 * (k & 3))) & 31u;
                        gs[k] = *(const int*)(aa_pos + (uint64_t)cbyte * P + ri + 2 * l);
                    }
                    const int w1 = *(const int*)(goRp + ri + 2 * l), w2 = *(const int*)(goCp + ri + 2 * l), w3 = *(const int*)(clCp + ri + 2 * l);
#pragma unroll
                    for (int m = 0; m < 4; m++) { pq.A[m] = gs[2 * m]; pq.B[m] = gs[2 * m + 1]; }
                    pq.vgoC = adds(w1, fc.ge2); pq.vgoR = w2; pq.vclR = w3;
                    pq.selE = 0x03020100; pq.selO = 0x07060504; pq.gselE = pq.gselO = 0x03020100;
                }
            }
        } else {
        const uint8_t* Vp = right ? qp : rp; const uint8_t* Cp = right ? rp : qp;
        vc = right ? pf_qv : pf_rv; cb_lo = right ? pf_rc.x : pf_qc.x; cb_hi = right ? pf_rc.y : pf_qc.y;
        if (__any(run && !pf_ok)) {   // a slot that has just taken a pair
            if (run && !pf_ok) {
                vc = *(const unsigned short*)(Vp + ri + 2 * l);
                const uint2 cb = *(const uint2*)(Cp + rj);   // the step's 8 column bytes (images are 4-byte aligned, positions multiples of 8)
                cb_lo = cb.x; cb_hi = cb.y;
            }
        }
        asm volatile("" : "+v"(vc));   // (consume the old prefetch before the next one is issued: the memory counter is in-order)
        if (run) {                      // for the step after this one, whichever way it goes
            pf_qv = *(const unsigned short*)(qp + si + 2 * l); pf_rv = *(const unsigned short*)(rp + (boot > 1 ? 0u : sj) + 2 * l);   // (first block: sj is not a position yet)
            pf_qc = *(const uint2*)(qp + si + QUAD_B); pf_rc = *(const uint2*)(rp + (uint32_t)(sj + QUAD_B));   // (32-bit sum: sj is "negative" during the first block)
            pf_ok = true;
        }
        }
        int Ad = right ? Dcol : Drow, Ac = right ? Ccol : Rrow, Pd = right ? Drow : Dcol, Pr = right ? Rrow : Ccol;
        QuadOut o;
        int dcol[STEP];
        const int loc_thr = best_max - off_n + ZERO;
        constexpr int NBOOT = QUAD_B / STEP;
        if (TRACE && run && l == 0 && (boot == 0 || boot == NBOOT)) {   // add_block(i, j, width, height, right) in matrix orientation (scan_block.rs:154,204,284)
            BlockRec br;   // (the first block: one record for the 32 x 32 rectangle, its four sub-steps' trace words are contiguous)
            br.i = right ? ri : rj; br.j = right ? rj : ri; br.h = (uint16_t)(right ? QUAD_B : STEP); br.w = (uint16_t)(right ? (boot ? QUAD_B : STEP) : QUAD_B);
            br.trace_base = trace_top | (right ? 0x80000000u : 0u);
            uint32_t* w = (uint32_t*)(bl + nblocks);
            wt(w, br.i); wt(w + 1, br.j); wt(w + 2, (uint32_t)br.h | ((uint32_t)br.w << 16)); wt(w + 3, br.trace_base);
        }
        quad_rect<KIND, TRACE, XDROP>(smem, fc, l, Ad, Ac, Pd, Pr, Pl, sink, vc & 0xff, (vc >> 8) & 0xff, cb_lo, cb_hi, corner, off_add,
                                      run ? (boot ? -1 : loc_thr) : 0x7fffffff, tr + trace_top, run, run && boot == NBOOT && l == 0, o, dcol, &pq);
        if (boot) {   // the first block's maximum so far: largest value, then smallest row % 16, largest column, largest row (later sub-steps hold the larger columns)
            const bool take = boot == NBOOT || o.mx > bmx || (o.mx == bmx && (o.row & 15) <= (brow & 15));
            bmx = take ? o.mx : bmx; brow = take ? o.row : brow; bcol = take ? (NBOOT - boot) * STEP + o.col : bcol;
            o.mx = bmx; o.row = brow;
        }
        const bool bsub = run && boot > 1, blast = run && boot == 1;

        // ---- what does the step call for? (scan_block.rs:332-558; nothing is committed yet)
        const int right_max = right ? o.act_max8 : o.pas_max8, down_max = right ? o.pas_max8 : o.act_max8;
        const int new_off_max = off_n + o.mx - ZERO;
        const bool improve = new_off_max > best_max;
        const uint32_t new_y = improve ? 0u : y_drop + 1;
        bool stop = q_out && r_out;                                                                   // end of the matrix
        if (XDROP) stop = stop || (!improve && new_off_max < best_max - x_drop && x_iter >= 1);      // X-drop termination
        stop = stop || (!q_out && !r_out && 2 * QUAD_B <= max_size && new_y > QUAD_B / STEP - 1);    // grow
        // the first block: no decisions between its sub-steps; after the last one anything but a plain shift step (an empty
        // improvement -- the driver would grow at once --, the end of the matrix) sends the pair to the per-pair kernel, from scratch
        const bool fresh = blast && (stop || !improve);
        const bool commit = run && !stop && !bsub && !fresh;
        // The last step of a global alignment: its columns stop at the end of the column sequence (scan_block.rs:1216-1224; the
        // trace index still advances over the whole rectangle), and the score is the vector border's entry at the end of the
        // vector sequence after the last computed column (scan_block.rs:560-570). The pair is complete: results are written here.
        const bool fin = !XDROP && run && boot == 0 && q_out && r_out;
        if (!XDROP && __any(fin)) {
            const uint32_t ncols = lenC - rj + 1;   // 1 .. 8 (columns rj .. lenC)
            int dsel = dcol[0];
#pragma unroll
            for (int k = 1; k < STEP; k++) dsel = ncols == (uint32_t)(k + 1) ? dcol[k] : dsel;
            const uint32_t idx = lenV - ri;         // 0 .. 31: both ends are inside the block
            const int v = __builtin_amdgcn_ds_bpermute((int)(((uint32_t)lane & 48u) + (idx >> 1)) << 2, dsel);
            const int sc16 = (idx & 1) ? v >> 16 : (int)(short)v;
            if (fin && l == 0) {
                bp.score[pair] = off_n + sc16 - ZERO; bp.query_idx[pair] = qlen; bp.reference_idx[pair] = rlen;
                if (bp.cells) bp.cells[pair] = cells0 + (unsigned long long)nsteps * (STEP * QUAD_B) + (unsigned long long)ncols * QUAD_B;
                if (bp.status) bp.status[pair] = 0;
                if (bp.nblocks_out) bp.nblocks_out[pair] = TRACE ? nblocks + 1 : 0;
                if (bp.trace_words_out) bp.trace_words_out[pair] = TRACE ? trace_top + STEP * QUAD_B / 8 : 0;
                if (bp.slot_out) bp.slot_out[pair] = pair;
                if (TRACE) bp.slot_info[pair] = SlotInfo{pair, nblocks + 1, qlen, rlen};
            }
            if (fin) pair = ~0u;
        }
        leave = leave || (run && stop && !bsub && !blast && !fin);
        if (fresh) { if (l == 0) { bp.cont_out_flag[pair] = 2u; enqueue(2u * pair + 2u); } pair = ~0u; boot = 0; }
        if (bsub) {
            Dcol = Ad; Ccol = Ac; Drow = Pd; Rrow = Pr;
            sj += STEP; nsteps++; boot--;
            if (TRACE) trace_top += STEP * QUAD_B / 8;
        }

        // ---- slots that leave: their state as it was at the top of this step
        if (__any(leave)) {
            if (leave) {
                // (agent-scope stores, see wt(): the per-pair kernel that takes the record may already be running, on another XCD)
                PairCont* c = bp.cont_out + pair;
                uint32_t* w = (uint32_t*)c;   // the 24 header words: lane k of the slot writes words k and 16 + k
                const unsigned long long cells = cells0 + (unsigned long long)nsteps * (STEP * QUAD_B);
                uint32_t h0 = 0, h1 = 0;
                switch (l) {
                    case 0: h0 = pair; h1 = (uint32_t)cells; break;            case 1: h0 = si; h1 = (uint32_t)(cells >> 32); break;
                    case 2: h0 = sj; h1 = budget; break;                        case 3: h0 = (uint32_t)dir; h1 = trace_top; break;
                    case 4: h0 = (uint32_t)prev_dir; h1 = nblocks; break;       case 5: h0 = (uint32_t)off; h1 = ck_tt; break;
                    case 6: h0 = (uint32_t)off_max; h1 = ck_nb; break;          case 7: h0 = (uint32_t)best_max; h1 = 0; break;
                    case 8: h0 = y_drop; break;       case 9: h0 = (uint32_t)x_iter; break;   case 10: h0 = (uint32_t)D_corner; break;
                    case 11: h0 = best_i; break;      case 12: h0 = best_j; break;            case 13: h0 = ck_i; break;
                    case 14: h0 = ck_j; break;        default: h0 = (uint32_t)ck_off; break;
                }
                wt(w + l, h0);
                if (l < 8) wt(w + 16 + l, h1);
                if (l == 0) bp.cont_out_flag[pair] = 1u;
                wt(&c->borders[0][l], (uint32_t)Dcol); wt(&c->borders[1][l], (uint32_t)Ccol); wt(&c->borders[2][l], (uint32_t)Drow); wt(&c->borders[3][l], (uint32_t)Rrow);
                wt(&c->ckpt[0][l], (uint32_t)ck0); wt(&c->ckpt[1][l], (uint32_t)ck1); wt(&c->ckpt[2][l], (uint32_t)ck2); wt(&c->ckpt[3][l], (uint32_t)ck3);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the records (and the pairs' trace words) have arrived before their queue entries go out
            if (leave) { if (l == 0) enqueue(2u * pair + 1u); pair = ~0u; }
        }
        // ---- slots that go on: commit the step
        if (commit) {
            Dcol = right ? Ad : Pd; Ccol = right ? Ac : Pr; Drow = right ? Pd : Ad; Rrow = right ? Pr : Ac;
            off = off_n; off_max = new_off_max; y_drop = new_y; prev_dir = blast ? DIR_GROW : dir; D_corner = blast ? 0 : o.corner_new;
            nsteps++; budget--;
            if (TRACE) { trace_top += STEP * QUAD_B / 8; nblocks++; }
            if (improve) {
                if (XDROP) {   // scan_block.rs:370-404
                    best_i = right ? si + (uint32_t)o.row : si + (QUAD_B - STEP) + (uint32_t)o.col;
                    best_j = right ? sj + (QUAD_B - STEP) + (uint32_t)o.col : sj + (uint32_t)o.row;
                    if (blast) best_j = (uint32_t)bcol;
                }
                if (QUAD_B < max_size) {
                    ck_i = si; ck_j = sj; ck_off = off; ck0 = Dcol; ck1 = Ccol; ck2 = Drow; ck3 = Rrow;
                    if (TRACE) { ck_tt = trace_top; ck_nb = nblocks; }
                }
                best_max = off_max;
            }
            if (XDROP) x_iter = (off_max < best_max - x_drop) ? x_iter + 1 : 0;
            const bool go_down = r_out || (!q_out && down_max > right_max);
            si += go_down ? (uint32_t)STEP : 0u; sj += go_down ? 0u : (uint32_t)STEP;
            dir = go_down ? DIR_DOWN : DIR_RIGHT;
            boot = 0;
        }
    }
    // this producer is done: every entry of the wave is in the queue before the count says so
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(bp.cq_ctrl + 32, 1u, BA_RLX_AGENT);
}

}  // namespace ba
