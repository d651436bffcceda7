// block_aligner_amd — per-pair driver (shift right/down, grow, shrink, X-drop), traceback, and kernels.
// Follows /root/reference/src/scan_block.rs:94-595 (align_core), 1003-1061 (border moves),
// 1428-1462 (trace stack), 1482-1672 (cigar_core). One wavefront per pair; all driver state is wave-uniform.
#pragma once
#include "ba_device.hpp"

namespace ba {

// ------------------------------------------------------------------ LDS border helpers (whole wave cooperates)
// Lanes exchange border data through LDS. The hardware executes one wave's LDS operations in program order; this
// keeps the compiler from reordering them across a hand-off point (no instruction is emitted beyond waits).
#ifndef BA_BIG
#define BA_BIG 0   // 1: the kernels of this TU (class BA_PMAX = 32) handle blocks of 4096 .. 32768 cells (borders in global memory, tiled fill)
#endif
constexpr bool kBig = BA_BIG != 0;
__device__ __forceinline__ void lds_sync() {
    // (big-block kernels keep the borders in global memory: a workgroup-scope fence drains this wave's stores and drops its
    // CU's stale L1 lines before another lane reads them back)
    if (kBig) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void lds_fill0(short* a, uint32_t n) {
    for (uint32_t k = 2 * lane_id(); k < n; k += 128) *(int*)(a + k) = 0;
    lds_sync();
}
// dst and src may be the two halves of one array (shrink); each 128-entry chunk is read before any of it is written
__device__ __forceinline__ void lds_copy(short* dst, const short* src, uint32_t n) {
    lds_sync();
    for (uint32_t k = 2 * lane_id(); k < n; k += 128) {
        const int v = *(const int*)(src + k);
        lds_sync();
        *(int*)(dst + k) = v;
    }
    lds_sync();
}
// max of the first 8 entries (scan_block.rs:1020-1022)
__device__ __forceinline__ int lds_prefix_max8(const short* a) {
    lds_sync();
    const int* p = (const int*)a;
    const s16x2 m = as_s(vmax(vmax(p[0], p[1]), vmax(p[2], p[3])));
    return uni(max((int)m.x, (int)m.y));
}
// max of the last 2 entries (scan_block.rs:1030-1032, SHRINK_SUFFIX_LEN = 2)
__device__ __forceinline__ int lds_suffix_max2(const short* a, uint32_t n) {
    lds_sync();
    s16x2 m = as_s(*(const int*)(a + n - 2));
    return uni(max((int)m.x, (int)m.y));
}
// buf[k] = buf[k+8] (+) off_add for k < n-8, buf[n-8..n) = temp[0..8); returns buf_old[7] (+) off_add
// (scan_block.rs:1040-1061)
__device__ __forceinline__ int lds_shift_and_offset(uint32_t n, short* b1, short* b2, const short* t1, const short* t2, int off_add) {
    const int oa = splat(off_add);
    lds_sync();
    const int corner = uni((int)as_s(adds(splat((int)b1[STEP - 1]), oa)).x);
    for (uint32_t base = 0; base < n; base += 128) {
        const uint32_t k = base + 2 * lane_id();
        int v1 = 0, v2 = 0;
        const bool in = k < n;
        if (in) {
            if (k + STEP < n) {
                v1 = adds(*(const int*)(b1 + k + STEP), oa);
                v2 = adds(*(const int*)(b2 + k + STEP), oa);
            } else {
                v1 = *(const int*)(t1 + (k + STEP - n));
                v2 = *(const int*)(t2 + (k + STEP - n));
            }
        }
        lds_sync();
        if (in) { *(int*)(b1 + k) = v1; *(int*)(b2 + k) = v2; }
        lds_sync();
    }
    return corner;
}

// ------------------------------------------------------------------ traceback (one lane walks the path)
struct Move { uint32_t op, di, dj, next; };
// OP_LUT of scan_block.rs:1532-1558 as branches; table: 0 = D, 1 = C, 2 = R, 3 = pending; t as the reference defines it,
// t2 bit 0 = "C opened here". The reference's trace2 bit 1 -- "the scan-direction gap that reaches this cell was opened
// in the cell above" -- is not stored per cell: a scan-direction gap move leaves the state PENDING (3) and the walker
// resolves it at the destination from that cell's own "opened here" bit (tb_resolve), which is the same value.
constexpr uint32_t TB_PENDING = 3;
__device__ __forceinline__ Move tb_lut(bool right_blk, uint32_t t, uint32_t t2, uint32_t table) {
    constexpr uint32_t OP_M = 1, OP_I = 4, OP_D = 5;
    const uint32_t gapA_op = right_blk ? OP_D : OP_I, gapB_op = right_blk ? OP_I : OP_D;
    const uint32_t A_di = right_blk ? 0 : 1, A_dj = right_blk ? 1 : 0;
    const uint32_t A_tab = right_blk ? 1 : 2, B_tab = right_blk ? 2 : 1;
    // "A" = the gap kind tracked by trace bit 0 / trace2 bit 0 (C for right blocks, R for down blocks); "B" = the
    // scan-direction gap (along the vectors)
    if (table == A_tab) return (t2 & 1) ? Move{gapA_op, A_di, A_dj, 0} : Move{gapA_op, A_di, A_dj, A_tab};
    if (table == B_tab) return Move{gapB_op, 1 - A_di, 1 - A_dj, TB_PENDING};
    if (t == 0) return Move{OP_M, 1, 1, 0};
    if (t & 1) return (t2 & 1) ? Move{gapA_op, A_di, A_dj, 0} : Move{gapA_op, A_di, A_dj, A_tab};
    return Move{gapB_op, 1 - A_di, 1 - A_dj, TB_PENDING};
}
// state on arrival at a cell: a pending scan-direction gap continues unless it was opened in this very cell
__device__ __forceinline__ uint32_t tb_resolve(bool right_blk, uint32_t table, uint32_t nib) {
    return table == TB_PENDING ? ((nib & 8u) ? 0u : (right_blk ? 2u : 1u)) : table;
}
// state after a move: a scan-direction gap leaving through the top row of a rectangle stays a gap (the reference's
// flag for the first vector of a column is 0, scan_block.rs:1103,1180)
__device__ __forceinline__ uint32_t tb_next(bool right_blk, uint32_t next, uint32_t v) {
    return (next == TB_PENDING && v == 0) ? (right_blk ? 2u : 1u) : next;
}

// Walk back from (i, j); emit run-length ops right-aligned into [out_lo, out_hi). Returns run count, or
// sets *status on failure. Executed by lane 0 only. (scan_block.rs:1482-1672)
__device__ __forceinline__ uint32_t traceback(const BlockRec* __restrict__ blocks, uint32_t nblocks, const uint32_t* __restrict__ trace,
                                     uint32_t i, uint32_t j, const uint8_t* __restrict__ q, const uint8_t* __restrict__ r, uint32_t flags,
                                     uint32_t* __restrict__ out, uint64_t out_lo, uint64_t out_hi, uint32_t* status) {
    const bool eq = flags & F_CIGAR_EQ, local = flags & F_LOCAL, fqs = flags & F_FQS;
    bool stop = false;               // LOCAL_START / FREE_QUERY_START_GAPS end the walk before (0, 0)
    uint64_t wp = out_hi;          // next free slot is wp - 1
    uint32_t run_op = 0, run_len = 0;
    uint32_t table = 0;
    uint32_t bidx = nblocks;
    while ((i > 0 || j > 0) && !stop) {
        BlockRec br;
        for (;;) {
            if (bidx == 0) { *status |= ST_TRACEBACK_LOST; return 0; }
            bidx--;
            br = blocks[bidx];
            if (i >= (br.i & 0x7fffffffu) && j >= br.j) break;
        }
        const bool right_blk = br.trace_base >> 31;
        const bool l2 = br.i >> 31;   // a slot's rectangle (ba_multi.hpp): words of 4 cells x 2 columns (BR_L2)
        br.i &= 0x7fffffffu;
        if (br.trace_base & 0x40000000u) { *status |= ST_TRACEBACK_LOST; return 0; }   // a speculative grow that was never materialised: must not be on a path
        const uint32_t tbase = br.trace_base & 0x3fffffffu;
        const uint32_t Hv = right_blk ? br.h : br.w;          // cells along the vector axis
        const uint32_t nch = Hv > 128 ? Hv / 128 : 1, nl = Hv > 128 ? 64 : Hv / 2;   // chunks of 128 cells, lanes per chunk
        while (i >= br.i && j >= br.j && (i > 0 || j > 0)) {
            const uint32_t ci = i - br.i, cj = j - br.j;
            const uint32_t v = right_blk ? ci : cj, w = right_blk ? cj : ci;
            const uint32_t chunk = v >> 7, lane = (v & 127) >> 1;
            if (right_blk && fqs && i == 0) { stop = true; break; }                     // scan_block.rs:1597-1599
            uint32_t nib;
            if (l2) {
                const uint32_t word = trace[tbase + (v >> 3) * 8 + (w >> 1) * 2 + ((v >> 2) & 1)];
                nib = ((word >> ((v & 3) * 8 + (w & 1) * 4)) ^ 15u) & 15u;
            } else {
                // (LOCAL_START batches: a trace word is followed by its cells' zero-mask word, place_rect)
                const uint32_t word = trace[tbase + (((w >> 2) * nch + chunk) * nl + lane) * (local ? 2u : 1u)];
                nib = ((word >> ((v & 1) * 16 + (w & 3) * 4)) ^ 15u) & 15u;   // all four bits are stored as "differs"
            }
            table = tb_resolve(right_blk, table, nib);
            if (local && table == 0) {                                                   // scan_block.rs:1604-1611
                if (l2) {   // a slot's rectangle (32 or 128 cells x 8 columns, ba_small.hpp / ba_multi.hpp): a bit per cell, the columns of a cell in one byte, after the trace words
                    const uint32_t z = trace[tbase + (uint32_t)br.h * br.w / 8 + (v >> 3) * 2 + ((v >> 2) & 1)];
                    if ((z >> ((v & 3) * 8 + w)) & 1) { stop = true; break; }
                } else {
                    const uint32_t z = trace[tbase + (((w >> 2) * nch + chunk) * nl + lane) * 2u + 1u];
                    if ((z >> ((v & 1) * 16 + (w & 3) * 4)) & 1) { stop = true; break; }
                }
            }
            const Move m = tb_lut(right_blk, nib & 3, (nib >> 2) & 1, table);
            uint32_t op = m.op;
            if (eq && op == 1) op = q[i] == r[j] ? 2 : 3;
            if (m.di > i || m.dj > j) { *status |= ST_TRACEBACK_LOST; return 0; }   // would leave the matrix: corrupt trace
            i -= m.di; j -= m.dj; table = tb_next(right_blk, m.next, v);
            if (op == run_op) run_len++;
            else {
                if (run_len) {
                    if (wp == out_lo) { *status |= ST_CIGAR_OVERFLOW; return 0; }
                    out[--wp] = (run_len << 4) | run_op;
                }
                run_op = op; run_len = 1;
            }
        }
    }
    if (run_len) {
        if (wp == out_lo) { *status |= ST_CIGAR_OVERFLOW; return 0; }
        out[--wp] = (run_len << 4) | run_op;
    }
    return (uint32_t)(out_hi - wp);
}

// ------------------------------------------------------------------ one path walked by a whole wave
// The lane walks above advance one path at ~0.55 us per cell whichever way they are used (a chain of ~200 dependent instructions and one
// memory round trip per rectangle), so a batch ends one longest-path walk after its last fill: 10 ms for the 8881-residue pair of the
// protein set. Here the wave's 64 lanes work on ONE path: they fetch the next 64 rectangle records (one per lane, kept in registers)
// and copy those rectangles' trace words into the wave's LDS region with coalesced loads, as many as fit; the walk itself -- same
// rules as traceback() above, scan_block.rs:1576-1670 -- then runs on wave-uniform values (scalar registers, scalar branches): a
// rectangle's record by v_readlane, the cell's trace word by one broadcast LDS read, the move table (OP_LUT, scan_block.rs:1532-1558)
// and 256-byte windows of both sequences in registers read by v_readlane, finished runs collected in a register (v_writelane) and
// stored 64 at a time. A rectangle larger than the LDS region is walked out of global memory. CIGAR_EQ is the only mode bit taken:
// the special modes keep traceback().
__device__ __forceinline__ uint32_t wave_excl_sum(uint32_t x, uint32_t lane) {
    uint32_t s = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)s, d, 64);
        s += lane >= (uint32_t)d ? o : 0u;
    }
    return s - x;
}
template <bool L2OK>
__device__ __forceinline__ uint32_t walk_wave(const BlockRec* __restrict__ blocks, uint32_t nblocks_in, const uint32_t* __restrict__ trace,
                                              uint32_t i_in, uint32_t j_in, const uint8_t* __restrict__ q, const uint8_t* __restrict__ r, bool eq,
                                              uint32_t* __restrict__ out, uint64_t out_lo, uint64_t out_hi, uint32_t* status,
                                              uint32_t* __restrict__ lds, uint32_t budget) {
    const uint32_t lane = (uint32_t)lane_id();
    uint32_t lutv0, lutv1;
    {
        const Move a = tb_lut(false, lane & 3, (lane >> 2) & 1, (lane >> 4) & 3), b = tb_lut(true, lane & 3, (lane >> 2) & 1, (lane >> 4) & 3);
        lutv0 = a.op | (a.di << 3) | (a.dj << 4) | (a.next << 5);
        lutv1 = b.op | (b.di << 3) | (b.dj << 4) | (b.next << 5);
    }
    uint32_t i = (uint32_t)uni((int)i_in), j = (uint32_t)uni((int)j_in), bidx = (uint32_t)uni((int)nblocks_in);
    uint32_t table = 0, run_op = 0, run_len = 0, nrun = 0, runbuf = 0, st = 0;
    uint64_t wp = out_hi;
    uint32_t qw0 = 0, rw0 = 0, qwin = 0, rwin = 0;
    auto load_q = [&]() { qw0 = i >= 252u ? (i - 252u) & ~3u : 0u; const uint32_t a = qw0 + 4u * lane; qwin = a <= i ? *(const uint32_t*)(q + a) : 0u; };
    auto load_r = [&]() { rw0 = j >= 252u ? (j - 252u) & ~3u : 0u; const uint32_t a = rw0 + 4u * lane; rwin = a <= j ? *(const uint32_t*)(r + a) : 0u; };
    if (eq) { load_q(); load_r(); }
    auto flush = [&]() {
        if (lane < nrun) out[wp - 1u - lane] = runbuf;
        wp -= nrun; nrun = 0;
    };
    auto emit = [&]() -> bool {   // false: the runs do not fit
        if (!run_len) return true;
        if (wp - nrun == out_lo) return false;
        asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(runbuf) : "s"(__builtin_amdgcn_readfirstlane((int)((run_len << 4) | run_op))), "s"(__builtin_amdgcn_readfirstlane((int)nrun)) : "m0");
        if (++nrun == 64u) flush();
        return true;
    };
    bool ok = true;
    while (ok && (i > 0 || j > 0)) {
        if (bidx == 0) { st = ST_TRACEBACK_LOST; ok = false; break; }
        // ---- stage: the next records, one per lane (lane k: record bidx - 1 - k), and their trace words
        const uint32_t nvalid = min(bidx, 64u);
        uint4 rec = uint4{0, 0, 0, 0};
        if (lane < nvalid) rec = *(const uint4*)(blocks + (bidx - 1u - lane));
        const bool untr = rec.w & 0x40000000u;
        const uint32_t words = untr ? 0u : (rec.z & 0xffffu) * (rec.z >> 16) / 8u;
        const uint32_t tb = rec.w & 0x3fffffffu;
        uint32_t toff = wave_excl_sum(lane < nvalid ? words : 0u, lane);
        const uint32_t nfit = (uint32_t)__popcll(__ballot(lane < nvalid && toff + words <= budget));
        // (a stack's rectangles are stacked: record k's words end where record k - 1's begin -- then one flat copy)
        const uint32_t tb_up = (uint32_t)__shfl_up((int)tb, 1, 64);
        const bool flat = nfit > 1 && !__any(lane > 0 && lane < nfit && tb + words != tb_up);
        if (nfit > 0) {
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)(toff + words), (int)(nfit - 1u));
            if (flat) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)tb, (int)(nfit - 1u));
                for (uint32_t x = lane; x < total; x += 64u) lds[x] = trace[lo + x];
                toff = tb - lo;
            } else {
                for (uint32_t k = 0; k < nfit; k++) {
                    const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)tb, (int)k), sz = (uint32_t)__builtin_amdgcn_readlane((int)words, (int)k);
                    const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)toff, (int)k);
                    for (uint32_t x = lane; x < sz; x += 64u) lds[off + x] = trace[base + x];
                }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        const bool direct = nfit == 0;                 // the top rectangle alone exceeds the region: walk it out of global memory
        const uint32_t ntake = direct ? 1u : nfit;
        auto walk_rects = [&](auto direct_c) {
        constexpr bool DIRECT = decltype(direct_c)::value;
        uint32_t cur = 0;
        while (cur < ntake && ok && (i > 0 || j > 0)) {
            const uint32_t rx = (uint32_t)__builtin_amdgcn_readlane((int)rec.x, (int)cur), bj = (uint32_t)__builtin_amdgcn_readlane((int)rec.y, (int)cur);
            const uint32_t bi = rx & 0x7fffffffu;
            if (!(i >= bi && j >= bj)) { cur++; continue; }
            const uint32_t rz = (uint32_t)__builtin_amdgcn_readlane((int)rec.z, (int)cur), rw = (uint32_t)__builtin_amdgcn_readlane((int)rec.w, (int)cur);
            if (rw & 0x40000000u) { st = ST_TRACEBACK_LOST; ok = false; break; }   // a speculative grow that was never materialised: must not be on a path
            const uint32_t off = DIRECT ? (rw & 0x3fffffffu) : (uint32_t)__builtin_amdgcn_readlane((int)toff, (int)cur);
            const bool right = rw >> 31, l2 = L2OK && (rx >> 31);
            const uint32_t Hv = right ? (rz & 0xffffu) : (rz >> 16);
            const uint32_t nch = Hv > 128u ? Hv / 128u : 1u, nl = Hv > 128u ? 64u : Hv / 2u;
            const uint32_t lutr = right ? lutv1 : lutv0;
            while (i >= bi && j >= bj && (i > 0 || j > 0)) {
                const uint32_t ci = i - bi, cj = j - bj;
                const uint32_t v = right ? ci : cj, w = right ? cj : ci;
                uint32_t widx, sh;
                if (l2) { widx = (v >> 3) * 8u + (w >> 1) * 2u + ((v >> 2) & 1u); sh = (v & 3u) * 8u + (w & 1u) * 4u; }
                else { widx = ((w >> 2) * nch + (v >> 7)) * nl + ((v & 127u) >> 1); sh = (v & 1u) * 16u + (w & 3u) * 4u; }
                uint32_t word;
                // ---- round 5: a run of diagonal moves at once. In state D a cell whose C and R flags both say "differs" is a match / mismatch and
                // leaves the state D (OP_LUT, scan_block.rs:1532-1558): lane k looks at cell (v - k, w - k) of this rectangle, the number of leading
                // lanes that agree is the length of the run; with CIGAR_EQ the lanes also compare their cells' bytes and the run is cut into = / X
                // runs from the two masks. (A cell on wave-uniform values costs ~700 cycles -- ~75 scalar instructions at one issue per ~5 cycles and
                // their branches --, the pass about as much as one cell.)
                if (table == 0u && i > 0u && j > 0u && min(v, w) >= 1u) {
                    const uint32_t lim = min(min(v, w), min(i, j) - 1u);   // cells 0 .. lim: inside the rectangle, and a diagonal move from them stays inside the matrix
                    const bool valid = lane <= lim;
                    const uint32_t vv = v - lane, ww = w - lane;
                    uint32_t widx_k, sh_k;
                    if (l2) { widx_k = (vv >> 3) * 8u + (ww >> 1) * 2u + ((vv >> 2) & 1u); sh_k = (vv & 3u) * 8u + (ww & 1u) * 4u; }
                    else { widx_k = ((ww >> 2) * nch + (vv >> 7)) * nl + ((vv & 127u) >> 1); sh_k = (vv & 1u) * 16u + (ww & 3u) * 4u; }
                    uint32_t word_k = 0;
                    if (valid) { if constexpr (DIRECT) word_k = trace[off + widx_k]; else word_k = lds[off + widx_k]; }
                    const unsigned long long cm = __ballot(valid && ((word_k >> sh_k) & 3u) == 3u);
                    uint32_t n = cm == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~cm);
                    if (eq && n >= 2u) {   // the cells' bytes out of the sequence windows (256 bytes each, four per lane)
                        if (i < qw0) load_q();
                        if (j < rw0) load_r();
                        n = min(n, min(i - qw0, j - rw0) + 1u);
                    }
                    if (n >= 2u) {
                        bool fits = true;
                        if (!eq) {
                            if (run_op == 1u) run_len += n;
                            else { fits = emit(); run_op = 1u; run_len = n; }
                        } else {
                            const uint32_t qa = i - lane - qw0, ra = j - lane - rw0;   // (lanes < n: inside the windows)
                            const uint32_t qd = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(qa & ~3u), (int)qwin), rd = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(ra & ~3u), (int)rwin);
                            const bool same = ((qd >> ((qa & 3u) * 8u)) & 255u) == ((rd >> ((ra & 3u) * 8u)) & 255u);
                            unsigned long long bits = __ballot(lane < n && same);
                            uint32_t rem = n;
                            while (rem && fits) {
                                const bool e = bits & 1ull;
                                const unsigned long long x = e ? ~bits : bits;
                                const uint32_t len = min(x ? (uint32_t)__builtin_ctzll(x) : 64u, rem), op2 = e ? 2u : 3u;
                                if (op2 == run_op) run_len += len;
                                else { fits = emit(); run_op = op2; run_len = len; }
                                bits = len < 64u ? bits >> len : 0ull; rem -= len;
                            }
                        }
                        if (!fits) { st = ST_CIGAR_OVERFLOW; ok = false; break; }
                        i -= n; j -= n;   // (the state stays D)
                        continue;
                    }
                    word = (uint32_t)__builtin_amdgcn_readlane((int)word_k, 0);   // (lane 0 looked at the current cell)
                } else {
                if constexpr (DIRECT) word = (uint32_t)uni((int)trace[off + widx]); else word = (uint32_t)uni((int)lds[off + widx]);
                }
                const uint32_t nib = ((word >> sh) ^ 15u) & 15u;                                // all four bits are stored as "differs"
                table = tb_resolve(right, table, nib);
                const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)lutr, (int)((table << 4) | nib));
                uint32_t op = m & 7u;
                const uint32_t di = (m >> 3) & 1u, dj = (m >> 4) & 1u;
                if (eq && op == 1u) {
                    if (i < qw0) load_q();
                    if (j < rw0) load_r();
                    const uint32_t qd = (uint32_t)__builtin_amdgcn_readlane((int)qwin, (int)((i - qw0) >> 2)), rd = (uint32_t)__builtin_amdgcn_readlane((int)rwin, (int)((j - rw0) >> 2));
                    op = ((qd >> ((i & 3u) * 8u)) & 255u) == ((rd >> ((j & 3u) * 8u)) & 255u) ? 2u : 3u;
                }
                if (di > i || dj > j) { st = ST_TRACEBACK_LOST; ok = false; break; }            // would leave the matrix: corrupt trace
                i -= di; j -= dj; table = tb_next(right, m >> 5, v);
                if (op == run_op) run_len++;
                else {
                    if (!emit()) { st = ST_CIGAR_OVERFLOW; ok = false; break; }
                    run_op = op; run_len = 1;
                }
            }
        }
        };
        if (direct) walk_rects(std::true_type{}); else walk_rects(std::false_type{});
        bidx -= ntake;
    }
    if (ok && !emit()) { st = ST_CIGAR_OVERFLOW; ok = false; }
    if (!ok) { *status |= st; return 0; }
    flush();
    return (uint32_t)(out_hi - wp);
}

// ------------------------------------------------------------------ traceback lanes (batch path)
// In a TRACE batch the fill waves do not walk their own tracebacks: a walk is ~(|q|+|r|) dependent steps and would
// idle 63 lanes for as long as the fill itself takes. Finished trace stacks are handed to traceback waves of the same
// persistent launch (wave 0 of every tb_stride-th workgroup) through a global ring (agent-scope release / acquire,
// MI355X guide G16); there every LANE walks one alignment out of per-lane LDS windows, so 64 walks overlap per wave.
#define BA_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct TbLane {
    uint32_t slot, pair, i, j, table, bidx, run_op, run_len, status;
    uint64_t wp, lo;
    const BlockRec* blocks; const uint32_t* trace; const uint8_t* q; const uint8_t* r;
    uint32_t bi, bj, tbase, zoff, nch, nl; bool right, in_rect, l2;
    // look-ahead state: the next rectangle records; what the LDS record of this lane currently holds: a 5-lane x
    // 2-column-group window of trace words (chunk tw_chunk, column groups tw_g and tw_g - 1, lanes from tw_lane0) and
    // 16-byte windows of both sequences ([qw0, qw0 + 16), [rw0, rw0 + 16); 0xffffffff = empty)
    uint4 nrec[4]; uint32_t nq;   // the next nq records of the stack (below bidx), fetched ahead
    uint32_t tw_chunk, tw_g, tw_lane0; bool tw_ok;
    uint32_t qw0, rw0;
    // tb_step_fast (round 6): finished runs wait here for the next call's loads to be issued (bit s of pmask: site s of the call produced
    // one; pv = the run, po = its place in dwords from the pair's first)
    uint32_t pv[6], po[6], pmask;
};

__device__ __forceinline__ void tb_emit(TbLane& t, uint32_t* __restrict__ out) {
    if (t.run_len) {
        if (t.wp == t.lo) { t.status |= ST_CIGAR_OVERFLOW; t.i = t.j = 0; return; }
        out[--t.wp] = (t.run_len << 4) | t.run_op;
    }
}
__device__ __forceinline__ void tb_fail(TbLane& t) { t.status |= ST_TRACEBACK_LOST; t.i = t.j = 0; t.run_len = 0; }

// One iteration of scan_block.rs:1576-1670 for one lane: move to the rectangle that holds the current cell if needed,
// bring the trace words around the cell (and the sequence bytes below it) into this lane's LDS record -- at most one
// round of global-memory latency per call -- then walk up to TB_CELLS_PER_STEP cells out of LDS: one byte read for the cell's trace
// bits, one for the move (OP_LUT of scan_block.rs:1532-1558 as a 128-entry table), two for the =/X comparison.
// The lanes of a traceback wave walk in lockstep, so a call costs what its longest-walking lane costs. A path crosses a
// rectangle in at most 8 diagonal cells (4 on average), and the wave's instruction stream -- not memory latency -- is
// what a walk costs the fill waves it shares a SIMD with: measured at config 3, 12 cells per call 1142 GCUPS, 8: 1145,
// 6: 1154, 4: 1156, 3: 1154 (more calls per walk, each much shorter).
constexpr int TB_CELLS_PER_STEP = 4;
#ifndef BA_WALK_CELLS
#define BA_WALK_CELLS 8
#endif
#ifndef BA_RING_DEPTH
#define BA_RING_DEPTH 2   // (config 3, same box: 1503 -> 1507 GCUPS)
#endif
#ifndef BA_WALK_DEPTH
#define BA_WALK_DEPTH 4
#endif
#ifndef BA_HELPER_CELLS
#define BA_HELPER_CELLS 4   // cells per call of an emptied fill wave's walking lanes (TB_CELLS_PER_STEP: the same code as the dedicated waves)
#endif
// DEPTH: rectangle records fetched ahead. A path skips rectangles (a right strip's left neighbour on the path is an earlier right
// strip, the down strips in between are not on it): with DEPTH > 1 a call walks down the stack past them instead of spending one
// call -- one memory round trip of its wave -- on each.
// LB: bytes of a lane's LDS record. TB_LANE_BYTES: 10 trace words + the two sequence windows. TB_LANE_BYTES_L2 (k_multi): 16 trace words, so
// that a window also holds the words of a slot's rectangles (4 cells x 2 columns each: two 8-cell lanes x two column groups = 64 bytes).
template <int CELLS = TB_CELLS_PER_STEP, int DEPTH = 1, int LB = (int)TB_LANE_BYTES>
__device__ __forceinline__ void tb_step(TbLane& t, uint32_t flags, uint32_t* __restrict__ out, unsigned char* lrec,
                                        const unsigned char* lut, unsigned long long* tacc = nullptr) {
    constexpr bool LOCREC = LB == (int)TB_LANE_BYTES_LOC;   // records with room for the zero-mask bits: the only ones LOCAL_START walks use
    constexpr bool L2OK = LB == (int)TB_LANE_BYTES_L2 || LOCREC;
    constexpr int SEQ_Q = LOCREC ? 80 : (L2OK ? 64 : 40), SEQ_R = SEQ_Q + 16;   // byte offsets of the sequence windows in the record
    const bool eq = flags & F_CIGAR_EQ, local = LOCREC && (flags & F_LOCAL), fqs = flags & F_FQS;
    BA_TSTAMP(ts0);
    if (!t.in_rect || !(t.i >= t.bi && t.j >= t.bj)) {
        bool found = false, untraced = false;
#pragma unroll
        for (int a = 0; a < DEPTH; a++) {
            if (!found && (a == 0 || t.nq)) {   // (beyond the first record only out of the look-ahead queue: one round of memory latency per call)
                if (t.bidx == 0) { tb_fail(t); return; }
                uint4 rec;
                if (t.nq) {
                    rec = t.nrec[0];
#pragma unroll
                    for (int k = 0; k + 1 < DEPTH; k++) t.nrec[k] = t.nrec[k + 1];
                    t.nq--;
                } else rec = *(const uint4*)(t.blocks + t.bidx - 1);
                t.bidx--;
                t.l2 = L2OK && (rec.x >> 31); t.bi = rec.x & 0x7fffffffu; t.bj = rec.y;
                const uint32_t h = rec.z & 0xffffu, w = rec.z >> 16;
                t.in_rect = t.i >= t.bi && t.j >= t.bj;
                t.zoff = h * w / 8;
                t.right = rec.w >> 31;
                t.tbase = rec.w & 0x3fffffffu;
                const uint32_t Hv = t.right ? h : w;
                t.nch = Hv > 128 ? Hv / 128 : 1; t.nl = Hv > 128 ? 64 : Hv / 2;
                untraced = rec.w & 0x40000000u;
                found = t.in_rect;
            }
        }
        if (t.nq == 0 && t.bidx > 0) {   // the next records, in flight while this call waits for its trace words
#pragma unroll
            for (int k = 0; k < DEPTH; k++) if (t.bidx > (uint32_t)k) t.nrec[k] = *(const uint4*)(t.blocks + t.bidx - 1 - k);
            t.nq = min((uint32_t)DEPTH, t.bidx);
        }
        t.tw_ok = false;
        if (!t.in_rect) return;
        if (untraced) { tb_fail(t); return; }   // a speculative grow that was never materialised: must not be on a path
    }
    {
        const uint32_t ci = t.i - t.bi, cj = t.j - t.bj;
        const uint32_t v = t.right ? ci : cj, w = t.right ? cj : ci;
        const uint32_t chunk = v >> 7, lc = (v & 127) >> 1, g = w >> 2;
        if (L2OK && t.l2) {
            // a slot's rectangle (128 x 8 cells, one chunk): a lane's 8 cells x 8 columns are 32 contiguous bytes (word lane8 * 8 +
            // (column >> 1) * 2 + cell quad); the window is the two 8-cell lanes ending at the cell's: 64 contiguous bytes, four
            // 16-byte loads into one cache line (mostly). tw_lane0: the first lane8.
            const uint32_t L8 = v >> 3;
            if (!(t.tw_ok && t.tw_chunk == 0xffffu && L8 - t.tw_lane0 < 2u)) {
                t.tw_chunk = 0xffffu; t.tw_g = 0; t.tw_lane0 = L8 ? L8 - 1 : 0u; t.tw_ok = true;
                const uint32_t* wp = t.trace + t.tbase + t.tw_lane0 * 8;
                uint4 a0, a1, a2, a3;
                __builtin_memcpy(&a0, wp, 16); __builtin_memcpy(&a1, wp + 4, 16); __builtin_memcpy(&a2, wp + 8, 16); __builtin_memcpy(&a3, wp + 12, 16);
                uint4* lw = (uint4*)lrec;
                lw[0] = a0; lw[1] = a1; lw[2] = a2; lw[3] = a3;
                if (local) {   // the two lanes' zero-mask words (two per lane behind the rectangle's trace words), at record bytes 64 .. 79
                    uint4 z;
                    __builtin_memcpy(&z, t.trace + t.tbase + t.zoff + t.tw_lane0 * 2, 16);
                    lw[4] = z;
                }
            }
        } else
        if (local) {
            // LOCAL_START batches, per-pair rectangles: every trace word is followed by its cells' zero-mask word (place_rect) -- the same window
            // of two column groups x five lanes is 2 x 10 contiguous words
            if (!(t.tw_ok && chunk == t.tw_chunk && t.tw_g - g < 2u && lc - t.tw_lane0 < 5u)) {
                t.tw_chunk = chunk; t.tw_g = g; t.tw_lane0 = min(lc >= 4 ? lc - 4 : 0u, t.nl - 5u); t.tw_ok = true;
                const uint32_t* hi = t.trace + t.tbase + ((g * t.nch + chunk) * t.nl + t.tw_lane0) * 2u;
                const uint32_t* lo = g ? hi - t.nch * t.nl * 2u : hi;
                uint4 h0, h1, l0, l1; uint2 h2, l2;
                __builtin_memcpy(&h0, hi, 16); __builtin_memcpy(&h1, hi + 4, 16); __builtin_memcpy(&h2, hi + 8, 8);
                __builtin_memcpy(&l0, lo, 16); __builtin_memcpy(&l1, lo + 4, 16); __builtin_memcpy(&l2, lo + 8, 8);
                uint32_t* lw = (uint32_t*)lrec;
                lw[0] = h0.x; lw[1] = h0.y; lw[2] = h0.z; lw[3] = h0.w; lw[4] = h1.x; lw[5] = h1.y; lw[6] = h1.z; lw[7] = h1.w; lw[8] = h2.x; lw[9] = h2.y;
                lw[10] = l0.x; lw[11] = l0.y; lw[12] = l0.z; lw[13] = l0.w; lw[14] = l1.x; lw[15] = l1.y; lw[16] = l1.z; lw[17] = l1.w; lw[18] = l2.x; lw[19] = l2.y;
            }
        } else
        if (!(t.tw_ok && chunk == t.tw_chunk && t.tw_g - g < 2u && lc - t.tw_lane0 < 5u)) {
            // the two column groups ending at the cell's and the five lanes ending at the cell's
            // (five consecutive words per group: one 16-byte and one 4-byte load each -- the walks of a wave are in 64 different
            // places, so every load instruction is 64 transactions; the window is pushed down where it would pass the last lane)
            t.tw_chunk = chunk; t.tw_g = g; t.tw_lane0 = min(lc >= 4 ? lc - 4 : 0u, t.nl - 5u); t.tw_ok = true;   // (nl >= 8)
            const uint32_t* hi = t.trace + t.tbase + (g * t.nch + chunk) * t.nl + t.tw_lane0;
            const uint32_t* lo = g ? hi - t.nch * t.nl : hi;
            uint32_t wv[10];
            {
                uint4 h4, l4;
                __builtin_memcpy(&h4, hi, 16); __builtin_memcpy(&l4, lo, 16);
                wv[0] = h4.x; wv[1] = h4.y; wv[2] = h4.z; wv[3] = h4.w; wv[4] = hi[4];
                wv[5] = l4.x; wv[6] = l4.y; wv[7] = l4.z; wv[8] = l4.w; wv[9] = lo[4];
            }
#pragma unroll
            for (int k = 0; k < 10; k++) ((uint32_t*)lrec)[k] = wv[k];
        }
        if (eq) {   // refill the sequence windows when fewer than 8 positions are left below the current one
            if (t.qw0 == 0xffffffffu || t.i < t.qw0 + 8 || t.i >= t.qw0 + 16) {
                t.qw0 = t.i >= 12 ? (t.i - 12) & ~3u : 0;
                const uint32_t* p = (const uint32_t*)(t.q + t.qw0);
                uint32_t b4[4];
#pragma unroll
                for (int k = 0; k < 4; k++) b4[k] = p[k];
#pragma unroll
                for (int k = 0; k < 4; k++) ((uint32_t*)lrec)[SEQ_Q / 4 + k] = b4[k];
            }
            if (t.rw0 == 0xffffffffu || t.j < t.rw0 + 8 || t.j >= t.rw0 + 16) {
                t.rw0 = t.j >= 12 ? (t.j - 12) & ~3u : 0;
                const uint32_t* p = (const uint32_t*)(t.r + t.rw0);
                uint32_t b4[4];
#pragma unroll
                for (int k = 0; k < 4; k++) b4[k] = p[k];
#pragma unroll
                for (int k = 0; k < 4; k++) ((uint32_t*)lrec)[SEQ_R / 4 + k] = b4[k];
            }
        }
    }
#ifdef BA_TIMING
    const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
    if (tacc) { tacc[0] += ts1 - ts0; tacc[1] += ts2 - ts1; }
#endif
    {
        // Straight-line, predicated: the lanes of the wave are at different places of different walks, so every `break`
        // of a plain loop is an exec-mask region that all of them pay for (it had ~15 per cell). `alive` = this
        // lane still walks in this call; a lane that is not alive computes on clamped indices and commits nothing.
        bool alive = true;
#pragma unroll
        for (int s = 0; s < CELLS; s++) {
            alive = alive && (t.i > 0 || t.j > 0) && t.i >= t.bi && t.j >= t.bj;
            const uint32_t ci = t.i - t.bi, cj = t.j - t.bj;
            const uint32_t v = t.right ? ci : cj, w = t.right ? cj : ci;
            const uint32_t lc = (v & 127) >> 1;
            if (fqs) {                                                                      // scan_block.rs:1597-1599
                const bool stop = alive && t.right && t.i == 0;
                if (stop) t.i = t.j = 0;
                alive = alive && !stop;
            }
            const uint32_t gi = t.tw_g - (w >> 2);
            uint32_t baddr, zaddr = 0, zbit = 0;   // (zaddr / zbit: LOCAL_START -- the cell's zero-mask bit in the record)
            if (L2OK && t.l2) {
                const uint32_t k8 = (v >> 3) - t.tw_lane0;
                alive = alive && t.tw_chunk == 0xffffu && k8 <= 1u;
                baddr = k8 * 32 + (w >> 1) * 8 + ((v >> 2) & 1) * 4 + (v & 3);
                zaddr = 64 + k8 * 8 + ((v >> 2) & 1) * 4 + (v & 3); zbit = w;
            } else {
                const uint32_t k = lc - t.tw_lane0;
                alive = alive && (v >> 7) == t.tw_chunk && gi <= 1u && k <= 4u;             // else: left the window, the next call reloads it
                baddr = local ? gi * 40 + k * 8 + (v & 1) * 2 + ((w & 3) >> 1) : gi * 20 + k * 4 + (v & 1) * 2 + ((w & 3) >> 1);
                zaddr = baddr + 4; zbit = (w & 1) * 4;
            }
            const uint32_t qo = t.i - t.qw0, ro = t.j - t.rw0;
            const uint32_t byte = lrec[alive ? baddr : 0u];
            const uint32_t qb = lrec[SEQ_Q + (qo & 15u)], rb = lrec[SEQ_R + (ro & 15u)];    // for a match at this cell (read alongside, used if needed)
            const uint32_t nib = ((byte >> ((w & 1) * 4)) ^ 15u) & 15u;                      // all four bits stored as "differs"
            const uint32_t table = tb_resolve(t.right, t.table, nib);
            if (local) {                                                                    // scan_block.rs:1604-1611: the path starts at a cell whose D is the zero
                const uint32_t zb = lrec[alive ? zaddr : 0u];
                const bool stop = alive && table == 0 && ((zb >> zbit) & 1u);
                if (stop) t.i = t.j = 0;
                alive = alive && !stop;
            }
            const uint32_t m = lut[((uint32_t)t.right << 6) | (table << 4) | nib];          // op | di << 3 | dj << 4 | next << 5
            uint32_t op = m & 7u;
            const uint32_t di = (m >> 3) & 1u, dj = (m >> 4) & 1u;
            const bool is_match = eq && op == 1;
            alive = alive && !(is_match && (qo > 15u || ro > 15u));                         // the next call refills the sequence windows
            if (is_match) op = qb == rb ? 2u : 3u;
            if (alive && (di > t.i || dj > t.j)) { tb_fail(t); alive = false; }             // would leave the matrix: corrupt trace
            const bool same = op == t.run_op;
            if (alive && !same && t.run_len) {                                              // the run ends: the one store of the walk
                if (t.wp == t.lo) { t.status |= ST_CIGAR_OVERFLOW; t.i = t.j = 0; alive = false; }
                else out[--t.wp] = (t.run_len << 4) | t.run_op;
            }
            t.i -= alive ? di : 0u; t.j -= alive ? dj : 0u;
            t.table = alive ? tb_next(t.right, m >> 5, v) : t.table;
            t.run_len = alive ? (same ? t.run_len + 1 : 1u) : t.run_len;
            t.run_op = alive ? op : t.run_op;
        }
    }
}

// ------------------------------------------------------------------ round 6: tb_step_fast
// What a lane's walk cost in round 5 (tb_step above, 8 cells per call): ~1270 vector + ~780 scalar instructions per call on the wave's
// instruction stream (k_multi's traceback waves; ~130 per cell, and ~270 in the loop over the records fetched ahead), up to three memory
// round trips one after the other (trace window, then the query bytes, then the reference bytes -- each load waited for before the next was
// issued), and a store per finished run wherever it finished. Nine cells in ten of a path are matches / mismatches, so:
//  * tb_diag: in state D a cell whose C and R flags both say "differs" is a diagonal move that leaves the state D (OP_LUT,
//    scan_block.rs:1532-1558). The (up to) eight cells of the diagonal below the current one inside the lane's window are read with eight
//    byte reads at addresses out of two small tables, their "is a diagonal move" bits come from a handful of packed operations, the cells'
//    sequence bytes are compared (CIGAR_EQ: scan_block.rs:1620-1628) and the = / X runs appended to the lane's run state in BA_TB_DIAG_RUNS
//    straight-line rounds. A call is [tb_diag] [BA_TB_FCELLS cells of the general code: a gap] [tb_diag].
//  * the rectangle that holds the cell: compares and selects over the DEPTH records fetched ahead, the queue shifted by selects, only the
//    records that were passed fetched again.
//  * ONE memory round trip per call: records, trace window and both sequence windows are issued together, then waited for.
//  * no store behind the call's loads: gfx9 counts loads and stores in one in-order counter (vmcnt), and the compiler must wait for
//    everything when a store may or may not have been issued after the load it waits for. A finished run therefore stays in the lane's
//    registers (TbLane::pv / po, one slot per place of the call that can finish one) and is stored in the NEXT call, right behind that call's
//    loads -- the wait for those covers it, and nothing younger than a load is outstanding when the loop comes round.
// (BA_TB_FAST = 0: the round-5 walk, for same-box A/B. First version of this round -- tb_diag inside tb_step, stores where the runs
// ended, every record fetched again in every call: 27 % fewer instructions per call and config 3 at 176 instead of 162 ms, the 12 500-pair
// batch at 53 instead of 41 ms: the stores of a call's last runs were waited for at the top of the next call, a full round trip each.)
#ifndef BA_TB_FAST
#define BA_TB_FAST 1
#endif
#ifndef BA_TB_FCELLS
#define BA_TB_FCELLS 1
#endif
#ifndef BA_TB_DIAG_RUNS
#define BA_TB_DIAG_RUNS 2
#endif
#ifndef BA_TB_DIAG2
#define BA_TB_DIAG2 1   // the second tb_diag of a call (behind the general cells)
#endif
#ifndef BA_TB_FDEPTH
#define BA_TB_FDEPTH 3  // records fetched ahead
#endif
// sixteen bytes at a 4-byte aligned address, as one load (values, not a copy through memory: the registers stay registers)
typedef uint32_t tb_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 tb_ld16(const void* p) {
    typedef tb_u32x4 __attribute__((aligned(4))) V;
    const tb_u32x4 v = *(const V*)p;
    return uint4{v.x, v.y, v.z, v.w};
}
// v_ffbl_b32: the position of the lowest set bit, 0xffffffff for 0
__device__ __forceinline__ uint32_t ffbl_u32(uint32_t x) { uint32_t r; asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x)); return r; }
// the walk's first records, fetched when the lane takes the walk (tb_step_fast only looks at records fetched ahead)
__device__ __forceinline__ void tb_prefetch2(TbLane& t) {
    if (t.bidx > 0u) t.nrec[0] = *(const uint4*)(t.blocks + t.bidx - 1);
    if (t.bidx > 1u) t.nrec[1] = *(const uint4*)(t.blocks + t.bidx - 2);
    t.nq = min(2u, t.bidx);
}
// Tables of tb_diag behind the lanes' records (8-byte aligned): per (position, k) the byte offset of the cell k steps up the diagonal.
//   slot rectangles (records of TB_LANE_BYTES_L2: a window of 16 rows x 8 columns, byte = (row >> 3) * 32 + (column >> 1) * 8 + (row & 7)):
//     F[u][k], u = row in the window (16 x 8 bytes), then G[w][k], w = column (8 x 8 bytes)
//   per-pair rectangles (records of TB_LANE_BYTES: 5 lanes x 2 column groups, byte = (newer group ? 0 : 20) + row * 2 + ((column & 3) >> 1)):
//     C[x][k], x = column in the window's 8 columns (8 x 8 bytes); the row term is arithmetic
template <bool L2W>
__device__ __forceinline__ void tb_diag_lut_fill(unsigned char* dlut) {
    if constexpr (L2W) {
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const uint32_t idx = (uint32_t)lane_id() + 64u * e;   // 0 .. 191
            const uint32_t k = idx & 7u, p = idx >> 3;              // p < 16: F, else G
            const uint32_t x = (p - k) & 15u, c = ((p - 16u) - k) & 7u;
            dlut[idx] = (unsigned char)(p < 16u ? (((x & 8u) << 2) | (x & 7u)) : ((c >> 1) << 3));
        }
    }
    {   // C: alone (records of TB_LANE_BYTES) or behind F and G (a kernel of slot rectangles also walks the rectangles its solo driver wrote)
        const uint32_t idx = (uint32_t)lane_id(), k = idx & 7u, x = ((idx >> 3) - k) & 7u;
        dlut[(L2W ? TB_DIAG_LUT_L2 - TB_DIAG_LUT_STD : 0u) + idx] = (unsigned char)((x < 4u ? 20u : 0u) + ((x & 3u) >> 1));
    }
}
// a finished run into slot S of the lane's deferred stores (tb_flush); ovf: the pair's range is full
template <int S>
__device__ __forceinline__ void tb_push(TbLane& t, const bool cond, const uint32_t val, bool& ovf) {
    const bool room = t.wp != t.lo, go = cond && room;
    t.wp -= go ? 1u : 0u;
    t.pv[S] = go ? val : t.pv[S];
    t.po[S] = go ? (uint32_t)t.wp - (uint32_t)t.lo : t.po[S];
    t.pmask |= go ? (1u << S) : 0u;
    ovf = ovf || (cond && !room);
}
__device__ __forceinline__ void tb_flush(TbLane& t, uint32_t* __restrict__ out) {
#pragma unroll
    for (int s = 0; s < 6; s++) if ((t.pmask >> s) & 1u) out[t.lo + t.po[s]] = t.pv[s];
    t.pmask = 0;
}

// (see above) t: the lane's walk; dlut: the tables; lrec: the lane's record; S0: the first of this call site's two slots. Every lane of the
// wave executes this; a lane that cannot use it (another state, no window, a rectangle of the other layout) commits nothing.
template <bool L2W, int SEQ_Q, int SEQ_R, int S0>
__device__ __forceinline__ void tb_diag(TbLane& t, const bool eq, const unsigned char* lrec, const unsigned char* dlut, bool& ovf) {
    static_assert(BA_TB_DIAG_RUNS >= 1 && BA_TB_DIAG_RUNS <= 2, "two slots per call site");
    const uint32_t ci = t.i - t.bi, cj = t.j - t.bj;
    const uint32_t v = t.right ? ci : cj, w = t.right ? cj : ci;
    bool ok = t.in_rect && t.tw_ok && t.i >= t.bi && t.j >= t.bj && (t.table == 0u || t.table == TB_PENDING);
    uint32_t K = min(min(t.i, t.j), min(v, w) + 1u);   // cells 0 .. K - 1 of the diagonal: inside the rectangle, and a diagonal move from them stays inside the matrix
    uint32_t alo, ahi;                                 // the eight cells' byte offsets in the record, packed
    {
        // a per-pair rectangle's window (5 lanes x 2 column groups): row u / column x inside it (x: + 4 = the newer group; w <= 4 tw_g + 3)
        const uint32_t u = (v & 127u) - 2u * t.tw_lane0, x = w + 4u - 4u * t.tw_g;
        const bool ok_s = !t.l2 && (v >> 7) == t.tw_chunk && u < 10u && x < 8u;
        const uint32_t K_s = min(u, x) + 1u;
        const uint2 C = *(const uint2*)(dlut + (L2W ? TB_DIAG_LUT_L2 - TB_DIAG_LUT_STD : 0u) + (x & 7u) * 8u);
        const uint32_t r2 = (2u * (u & 15u)) * 0x01010101u;   // (a borrow out of a byte only reaches cells further up, which are then outside the window too)
        alo = ((r2 - 0x06040200u) + C.x) & 0x3f3f3f3fu; ahi = ((r2 - 0x0e0c0a08u) + C.y) & 0x3f3f3f3fu;
        if constexpr (L2W) {
            // a slot rectangle's window (16 rows x 8 columns)
            const uint32_t u2 = v - 8u * t.tw_lane0;
            const bool ok_2 = t.l2 && t.tw_chunk == 0xffffu && u2 < 16u;
            const uint2 F = *(const uint2*)(dlut + (u2 & 15u) * 8u), G = *(const uint2*)(dlut + 128u + (w & 7u) * 8u);
            alo = t.l2 ? F.x + G.x : alo; ahi = t.l2 ? F.y + G.y : ahi;
            ok = ok && (ok_2 || ok_s);
            K = min(K, t.l2 ? u2 + 1u : K_s);
        } else {
            ok = ok && ok_s;
            K = min(K, K_s);
        }
    }
    const uint32_t qo = t.i - t.qw0, ro = t.j - t.rw0;
    if (eq) { ok = ok && qo <= 15u && ro <= 15u; K = min(K, min(qo, ro) + 1u); }
    K = ok ? min(K, 8u) : 0u;
    uint32_t b[8];
#pragma unroll
    for (int k = 0; k < 8; k++) b[k] = lrec[((k < 4 ? alo : ahi) >> (8 * (k & 3))) & 0xffu];
    // "both low flags of the cell's nibble set" for the eight cells: the nibble alternates with the column's parity
    const uint32_t sh0 = (w & 1u) * 4u;
    const uint32_t e4 = (b[0] | (b[2] << 8) | (b[4] << 16) | (b[6] << 24)) >> sh0, o4 = (b[1] | (b[3] << 8) | (b[5] << 16) | (b[7] << 24)) >> (4u - sh0);
    const uint32_t z = ((e4 & (e4 >> 1)) & 0x01010101u) | (((o4 & (o4 >> 1)) & 0x01010101u) << 4);   // bit 4k: cell k is a match / mismatch in state D
    uint32_t n = ffbl_u32(~z & 0x11111111u) >> 2;   // (every cell is one: ffbl = -1; K <= 8 bounds it below)
    // a pending scan-direction gap ends at this cell if it was opened here (tb_resolve: stored bit 3 clear), else the general code goes on with the gap
    const bool d_state = t.table == 0u || !((e4 >> 3) & 1u);
    n = d_state ? min(n, K) : 0u;
    uint32_t ne = 0;   // bit 4k: the bytes of cell k differ
    if (eq) {
        const unsigned char* qb = lrec + SEQ_Q + (ok ? qo : 7u) - 7u; const unsigned char* rb = lrec + SEQ_R + (ok ? ro : 7u) - 7u;
#pragma unroll
        for (int k = 0; k < 8; k++) ne |= min((uint32_t)qb[7 - k] ^ (uint32_t)rb[7 - k], 1u) << (4 * k);
    }
    uint32_t pos = 0;
#pragma unroll
    for (int it = 0; it < (BA_TB_DIAG_RUNS); it++) {
        if (it > 0 && !eq) break;   // (without CIGAR_EQ the cells are one run of M)
        const uint32_t rem = n - pos;
        const uint32_t cur = ne >> ((4u * pos) & 31u);
        const uint32_t is_x = cur & 1u;
        const uint32_t y = (cur ^ (0u - is_x)) & 0x11111111u;     // bit 4k: cell pos + k is of the other kind
        const uint32_t len = min(ffbl_u32(y) >> 2, rem);           // (no such cell: ffbl = -1, more than any rem; rem = 0: nothing happens below)
        const uint32_t op = eq ? 2u + is_x : 1u;
        const bool same = op == t.run_op;
        if (it == 0) tb_push<S0>(t, rem > 0u && !same && t.run_len != 0u, (t.run_len << 4) | t.run_op, ovf);
        else tb_push<S0 + 1>(t, rem > 0u && !same && t.run_len != 0u, (t.run_len << 4) | t.run_op, ovf);
        t.run_len = same ? t.run_len + len : (rem > 0u ? len : t.run_len);
        t.run_op = rem > 0u ? op : t.run_op;
        pos += len;
    }
    t.i -= pos; t.j -= pos;
    t.table = pos ? 0u : t.table;
}

// One call of a lane's walk (CIGAR_EQ is the only mode bit: LOCAL_START / FREE_QUERY_START_GAPS walks keep tb_step). LB: TB_LANE_BYTES
// (per-pair rectangles only) or TB_LANE_BYTES_L2 (slot rectangles too; tb_diag then takes only those).
template <int DEPTH, int LB>
__device__ __forceinline__ void tb_step_fast(TbLane& t, const bool eq, uint32_t* __restrict__ out, unsigned char* lrec, const unsigned char* lut, const unsigned char* dlut) {
    static_assert(DEPTH >= 2 && DEPTH <= 4 && (BA_TB_FCELLS) >= 1 && (BA_TB_FCELLS) <= 2, "");
    constexpr bool L2OK = LB == (int)TB_LANE_BYTES_L2;
    constexpr int SEQ_Q = L2OK ? 64 : 40, SEQ_R = SEQ_Q + 16;   // byte offsets of the sequence windows in the record
    // ---- 1. the rectangle that holds the cell (scan_block.rs:1578-1590): the first of the records fetched ahead that does
    if (!t.in_rect || !(t.i >= t.bi && t.j >= t.bj)) {
        if (t.bidx == 0) { tb_fail(t); return; }
        bool found = false;
        uint32_t take = t.nq;                    // none of them: all are passed
        uint4 rec = t.nrec[0];
#pragma unroll
        for (int a = DEPTH - 1; a >= 0; a--) {   // (last to first: the first record that holds the cell wins)
            const bool c = (uint32_t)a < t.nq && t.i >= (t.nrec[a].x & 0x7fffffffu) && t.j >= t.nrec[a].y;
            rec.x = c ? t.nrec[a].x : rec.x; rec.y = c ? t.nrec[a].y : rec.y; rec.z = c ? t.nrec[a].z : rec.z; rec.w = c ? t.nrec[a].w : rec.w;
            take = c ? (uint32_t)a + 1u : take;
            found = found || c;
        }
        // the queue moves up by `take`; what it lacks then is fetched (for the next call)
        const uint32_t nq_old = t.nq;
        t.bidx -= take;
        uint4 nn[DEPTH];
#pragma unroll
        for (int k = 0; k < DEPTH; k++) {
            nn[k] = t.nrec[k];
            bool have = false;
#pragma unroll
            for (int sft = 1; k + sft < DEPTH; sft++) {
                const bool m = take == (uint32_t)sft && (uint32_t)(k + sft) < nq_old;
                nn[k].x = m ? t.nrec[k + sft].x : nn[k].x; nn[k].y = m ? t.nrec[k + sft].y : nn[k].y; nn[k].z = m ? t.nrec[k + sft].z : nn[k].z; nn[k].w = m ? t.nrec[k + sft].w : nn[k].w;
                have = have || m;
            }
            if (!have && t.bidx > (uint32_t)k) nn[k] = *(const uint4*)(t.blocks + t.bidx - 1 - k);
        }
#pragma unroll
        for (int k = 0; k < DEPTH; k++) t.nrec[k] = nn[k];
        t.nq = min((uint32_t)DEPTH, t.bidx);
        t.in_rect = found;
        t.l2 = L2OK && (rec.x >> 31); t.bi = rec.x & 0x7fffffffu; t.bj = rec.y;
        t.right = rec.w >> 31;
        t.tbase = rec.w & 0x3fffffffu;
        const uint32_t Hv = t.right ? (rec.z & 0xffffu) : (rec.z >> 16);
        t.nch = Hv > 128 ? Hv / 128 : 1; t.nl = Hv > 128 ? 64 : Hv / 2;
        t.tw_ok = false;
        if (!found) return;
        if (rec.w & 0x40000000u) { tb_fail(t); return; }   // a speculative grow that was never materialised: must not be on a path
    }
    // ---- 2. what the cell's neighbourhood needs from memory: the trace window and the sequence windows, all loads issued before any is waited for
    const uint32_t ci0 = t.i - t.bi, cj0 = t.j - t.bj;
    const uint32_t v0 = t.right ? ci0 : cj0, w0 = t.right ? cj0 : ci0;
    const bool is_l2 = L2OK && t.l2;
    bool ld_l2 = false, ld_std = false;
    if constexpr (L2OK) {
        // a slot's rectangle (128 or 32 x 8 cells, one chunk): a lane's 8 cells x 8 columns are 32 contiguous bytes (word lane8 * 8 + (column >> 1) * 2 + cell quad);
        // the window is the two 8-cell lanes ending at the cell's: 64 contiguous bytes. tw_lane0: the first lane8
        const uint32_t L8 = v0 >> 3;
        ld_l2 = is_l2 && !(t.tw_ok && t.tw_chunk == 0xffffu && L8 - t.tw_lane0 < 2u);
        if (ld_l2) { t.tw_chunk = 0xffffu; t.tw_g = 0; t.tw_lane0 = L8 ? L8 - 1 : 0u; t.tw_ok = true; }
    }
    {
        // per-pair rectangles: the two column groups ending at the cell's and the five lanes ending at the cell's (five consecutive words per group)
        const uint32_t chunk = v0 >> 7, lc = (v0 & 127) >> 1, g = w0 >> 2;
        ld_std = !is_l2 && !(t.tw_ok && chunk == t.tw_chunk && t.tw_g - g < 2u && lc - t.tw_lane0 < 5u);
        if (ld_std) { t.tw_chunk = chunk; t.tw_g = g; t.tw_lane0 = min(lc >= 4 ? lc - 4 : 0u, t.nl - 5u); t.tw_ok = true; }   // (nl >= 8)
    }
    // four 16-byte loads either way (one region of straight-line loads: registers shared between two regions would make the second wait for the
    // first's loads): slot rectangle -- the window's 64 bytes; per-pair rectangle -- the newer group's words 0..3, the older group's, then words 4.. of
    // each (of which only the first is used; a rectangle's words are followed by 64 words of slack at the end of the slot)
    uint4 wa = {0, 0, 0, 0}, wb = {0, 0, 0, 0}, wc = {0, 0, 0, 0}, wd = {0, 0, 0, 0};
    {
        const uint32_t* tb0 = t.trace + t.tbase;
        const uint32_t o_hi = (t.tw_g * t.nch + t.tw_chunk) * t.nl + t.tw_lane0, o_lo = t.tw_g ? o_hi - t.nch * t.nl : o_hi;
        const uint32_t o_l2 = t.tw_lane0 * 8;
        const uint32_t o0 = is_l2 ? o_l2 : o_hi, o1 = is_l2 ? o_l2 + 4 : o_lo, o2 = is_l2 ? o_l2 + 8 : o_hi + 4, o3 = is_l2 ? o_l2 + 12 : o_lo + 4;
        if (ld_l2 || ld_std) { wa = tb_ld16(tb0 + o0); wb = tb_ld16(tb0 + o1); wc = tb_ld16(tb0 + o2); wd = tb_ld16(tb0 + o3); }
    }
    uint4 qv = {0, 0, 0, 0}, rv = {0, 0, 0, 0};
    const bool ld_q = eq && (t.qw0 == 0xffffffffu || t.i < t.qw0 + 8 || t.i >= t.qw0 + 16);   // fewer than 8 positions left below the current one
    const bool ld_r = eq && (t.rw0 == 0xffffffffu || t.j < t.rw0 + 8 || t.j >= t.rw0 + 16);
    if (ld_q) { t.qw0 = t.i >= 12 ? (t.i - 12) & ~3u : 0; qv = tb_ld16(t.q + t.qw0); }
    if (ld_r) { t.rw0 = t.j >= 12 ? (t.j - 12) & ~3u : 0; rv = tb_ld16(t.r + t.rw0); }
    // ---- 3. the runs the previous call finished: stored behind this call's loads
    tb_flush(t, out);
    // ---- 4. into the lane's record
    if constexpr (L2OK) if (ld_l2) { uint4* lw = (uint4*)lrec; lw[0] = wa; lw[1] = wb; lw[2] = wc; lw[3] = wd; }
    if (ld_std) {
        uint32_t* lw = (uint32_t*)lrec;
        lw[0] = wa.x; lw[1] = wa.y; lw[2] = wa.z; lw[3] = wa.w; lw[4] = wc.x; lw[5] = wb.x; lw[6] = wb.y; lw[7] = wb.z; lw[8] = wb.w; lw[9] = wd.x;
    }
    if (ld_q) { uint32_t* lw = (uint32_t*)lrec + SEQ_Q / 4; lw[0] = qv.x; lw[1] = qv.y; lw[2] = qv.z; lw[3] = qv.w; }
    if (ld_r) { uint32_t* lw = (uint32_t*)lrec + SEQ_R / 4; lw[0] = rv.x; lw[1] = rv.y; lw[2] = rv.z; lw[3] = rv.w; }
    // ---- 5. the walk: runs of diagonal moves, a gap cell (the general code: tb_step's), runs of diagonal moves
    bool ovf = false;
    tb_diag<L2OK, SEQ_Q, SEQ_R, 0>(t, eq, lrec, dlut, ovf);
    {
        bool alive = true;
#pragma unroll
        for (int s = 0; s < (BA_TB_FCELLS); s++) {
            alive = alive && (t.i > 0 || t.j > 0) && t.i >= t.bi && t.j >= t.bj;
            const uint32_t ci = t.i - t.bi, cj = t.j - t.bj;
            const uint32_t v = t.right ? ci : cj, w = t.right ? cj : ci;
            const uint32_t lc = (v & 127) >> 1;
            const uint32_t gi = t.tw_g - (w >> 2);
            uint32_t baddr;
            if (L2OK && t.l2) {
                const uint32_t k8 = (v >> 3) - t.tw_lane0;
                alive = alive && t.tw_chunk == 0xffffu && k8 <= 1u;
                baddr = k8 * 32 + (w >> 1) * 8 + ((v >> 2) & 1) * 4 + (v & 3);
            } else {
                const uint32_t k = lc - t.tw_lane0;
                alive = alive && (v >> 7) == t.tw_chunk && gi <= 1u && k <= 4u;             // else: left the window, the next call reloads it
                baddr = gi * 20 + k * 4 + (v & 1) * 2 + ((w & 3) >> 1);
            }
            const uint32_t qo = t.i - t.qw0, ro = t.j - t.rw0;
            const uint32_t byte = lrec[alive ? baddr : 0u];
            const uint32_t qb = lrec[SEQ_Q + (qo & 15u)], rb = lrec[SEQ_R + (ro & 15u)];    // for a match at this cell (read alongside, used if needed)
            const uint32_t nib = ((byte >> ((w & 1) * 4)) ^ 15u) & 15u;                      // all four bits stored as "differs"
            const uint32_t table = tb_resolve(t.right, t.table, nib);
            const uint32_t m = lut[((uint32_t)t.right << 6) | (table << 4) | nib];          // op | di << 3 | dj << 4 | next << 5
            uint32_t op = m & 7u;
            const uint32_t di = (m >> 3) & 1u, dj = (m >> 4) & 1u;
            const bool is_match = eq && op == 1;
            alive = alive && !(is_match && (qo > 15u || ro > 15u));                         // the next call refills the sequence windows
            if (is_match) op = qb == rb ? 2u : 3u;
            if (alive && (di > t.i || dj > t.j)) { tb_fail(t); alive = false; }             // would leave the matrix: corrupt trace
            const bool same = op == t.run_op;
            if (s == 0) tb_push<2>(t, alive && !same && t.run_len != 0u, (t.run_len << 4) | t.run_op, ovf);
            else tb_push<3>(t, alive && !same && t.run_len != 0u, (t.run_len << 4) | t.run_op, ovf);
            t.i -= alive ? di : 0u; t.j -= alive ? dj : 0u;
            t.table = alive ? tb_next(t.right, m >> 5, v) : t.table;
            t.run_len = alive ? (same ? t.run_len + 1 : 1u) : t.run_len;
            t.run_op = alive ? op : t.run_op;
        }
    }
    if constexpr ((BA_TB_DIAG2) != 0) tb_diag<L2OK, SEQ_Q, SEQ_R, 4>(t, eq, lrec, dlut, ovf);
    if (ovf) { t.status |= ST_CIGAR_OVERFLOW; t.i = t.j = 0; }
}

// `nlanes` lanes of the wave take part. dedicated: a traceback wave proper. Otherwise a fill wave that found the work
// counter exhausted: the batch ends with one full walk latency after the last fill, and a walk is a chain of ~2 500
// dependent iterations whose length grows with the number of lanes walking in lockstep, so the last `tb_reserve`
// hand-offs are left to these late helpers, one lane per wave on a SIMD that has nothing else left to do (the
// dedicated waves stop claiming tickets once the ticket counter reaches that reserve).
template <int LB = (int)TB_LANE_BYTES, int CELLS = TB_CELLS_PER_STEP, int DEPTH = BA_RING_DEPTH>
__device__ __forceinline__ void traceback_consumer(const BatchParams& bp, uint32_t flag_mask, unsigned char* tb_lds, uint32_t nlanes, bool dedicated, int prio = -1) {
    enum { IDLE = 0, WAIT = 1, WALK = 2, RETIRED = 3 };
    int phase = (uint32_t)lane_id() < nlanes ? IDLE : RETIRED;
    uint32_t claimed = 0, pend = 0;
    const uint32_t reserve_from = bp.n > bp.tb_reserve ? bp.n - bp.tb_reserve : 0u;
    TbLane t{};
    const uint32_t eq = bp.flags & flag_mask;   // mode bits the walk looks at: CIGAR_EQ, LOCAL_START, FREE_QUERY_START_GAPS
    uint32_t* head = bp.tb_ctrl + 32;
    // this lane's LDS record and the move table (scan_block.rs:1532-1558), two entries built per lane
    unsigned char* lut = tb_lds;
    unsigned char* lrec = tb_lds + TB_LUT_BYTES + (uint32_t)lane_id() * (uint32_t)LB;
    // round 6: tb_step_fast for every walk that needs no mode bit but CIGAR_EQ; tb_diag's tables sit behind the records of the lanes that take part
    constexpr bool FASTP = (BA_TB_FAST) != 0 && LB != (int)TB_LANE_BYTES_LOC;
    unsigned char* dlut = tb_lds + ((TB_LUT_BYTES + nlanes * (uint32_t)LB + 7u) & ~7u);
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const uint32_t idx = (uint32_t)lane_id() + 64u * e;
        const Move mv = tb_lut(idx >> 6, idx & 3, (idx >> 2) & 1, (idx >> 4) & 3);   // (bit 3 of the cell's nibble is resolved before the lookup)
        lut[idx] = (unsigned char)(mv.op | (mv.di << 3) | (mv.dj << 4) | (mv.next << 5));
    }
    if constexpr (FASTP) tb_diag_lut_fill<LB == (int)TB_LANE_BYTES_L2>(dlut);
    lds_sync();
    // A walk is a long dependent chain of short instructions sharing its SIMD with VALU-saturating fill waves; without
    // priority it gets a quarter of the issue slots and every pending walk pins a whole trace slot meanwhile.
#ifndef BA_TB_PRIO
#define BA_TB_PRIO 3
#endif
#ifndef MQ_TB_PRIO
#define MQ_TB_PRIO 1   // (k_multi's traceback waves)
#endif
    if (prio == 1) __builtin_amdgcn_s_setprio(MQ_TB_PRIO); else __builtin_amdgcn_s_setprio(BA_TB_PRIO);   // (k_multi: below its solo mode's priority, ten trace slots per wave give the walks slack: +0.8 %)
#ifdef BA_TIMING
    unsigned long long c_sec[3] = {};
    unsigned long long c_iters = 0, c_walk_lanes = 0, c_walk_iters = 0, c_poll_iters = 0, c_walk_ticks = 0, c_t0 = __builtin_amdgcn_s_memtime();
#endif
    for (;;) {
#ifdef BA_TIMING
        const unsigned long long it0 = __builtin_amdgcn_s_memtime();
        const bool any_walk = __any(phase == WALK);
        c_iters++; c_walk_lanes += __popcll(__ballot(phase == WALK)); c_walk_iters += any_walk; c_poll_iters += __any(phase == WAIT);
#endif
        if (phase == IDLE) {
            // (a dedicated lane looks before it claims: a ticket, once taken, has to be served by this lane)
            if (dedicated && __hip_atomic_load(head, BA_RLX_AGENT) >= reserve_from) phase = RETIRED;
            else {
                claimed = __hip_atomic_fetch_add(head, 1u, BA_RLX_AGENT);
                phase = claimed >= bp.n ? RETIRED : WAIT;     // every pair yields exactly one task
            }
            pend = 0;
        }
        // `pend` = this lane's ring entry as read at the END of the previous iteration: the poll's round trip to L2
        // overlaps with the other lanes' walk instead of stalling every iteration in which some lane is waiting
        const bool got = phase == WAIT && pend != 0;
        const bool walking = phase == WALK;
        const uint32_t entry = pend;
        if (__any(got)) {
            // ONE acquire per poll round that found work, then plain loads (guide G16): drops stale L1 lines of
            // arenas this CU read during earlier walks
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (got) __hip_atomic_store(bp.tb_queue + (claimed & bp.tb_qmask), 0u, BA_RLX_AGENT);
            if (got && (entry & 0x80000000u)) {           // a pair that produced no trace stack (slot time-out)
                bp.cig_len[entry & 0x7fffffffu] = 0;
                phase = IDLE;
            } else if (got) {
                t = TbLane{};
                t.qw0 = t.rw0 = 0xffffffffu; t.tw_ok = false;
                t.slot = entry - 1;
                const SlotInfo si = bp.slot_info[t.slot];
                t.pair = si.pair; t.i = si.end_i; t.j = si.end_j; t.bidx = si.nblocks;
                t.blocks = bp.blocks + (uint64_t)t.slot * bp.blocks_stride;
                t.trace = bp.trace_arena + (uint64_t)t.slot * bp.trace_stride;
                t.q = bp.pool + bp.q_off[t.pair]; t.r = bp.pool + bp.r_off[t.pair];
                t.lo = bp.cig_off[t.pair]; t.wp = bp.cig_off[t.pair + 1];
                t.status = bp.status[t.pair];
                if (t.status || (bp.flags & 0x200u)) t.i = t.j = 0;          // the fill failed (or development switch: skip the walk)
                if constexpr (FASTP) tb_prefetch2(t);
                phase = WALK;
            }
        }
        if (walking) {
#ifdef BA_TIMING
            const unsigned long long tq0 = __builtin_amdgcn_s_memtime();
            if (t.i > 0 || t.j > 0) {
                if constexpr (FASTP) tb_step_fast<BA_TB_FDEPTH, LB>(t, eq != 0u, bp.cig_ops, lrec, lut, dlut);
                else tb_step<CELLS, DEPTH, LB>(t, eq, bp.cig_ops, lrec, lut, c_sec);
            }
            c_sec[2] += __builtin_amdgcn_s_memtime() - tq0;
#else
            // (an emptied fill wave's few lanes -- the batch's last walks, each a chain of memory round trips that ends the launch --
            // take more cells per call and look further ahead; dedicated waves share their SIMD with fill waves: see tb_step)
            if (t.i > 0 || t.j > 0) {
                if constexpr (FASTP) tb_step_fast<BA_TB_FDEPTH, LB>(t, eq != 0u, bp.cig_ops, lrec, lut, dlut);
                else
                if (BA_HELPER_CELLS != CELLS && !dedicated) tb_step<BA_HELPER_CELLS, BA_WALK_DEPTH, LB>(t, eq, bp.cig_ops, lrec, lut);
                else tb_step<CELLS, DEPTH, LB>(t, eq, bp.cig_ops, lrec, lut);
            }
#endif
            if (!(t.i > 0 || t.j > 0)) {
                if constexpr (FASTP) tb_flush(t, bp.cig_ops);
                tb_emit(t, bp.cig_ops);
                bp.cig_len[t.pair] = t.status ? 0u : (uint32_t)(bp.cig_off[t.pair + 1] - t.wp);
                if (t.status) bp.status[t.pair] = t.status;
                __hip_atomic_store(bp.slot_free + t.slot, 1u, BA_RLX_AGENT);   // the arena may be reused
                phase = IDLE;
            }
        }
        if (phase == WAIT) pend = __hip_atomic_load(bp.tb_queue + (claimed & bp.tb_qmask), BA_RLX_AGENT);
#ifdef BA_TIMING
        if (any_walk) c_walk_ticks += __builtin_amdgcn_s_memtime() - it0;
#endif
        if (__all(phase == RETIRED)) break;
        if (!__any(phase == WALK)) __builtin_amdgcn_s_sleep(32);               // nothing to walk: poll gently
    }
#ifdef BA_TIMING
    if (bp.prof && dedicated && is_lane(0)) {
        atomicAdd(bp.prof + 20, c_iters); atomicAdd(bp.prof + 21, c_walk_lanes); atomicAdd(bp.prof + 22, c_walk_iters);
        atomicAdd(bp.prof + 23, c_poll_iters); atomicAdd(bp.prof + 24, c_walk_ticks);
        atomicAdd(bp.prof + 25, __builtin_amdgcn_s_memtime() - c_t0); atomicAdd(bp.prof + 26, 1ull);
        atomicAdd(bp.prof + 27, c_sec[0]); atomicAdd(bp.prof + 28, c_sec[1]); atomicAdd(bp.prof + 29, c_sec[2]);
    }
#endif
}

// A fill wave that waits for a trace slot while the traceback side makes no progress (its waves may not be on the device yet: a launch
// that is only partly resident) walks a pending traceback itself instead of giving up: it takes the ring's head entry -- only if the
// entry is there: compare-and-swap on the head counter after seeing it, so nothing is ever claimed that might not come -- and walks it
// with lane 0 out of its own LDS region. Returns true if a walk was done (one of this wave's slots may be free now).
template <int LB = (int)TB_LANE_BYTES>
__device__ __forceinline__ bool traceback_help_one(const BatchParams& bp, uint32_t flag_mask, unsigned char* tb_lds) {
    uint32_t entry = 0;
    if (is_lane(0)) {
        uint32_t h = __hip_atomic_load(bp.tb_ctrl + 32, BA_RLX_AGENT);
        if (h < bp.n) {
            const uint32_t e = __hip_atomic_load(bp.tb_queue + (h & bp.tb_qmask), BA_RLX_AGENT);
            if (e && __hip_atomic_compare_exchange_strong(bp.tb_ctrl + 32, &h, h + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                entry = e;
                __hip_atomic_store(bp.tb_queue + (h & bp.tb_qmask), 0u, BA_RLX_AGENT);
            }
        }
    }
    entry = (uint32_t)uni((int)entry);
    if (!entry) return false;
    if (entry & 0x80000000u) {   // a pair that produced no trace stack
        if (is_lane(0)) bp.cig_len[entry & 0x7fffffffu] = 0;
        return true;
    }
    unsigned char* lut = tb_lds;
    unsigned char* lrec = tb_lds + TB_LUT_BYTES;
    lds_sync();
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const uint32_t idx = (uint32_t)lane_id() + 64u * e;
        const Move mv = tb_lut(idx >> 6, idx & 3, (idx >> 2) & 1, (idx >> 4) & 3);
        lut[idx] = (unsigned char)(mv.op | (mv.di << 3) | (mv.dj << 4) | (mv.next << 5));
    }
    lds_sync();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (is_lane(0)) {
        const uint32_t eq = bp.flags & flag_mask;
        TbLane t{};
        t.qw0 = t.rw0 = 0xffffffffu; t.tw_ok = false;
        t.slot = entry - 1;
        const SlotInfo si = bp.slot_info[t.slot];
        t.pair = si.pair; t.i = si.end_i; t.j = si.end_j; t.bidx = si.nblocks;
        t.blocks = bp.blocks + (uint64_t)t.slot * bp.blocks_stride;
        t.trace = bp.trace_arena + (uint64_t)t.slot * bp.trace_stride;
        t.q = bp.pool + bp.q_off[t.pair]; t.r = bp.pool + bp.r_off[t.pair];
        t.lo = bp.cig_off[t.pair]; t.wp = bp.cig_off[t.pair + 1];
        t.status = bp.status[t.pair];
        if (t.status || (bp.flags & 0x200u)) t.i = t.j = 0;
        while (t.i > 0 || t.j > 0) tb_step<TB_CELLS_PER_STEP, BA_RING_DEPTH, LB>(t, eq, bp.cig_ops, lrec, lut);
        tb_emit(t, bp.cig_ops);
        bp.cig_len[t.pair] = t.status ? 0u : (uint32_t)(bp.cig_off[t.pair + 1] - t.wp);
        if (t.status) bp.status[t.pair] = t.status;
        __hip_atomic_store(bp.slot_free + t.slot, 1u, BA_RLX_AGENT);
    }
    lds_sync();
    return true;
}

// An emptied fill wave at the end of a batch: it serves hand-offs like a traceback lane (a ticket, then the entry), but with all of
// its lanes on one path at a time (walk_wave) -- the batch ends one walk after its last fill, and this walk is the short one.
template <bool L2OK>
__device__ __forceinline__ void traceback_helper_wave(const BatchParams& bp, uint32_t* lds, uint32_t budget) {
    uint32_t* head = bp.tb_ctrl + 32;
    for (;;) {
        uint32_t claimed = 0;
        if (is_lane(0)) claimed = __hip_atomic_fetch_add(head, 1u, BA_RLX_AGENT);
        claimed = (uint32_t)uni((int)claimed);
        if (claimed >= bp.n) break;
        uint32_t entry = 0;
        for (;;) {
            if (is_lane(0)) entry = __hip_atomic_load(bp.tb_queue + (claimed & bp.tb_qmask), BA_RLX_AGENT);
            entry = (uint32_t)uni((int)entry);
            if (entry) break;
            __builtin_amdgcn_s_sleep(32);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (is_lane(0)) __hip_atomic_store(bp.tb_queue + (claimed & bp.tb_qmask), 0u, BA_RLX_AGENT);
        if (entry & 0x80000000u) { if (is_lane(0)) bp.cig_len[entry & 0x7fffffffu] = 0; continue; }   // a pair that produced no trace stack
        const uint32_t slot = entry - 1;
        const SlotInfo si = bp.slot_info[slot];
        const uint32_t pair = (uint32_t)uni((int)si.pair);
        uint32_t st = bp.status[pair], ncig = 0;
        lds_sync();
        if (!st && !(bp.flags & 0x200u))
            ncig = walk_wave<L2OK>(bp.blocks + (uint64_t)slot * bp.blocks_stride, si.nblocks, bp.trace_arena + (uint64_t)slot * bp.trace_stride, si.end_i, si.end_j,
                                   bp.pool + bp.q_off[pair], bp.pool + bp.r_off[pair], (bp.flags & F_CIGAR_EQ) != 0, bp.cig_ops, bp.cig_off[pair], bp.cig_off[pair + 1],
                                   &st, lds, budget);
        lds_sync();
        if (is_lane(0)) {
            bp.cig_len[pair] = st ? 0u : ncig;
            if (st) bp.status[pair] = st;
            __hip_atomic_store(bp.slot_free + slot, 1u, BA_RLX_AGENT);   // the arena may be reused
        }
    }
}

// Pair-slot batches (every pair's trace stack stays in its own region until the batch ends): the tracebacks of the whole
// batch after its fill kernels, one pair per LANE, all 64 lanes of every wave walking (tb_step). Lanes that finish take the
// next pairs of the batch order (longest first, so the lanes of a wave walk paths of similar length) with one atomic per wave.
// L2OK: the batch holds slot rectangles of k_small (words of 4 cells x 2 columns, see multi_rect): the lanes' records are TB_LANE_BYTES_L2 bytes
// LOC: a LOCAL_START / FREE_QUERY_START_GAPS batch of k_small (records with room for the zero-mask bits; the early stops of scan_block.rs:1597-1611)
template <bool L2OK = false, bool LOC = false>
__device__ __forceinline__ void traceback_all(const BatchParams& bp, uint32_t flag_mask, unsigned char* tb_lds) {
    constexpr int LB = LOC ? (int)TB_LANE_BYTES_LOC : (L2OK ? (int)TB_LANE_BYTES_L2 : (int)TB_LANE_BYTES);
    constexpr uint32_t WAVE_LDS = LOC ? TB_LDS_BYTES_LOC : (L2OK ? TB_LDS_BYTES_L2 : TB_LDS_BYTES);
    const uint32_t eq = bp.flags & flag_mask;
    unsigned char* lut = tb_lds;
    unsigned char* lrec = tb_lds + TB_LUT_BYTES + (uint32_t)lane_id() * (uint32_t)LB;
    constexpr bool FASTP = (BA_TB_FAST) != 0 && !LOC;   // (round 6: tb_step_fast; tb_diag's tables behind the records)
    unsigned char* dlut = tb_lds + ((TB_LUT_BYTES + 64u * (uint32_t)LB + 7u) & ~7u);
    static_assert(LOC || ((TB_LUT_BYTES + 64u * (uint32_t)LB + 7u) & ~7u) + (L2OK ? TB_DIAG_LUT_L2 : TB_DIAG_LUT_STD) <= WAVE_LDS, "k_walk LDS: tb_diag's tables");
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const uint32_t idx = (uint32_t)lane_id() + 64u * e;
        const Move mv = tb_lut(idx >> 6, idx & 3, (idx >> 2) & 1, (idx >> 4) & 3);
        lut[idx] = (unsigned char)(mv.op | (mv.di << 3) | (mv.dj << 4) | (mv.next << 5));
    }
    if constexpr (FASTP) tb_diag_lut_fill<L2OK>(dlut);
    lds_sync();
    TbLane t{};
    bool walking = false, more = true;
    // small-block batches walk in two launches: the pairs k_quad finished (cont_mode 1: flag 0) while the per-pair kernel is still
    // at work on the others, then those (cont_mode 2). 0: every pair.
    const uint32_t filter = bp.cont_mode;
    // The batch order is longest first and the launch ends with its longest walk: the first bp.walk_wave_n paths (ba_host.cpp plan_walks)
    // go one to a wave (walk_wave), one at a time from their own counter (work_counter[1]); the lanes start behind them. Few: such a walk
    // is scalar code, and a CU has one scalar unit for all its waves.
    const uint32_t spec_flags = bp.flags & (F_LOCAL | F_FQS);
    const uint32_t n_wave = (LOC || spec_flags) ? 0u : min(bp.walk_wave_n, bp.n);
    if (n_wave) {
        __builtin_amdgcn_s_setprio(3);   // (the launch's longest chains: ahead of the lanes' walks on the same SIMD -- protein set with traceback +4 %)
        for (;;) {
            uint32_t p = 0;
            if (is_lane(0)) p = atomicAdd(bp.work_counter + 1, 1u);
            p = (uint32_t)uni((int)p);
            if (p >= n_wave) break;
            bool mine = true;
            if (filter) mine = (bp.cont_in_flag[p] == 0) == (filter == 1);
            SlotInfo si{0, 0, 0, 0};
            if (mine) { si = bp.slot_info[p]; mine = si.nblocks != ~0u; }
            if (!mine) continue;
            uint32_t st = bp.status[p], ncig = 0;
            if (!st && !(bp.flags & 0x200u))
                ncig = walk_wave<L2OK>(bp.blocks + bp.blocks_off[p], si.nblocks, bp.trace_arena + bp.trace_off[p], si.end_i, si.end_j, bp.pool + bp.q_off[p],
                                       bp.pool + bp.r_off[p], eq != 0, bp.cig_ops, bp.cig_off[p], bp.cig_off[p + 1], &st, (uint32_t*)tb_lds, WAVE_LDS / 4u);
            if (is_lane(0)) { bp.cig_len[p] = st ? 0u : ncig; if (st) bp.status[p] = st; }
        }
        __builtin_amdgcn_s_setprio(0);
        lds_sync();
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const uint32_t idx = (uint32_t)lane_id() + 64u * e;
            const Move mv = tb_lut(idx >> 6, idx & 3, (idx >> 2) & 1, (idx >> 4) & 3);
            lut[idx] = (unsigned char)(mv.op | (mv.di << 3) | (mv.dj << 4) | (mv.next << 5));
        }
        if constexpr (FASTP) tb_diag_lut_fill<L2OK>(dlut);
        lds_sync();
    }
    uint32_t w_next = 0, w_end = 0;   // this wave's share of the batch order: 64 pairs per atomic (a returning atomic is a full memory
                                      // round trip that every lane of the wave waits for: one per finished lane would double the walk time)
    for (;;) {
        if (!__all(walking) && (more || w_next != w_end)) {
            if (w_next == w_end) {
                uint32_t base = 0;
                if (is_lane(0)) base = atomicAdd(bp.work_counter, 64u);
                base = (uint32_t)uni((int)base) + n_wave;
                if (base >= bp.n) { more = false; base = 0; w_next = w_end = 0; }
                else { w_next = base; w_end = min(base + 64u, bp.n); }
            }
            const unsigned long long im = __ballot(!walking);
            const uint32_t rank = (uint32_t)__popcll(im & ((1ull << lane_id()) - 1ull));
            const uint32_t take = min((uint32_t)__popcll(im), w_end - w_next);
            const uint32_t p = w_next + rank;
            w_next += take;
            bool mine = !walking && rank < take;
            if (filter && mine) mine = (bp.cont_in_flag[p] == 0) == (filter == 1);
            SlotInfo si{0, 0, 0, 0};
            if (mine) { si = bp.slot_info[p]; mine = si.nblocks != ~0u; }   // (~0: the per-pair kernel has walked this path itself)
            if (mine) {
                t = TbLane{};
                t.qw0 = t.rw0 = 0xffffffffu; t.tw_ok = false;
                t.slot = p; t.pair = p; t.i = si.end_i; t.j = si.end_j; t.bidx = si.nblocks;
                t.blocks = bp.blocks + bp.blocks_off[p];
                t.trace = bp.trace_arena + bp.trace_off[p];
                t.q = bp.pool + bp.q_off[p]; t.r = bp.pool + bp.r_off[p];
                t.lo = bp.cig_off[p]; t.wp = bp.cig_off[p + 1];
                t.status = bp.status[p];
                if (t.status || (bp.flags & 0x200u)) t.i = t.j = 0;   // the fill failed (or development switch: skip the walk)
                if constexpr (FASTP) tb_prefetch2(t);
                walking = true;
            }
        }
        if (!__any(walking)) break;
        if (walking) {
            // (no fill wave shares the SIMD here: a call walks on while its window lasts; the mode bits as constants: a walk is a
            // serial chain of ~200 instructions per cell whose length, for the batch's longest pair, ends the launch)
            if (t.i > 0 || t.j > 0) {
                // (k_small's LOCAL_START / FREE_QUERY_START_GAPS batches: the early stops, scan_block.rs:1597-1611)
                if constexpr (LOC) tb_step<BA_WALK_CELLS, BA_WALK_DEPTH, LB>(t, spec_flags | (eq ? (uint32_t)F_CIGAR_EQ : 0u), bp.cig_ops, lrec, lut);
                else if constexpr (FASTP) { if (eq) tb_step_fast<BA_TB_FDEPTH, LB>(t, true, bp.cig_ops, lrec, lut, dlut); else tb_step_fast<BA_TB_FDEPTH, LB>(t, false, bp.cig_ops, lrec, lut, dlut); }
                else
                if (eq) tb_step<BA_WALK_CELLS, BA_WALK_DEPTH, LB>(t, (uint32_t)F_CIGAR_EQ, bp.cig_ops, lrec, lut);
                else tb_step<BA_WALK_CELLS, BA_WALK_DEPTH, LB>(t, 0u, bp.cig_ops, lrec, lut);
            }
            if (!(t.i > 0 || t.j > 0)) {
                if constexpr (FASTP) tb_flush(t, bp.cig_ops);
                tb_emit(t, bp.cig_ops);
                bp.cig_len[t.pair] = t.status ? 0u : (uint32_t)(bp.cig_off[t.pair + 1] - t.wp);
                if (t.status) bp.status[t.pair] = t.status;
                walking = false;
            }
        }
    }
}

// ------------------------------------------------------------------ exchange with the multi-pair kernel (ba_multi.hpp)
// A pair's driver state at the top of the driver loop (scan_block.rs:123-130's locals), wave-uniform: what travels between a
// slot of k_multi -- where it lives in the slot's lanes -- and Aligner::run.
struct PairState {
    uint32_t si, sj; int dir, prev_dir, off, off_max, best_max; uint32_t y_drop_iter; int x_drop_iter, D_corner;
    uint32_t best_i, best_j, ck_i, ck_j; int ck_off; uint32_t ck_tt, ck_nb;
    unsigned long long cells; uint32_t step_budget, trace_top, nblocks, status;
    int exited;   // out of Aligner::run: 1 = the pair goes on in its slot (this state, borders in LDS), 0 = it is finished
};
constexpr uint32_t MQ_B = 128;   // block size of a slot of k_multi
// How a pair enters Aligner::run from k_multi (all passed and returned BY VALUE: through references or pointers the driver's
// loop scalars are demoted to memory and its wave-uniform logic to vector code)
enum { MM_NONE = -1, MM_FRESH = 0, MM_RESUME = 1 };

// ------------------------------------------------------------------ driver
// SPECIAL: the batch uses LOCAL_START / FREE_QUERY_START_GAPS / FREE_QUERY_END_GAPS. A separate instantiation, so the
// common kernels carry none of that state (it costs registers: +30 % spills when folded into one kernel).
struct NoState { int exited; };
// MULTI: the instantiation k_multi (ba_multi.hpp) / k_small (ba_small.hpp) runs in its solo mode (entry / exit with a PairState; everything
// else is the per-pair driver); SLOT_B: the block size at which a pair goes back to its slot (128: k_multi, 32: k_small)
template <int PMAX, int KIND, bool TRACE, bool XDROP, bool SPECIAL, bool MULTI = false, int SLOT_B_ = 128>
struct Aligner {
    static constexpr uint32_t SLOT_B = (uint32_t)SLOT_B_;
    typedef std::conditional_t<MULTI, PairState, NoState> RunOut;
    // The batch descriptor lives in device memory. Only the scalars the step loop needs are copied into registers;
    // everything else (a couple of dozen per-pair output pointers) is re-read where it is used, once per pair, so it
    // does not sit in SGPRs across the whole persistent loop and get spilled to VGPR lanes.
    uint32_t h_flags, h_max_size; int h_x_drop;
    // cold fields: scalar loads from the kernel-argument segment at the point of use (the empty asm keeps the compiler
    // from hoisting them out of the persistent loop and pinning ~50 SGPRs)
    typedef const __attribute__((address_space(4))) BatchParams* ColdPtr;
    __device__ __forceinline__ static ColdPtr coldp() {
        ColdPtr p = (ColdPtr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(p));
        return p;
    }
    const WaveLds& L;
    const FillConsts& fc;
    const uint8_t* q; const uint8_t* r;
    uint32_t qlen, rlen;
    uint32_t* trace; BlockRec* blocks; short* ckpt;   // this wave's slot in the global scratch arenas
    short* big_top = nullptr;                         // big-block kernels: the row hand-off arrays of the tiled fill (2 x big_array_shorts)
    uint32_t trace_top = 0, nblocks = 0;
    int parked = 0;   // lanes: 0 i_ckpt, 1 j_ckpt, 2 off_ckpt, 3 ck_trace_top, 4 ck_nblocks, 5 best_i, 6 best_j, 7 pair, 8 slot, 9-11 the first grow rectangle's max / row / col (see park()), 12 / 13 the capacity of the pair's trace region (words) / record list
    uint32_t status = 0;
    unsigned long long cells = 0;
    // sequence bytes for the next shift step, fetched one step ahead for both possible directions: this lane's two
    // vector-axis bytes (vector loads) and the step's 8 column bytes (one scalar load each, so they arrive wave-uniform)
    int pf_qv = 0, pf_rv = 0;
    unsigned long long pf_qc = 0, pf_rc = 0;
    bool pf_ok = false;   // the prefetched bytes serve the next step if it is a shift by 8 from where they were fetched
#ifdef BA_TIMING
    unsigned long long prof[48] = {};
#else
    unsigned long long* prof = nullptr;
#endif
    typedef const __attribute__((address_space(4))) unsigned long long* ConstU64;
    __device__ __forceinline__ static unsigned long long load_cols(const uint8_t* p) {   // 8 sequence bytes at a 4-byte aligned position
        return *(ConstU64)(uintptr_t)p;
    }
    __device__ __forceinline__ void prefetch_seq(uint32_t si, uint32_t sj, uint32_t B) {
        const int lane = lane_id();
        pf_qv = *(const unsigned short*)(q + si + 2 * lane);
        pf_rv = *(const unsigned short*)(r + sj + 2 * lane);
        pf_qc = load_cols(q + si + B);
        pf_rc = load_cols(r + sj + B);
        pf_ok = true;
    }
    // While the block is a single chunk (<= 128 cells) the whole checkpoint is four VGPRs (D_col, C_col, D_row, R_row order):
    // a fast run parks its border registers here instead of storing to memory.
    int ck_reg[4] = {0, 0, 0, 0};
    bool ck_in_regs = false;
    bool chain = false;   // TRACE: rectangles of speculative (untraced) grows are on the trace stack, see run()

    // A shift step is taken by the register path unless its columns could break early at the end of the matrix
    // (scan_block.rs:1216-1224; never with X-drop): those go through place_rect, which implements the break.
    // (the break needs both: vectors that reach past the end of their sequence and a column at or past the end of the other)
    __device__ __forceinline__ static bool fast_eligible(uint32_t ri, uint32_t B, uint32_t lenV, uint32_t rj, uint32_t lenC) {
        return XDROP || ri + B <= lenV || rj + STEP <= lenC;
    }

    enum { RUN_EXIT_POST = 0, RUN_EXIT_TOP = 1, RUN_EXIT_FATAL = 2 };
    // A run of plain shift steps of a single-chunk block, borders in registers (see fast_rect). Entered with the first step
    // already set up by the driver loop (off, off_add, corner); restates scan_block.rs:332-558 for the steps after which
    // nothing but another shift follows, and hands the first step that needs more (exit = POST: its fill is done, `dir`,
    // the returned maximum and `fo` describe it) or the first one it cannot take (exit = TOP) back to the driver loop.
    struct RunState {   // the driver's loop-carried scalars, passed and returned by value (all wave-uniform)
        uint32_t si, sj; int dir, prev_dir, off, prev_off, off_max, off_add, best_max; uint32_t y_drop_iter; int x_drop_iter, D_corner;
        uint32_t step_budget; int run_exit; Best cur; FastOut fo;
    };
    __device__ __forceinline__ RunState fast_run(const RunState in, int corner, const uint32_t B, const uint32_t min_size, const uint32_t max_size, const bool one_step = false) {
        uint32_t si = in.si, sj = in.sj; int dir = in.dir, prev_dir = in.prev_dir, off = in.off, prev_off = in.prev_off, off_max = in.off_max;
        int off_add = in.off_add, best_max = in.best_max; uint32_t y_drop_iter = in.y_drop_iter; int x_drop_iter = in.x_drop_iter, D_corner = in.D_corner;
        uint32_t step_budget = in.step_budget; int run_exit = RUN_EXIT_POST; FastOut fo{};
        constexpr int PR_DIST = (int)(lds_array_bytes_h(kBig ? 128u : (uint32_t)PMAX * 128u) / 2);   // D_row -> R_row and D_col -> C_col, in entries
        const int lane = lane_id();
        const int nl = (int)(B >> 1);
        int Dcol, Ccol, Drow, Rrow;
        lds_sync();
        Dcol = *(const int*)(L.D_col + 2 * lane); Ccol = *(const int*)(L.C_col + 2 * lane);
        Drow = *(const int*)(L.D_row + 2 * lane); Rrow = *(const int*)(L.R_row + 2 * lane);
        lds_sync();
        // latest values of the driver state that only improving steps write (parked on the way out, if any step improved)
        bool improved = false;
        int n_best_i = 0, n_best_j = 0, n_ck_i = 0, n_ck_j = 0, n_ck_off = 0, n_ck_tt = 0, n_ck_nb = 0;
        Best cur{0, 0, 0};
        for (;;) {
            const bool right = dir == DIR_RIGHT;
            const uint8_t* seqV = right ? q : r; const uint8_t* seqC = right ? r : q;
            const uint32_t ri = right ? si : sj, rj = (right ? sj : si) + B - STEP;
            // sequence bytes: prefetched by the previous step if it predicted this position, else fetched now
            int vc; unsigned long long cb;
            if (pf_ok) { vc = right ? pf_qv : pf_rv; cb = right ? pf_rc : pf_qc; }
            else {
                vc = 2 * lane < (int)B ? (int)*(const unsigned short*)(seqV + ri + 2 * lane) : 0;
                cb = load_cols(seqC + rj);
            }
            // pin the consumption of the old prefetch here: the memory counter is in-order, so the next prefetch must be
            // issued only after the wait for the previous one
            asm volatile("" : "+v"(vc));
            const uint32_t tb = trace_top;
            if (TRACE) {   // add_block(i, j, width, height, right) in matrix orientation (scan_block.rs:154,204)
                if (right) add_block(ri, rj, STEP, B, true);
                else add_block(rj, ri, B, STEP, false);
                if (status) { run_exit = RUN_EXIT_FATAL; break; }
            }
            uint32_t* tout = TRACE ? trace + tb : nullptr;
            const int loc_thr = best_max - off + ZERO;   // a rectangle maximum above this raises the best score: its location is needed
            // (the special-mode kernels: LOCAL_START's floor -- the relative zero of this step's offset -- and zero mask; FREE_QUERY_START_GAPS steps get here
            // only below row 0 of the matrix, where they are plain steps: see run())
            const bool sp_local = SPECIAL && (h_flags & F_LOCAL);
            const int rz2 = SPECIAL ? (sp_local ? splat(clamp16(-off + ZERO)) : (int)0x80008000u) : 0;
#define BA_FAST(LANES) do { if (right) fast_rect<KIND, TRACE, XDROP, LANES, PR_DIST, SPECIAL>(L.table, fc, Dcol, Ccol, Drow, Rrow, L.D_row, L.D_col, vc & 0xff, (vc >> 8) & 0xff, cb, nl, corner, off_add, loc_thr, tout, fo, rz2, sp_local); \
                           else fast_rect<KIND, TRACE, XDROP, LANES, PR_DIST, SPECIAL>(L.table, fc, Drow, Rrow, Dcol, Ccol, L.D_col, L.D_row, vc & 0xff, (vc >> 8) & 0xff, cb, nl, corner, off_add, loc_thr, tout, fo, rz2, sp_local); } while (0)
            if (B == 128) BA_FAST(64); else if (B == 32) BA_FAST(16); else if (B == 64) BA_FAST(32); else BA_FAST(0);
#undef BA_FAST
            cells += (unsigned long long)(STEP * B);
            prefetch_seq(si, sj, B);   // for the next step, behind this step's stores
            cur = Best{fo.mx, fo.row, fo.col};

            // ---- what does this step call for? (no driver state has been touched yet)
            const int right_max = right ? fo.act_max8 : fo.pas_max8, down_max = right ? fo.pas_max8 : fo.act_max8;
            const int mx = fo.mx;
            const int new_off_max = off + mx - ZERO;
            const bool improve = new_off_max > best_max;
            const uint32_t new_y = improve ? 0u : y_drop_iter + 1;
            const bool q_out = si + B > qlen, r_out = sj + B > rlen;
            bool leave = q_out && r_out;                                                                      // end of the matrix
            if (XDROP) leave = leave || (!improve && new_off_max < best_max - h_x_drop && x_drop_iter >= 1);   // X-drop termination
            if (!q_out && !r_out) {
                leave = leave || (2 * B <= max_size && new_y > B / STEP - 1);                                  // grow
                if (B > min_size && new_y == 0) {                                                              // shrink (scan_block.rs:505-513)
                    const s16x2 a = as_s(__builtin_amdgcn_readlane(Drow, nl - 1)), b = as_s(__builtin_amdgcn_readlane(Dcol, nl - 1));
                    leave = leave || max(max((int)a.x, (int)a.y), max((int)b.x, (int)b.y)) >= mx;
                }
            }
            if (TRACE && chain && improve) leave = true;   // untraced grow rectangles below: the generic post-processing rolls back
            if (leave) { run_exit = RUN_EXIT_POST; break; }

            // ---- a plain step: scan_block.rs:332-450 without the grow / shrink / termination branches
            off_max = new_off_max; y_drop_iter = new_y; prev_dir = dir; D_corner = fo.corner_new;
            if (improve) {
                improved = true;
                if (XDROP) {   // scan_block.rs:370-404
                    if (right) { n_best_i = (int)(si + (uint32_t)cur.row); n_best_j = (int)(sj + (B - STEP) + (uint32_t)cur.col); }
                    else { n_best_i = (int)(si + (B - STEP) + (uint32_t)cur.col); n_best_j = (int)(sj + (uint32_t)cur.row); }
                }
                if (B < max_size) {   // checkpoint (scan_block.rs:406-427): four register copies
                    n_ck_i = (int)si; n_ck_j = (int)sj; n_ck_off = off;
                    ck_reg[0] = Dcol; ck_reg[1] = Ccol; ck_reg[2] = Drow; ck_reg[3] = Rrow; ck_in_regs = true;
                    if (TRACE) { n_ck_tt = (int)trace_top; n_ck_nb = (int)nblocks; }
                }
                best_max = off_max;
            }
            if (XDROP) {
                if (off_max < best_max - h_x_drop) x_drop_iter++;   // (the second consecutive hit left the run above; X_DROP_ITER = 2)
                else x_drop_iter = 0;
            }
            const bool go_down = r_out || (!q_out && down_max > right_max);   // forced at the matrix edge, else greedy (ties -> right)
            si += go_down ? (uint32_t)STEP : 0u; sj += go_down ? 0u : (uint32_t)STEP;
            dir = go_down ? DIR_DOWN : DIR_RIGHT;

            // ---- set up the next step (what the top of the driver loop does)
            // (one_step: the multi-pair kernel takes the pair back into its slot for the plain steps that follow)
            if (one_step || !fast_eligible(dir == DIR_RIGHT ? si : sj, B, dir == DIR_RIGHT ? qlen : rlen, (dir == DIR_RIGHT ? sj : si) + B - STEP, dir == DIR_RIGHT ? rlen : qlen)) { run_exit = RUN_EXIT_TOP; break; }
            if (--step_budget == 0) { status |= ST_WATCHDOG; run_exit = RUN_EXIT_FATAL; break; }
#ifdef BA_TIMING
            prof[16]++;
#endif
            prev_off = off; off = off_max;
            off_add = clamp16(prev_off - off);
            corner = prev_dir != dir ? (int)as_s(adds(splat(D_corner), splat(off_add))).x : 0;
        }
        // borders back to LDS for the generic code
        lds_sync();
        if (lane < nl) {
            *(int*)(L.D_col + 2 * lane) = Dcol; *(int*)(L.C_col + 2 * lane) = Ccol;
            *(int*)(L.D_row + 2 * lane) = Drow; *(int*)(L.R_row + 2 * lane) = Rrow;
        }
        lds_sync();
        if (improved) {
            if (XDROP) { park<5>(parked, n_best_i); park<6>(parked, n_best_j); }
            if (B < max_size) {
                park<0>(parked, n_ck_i); park<1>(parked, n_ck_j); park<2>(parked, n_ck_off);
                if (TRACE) { park<3>(parked, n_ck_tt); park<4>(parked, n_ck_nb); }
            }
        }
        RunState out;
        out.si = si; out.sj = sj; out.dir = dir; out.prev_dir = prev_dir; out.off = off; out.prev_off = prev_off; out.off_max = off_max;
        out.off_add = off_add; out.best_max = best_max; out.y_drop_iter = y_drop_iter; out.x_drop_iter = x_drop_iter; out.D_corner = D_corner;
        out.step_budget = step_budget; out.run_exit = run_exit; out.cur = cur; out.fo = fo;
        return out;
    }

    __device__ __forceinline__ Aligner(const BatchParams& b, const WaveLds& L_, const FillConsts& fc_) : L(L_), fc(fc_) {
        h_flags = b.flags; h_max_size = b.max_size; h_x_drop = b.x_drop;
    }

    __device__ __forceinline__ void add_block(uint32_t i, uint32_t j, uint32_t w, uint32_t h, bool right, bool untraced = false) {
        if (nblocks >= (uint32_t)unpark<13>(parked)) { status |= ST_BLOCKS_OVERFLOW; return; }
        // LOCAL_START: every trace word is followed by its cells' zero-mask word (place_rect)
        const uint32_t words = (w * h / 8) * ((SPECIAL && (h_flags & F_LOCAL)) ? 2u : 1u);
        if (trace_top + words + 64 > (uint32_t)unpark<12>(parked)) { status |= ST_TRACE_OVERFLOW; return; }   // (64 words of slack stay free: see the host's trace_stride)
        {   // every lane stores the same 16 bytes to the same address (one transaction): no exec-mask region per step
            BlockRec br; br.i = i; br.j = j; br.h = (uint16_t)h; br.w = (uint16_t)w;
            br.trace_base = trace_top | (right ? 0x80000000u : 0u) | (untraced ? 0x40000000u : 0u);   // (untraced: space reserved, flags not computed)
            blocks[nblocks] = br;
        }
        nblocks++;
        trace_top += words;
    }

    // The block borders of the best-so-far position (scan_block.rs:406-427) are parked in this wave's slot of a global
    // scratch arena (L2-resident, 8 * max_size bytes): they are written on every improving step but read back only
    // when the block grows, and keeping them out of LDS doubles the number of resident waves per CU.
    __device__ __forceinline__ void save_ckpt_borders(uint32_t n) {
        ck_in_regs = false;
        lds_sync();
        const uint32_t ms = h_max_size;
        for (uint32_t k = 2 * lane_id(); k < n; k += 128) {
            *(int*)(ckpt + k) = *(const int*)(L.D_col + k);
            *(int*)(ckpt + ms + k) = *(const int*)(L.C_col + k);
            *(int*)(ckpt + 2 * ms + k) = *(const int*)(L.D_row + k);
            *(int*)(ckpt + 3 * ms + k) = *(const int*)(L.R_row + k);
        }
    }
    // The checkpoint borders to (save = true) or from the second half of this wave's checkpoint arena: the base of a chain of
    // speculative grows.
    __device__ __forceinline__ void copy_ckpt(bool save, uint32_t n) {
        const uint32_t ms = h_max_size;
        short* second = ckpt + 4 * ms;
        if (save && ck_in_regs) {
            const uint32_t k = 2 * lane_id();
            if (k < n) for (int a = 0; a < 4; a++) *(int*)(second + a * ms + k) = ck_reg[a];
            return;
        }
        for (uint32_t k = 2 * lane_id(); k < n; k += 128)
            for (int a = 0; a < 4; a++) {
                if (save) *(int*)(second + a * ms + k) = ckpt_load(ckpt + a * ms + k);
                else *(int*)(ckpt + a * ms + k) = ckpt_load(second + a * ms + k);
            }
        if (!save) ck_in_regs = false;
    }
    __device__ __forceinline__ void restore_ckpt_borders(uint32_t n) {
        if (ck_in_regs) {   // checkpoint of a single-chunk block: registers -> LDS
            const uint32_t k = 2 * lane_id();
            lds_sync();
            if (k < n) {
                *(int*)(L.D_col + k) = ck_reg[0]; *(int*)(L.C_col + k) = ck_reg[1];
                *(int*)(L.D_row + k) = ck_reg[2]; *(int*)(L.R_row + k) = ck_reg[3];
            }
            lds_sync();
            return;
        }
        // The wave reads back its own checkpoint stores: loads that bypass this CU's L1 (agent-scope relaxed = `sc1`, served by
        // L2, where the wave's earlier stores to the same addresses have arrived in program order) instead of an agent-scope
        // fence -- a fence writes back and invalidates for the whole XCD, and short pairs grow often.
        const uint32_t ms = h_max_size;
        for (uint32_t k = 2 * lane_id(); k < n; k += 128) {
            *(int*)(L.D_col + k) = ckpt_load(ckpt + k);
            *(int*)(L.C_col + k) = ckpt_load(ckpt + ms + k);
            *(int*)(L.D_row + k) = ckpt_load(ckpt + 2 * ms + k);
            *(int*)(L.R_row + k) = ckpt_load(ckpt + 3 * ms + k);
        }
        lds_sync();
    }
    __device__ __forceinline__ static int ckpt_load(const short* p) { return __hip_atomic_load((const int*)p, BA_RLX_AGENT); }
    __device__ __forceinline__ static short ckpt_load16(const short* p) { return __hip_atomic_load(p, BA_RLX_AGENT); }   // (past the L1: the wave reads back its own stores)

    // Wait until this wave's next trace slot has been walked (its previous tenant's traceback is done). The wait never gives up
    // (round 4: no pair comes back failed for a slow traceback side): whenever nobody has claimed a hand-off for a few milliseconds the
    // wave walks the ring's head entry itself and looks again -- the entries ahead of its own slots are either unclaimed (it walks
    // them) or claimed by a resident wave that is walking them.
    // tb_lds: this wave's LDS region (free between pairs): see traceback_help_one.
    __device__ __forceinline__ void acquire_slot(uint32_t slot, const BatchParams& bp, unsigned char* tb_lds) {
        uint32_t seen = 0, idle = 0;
        for (;;) {
            uint32_t f = 0, head = 0;
            if (is_lane(0)) {
                f = __hip_atomic_load(coldp()->slot_free + slot, BA_RLX_AGENT);
                head = __hip_atomic_load(coldp()->tb_ctrl + 32, BA_RLX_AGENT);
            }
            if (uni((int)f)) {
                if (is_lane(0)) __hip_atomic_store(coldp()->slot_free + slot, 0u, BA_RLX_AGENT);
                return;
            }
            head = (uint32_t)uni((int)head);
            if (head != seen) { seen = head; idle = 0; }
            // nobody has taken a traceback for a few milliseconds: walk one here (a launch whose traceback waves are not resident)
            else if ((++idle & 2047u) == 0 && traceback_help_one<SPECIAL ? (int)TB_LANE_BYTES_LOC : (int)TB_LANE_BYTES>(bp, SPECIAL ? ~0u : (uint32_t)F_CIGAR_EQ, tb_lds)) idle = 0;
            __builtin_amdgcn_s_sleep(64);
        }
    }
    // Publish a finished trace stack: plain stores -> release fence -> drained -> queue entry (guide G16 flag form).
    __device__ __forceinline__ void hand_off(uint32_t slot, uint32_t pair, uint32_t end_i, uint32_t end_j, bool null_task = false) {
        if (is_lane(0)) {
            if (!null_task) coldp()->slot_info[slot] = SlotInfo{pair, nblocks, end_i, end_j};
            coldp()->status[pair] = status;
        }
#ifndef BA_X_NOFENCE   // (development, results invalid: what the release costs)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (is_lane(0)) {
            const uint32_t tpos = __hip_atomic_fetch_add(coldp()->tb_ctrl, 1u, BA_RLX_AGENT) & coldp()->tb_qmask;
            while (__hip_atomic_load(coldp()->tb_queue + tpos, BA_RLX_AGENT) != 0) __builtin_amdgcn_s_sleep(8);
            __hip_atomic_store(coldp()->tb_queue + tpos, null_task ? (0x80000000u | pair) : slot + 1, BA_RLX_AGENT);
        }
    }

    // (always inlined: as a real call the Aligner object and everything it references would live in scratch memory)
    // mmode / min / allow_quad / forced_in: the multi-pair kernel's entry (MM_*): resume from `min` (borders already in LDS, checkpoint
    // in ck_reg: see import_slot), leave again -- returning the state, exited = 1 -- once the next step is a plain shift step at MQ_B
    // cells (allow_quad), after taking the step at the top here whatever it is (forced_in: the slot rolled it back)
    __device__ __forceinline__ RunOut run(uint32_t pair_in, uint32_t slot_in, bool batch_traceback, const PairCont* resume = nullptr, const int mmode_in = MM_NONE,
                                          const RunOut min = RunOut{}, const bool allow_quad = false, const bool forced_in = false, const bool walk_now = false) {
        RunOut mout{}; bool forced = MULTI && forced_in;
        const int mmode = MULTI ? mmode_in : (int)MM_NONE;
        BA_TSTAMP(tq0);
        q = coldp()->pool + coldp()->q_off[pair_in]; r = coldp()->pool + coldp()->r_off[pair_in];
        qlen = coldp()->q_len[pair_in]; rlen = coldp()->r_len[pair_in];
        const uint32_t min_size = coldp()->min_size, max_size = h_max_size;
        ProfileView pv{};
        if constexpr (KIND == KIND_PROFILE) {   // r points at the pair's AAProfile image (ba_params.h)
            pv.P = profile_positions(rlen, max_size);
            pv.pos_aa = (const signed char*)r;
            pv.aa_pos = (const short*)(r + (uint64_t)pv.P * 32);
            pv.goC = pv.aa_pos + (uint64_t)pv.P * 32; pv.clC = pv.goC + pv.P; pv.goR = pv.clC + pv.P;
        }
        const uint32_t special = SPECIAL ? h_flags & (F_LOCAL | F_FQS | F_FQE) : 0u;
        const bool FQE = SPECIAL && (h_flags & F_FQE);
        FqeOut fq{0, 0};
        // scratch reset (scan_block.rs:1322-1339): borders to MIN = 0. The checkpoint copies need no reset: they are
        // always written (first iteration is a grow, scan_block.rs:313-322) before they can be read.
        // (round 6: a launch of the 2048-cell class may carry pairs whose block range ends above it -- the host's bet that few of them grow that far,
        // ba_host.cpp batch_build; CLASS_CAP: what this kernel's border arrays hold)
        // (only the 2048-cell class is compiled with it: every register of the smaller classes' drivers is spoken for -- the two lines in all classes moved
        // k_multi<8>'s solo driver from 315 to 336 SGPR spills and config 3 from 158.2 to 161.1 ms)
        constexpr bool CLASS_BET = !kBig && PMAX >= 16;
        constexpr uint32_t CLASS_CAP = (uint32_t)PMAX * 128u;
        uint32_t fill_n = max_size;
        if constexpr (CLASS_BET) fill_n = max_size < CLASS_CAP ? max_size : CLASS_CAP;
        if (!MULTI || mmode != MM_RESUME) { lds_fill0(L.D_col, fill_n); lds_fill0(L.C_col, fill_n); lds_fill0(L.D_row, fill_n); lds_fill0(L.R_row, fill_n); }
        short* temp1 = L.misc + 16; short* temp2 = L.misc + 32;
        lds_fill0(temp1, 32);

        uint32_t si = 0, sj = 0;
        int best_max = 0;
        parked = 0;
        park<7>(parked, (int)pair_in); park<8>(parked, (int)slot_in);   // not needed again before the pair is done
        if (TRACE) {   // room for this pair's trace words and rectangle records: the ring slot's, or the pair's own region
            uint64_t tcap = coldp()->trace_stride, bcap = coldp()->blocks_stride;
            if (coldp()->trace_off) {
                tcap = coldp()->trace_off[pair_in + 1] - coldp()->trace_off[pair_in];
                bcap = coldp()->blocks_off[pair_in + 1] - coldp()->blocks_off[pair_in];
            }
            park<12>(parked, (int)(tcap < 0x7fffffffull ? tcap : 0x7fffffffull)); park<13>(parked, (int)(bcap < 0x7fffffffull ? bcap : 0x7fffffffull));
        }
        int prev_dir = DIR_GROW, dir = DIR_GROW;
        uint32_t prev_size = 0, block_size = min_size;
        int off = 0, prev_off = 0, off_max = 0;
        uint32_t y_drop_iter = 0; int x_drop_iter = 0;
        int D_corner = 0;
        int gphase = 0;                 // 0 = first rectangle of this driver step, 1 = second rectangle of a grow
        int off_add = 0;

        uint32_t step_budget = 64u * ((qlen + rlen) / STEP + 64u);   // watchdog: far above any legal run
#ifdef BA_TIMING
        uint32_t steps = 0;
#endif
        if (resume) {   // a pair that comes back from the small-block kernel: its state at the top of the loop
            const int lane = lane_id();
            // The record may have been written while this kernel was already running (queue mode, see k_align): loads that bypass
            // this CU's L1 (agent-scope relaxed = sc1, served by L2, where the producer's release has put the record) -- and never
            // the scalar cache, whose lines may predate the record: one vector load of the 24 header words, lane k holding word k.
            const int hdr = lane < 24 ? (int)__hip_atomic_load((const uint32_t*)resume + lane, BA_RLX_AGENT) : 0;
#define BA_HDR(k) __builtin_amdgcn_readlane(hdr, k)
            si = (uint32_t)BA_HDR(1); sj = (uint32_t)BA_HDR(2); dir = BA_HDR(3); prev_dir = BA_HDR(4); off = BA_HDR(5); off_max = BA_HDR(6);
            best_max = BA_HDR(7); y_drop_iter = (uint32_t)BA_HDR(8); x_drop_iter = BA_HDR(9); D_corner = BA_HDR(10);
            cells = (unsigned long long)(uint32_t)BA_HDR(16) | ((unsigned long long)(uint32_t)BA_HDR(17) << 32); step_budget = (uint32_t)BA_HDR(18);
            park<5>(parked, BA_HDR(11)); park<6>(parked, BA_HDR(12));
            park<0>(parked, BA_HDR(13)); park<1>(parked, BA_HDR(14)); park<2>(parked, BA_HDR(15));
            if (TRACE) {
                trace_top = (uint32_t)BA_HDR(19); nblocks = (uint32_t)BA_HDR(20); status = (uint32_t)BA_HDR(23);
                park<3>(parked, BA_HDR(21)); park<4>(parked, BA_HDR(22));
            }
#undef BA_HDR
            if (lane < 16) {
#define BA_REC(field, k) (int)__hip_atomic_load(&resume->field[k][lane], BA_RLX_AGENT)
                *(int*)(L.D_col + 2 * lane) = BA_REC(borders, 0); *(int*)(L.C_col + 2 * lane) = BA_REC(borders, 1);
                *(int*)(L.D_row + 2 * lane) = BA_REC(borders, 2); *(int*)(L.R_row + 2 * lane) = BA_REC(borders, 3);
                ck_reg[0] = BA_REC(ckpt, 0); ck_reg[1] = BA_REC(ckpt, 1); ck_reg[2] = BA_REC(ckpt, 2); ck_reg[3] = BA_REC(ckpt, 3);
#undef BA_REC
            }
            ck_in_regs = true;
            lds_sync();
        }
        if constexpr (MULTI) if (mmode == MM_RESUME) {   // a pair that comes back from its slot of the multi-pair kernel (ba_multi.hpp): borders in LDS, checkpoint in ck_reg
            const PairState& st = min;
            si = st.si; sj = st.sj; dir = st.dir; prev_dir = st.prev_dir; off = st.off; off_max = st.off_max; best_max = st.best_max;
            y_drop_iter = st.y_drop_iter; x_drop_iter = st.x_drop_iter; D_corner = st.D_corner; cells = st.cells; step_budget = st.step_budget;
            park<5>(parked, (int)st.best_i); park<6>(parked, (int)st.best_j);
            park<0>(parked, (int)st.ck_i); park<1>(parked, (int)st.ck_j); park<2>(parked, st.ck_off);
            if (TRACE) { trace_top = st.trace_top; nblocks = st.nblocks; status = st.status; park<3>(parked, (int)st.ck_tt); park<4>(parked, (int)st.ck_nb); }
        }
        // speculative grows (see the grow transition below): the checkpoint the chain started from
        uint32_t ub_size = 0, ub_tt = 0, ub_nb = 0, ub_budget = 0; int ub_xiter = 0; unsigned long long ub_cells = 0;
        bool no_spec = false;
        chain = false;
        auto rollback = [&]() {   // back to the chain's base checkpoint; the grow it was about to take is taken again, traced
            block_size = ub_size; x_drop_iter = ub_xiter; cells = ub_cells; step_budget = ub_budget;
            copy_ckpt(false, block_size);
            chain = false; no_spec = true;
            prev_size = block_size; block_size *= 2; dir = DIR_GROW; gphase = 0;
            si = (uint32_t)unpark<0>(parked); sj = (uint32_t)unpark<1>(parked); off = unpark<2>(parked);
            restore_ckpt_borders(prev_size);
            trace_top = ub_tt; nblocks = ub_nb; park<3>(parked, (int)ub_tt); park<4>(parked, (int)ub_nb);
            y_drop_iter = 0; pf_ok = false;
        };
        BA_TSTAMP(tr0);
        BA_TADD(prof, 47, tq0, tr0);
        for (;;) {
            BA_TSTAMP(ts0);
            // ---- set up the next rectangle (one fill call site for all four kinds of rectangle)
            const uint8_t* seqV; const uint8_t* seqC; uint32_t lenV, lenC, ri, rj, rw, rh;
            short *Dc, *Cc, *Dr, *Rr; int corner = 0; bool right;
            if (gphase == 0) {
                if (--step_budget == 0) { status |= ST_WATCHDOG; break; }
#ifdef BA_TIMING
                steps++;
#endif
                prev_off = off;
            }
            if (dir == DIR_RIGHT) {
                off = off_max;
                off_add = clamp16(prev_off - off);
                seqV = q; seqC = r; lenV = qlen; lenC = rlen; ri = si; rj = sj + block_size - STEP; rw = STEP; rh = block_size;
                Dc = L.D_col; Cc = L.C_col; Dr = temp1; Rr = temp2; right = true;
                corner = prev_dir == DIR_DOWN ? (int)as_s(adds(splat(D_corner), splat(off_add))).x : 0;
            } else if (dir == DIR_DOWN) {
                off = off_max;
                off_add = clamp16(prev_off - off);
                seqV = r; seqC = q; lenV = rlen; lenC = qlen; ri = sj; rj = si + block_size - STEP; rw = STEP; rh = block_size;
                Dc = L.D_row; Cc = L.R_row; Dr = temp1; Rr = temp2; right = false;
                corner = prev_dir == DIR_RIGHT ? (int)as_s(adds(splat(D_corner), splat(off_add))).x : 0;
            } else if (gphase == 0) {   // grow, rectangle below the old block (scan_block.rs:260-278)
                D_corner = 0; off_add = 0;
                seqV = r; seqC = q; lenV = rlen; lenC = qlen; ri = sj; rj = si + prev_size; rw = block_size - prev_size; rh = prev_size;
                Dc = L.D_row; Cc = L.R_row; Dr = L.D_col + prev_size; Rr = L.C_col + prev_size; right = false;
            } else {                    // grow, rectangle right of the old block, full new height (scan_block.rs:287-305)
                off_add = 0;
                seqV = q; seqC = r; lenV = qlen; lenC = rlen; ri = si; rj = sj + prev_size; rw = block_size - prev_size; rh = block_size;
                Dc = L.D_col; Cc = L.C_col; Dr = L.D_row + prev_size; Rr = L.R_row + prev_size; right = true;
            }
            // bit 8: development switch, generic path only; profiles and the special modes also take the generic path
            BA_TSTAMP(tsa);
            // (plain: a shift step of a single-chunk block that cannot break early -- what the register path and a slot of the multi-pair kernels take)
            // (LOCAL_START / FREE_QUERY_START_GAPS steps are a slot's too -- k_small's special instantiations --, FREE_QUERY_END_GAPS ones are not; none takes the register path)
            // (round 5, later: the register path takes LOCAL_START steps -- fast_rect<.., SP> -- and FREE_QUERY_START_GAPS steps below row 0 of the matrix,
            // which are plain steps; FREE_QUERY_END_GAPS keeps the generic code)
            // (round 6: a 256-cell slot of k_multi takes plain steps of 256 rows -- two chunks: never the register path)
            const bool plain0 = !kBig && dir != DIR_GROW && (rh <= 128 || (MULTI && SLOT_B > 128 && rh == SLOT_B)) && !(h_flags & 0x100u) && !(special & F_FQE) && fast_eligible(ri, rh, lenV, rj, lenC);
            const bool plain = plain0 && (!special || MULTI);
            const bool fast = plain0 && rh <= 128 && KIND != KIND_PROFILE && !((special & F_FQS) && si == 0);
            if (!fast) pf_ok = false;   // prefetched bytes only serve a shift step that directly follows the one that fetched them
            if constexpr (MULTI) if (mmode != MM_NONE && allow_quad && plain && !forced && block_size == SLOT_B && min_size == SLOT_B && !chain && !no_spec) {
                // ---- the pair goes (back) to its slot of the multi-pair kernel: plain shift steps at MQ_B cells are taken there,
                // four pairs to a wave. The borders are in LDS (cell order); the checkpoint follows them at entries MQ_B .. 2 MQ_B.
                step_budget++;   // (the step this iteration counted has not been taken)
                PairState& st = mout;
                st.si = si; st.sj = sj; st.dir = dir; st.prev_dir = prev_dir; st.off = prev_off; st.off_max = off_max; st.best_max = best_max;
                st.y_drop_iter = y_drop_iter; st.x_drop_iter = x_drop_iter; st.D_corner = D_corner; st.cells = cells; st.step_budget = step_budget;
                st.best_i = (uint32_t)unpark<5>(parked); st.best_j = (uint32_t)unpark<6>(parked);
                st.ck_i = (uint32_t)unpark<0>(parked); st.ck_j = (uint32_t)unpark<1>(parked); st.ck_off = unpark<2>(parked);
                st.ck_tt = (uint32_t)unpark<3>(parked); st.ck_nb = (uint32_t)unpark<4>(parked);
                st.trace_top = trace_top; st.nblocks = nblocks; st.status = status;
                if (SLOT_B < max_size) {
                    const uint32_t ms = h_max_size;
                    lds_sync();
#pragma unroll
                    for (uint32_t k0 = 0; k0 < (SLOT_B > 128u ? SLOT_B : 128u); k0 += 128u) {   // (a checkpoint of more than 128 cells is never in registers)
                    const uint32_t k = k0 + 2 * lane_id();
                    if (SLOT_B >= 128 || k < SLOT_B) {
                    if (SLOT_B <= 128 && ck_in_regs) {
                        *(int*)(L.D_col + SLOT_B + k) = ck_reg[0]; *(int*)(L.C_col + SLOT_B + k) = ck_reg[1];
                        *(int*)(L.D_row + SLOT_B + k) = ck_reg[2]; *(int*)(L.R_row + SLOT_B + k) = ck_reg[3];
                    } else {
                        *(int*)(L.D_col + SLOT_B + k) = ckpt_load(ckpt + k); *(int*)(L.C_col + SLOT_B + k) = ckpt_load(ckpt + ms + k);
                        *(int*)(L.D_row + SLOT_B + k) = ckpt_load(ckpt + 2 * ms + k); *(int*)(L.R_row + SLOT_B + k) = ckpt_load(ckpt + 3 * ms + k);
                    }
                    }
                    }
                    lds_sync();
                }
                mout.exited = 1;
                return mout;
            }
            if (MULTI) forced = false;
            BA_TSTAMP(tsb);
            const uint32_t tb = trace_top;
            const bool spec = TRACE && chain && dir == DIR_GROW;   // this grow step runs without trace flags and location bookkeeping
            if (TRACE && !fast) {   // add_block(i, j, width, height, right) in matrix orientation (scan_block.rs:154,204,257,284)
                if (right) add_block(ri, rj, rw, rh, true, spec);
                else add_block(rj, ri, rh, rw, false, spec);
                if (status) break;
            }
            uint32_t* tout = TRACE ? trace + tb : nullptr;
            const int rz = clamp16(-off + ZERO);
            BA_TSTAMP(ts1);
            BA_TADD(prof, 32, ts0, tsa); BA_TADD(prof, 33, tsa, tsb); BA_TADD(prof, 34, tsb, ts1);
            Best cur{0, 0, 0};
            FastOut fo{}; int run_exit = RUN_EXIT_POST;
            const uint32_t sp = special ? ((h_flags & F_LOCAL) ? SP_LOCAL : 0u) | (((h_flags & F_FQS) && right) ? SP_FQS_ROW0 : 0u) | (FQE ? SP_FQE : 0u) : 0u;
#define BA_PLACE1(N, PD) cur = place_rect<N, KIND, TRACE, XDROP, PD>(L, fc, seqV, seqC, lenV, lenC, ri, rj, rw, rh, Dc, Cc, Dr, Rr, corner, rz, off_add, tout, cells, prof, sp, &fq, &pv)
#define BA_PLACE_T(N) do { if constexpr (KIND == KIND_PROFILE) { if (right) BA_PLACE1(N, 1); else BA_PLACE1(N, 2); } else BA_PLACE1(N, 0); } while (0)
#define BA_PLACE_S(N) cur = place_rect<N, KIND, false, XDROP, 0, false>(L, fc, seqV, seqC, lenV, lenC, ri, rj, rw, rh, Dc, Cc, Dr, Rr, corner, rz, off_add, nullptr, cells, prof, sp, &fq, &pv)
#define BA_PLACE(N) do { if constexpr (TRACE && KIND != KIND_PROFILE && !SPECIAL && !kBig) { if (spec) BA_PLACE_S(N); else BA_PLACE_T(N); } else BA_PLACE_T(N); } while (0)
            if (kBig && rh > BIG_TILE) {
                // ---- row tiles of BIG_TILE cells (TileCtx): the tile above hands its last row over through big_top
                const uint32_t ntiles = rh / BIG_TILE;
                int corner_t[16];                       // D above each tile in the column left of the rectangle: the old border, re-based
                lds_sync();
#pragma unroll
                for (uint32_t t = 1; t < 16; t++)
                    corner_t[t] = t < ntiles ? uni((int)as_s(adds(splat((int)Dc[t * BIG_TILE - 1]), splat(off_add))).x) : 0;
                const bool brk = !XDROP && !(sp & SP_FQE) && (ri + rh > lenV);
                for (uint32_t t = 0; t < ntiles; t++) {
                    TileCtx tc;
                    tc.ch_base = (int)(t * (BIG_TILE / 128)); tc.nch_total = (int)(rh / 128); tc.first = t == 0; tc.last = t + 1 == ntiles;
                    tc.corner0 = 0;
#pragma unroll
                    for (uint32_t u = 1; u < 16; u++) if (u == t) tc.corner0 = corner_t[u];
                    tc.topD = big_top; tc.topR = big_top + big_array_shorts(h_max_size); tc.break_armed = brk;
                    tc.fqR = big_top + 2 * big_array_shorts(h_max_size); tc.fqT = big_top + 3 * big_array_shorts(h_max_size);
                    short* oD = tc.last ? Dr : big_top; short* oR = tc.last ? Rr : big_top + big_array_shorts(h_max_size);
                    Best part;
#define BA_TILE1(PD) part = place_rect<(int)(BIG_TILE / 128), KIND, TRACE, XDROP, PD>(L, fc, seqV, seqC, lenV, lenC, ri + t * BIG_TILE, rj, rw, BIG_TILE, \
                                   Dc + t * BIG_TILE, Cc + t * BIG_TILE, oD, oR, t == 0 ? corner : 0, rz, off_add, tout, cells, prof, sp, nullptr, &pv, &tc)
                    if constexpr (KIND == KIND_PROFILE) { if (right) BA_TILE1(1); else BA_TILE1(2); } else BA_TILE1(0);
#undef BA_TILE1
                    part.row += (int)(t * BIG_TILE);
                    // the rectangle's maximum and its location: largest value, then smallest row % 16, largest column, largest row
                    // (the order in which place_rect resolves ties inside a tile; tiles are multiples of 16 rows)
                    bool take = t == 0 || part.mx > cur.mx;
                    if (XDROP && t > 0 && part.mx == cur.mx) {
                        const int a = part.row & 15, b = cur.row & 15;
                        take = a < b || (a == b && (part.col > cur.col || (part.col == cur.col && part.row > cur.row)));
                    }
                    if (take) cur = part;
                    lds_sync();
                }
                if (sp & SP_FQE) {
                    // FREE_QUERY_END_GAPS (scan_block.rs:1189-1201) from the per-column arrays of the tiles: in the reference's order
                    // (columns outer) a tracked vector of column j records j when its value ties or raises the maximum of everything
                    // before it -- all earlier columns (A, starting at MIN = 0) and the cells above it in column j, which is what fqT[j]
                    // was filtered by. M = the overall maximum; j = the last recording column.
                    const short* fR = big_top + 2 * big_array_shorts(h_max_size); const short* fT = big_top + 3 * big_array_shorts(h_max_size);
                    int A = 0, jbest = 0;
                    for (uint32_t j0 = 0; j0 < rw; j0 += 64) {
                        const uint32_t j = j0 + (uint32_t)lane_id();
                        const bool in = j < rw;
                        const int Rj = in ? (int)ckpt_load16(fR + j) : -32768, Tj = in ? (int)ckpt_load16(fT + j) : -32768;
                        const int incl = wave_prefix_max(Rj);                                  // max of R over this chunk's columns up to and including j
                        int excl = __builtin_amdgcn_update_dpp(-32768, incl, 0x138, 0xf, 0xf, false);   // ... up to j - 1 (wave_shr:1, lane 0: nothing)
                        excl = max(excl, A);
                        const unsigned long long hits = __ballot(in && Tj >= excl);
                        if (hits) jbest = (int)(j0 + 63u - (uint32_t)__builtin_clzll(hits));
                        A = max(A, __builtin_amdgcn_readlane(incl, 63));
                    }
                    fq.M = A; fq.j = jbest;
                }
            }
            else if (fast) {
                if constexpr (KIND != KIND_PROFILE) {
                    // ---- a run of plain shift steps with the four borders in registers (fast_rect). The run ends -- with the last
                    // step's results handed to the generic post-processing below -- as soon as a step calls for anything but
                    // another shift: X-drop termination, the end of the matrix, a grow, a shrink, an early column break.
                    RunState rs;
                    rs.si = si; rs.sj = sj; rs.dir = dir; rs.prev_dir = prev_dir; rs.off = off; rs.prev_off = prev_off; rs.off_max = off_max;
                    rs.off_add = off_add; rs.best_max = best_max; rs.y_drop_iter = y_drop_iter; rs.x_drop_iter = x_drop_iter; rs.D_corner = D_corner;
                    rs.step_budget = step_budget; rs.run_exit = RUN_EXIT_POST; rs.cur = cur; rs.fo = fo;
                    rs = fast_run(rs, corner, block_size, min_size, max_size, MULTI && mmode != MM_NONE && allow_quad && block_size == SLOT_B && min_size == SLOT_B);
                    si = rs.si; sj = rs.sj; dir = rs.dir; prev_dir = rs.prev_dir; off = rs.off; prev_off = rs.prev_off; off_max = rs.off_max;
                    off_add = rs.off_add; best_max = rs.best_max; y_drop_iter = rs.y_drop_iter; x_drop_iter = rs.x_drop_iter; D_corner = rs.D_corner;
                    step_budget = rs.step_budget; run_exit = rs.run_exit; cur = rs.cur; fo = rs.fo;
                    right = dir == DIR_RIGHT;
                    if (run_exit == RUN_EXIT_TOP) continue;     // the next step is not a fast one: back to the top with dir / si / sj set
                    if (run_exit == RUN_EXIT_FATAL) break;
                }
            }
            else if (rh <= 128) BA_PLACE(1);
            else if (PMAX >= 2 && rh == 256) BA_PLACE(2);
            // (512 rows and more: eight cells per lane, see place_rect8 -- sequence kinds without special modes)
#define BA_PLACE8(N8) do { if constexpr (KIND != KIND_PROFILE && !SPECIAL && !kBig) { \
                if (TRACE && spec) cur = place_rect8<N8, KIND, false, XDROP, false>(L, fc, seqV, seqC, lenV, lenC, ri, rj, rw, Dc, Cc, Dr, Rr, corner, rz, off_add, nullptr, cells); \
                else cur = place_rect8<N8, KIND, TRACE, XDROP>(L, fc, seqV, seqC, lenV, lenC, ri, rj, rw, Dc, Cc, Dr, Rr, corner, rz, off_add, tout, cells); } } while (0)
            else if (PMAX >= 4 && rh == 512) { if constexpr (KIND != KIND_PROFILE && !SPECIAL && !kBig) BA_PLACE8(1); else BA_PLACE(4); }
            else if (PMAX >= 8 && rh == 1024) { if constexpr (KIND != KIND_PROFILE && !SPECIAL && !kBig) BA_PLACE8(2); else BA_PLACE(8); }
            else if (PMAX >= 16 && rh == 2048) { if constexpr (KIND != KIND_PROFILE && !SPECIAL && !kBig) BA_PLACE8(4); else BA_PLACE(16); }
#undef BA_PLACE8
#undef BA_PLACE
#undef BA_PLACE_S
#undef BA_PLACE_T
#undef BA_PLACE1
            BA_TSTAMP(ts2);
            BA_TADD(prof, 12, ts0, ts1); BA_TADD(prof, 13, ts1, ts2);
            if (dir == DIR_GROW && gphase == 0) {   // first rectangle of a grow: its maximum waits (parked) for the second one
                park<9>(parked, cur.mx); park<10>(parked, cur.row); park<11>(parked, cur.col);
                gphase = 1; continue;
            }
            gphase = 0;

            // ---- the rest of the driver step
            int right_max, down_max;
            if (fast) {
                right_max = right ? fo.act_max8 : fo.pas_max8;
                down_max = right ? fo.pas_max8 : fo.act_max8;
                D_corner = fo.corner_new;
            } else if (dir == DIR_RIGHT) {
                right_max = lds_prefix_max8(L.D_col);
                D_corner = lds_shift_and_offset(block_size, L.D_row, L.R_row, temp1, temp2, off_add);
                down_max = lds_prefix_max8(L.D_row);
            } else if (dir == DIR_DOWN) {
                down_max = lds_prefix_max8(L.D_row);
                D_corner = lds_shift_and_offset(block_size, L.D_col, L.C_col, temp1, temp2, off_add);
                right_max = lds_prefix_max8(L.D_col);
            } else {
                right_max = lds_prefix_max8(L.D_col);
                down_max = lds_prefix_max8(L.D_row);
                save_ckpt_borders(block_size);
                if (TRACE) { park<3>(parked, (int)trace_top); park<4>(parked, (int)nblocks); }
            }

            BA_TSTAMP(tpa);
            const int this_dir = dir;
            prev_dir = dir;
            // FREE_QUERY_END_GAPS: only the vector lane that holds the last query row counts (scan_block.rs:333-339)
            const bool was_grow = dir == DIR_GROW;
            const int D_max_max = FQE ? fq.M : cur.mx, grow_max = was_grow ? unpark<9>(parked) : 0;   // (0 = MIN: no grow rectangle)
            const int mx = max(D_max_max, grow_max);
            off_max = off + mx - ZERO;
            if (TRACE && chain && off_max > best_max) { rollback(); continue; }   // the path to a new best may cross the untraced rectangles
            if (TRACE && no_spec && off_max > best_max) no_spec = false;         // the re-run has reached the step that called for it
            y_drop_iter++;
            bool grow_no_max = this_dir == DIR_GROW;

            if (off_max > best_max) {
                if (FQE) {     // scan_block.rs:354-368
                    park<5>(parked, (int)qlen);
                    if (this_dir == DIR_RIGHT) park<6>(parked, (int)(sj + (block_size - STEP) + (uint32_t)fq.j));
                    else if (this_dir == DIR_GROW) park<6>(parked, (int)(sj + prev_size + (uint32_t)fq.j));
                    else status |= ST_MODE;   // the reference panics: min block size > query length rules out down steps
                }
                if (XDROP) {   // scan_block.rs:370-404
                    uint32_t bi_, bj_;
                    if (this_dir == DIR_RIGHT) { bi_ = si + cur.row; bj_ = sj + (block_size - STEP) + cur.col; }
                    else if (this_dir == DIR_DOWN) { bi_ = si + (block_size - STEP) + cur.col; bj_ = sj + cur.row; }
                    else if (D_max_max >= grow_max) { bi_ = si + cur.row; bj_ = sj + prev_size + cur.col; }
                    else { bi_ = si + prev_size + (uint32_t)unpark<11>(parked); bj_ = sj + (uint32_t)unpark<10>(parked); }
                    park<5>(parked, (int)bi_); park<6>(parked, (int)bj_);
                }
                if (block_size < max_size) {
                    park<0>(parked, (int)si); park<1>(parked, (int)sj); park<2>(parked, off);
                    save_ckpt_borders(block_size);   // (a fast run has put the borders back into LDS before handing its last step over)
                    if (TRACE) { park<3>(parked, (int)trace_top); park<4>(parked, (int)nblocks); }
                    grow_no_max = false;
                }
                best_max = off_max;
                y_drop_iter = 0;
            }
            BA_TSTAMP(tpb);
            if (XDROP) {
                if (off_max < best_max - h_x_drop) {
                    if (x_drop_iter < 1) x_drop_iter++;   // X_DROP_ITER = 2
                    else break;
                } else x_drop_iter = 0;
            }
            BA_TSTAMP(ts3);
            BA_TADD(prof, 14, ts2, ts3);
            BA_TADD(prof, 35, ts2, tpa); BA_TADD(prof, 36, tpa, tpb); BA_TADD(prof, 37, tpb, ts3);
            if (si + block_size > qlen && sj + block_size > rlen) {
                if (TRACE && chain && !XDROP) { rollback(); continue; }   // a global alignment's path starts at the end cell: everything below must be traced
                break;
            }
            if (sj + block_size > rlen) { si += STEP; dir = DIR_DOWN; continue; }
            if (si + block_size > qlen) { sj += STEP; dir = DIR_RIGHT; continue; }

            const uint32_t next_size = block_size * 2;
            if (next_size <= max_size && (y_drop_iter > block_size / STEP - 1 || grow_no_max)) {
                if constexpr (CLASS_BET) { if (next_size > CLASS_CAP) { status |= ST_CLASS_OVERFLOW; break; } }   // (the pair is run again in the row-tiled class: nothing of this run is kept)
                // (X-drop only: a global alignment's path starts at its end cell, so every chain of untraced grows would be rolled back at the
                // end of the matrix at the latest -- the protein set's growers did all their large-block work twice)
                if (TRACE && XDROP && KIND != KIND_PROFILE && !SPECIAL && !kBig && !chain && !no_spec && !(h_flags & 0x400u)) {
                    // Speculative grows. A grow that does not raise the best score is followed at once by the next grow, and the
                    // sequence that closes an X-drop alignment (128 -> 1024 in config 3: 29 % of its cells) never does: its rectangles
                    // can be on no path. From here on grow steps reserve their trace space but compute neither trace flags nor
                    // the location bookkeeping; should any later step raise the best score (or a global alignment end) while such
                    // rectangles are on the stack, the driver returns to this point -- the checkpoint below -- and repeats
                    // the steps since, traced (results and the cell count are those of the single, traced execution).
                    ub_size = block_size; ub_tt = (uint32_t)unpark<3>(parked); ub_nb = (uint32_t)unpark<4>(parked);
                    ub_xiter = x_drop_iter; ub_cells = cells; ub_budget = step_budget;
                    copy_ckpt(true, block_size);
                    chain = true;
                }
                prev_size = block_size; block_size = next_size; dir = DIR_GROW;
                si = (uint32_t)unpark<0>(parked); sj = (uint32_t)unpark<1>(parked); off = unpark<2>(parked);
                restore_ckpt_borders(prev_size);
                if (TRACE) { trace_top = (uint32_t)unpark<3>(parked); nblocks = (uint32_t)unpark<4>(parked); }
                y_drop_iter = 0;
                continue;
            }
            if (block_size > min_size && y_drop_iter == 0) {   // SHRINK
                const int shrink_max = max(lds_suffix_max2(L.D_row, block_size), lds_suffix_max2(L.D_col, block_size));
                if (shrink_max >= mx) {
                    pf_ok = false;   // the block moves and halves: the prefetched bytes are for the old position
                    prev_dir = DIR_GROW;
                    block_size /= 2;
                    lds_copy(L.D_col, L.D_col + block_size, block_size); lds_copy(L.C_col, L.C_col + block_size, block_size);
                    lds_copy(L.D_row, L.D_row + block_size, block_size); lds_copy(L.R_row, L.R_row + block_size, block_size);
                    si += block_size; sj += block_size;
                    park<0>(parked, (int)si); park<1>(parked, (int)sj); park<2>(parked, off);
                    save_ckpt_borders(block_size);
                    right_max = lds_prefix_max8(L.D_col);
                    down_max = lds_prefix_max8(L.D_row);
                    if (TRACE) { park<3>(parked, (int)trace_top); park<4>(parked, (int)nblocks); }
                    y_drop_iter = 0;
                }
            }
            if (down_max > right_max) { si += STEP; dir = DIR_DOWN; }
            else { sj += STEP; dir = DIR_RIGHT; }
        }

        BA_TSTAMP(tr1);
        BA_TADD(prof, 15, tr0, tr1);
#ifdef BA_TIMING
        prof[16] += steps;
#endif
        if (TRACE && XDROP && chain && coldp()->prof) {
            // The speculative (untraced) rectangles left on the stack: cells that needed neither trace flags nor location bookkeeping. Their
            // sum goes to a counter of the launch (bench.py's ops_required); they all sit behind the record count of the chain's base checkpoint
            // (ub_nb: the chain of grows that closed the alignment and the shift steps between them -- however many). Records read past the L1:
            // this wave's own stores.
            uint32_t sc = 0;
            const uint32_t first = ub_nb < nblocks ? ub_nb : nblocks;
            for (uint32_t k = first + (uint32_t)lane_id(); k < nblocks; k += 64u) {
                const uint32_t* rp = (const uint32_t*)(blocks + k);
                const uint32_t hw = __hip_atomic_load(rp + 2, BA_RLX_AGENT), tb = __hip_atomic_load(rp + 3, BA_RLX_AGENT);
                if (tb & 0x40000000u) sc += (hw & 0xffffu) * (hw >> 16);
            }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) sc += (uint32_t)__shfl_xor((int)sc, d, 64);
            if (is_lane(0) && sc) atomicAdd(coldp()->prof + 60, (unsigned long long)sc);
        }
        const uint32_t pair = (uint32_t)unpark<7>(parked), slot = (uint32_t)unpark<8>(parked);
        int score; uint32_t ri, rj;
        if (XDROP || FQE) { score = best_max; ri = (uint32_t)unpark<5>(parked); rj = (uint32_t)unpark<6>(parked); }
        else {
            lds_sync();
            if (dir == DIR_DOWN) score = off + uni((int)L.D_row[rlen - sj]) - ZERO;
            else score = off + uni((int)L.D_col[qlen - si]) - ZERO;
            ri = qlen; rj = rlen;
        }
        uint32_t ncig = 0;
        if (TRACE && batch_traceback) {
            if (is_lane(0)) {
                coldp()->score[pair] = score; coldp()->query_idx[pair] = ri; coldp()->reference_idx[pair] = rj;
                if (coldp()->cells) coldp()->cells[pair] = cells;
                if (coldp()->nblocks_out) coldp()->nblocks_out[pair] = nblocks;
                if (coldp()->trace_words_out) coldp()->trace_words_out[pair] = trace_top;
            }
            hand_off(slot, pair, ri, rj);
            return mout;
        }
        // pair-slot batches: pairs shorter than inline_len2 leave their paths to k_walk (one pair per lane) instead of this wave's lane 0
        // (walk_now: k_small's longest pairs, run one to a wave at the start of the launch -- their walks, the batch's longest, overlap with the fill)
        const bool walk_later = TRACE && coldp()->trace_off && qlen + rlen < coldp()->inline_len2 && !(MULTI && walk_now && !SPECIAL);   // (the special modes' inline walk is one lane's: always k_walk's)
        if (TRACE && coldp()->cig_ops && !status && !walk_later) {
            // the trace words and rectangle list were written with plain stores and this slot's arena was read
            // during the previous pair's traceback: drain the stores and drop stale L1 lines before reading back
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            if constexpr (SPECIAL) {
                if (is_lane(0)) {
                    uint32_t st = 0;
                    ncig = traceback(blocks, nblocks, trace, ri, rj, q, r, h_flags, coldp()->cig_ops, coldp()->cig_off[pair], coldp()->cig_off[pair + 1], &st);
                    status |= st;
                }
            } else {
                // the whole wave on this one path, out of the border arrays' LDS space (the pair is finished: nothing in it is live)
                uint32_t st = 0;
                lds_sync();
                ncig = walk_wave<MULTI>(blocks, nblocks, trace, ri, rj, q, r, (h_flags & F_CIGAR_EQ) != 0, coldp()->cig_ops, coldp()->cig_off[pair],
                                        coldp()->cig_off[pair + 1], &st, (uint32_t*)L.D_col, lds_array_bytes_h(kBig ? 128u : (uint32_t)PMAX * 128u));
                lds_sync();
                status |= st;
            }
        }
        if (is_lane(0)) {
            coldp()->score[pair] = score; coldp()->query_idx[pair] = ri; coldp()->reference_idx[pair] = rj;
            if (coldp()->cig_len) coldp()->cig_len[pair] = ncig;
            if (coldp()->cells) coldp()->cells[pair] = cells;
            if (coldp()->status) coldp()->status[pair] = status;
            if (coldp()->nblocks_out) coldp()->nblocks_out[pair] = nblocks;
            if (coldp()->trace_words_out) coldp()->trace_words_out[pair] = trace_top;
            if (coldp()->slot_out) coldp()->slot_out[pair] = slot;
            if (TRACE && coldp()->trace_off) coldp()->slot_info[pair] = SlotInfo{pair, walk_later ? nblocks : ~0u, ri, rj};   // pair-slot batches: k_walk's task (~0: nothing left to walk)
        }
        return mout;
    }

    // k_multi, before run(MM_RESUME): a slot's registers (8 cells per lane, cell order; lanes 0 .. 15 hold them: reg[0..3] D_col, [4..7]
    // C_col, [8..11] D_row, [12..15] R_row, ckr[] the checkpoint likewise) become this wave's LDS borders and ck_reg. ck_pre: the
    // checkpoint registers hold the state BEFORE the last improving step (the slot keeps neither the location of a step's maximum
    // nor the borders after it): that step -- direction ck_dir at block position (ck_i, ck_j) -- is taken once more here, for the
    // location (returned in best_i / best_j; scan_block.rs:370-404) and the borders the reference's checkpoint holds (406-427).
    __device__ __forceinline__ void import_slot(const int (&reg)[16], const int (&ckr)[16], uint32_t pair_in, bool have_ck, bool ck_pre, int ck_dir, int ck_offadd, int ck_corner,
                                                uint32_t ck_i, uint32_t ck_j, uint32_t& best_i, uint32_t& best_j, int ck_off = 0) {
        constexpr int SLOT_LANES = (int)SLOT_B / 8;   // the slot's lanes (8 cells each): 16 (k_multi) or 4 (k_small)
        const int lane = lane_id(), l8 = 8 * (lane & (SLOT_LANES - 1));
        const bool mine = lane < SLOT_LANES;
        q = coldp()->pool + coldp()->q_off[pair_in]; r = coldp()->pool + coldp()->r_off[pair_in];
        if (have_ck) {
            lds_sync();
            if (mine) {
                *(int4*)(L.D_col + l8) = int4{ckr[0], ckr[1], ckr[2], ckr[3]}; *(int4*)(L.C_col + l8) = int4{ckr[4], ckr[5], ckr[6], ckr[7]};
                *(int4*)(L.D_row + l8) = int4{ckr[8], ckr[9], ckr[10], ckr[11]}; *(int4*)(L.R_row + l8) = int4{ckr[12], ckr[13], ckr[14], ckr[15]};
            }
            lds_sync();
            constexpr bool WIDE = SLOT_B > 128;   // (round 6) a 256-cell slot: two chunks -- the checkpoint lives in the arena, its step goes through the generic rectangle code
            constexpr int NCH_S = WIDE ? (int)SLOT_B / 128 : 1;
            int cDc = 0, cCc = 0, cDr = 0, cRr = 0;
            if constexpr (!WIDE) {
                cDc = *(const int*)(L.D_col + 2 * lane); cCc = *(const int*)(L.C_col + 2 * lane);
                cDr = *(const int*)(L.D_row + 2 * lane); cRr = *(const int*)(L.R_row + 2 * lane);
                lds_sync();
            }
            if (ck_pre) {
                constexpr int PR_DIST = (int)(lds_array_bytes_h(kBig ? 128u : (uint32_t)PMAX * 128u) / 2);
                const bool cright = ck_dir == DIR_RIGHT;
                const uint8_t* seqV = cright ? q : r; const uint8_t* seqC = cright ? r : q;
                const uint32_t cri = cright ? ck_i : ck_j, crj = (cright ? ck_j : ck_i) + SLOT_B - STEP;
                FastOut fo{};
                if constexpr (KIND != KIND_PROFILE && !SPECIAL && !WIDE) {
                    constexpr int FL = (int)SLOT_B / 2;   // lanes of the per-pair layout that hold cells
                    const int vc = 2 * lane < (int)SLOT_B ? (int)*(const unsigned short*)(seqV + cri + 2 * lane) : 0;
                    const unsigned long long cb = load_cols(seqC + crj);
                    if (cright) fast_rect<KIND, false, XDROP, FL, PR_DIST>(L.table, fc, cDc, cCc, cDr, cRr, L.D_row, L.D_col, vc & 0xff, (vc >> 8) & 0xff, cb, FL, ck_corner, ck_offadd, -1, nullptr, fo);
                    else fast_rect<KIND, false, XDROP, FL, PR_DIST>(L.table, fc, cDr, cRr, cDc, cCc, L.D_col, L.D_row, vc & 0xff, (vc >> 8) & 0xff, cb, FL, ck_corner, ck_offadd, -1, nullptr, fo);
                } else {
                    // sequence-to-profile / the special modes / slots of more than one chunk: the step through the generic rectangle code, on the LDS borders the
                    // checkpoint was just written to (place_block[_profile_right / _down] + shift_and_offset, scan_block.rs:147-246); then the borders back into registers
                    const uint32_t pq_len = coldp()->q_len[pair_in], pr_len = coldp()->r_len[pair_in];
                    ProfileView pvv{};
                    if constexpr (KIND == KIND_PROFILE) {
                        pvv.P = profile_positions(pr_len, h_max_size);
                        pvv.pos_aa = (const signed char*)r;
                        pvv.aa_pos = (const short*)(r + (uint64_t)pvv.P * 32);
                        pvv.goC = pvv.aa_pos + (uint64_t)pvv.P * 32; pvv.clC = pvv.goC + pvv.P; pvv.goR = pvv.clC + pvv.P;
                    }
                    constexpr int PD_R = KIND == KIND_PROFILE ? 1 : 0, PD_D = KIND == KIND_PROFILE ? 2 : 0;
                    const uint32_t spb = SPECIAL ? (((h_flags & F_LOCAL) ? SP_LOCAL : 0u) | (((h_flags & F_FQS) && cright) ? SP_FQS_ROW0 : 0u)) : 0u;
                    const int crz = clamp16(-ck_off + ZERO);   // the step's relative zero (LOCAL_START / FREE_QUERY_START_GAPS)
                    short* t1 = L.misc + 16; short* t2 = L.misc + 32;
                    lds_fill0(t1, 32);
                    unsigned long long none = 0;
                    Best cur;
                    if (cright) {
                        cur = place_rect<NCH_S, KIND, false, XDROP, PD_R>(L, fc, q, r, pq_len, pr_len, cri, crj, STEP, SLOT_B, L.D_col, L.C_col, t1, t2, ck_corner, crz, ck_offadd, nullptr, none, nullptr, spb, nullptr, &pvv);
                        (void)lds_shift_and_offset(SLOT_B, L.D_row, L.R_row, t1, t2, ck_offadd);
                    } else {
                        cur = place_rect<NCH_S, KIND, false, XDROP, PD_D>(L, fc, r, q, pr_len, pq_len, cri, crj, STEP, SLOT_B, L.D_row, L.R_row, t1, t2, ck_corner, crz, ck_offadd, nullptr, none, nullptr, spb, nullptr, &pvv);
                        (void)lds_shift_and_offset(SLOT_B, L.D_col, L.C_col, t1, t2, ck_offadd);
                    }
                    fo.row = cur.row; fo.col = cur.col;
                    lds_sync();
                    if constexpr (!WIDE) {
                        cDc = *(const int*)(L.D_col + 2 * lane); cCc = *(const int*)(L.C_col + 2 * lane);
                        cDr = *(const int*)(L.D_row + 2 * lane); cRr = *(const int*)(L.R_row + 2 * lane);
                    }
                }
                if (XDROP) {
                    if (cright) { best_i = ck_i + (uint32_t)fo.row; best_j = ck_j + (SLOT_B - STEP) + (uint32_t)fo.col; }
                    else { best_i = ck_i + (SLOT_B - STEP) + (uint32_t)fo.col; best_j = ck_j + (uint32_t)fo.row; }
                }
                lds_sync();
            }
            if constexpr (WIDE) save_ckpt_borders(SLOT_B);   // (LDS -> this wave's checkpoint arena: restore_ckpt_borders / the slot's next entry read it there)
            else { ck_reg[0] = cDc; ck_reg[1] = cCc; ck_reg[2] = cDr; ck_reg[3] = cRr; ck_in_regs = true; }
        }
        lds_sync();
        if (mine) {
            *(int4*)(L.D_col + l8) = int4{reg[0], reg[1], reg[2], reg[3]}; *(int4*)(L.C_col + l8) = int4{reg[4], reg[5], reg[6], reg[7]};
            *(int4*)(L.D_row + l8) = int4{reg[8], reg[9], reg[10], reg[11]}; *(int4*)(L.R_row + l8) = int4{reg[12], reg[13], reg[14], reg[15]};
        }
        lds_sync();
    }
};

// Persistent kernel: WAVES_PER_WG independent waves per workgroup (they share only the read-only score table in LDS);
// every wave pulls pair indices from a global counter until the batch is exhausted.
// Waves per SIMD (second launch bound): 4 at 128 VGPRs for blocks up to 1024 cells (their columns live in registers), 2 for
// 2048; the kernels for blocks up to 256 cells need ~95 and are held to 80 for 6 waves (a few spills outside the
// columns; measured +13 % on the 1 kbp DNA and protein configurations; 64 registers / 8 waves spills into the columns).
template <int PMAX, int KIND, bool TRACE, bool XDROP, bool SPECIAL>
__global__ void __launch_bounds__(WAVES_PER_WG * 64, (PMAX >= 16 ? 2 : (PMAX <= 2 ? 6 : 4))) k_align(const BatchParams bp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id();
    const int wave = uni((int)threadIdx.x >> 6);   // wave-uniform: keeps every per-wave pointer (LDS borders, trace slot) in SGPRs
    // ---- workgroup-shared scoring table
    {
        char* tab = smem;
        if (KIND == KIND_NUC) {
            // packed {score(crow, a), score(crow, b)} per (a, b, crow): one 4-byte read yields both cells of a lane
            nuc_table_fill(tab, bp.matrix, (int)threadIdx.x, WAVES_PER_WG * 64);   // (layout: ba_device.hpp nuc_key_off)
        } else {
            const int nbytes = KIND == KIND_AA ? 27 * 32 : (KIND == KIND_BYTES ? 2 : 0);   // PROFILE: scores live in the pair's image
            for (int k = (int)threadIdx.x; k < nbytes; k += WAVES_PER_WG * 64) tab[k] = (char)bp.matrix[k];
        }
    }
    __syncthreads();
    // The LDS layout depends only on the kernel's block class (PMAX * 128 cells), not on the batch's max size: the
    // four border arrays then sit at compile-time offsets from one per-wave base, i.e. in the immediate offset field of
    // the DS instructions instead of in five SGPRs (which the step loop does not have: they were spilled and reloaded).
    constexpr uint32_t LCLS = kBig ? 128u : (uint32_t)PMAX * 128u;   // LDS layout class (big-block kernels keep only misc in LDS)
    constexpr uint32_t ab = lds_array_bytes_h(LCLS);
    char* base = smem + lds_table_bytes_h(KIND) + (uint32_t)wave * lds_wave_bytes_h(LCLS);
    WaveLds L;
    L.D_col = (short*)(base + 0 * ab); L.C_col = (short*)(base + 1 * ab);
    L.D_row = (short*)(base + 2 * ab); L.R_row = (short*)(base + 3 * ab);
    L.misc = (short*)(base + 4 * ab);
    L.table = smem;
    const FillConsts fc = make_fill_consts(lane, bp.gap_open, bp.gap_extend);
    const uint32_t stride = bp.tb_stride;
    const bool batch_traceback = TRACE && stride > 0;
#ifdef BA_TIMING
    // wall-clock (100 MHz) marks: launch start, last fill wave done, last traceback done -> the length of the traceback tail
    if (bp.prof && blockIdx.x == 0 && wave == 1 && is_lane(0)) atomicMax(bp.prof + 43, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
    if (batch_traceback && wave == 0 && blockIdx.x % stride == 0) {
        if (!(bp.flags & 0x800u))   // (development switch: the traceback waves leave at once, as if they were never resident -- see traceback_help_one)
        traceback_consumer<SPECIAL ? (int)TB_LANE_BYTES_LOC : (int)TB_LANE_BYTES>(bp, SPECIAL ? ~0u : (uint32_t)F_CIGAR_EQ,
                           (unsigned char*)smem + lds_table_bytes_h(KIND) + WAVES_PER_WG * lds_wave_bytes_h(LCLS), 64u, true);   // (the special modes' records also hold zero-mask bits)
#ifdef BA_TIMING
        if (bp.prof && is_lane(0)) atomicMax(bp.prof + 41, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
        return;
    }
    {
        // dense index among the fill waves (traceback waves of this and earlier workgroups skipped)
        const uint32_t cons_before = batch_traceback ? (blockIdx.x + stride - 1) / stride + (blockIdx.x % stride == 0 ? 1u : 0u) : 0u;
        const uint32_t fill_wave = blockIdx.x * WAVES_PER_WG + (uint32_t)wave - cons_before;
        uint32_t turn = 0;
        uint32_t w_next = 0, w_end = 0;   // this wave's share of the work counter: bp.work_chunk pairs per atomic
        bool closing = false;             // queue mode: every producer wave has been seen done
        // (score-only batches: the pairs that leave k_quad are the batch's longest serial chains -- protein set +3 %; with traceback the
        // same priority costs 1 - 3 %: the walks of k_walk and of this kernel are then the longer chains)
        if (!TRACE && bp.cont_mode == 2 && bp.cq_side) __builtin_amdgcn_s_setprio(3);
        if (bp.cont_mode == 2 && bp.cq_side) {
            // The launch beside k_quad: a queue ticket, once taken, is served (an abandoned one would be a lost pair), so no ticket
            // is taken before k_quad is known to be on the device -- from then on its resident waves finish the batch's pairs whatever
            // else waits, and every ticket resolves. Until then this wave holds nothing: if k_quad does not show up (several batches
            // launched at once can fill the machine with waiting workgroups) it leaves, and the launch after k_quad drains the queue.
            uint32_t spins = 0; bool up = false;
            for (;;) {
                uint32_t f = 0;
                if (is_lane(0)) f = __hip_atomic_load(bp.cq_ctrl + 20, BA_RLX_AGENT);
                if (uni((int)f)) { up = true; break; }
                if (++spins > (1u << 20)) break;
                __builtin_amdgcn_s_sleep(32);
            }
            if (!up) return;
        }
        for (;;) {
            BA_TSTAMP(tk0);
            if (bp.cont_mode != 2 && w_next == w_end) {
                uint32_t v = 0;
                if (is_lane(0)) v = atomicAdd(bp.work_counter, bp.work_chunk);
                w_next = (uint32_t)uni((int)v);
                if (w_next >= bp.n) break;
                w_end = min(w_next + bp.work_chunk, bp.n);
            }
            uint32_t pair = bp.cont_mode == 2 ? 0u : w_next++;
            const PairCont* rec = nullptr;
            if (bp.cont_mode == 2) {
                // Small-block batches: the pairs k_quad could not finish arrive through a queue while it is still running (entry =
                // 1 + 2 * pair + (1 if the pair is to be run from scratch, else resume from its record)). A ticket is a queue position;
                // the wave waits for its entry, and leaves once every producer wave is done and the queue ends before the ticket.
                {
                if (w_next >= w_end) {   // one ticket while the producers run -- an entry is work waiting --, eight once they are done
                    const uint32_t chunk = closing ? 8u : 1u;
                    uint32_t t0 = 0;
                    if (is_lane(0)) t0 = __hip_atomic_fetch_add(bp.cq_ctrl + 16, chunk, BA_RLX_AGENT);
                    w_next = (uint32_t)uni((int)t0); w_end = w_next + chunk;
                }
                const uint32_t ticket = w_next++;
                if (ticket >= bp.n) break;   // (at most one entry per pair)
                uint32_t e = 0, spins = 0;
                for (;;) {
                    uint32_t v = 0, done = 0;
                    if (is_lane(0)) { v = __hip_atomic_load(bp.cq_queue + ticket, BA_RLX_AGENT); done = __hip_atomic_load(bp.cq_ctrl + 32, BA_RLX_AGENT); }
                    v = (uint32_t)uni((int)v); done = (uint32_t)uni((int)done);
                    if (v) { e = v; break; }
                    if (closing) {   // all producers were done before this read found the entry still empty: is the queue shorter than the ticket?
                        uint32_t tail = 0;
                        if (is_lane(0)) tail = __hip_atomic_load(bp.cq_ctrl, BA_RLX_AGENT);
                        if (ticket >= (uint32_t)uni((int)tail)) break;
                    }
                    if (done >= bp.cq_producers) { closing = true; continue; }
                    if (++spins > (1u << 24)) {   // tens of seconds without the producers finishing: never hang, and never lose the pair
                        // silently -- report (the host fails the run). The launch beside k_quad takes tickets only once k_quad is
                        // running (see below), so a ticket's producers are on the device and finish on their own.
                        if (is_lane(0)) __hip_atomic_store(bp.cq_ctrl + 48, 1u, BA_RLX_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(32);
                }
                if (!e) break;
                pair = (e - 1u) >> 1;   // (the record, released before its entry, is read past this CU's L1: see run())
                if (!((e - 1u) & 1u)) rec = bp.cont_in + pair;
                }
            }
            uint32_t slot = fill_wave * bp.slots_per_wave + turn;
            if (++turn == bp.slots_per_wave) turn = 0;
            if (TRACE && bp.trace_off) slot = pair;   // pair-slot batches: the pair's own region (never shared, never waited for)
            Aligner<PMAX, KIND, TRACE, XDROP, SPECIAL> al(bp, L, fc);
            BA_TSTAMP(tw0);
            if (batch_traceback) al.acquire_slot(slot, bp, (unsigned char*)base);
            BA_TSTAMP(tw1);
#ifdef BA_TIMING
            if (bp.prof && is_lane(0)) atomicAdd(bp.prof + 17, tw1 - tw0);
#endif
            al.trace = bp.trace_arena + (uint64_t)slot * bp.trace_stride;
            al.blocks = bp.blocks + (uint64_t)slot * bp.blocks_stride;
            if (TRACE && bp.trace_off) { al.trace = bp.trace_arena + bp.trace_off[pair]; al.blocks = bp.blocks + bp.blocks_off[pair]; }
            al.ckpt = bp.ckpt + (uint64_t)(fill_wave + bp.ckpt_wave0) * 8 * bp.max_size;   // (second half: the base of a chain of speculative grows)
            if (kBig) {   // the four borders live in this wave's slice of the big arena, not in LDS
                short* bw = bp.big + (uint64_t)fill_wave * big_wave_shorts(bp.max_size);
                const uint64_t as = big_array_shorts(bp.max_size);
                L.D_col = bw; L.C_col = bw + as; L.D_row = bw + 2 * as; L.R_row = bw + 3 * as;
                al.big_top = bw + 4 * as;
            }
            BA_TSTAMP(tk1);
            al.run(pair, slot, batch_traceback, rec);
#ifdef BA_TIMING
            {   // development: 44 = pair taken -> run() entered, 45 = run(), 46 = pairs, 47 = run() before its step loop
                const unsigned long long tk2 = __builtin_amdgcn_s_memtime();
                al.prof[44] += tk1 - tk0; al.prof[45] += tk2 - tk1; al.prof[46] += 1;
                if (bp.prof && is_lane(0)) for (int k = 0; k < 48; k++) if (k != 17 && !(k >= 20 && k < 32) && !(k >= 40 && k < 44)) atomicAdd(bp.prof + k, al.prof[k]);
            }
#endif
        }
#ifdef BA_TIMING
        if (bp.prof && is_lane(0)) atomicMax(bp.prof + 40, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
    }
    // Traceback waves are spread over the chip (wave 0 of every tb_stride-th workgroup): a walk is all divergent
    // loads, and eight of them on one CU would queue behind that CU's single memory pipeline. A fill wave joins with one
    // lane once the batch has no pairs left for it; its record and move table go where its block borders were.
    if (batch_traceback) {
        lds_sync();
        traceback_consumer<SPECIAL ? (int)TB_LANE_BYTES_LOC : (int)TB_LANE_BYTES>(bp, SPECIAL ? ~0u : (uint32_t)F_CIGAR_EQ, (unsigned char*)base, 1u, false);
#ifdef BA_TIMING
        if (bp.prof && is_lane(0)) atomicMax(bp.prof + 41, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
    }
}

}  // namespace ba
