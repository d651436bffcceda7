// block_aligner_amd — device side of the adaptive block aligner for gfx950 (CDNA4, wave64).
//
// One wavefront aligns one sequence pair. The block's vector axis (rows for a right shift, columns for a
// down shift) is spread over the 64 lanes, K consecutive cells per lane held as K/2 packed 2 x i16 VGPRs
// (v_pk_add_i16 clamp / v_pk_max_i16). The four block borders and their checkpoint copies live in LDS;
// the 32-bit block offset and all driver state are wave-uniform (SGPRs). The in-column gap recurrence is a
// max-plus prefix scan: lane-local serial scan + a 6-step DPP prefix-max over lanes on 32-bit re-based values.
//
// Semantics follow the reference's AVX2 (L = 16) backend bit for bit:
//   driver        /root/reference/src/scan_block.rs:94-595
//   block fill    /root/reference/src/scan_block.rs:1083-1228 (+ avx2.rs:297-338 scan incl. its zero-shift-in quirk)
//   border moves  /root/reference/src/scan_block.rs:1003-1061
//   trace/CIGAR   /root/reference/src/scan_block.rs:1344-1672
// The trace encoding, LDS layout and work distribution are this implementation's own.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ba_params.h"

namespace ba {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------ cross-lane helpers (wave64)
__device__ __forceinline__ int lane_id() { return (int)threadIdx.x & 63; }
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
// Single-lane side effects (global stores, the work-queue atomic) sit inside wave-uniform control flow. The lane id is
// laundered through an empty asm at every such site: if the compiler can prove two `lane == 0` tests equal it
// jump-threads one into the other across the persistent loop's back edge, peels lane 0 out of the loop and the
// other 63 lanes spin on a stale pair index (observed with ROCm 7.2 clang 22 on gfx950).
__device__ __forceinline__ bool is_lane(int k) {
    int l = lane_id();
    asm volatile("" : "+v"(l));
    return l == k;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_keep(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
// lane l <- lane l-1; lane 0 <- fill
__device__ __forceinline__ int wave_shr1(int src, int fill) { return dpp_keep<0x138, 0xf>(fill, src); }

// inclusive prefix max over the 64 lanes (row_shr 1/2/4/8, row_bcast:15, row_bcast:31)
__device__ __forceinline__ int wave_prefix_max(int v) {
    v = max(v, dpp_keep<0x111, 0xf>(v, v));
    v = max(v, dpp_keep<0x112, 0xf>(v, v));
    v = max(v, dpp_keep<0x114, 0xf>(v, v));
    v = max(v, dpp_keep<0x118, 0xf>(v, v));
    v = max(v, dpp_keep<0x142, 0xa>(v, v));
    v = max(v, dpp_keep<0x143, 0xc>(v, v));
    return v;
}
__device__ __forceinline__ int wave_max(int v) { return __builtin_amdgcn_readlane(wave_prefix_max(v), 63); }
__device__ __forceinline__ int wave_min(int v) { return -wave_max(-v); }

__device__ __forceinline__ s16x2 as_s(int x) { return __builtin_bit_cast(s16x2, x); }
__device__ __forceinline__ int as_i(s16x2 x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ u16x2 as_u(s16x2 x) { return __builtin_bit_cast(u16x2, x); }
__device__ __forceinline__ s16x2 splat(int v) { return s16x2{(short)v, (short)v}; }
__device__ __forceinline__ s16x2 adds(s16x2 a, s16x2 b) { return __builtin_elementwise_add_sat(a, b); }
__device__ __forceinline__ s16x2 subs(s16x2 a, s16x2 b) { return __builtin_elementwise_sub_sat(a, b); }
__device__ __forceinline__ s16x2 vmax(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
// 1 where the 16-bit halves differ, 0 where equal
__device__ __forceinline__ u16x2 neq01(s16x2 a, s16x2 b) {
    return __builtin_elementwise_min((u16x2)(as_u(a) - as_u(b)), u16x2{1, 1});
}
__device__ __forceinline__ int clamp16(int x) { return x < -32768 ? -32768 : (x > 32767 ? 32767 : x); }

// ------------------------------------------------------------------ per-wave LDS image
struct WaveLds {
    short* D_col; short* C_col; short* D_row; short* R_row;
    short* D_col_ck; short* C_col_ck; short* D_row_ck; short* R_row_ck;
    short* vtab;          // 16 scan artefact constants (avx2.rs:315-338)
    const int8_t* mat;    // scoring table copy
};
__host__ __device__ inline uint32_t lds_array_bytes(uint32_t max_size) { return max_size * 2 + 32; }
__host__ __device__ inline uint32_t lds_wave_bytes(uint32_t max_size) { return 8 * lds_array_bytes(max_size) + 128 + 896; }

struct Best { int mx; int row; int col; };   // rect max (i16 value) and, for X-drop, its resolved location

// ------------------------------------------------------------------ block fill
// Fills a width x height rectangle column by column; lanes own K = 2P consecutive cells of the vector axis.
// seqV runs along the vector axis, seqC supplies one byte per column (scan_block.rs:1083-1228; for a down
// shift the caller swaps the sequences exactly as the reference does).
template <int P, int KIND, bool TRACE, bool XDROP>
__device__ __forceinline__ Best place_block(const WaveLds& L, const uint8_t* __restrict__ seqV, const uint8_t* __restrict__ seqC,
                            uint32_t lenV, uint32_t lenC, uint32_t start_i, uint32_t start_j, uint32_t width, uint32_t height,
                            short* Dc, short* Cc, short* Dr, short* Rr, int corner, int rel_zero, int off_add,
                            int gap_open, int gap_extend, uint32_t* __restrict__ trace_out, unsigned long long& cells) {
    constexpr int K = 2 * P;
    const int lane = lane_id();
    const int nl = (int)(height / K);          // active lanes (64 when height >= 128)
    const bool active = lane < nl;
    const int r0 = lane * K;
    Best res{0, 0, 0};                          // MIN = 0 (avx2.rs:16)
    if (width == 0 || height == 0) return res;

    const s16x2 go2 = splat(gap_open), ge2 = splat(gap_extend), ome2 = subs(splat(gap_open), splat(gap_extend));
    const s16x2 offa = splat(off_add);

    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    s16x2 d[P], c[P], dmax[P], vconst[P];
    u16x2 jlast[P];
    int qidx[K];
#pragma unroll
    for (int p = 0; p < P; p++) {
        int dv = 0, cv = 0;
        if (active) { dv = *(const int*)(Dc + r0 + 2 * p); cv = *(const int*)(Cc + r0 + 2 * p); }
        d[p] = adds(as_s(dv), offa);            // just_offset folded into the load (scan_block.rs:1003-1012)
        c[p] = adds(as_s(cv), offa);
        dmax[p] = splat(0);
        jlast[p] = u16x2{0, 0};
        vconst[p] = as_s(*(const int*)(L.vtab + ((r0 + 2 * p) & 15)));
#pragma unroll
        for (int h = 0; h < 2; h++) {
            int b = active ? (int)seqV[start_i + r0 + 2 * p + h] : 0;
            qidx[2 * p + h] = KIND == KIND_NUC ? (b & 15) : (KIND == KIND_AA ? (b & 31) : b);
        }
    }
    const int KG = K * gap_extend;
    const bool break_armed = !XDROP && (start_i + height > lenV);
    uint32_t tacc[P];
#pragma unroll
    for (int p = 0; p < P; p++) tacc[p] = 0;
    int corner_cur = corner;
    uint32_t j = 0;
    int cvec = 0;   // the next 8 column bytes, one per lane (lanes 0..7), fetched once per 8 columns
    for (; j < width; j++) {
        if ((j & 7) == 0) cvec = (int)seqC[start_j + j + (lane & 7)];
        const int cb = __builtin_amdgcn_readlane(cvec, (int)(j & 7));
        // ---- substitution scores for this column
        s16x2 sc[P];
#pragma unroll
        for (int p = 0; p < P; p++) {
            int s0, s1;
            if (KIND == KIND_NUC) { const int8_t* row = L.mat + (cb & 7) * 16; s0 = row[qidx[2 * p]]; s1 = row[qidx[2 * p + 1]]; }
            else if (KIND == KIND_AA) { const int8_t* row = L.mat + cb * 32; s0 = row[qidx[2 * p]]; s1 = row[qidx[2 * p + 1]]; }
            else { s0 = qidx[2 * p] == cb ? L.mat[0] : L.mat[1]; s1 = qidx[2 * p + 1] == cb ? L.mat[0] : L.mat[1]; }
            sc[p] = s16x2{(short)s0, (short)s1};
        }
        // ---- D00: the previous column shifted down by one cell; lane 0 takes the corner (MIN after column 0)
        const int prev_last = wave_shr1(as_i(d[P - 1]), (int)((uint32_t)corner_cur << 16));
        corner_cur = 0;
        s16x2 x[P], cnew[P], copen[P], dpart[P];
#pragma unroll
        for (int p = 0; p < P; p++) {
            const int below = p == 0 ? prev_last : as_i(d[p - 1]);
            const s16x2 d00 = as_s(__builtin_amdgcn_alignbit(as_i(d[p]), below, 16));
            s16x2 d11 = adds(d00, sc[p]);
            if (p == 0 && start_i == 0 && start_j + j == 0 && lane == 0) d11.x = (short)rel_zero;   // cell (0,0), scan_block.rs:1130-1132
            copen[p] = adds(d[p], go2);
            cnew[p] = vmax(adds(c[p], ge2), copen[p]);
            d11 = vmax(d11, cnew[p]);
            dpart[p] = d11;
            x[p] = adds(d11, ome2);             // D11_open
        }
        // ---- R11: max-plus scan down the column. lane-local serial part:
        s16x2 r[P];
#pragma unroll
        for (int p = 0; p < P; p++) {
            const s16x2 t2 = adds(x[p], ge2);
            s16x2 m1 = vmax(x[p], s16x2{t2.x, t2.x});
            if (p > 0) {
                const s16x2 up = s16x2{r[p - 1].y, r[p - 1].y};
                m1 = vmax(m1, adds(up, s16x2{(short)gap_extend, (short)(2 * gap_extend)}));
            }
            r[p] = m1;
        }
        // cross-lane part on 32-bit values re-based by lane * K * g, so the distance term becomes a plain max
        {
            const int A = (int)r[P - 1].y;
            const int pm = wave_prefix_max(A - lane * KG);
            const int pmx = wave_shr1(pm, -(1 << 29));
            int cin = max(lane * KG, (lane - 1) * KG + pmx);   // lane*KG = the MIN (0) carry above row 0 of the column
            cin = max(cin, -32768);
            const s16x2 cin2 = splat(cin);
#pragma unroll
            for (int p = 0; p < P; p++) {
                const s16x2 t = adds(cin2, s16x2{(short)((2 * p + 1) * gap_extend), (short)((2 * p + 2) * gap_extend)});
                r[p] = vmax(vmax(r[p], t), vconst[p]);
            }
        }
        // ---- finish D11, bookkeeping
        uint32_t tnib[P];
        int prev_neqR = 0;
        if (TRACE) prev_neqR = wave_shr1(as_i(__builtin_bit_cast(s16x2, neq01(r[P - 1], x[P - 1]))), 1 << 16);
#pragma unroll
        for (int p = 0; p < P; p++) {
            const s16x2 d11 = vmax(dpart[p], r[p]);
            if (TRACE) {
                const u16x2 nC = neq01(d11, cnew[p]);
                const u16x2 nR = neq01(d11, r[p]);
                const u16x2 nCo = neq01(cnew[p], copen[p]);
                const u16x2 nRo = neq01(r[p], x[p]);
                // "R opened" flag belongs to the cell below it (scan_block.rs:1179-1182): shift down by one cell
                const u16x2 nRs = __builtin_bit_cast(u16x2, __builtin_amdgcn_alignbit(__builtin_bit_cast(int, nRo), prev_neqR, 16));
                prev_neqR = __builtin_bit_cast(int, nRo);
                const u16x2 nib = nC + nR * (u16x2){2, 2} + nCo * (u16x2){4, 4} + nRs * (u16x2){8, 8};
                tnib[p] = __builtin_bit_cast(uint32_t, nib);
            }
            dmax[p] = vmax(dmax[p], d11);
            if (XDROP) {
                const u16x2 ne = neq01(dmax[p], d11);
                const u16x2 jj = u16x2{(unsigned short)j, (unsigned short)j};
                jlast[p] = (u16x2)(jlast[p] - jj) * ne + jj;
            }
            d[p] = d11;
            c[p] = cnew[p];
        }
        if (TRACE) {
            const uint32_t sh = (j & 3) * 4;
#pragma unroll
            for (int p = 0; p < P; p++) tacc[p] |= tnib[p] << sh;
            if ((j & 3) == 3) {
                if (active) {
#pragma unroll
                    for (int p = 0; p < P; p++) trace_out[((j >> 2) * P + p) * nl + lane] = tacc[p];
                }
#pragma unroll
                for (int p = 0; p < P; p++) tacc[p] = 0;
            }
        }
        // last cell of the column feeds the orthogonal border (scan_block.rs:1213-1214)
        if (is_lane(nl - 1)) { Dr[j] = d[P - 1].y; Rr[j] = r[P - 1].y; }
        cells += height;
        if (break_armed && start_j + j >= lenC) {   // scan_block.rs:1216-1224
            if (TRACE && (j & 3) != 3 && active) {
#pragma unroll
                for (int p = 0; p < P; p++) trace_out[((j >> 2) * P + p) * nl + lane] = tacc[p];
            }
            break;
        }
    }
    // ---- write the vector-axis border back
    if (active) {
#pragma unroll
        for (int p = 0; p < P; p++) { *(int*)(Dc + r0 + 2 * p) = as_i(d[p]); *(int*)(Cc + r0 + 2 * p) = as_i(c[p]); }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- rectangle max and (X-drop) its location: among cells equal to the max, smallest (row % 16), then
    // largest column, then largest row (avx2.rs:271-274 + scan_block.rs:1198-1200 last-writer-wins per lane)
    int lm = -32768;
#pragma unroll
    for (int p = 0; p < P; p++) lm = max(lm, max((int)dmax[p].x, (int)dmax[p].y));
    if (!active) lm = -32768;
    const int M = wave_max(lm);
    res.mx = M;
    if (XDROP) {
        int key = 0x7fffffff;
#pragma unroll
        for (int p = 0; p < P; p++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int v = h ? (int)dmax[p].y : (int)dmax[p].x;
                const int jl = h ? (int)jlast[p].y : (int)jlast[p].x;
                const int row = r0 + 2 * p + h;
                const int k = ((row & 15) << 24) | ((4095 - jl) << 12) | (4095 - row);
                if (active && v == M) key = min(key, k);
            }
        }
        key = wave_min(key);
        res.col = 4095 - ((key >> 12) & 4095);
        res.row = 4095 - (key & 4095);
    }
    return res;
}

}  // namespace ba
