// block_aligner_amd — device side of the adaptive block aligner for gfx950 (CDNA4, wave64).
//
// One wavefront aligns one sequence pair. A rectangle of the DP matrix is filled column by column; the column (the
// "vector axis": rows for a right shift, columns for a down shift) is cut into chunks of 128 cells, and inside a chunk
// lane l owns cells 2l and 2l+1 as one packed 2 x i16 VGPR (v_pk_add_i16 clamp / v_pk_max_i16). Chunks of a column are
// processed top to bottom, each with a 64-lane max-plus prefix scan built from six DPP-fused v_max_i32 on 32-bit
// values re-based by lane * 2g; the carry between chunks is a wave-uniform scalar. The four block borders live in
// LDS, the 32-bit block offset and all driver state are wave-uniform (SGPRs).
//
// Semantics follow the reference's AVX2 (L = 16) backend bit for bit:
//   block fill    /root/reference/src/scan_block.rs:1083-1228 (+ avx2.rs:297-338 scan incl. its zero-shift-in quirk)
//   border moves  /root/reference/src/scan_block.rs:1003-1061
// The trace encoding, LDS layout and work distribution are this implementation's own.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ba_params.h"

namespace ba {

// development aid: -DBA_TIMING builds accumulate s_memtime deltas per phase (slots documented in tools/gpu_timing.py)
#ifdef BA_TIMING
#define BA_TSTAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define BA_TADD(acc, slot, a, b) (acc)[slot] += (b) - (a)
#else
#define BA_TSTAMP(var) do { } while (0)
#define BA_TADD(acc, slot, a, b) do { } while (0)
#endif

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

// one wave's LDS operations execute in program order; this only keeps the compiler from reordering them across a hand-off
__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_keep(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
// lane l <- lane l-1; lane 0 <- fill
__device__ __forceinline__ int wave_shr1(int src, int fill) { return dpp_keep<0x138, 0xf>(fill, src); }
// lane l <- lane l-1; lane 0 <- 0 (bound_ctrl:0): a single v_mov_b32_dpp, no "old value" set-up move
__device__ __forceinline__ int wave_shr1_z(int src) { return __builtin_amdgcn_update_dpp(0, src, 0x138, 0xf, 0xf, true); }
// write a wave-uniform value into lane 0 of v
__device__ __forceinline__ int set_lane0(int v, int uniform_val) {
    asm volatile("v_writelane_b32 %0, %1, 0" : "+v"(v) : "s"(__builtin_amdgcn_readfirstlane(uniform_val)));
    return v;
}
// Rarely read wave-uniform state parked in the lanes of one VGPR (one v_writelane per update) instead of in SGPRs:
// the step loop is over the SGPR budget, and what the compiler picks to spill is not what is cold.
template <int LANE>
__device__ __forceinline__ void park(int& v, int uniform_val) {
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(__builtin_amdgcn_readfirstlane(uniform_val)), "n"(LANE));
}
template <int LANE>
__device__ __forceinline__ int unpark(int v) { return __builtin_amdgcn_readlane(v, LANE); }
// a[l-1] + b[l] in one instruction (lane 0: 0 + b[0])
__device__ __forceinline__ int add_shr1(int a, int b) {
    int t;
    asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(t) : "v"(a), "v"(b));
    return t;
}

// inclusive prefix max over the 64 lanes: row_shr 1/2/4/8 inside rows of 16, then row_bcast:15 / row_bcast:31.
// The DPP control is fused into v_max_i32 (lanes without a valid source keep their value); a DPP read needs two wait
// states after the VALU write of its source, and hipcc does not pad inside an asm statement, hence the s_nop 1.
__device__ __forceinline__ int wave_prefix_max(int v) {
    asm volatile(
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}
// the same scan when only the first 16 / 32 lanes hold cells (blocks of 32 / 64 cells): four / five steps
__device__ __forceinline__ int wave_prefix_max16(int v) {
    asm volatile(
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}
__device__ __forceinline__ int wave_prefix_max32(int v) {
    asm volatile(
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}
template <int LANES>   // 16, 32 or 64 lanes hold cells
__device__ __forceinline__ int prefix_max_lanes(int v) { return LANES == 16 ? wave_prefix_max16(v) : (LANES == 32 ? wave_prefix_max32(v) : wave_prefix_max(v)); }
__device__ __forceinline__ int wave_max(int v) { return __builtin_amdgcn_readlane(wave_prefix_max(v), 63); }
template <int LANES>
__device__ __forceinline__ int max_lanes(int v) { return __builtin_amdgcn_readlane(prefix_max_lanes<LANES>(v), LANES - 1); }
__device__ __forceinline__ int wave_min(int v) { return -wave_max(-v); }

__device__ __forceinline__ s16x2 as_s(int x) { return __builtin_bit_cast(s16x2, x); }
__device__ __forceinline__ int as_i(s16x2 x) { return __builtin_bit_cast(int, x); }
__device__ __forceinline__ int pk(int lo, int hi) { return (lo & 0xffff) | (int)((uint32_t)hi << 16); }
__device__ __forceinline__ int splat(int v) { return pk(v, v); }
__device__ __forceinline__ int adds(int a, int b) { return as_i(__builtin_elementwise_add_sat(as_s(a), as_s(b))); }
__device__ __forceinline__ int subs(int a, int b) { return as_i(__builtin_elementwise_sub_sat(as_s(a), as_s(b))); }
__device__ __forceinline__ int vmax(int a, int b) { return as_i(__builtin_elementwise_max(as_s(a), as_s(b))); }
__device__ __forceinline__ int vmaxu(int a, int b) {
    return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
// per 16-bit half: 1 where a != b, else 0. Forced to two packed ops (sub, unsigned min with 1): left to itself the
// compiler canonicalises umin(a - b, 1) into compare/select/permute sequences that cost five issue slots.
__device__ __forceinline__ int neq01(int a, int b, int ones) {
    int t;
    (void)ones;   // the 1 is an inline constant read through op_sel_hi (both halves take its low half): no SGPR
    asm("v_pk_sub_u16 %0, %1, %2\n\tv_pk_min_u16 %0, %0, 1 op_sel_hi:[1,0]" : "=&v"(t) : "v"(a), "v"(b));
    return t;
}
// per 16-bit half: 1 where a == b, else 0 (sub, then saturating 1 - diff)
__device__ __forceinline__ int eq01(int a, int b, int ones) {
    int t;
    (void)ones;
    asm("v_pk_sub_u16 %0, %1, %2\n\tv_pk_sub_u16 %0, 1, %0 op_sel_hi:[0,1] clamp" : "=&v"(t) : "v"(a), "v"(b));
    return t;
}
// bit 15 of each half (the sign of a 16-bit lane), everything else cleared; the literal form keeps it a 2-cycle VOP2
__device__ __forceinline__ int sign_bits(int x) {
    int t;
    asm("v_and_b32_e32 %0, 0x80008000, %1" : "=v"(t) : "v"(x));
    return t;
}
// (a & mask) | (b & ~mask) as ONE instruction (left to itself the compiler re-expands the pattern into and / or forms)
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) {
    uint32_t t;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(t) : "s"(mask), "v"(a), "v"(b));
    return t;
}
__device__ __forceinline__ int pk_mul(int a, int m) {
    int t;
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(t) : "v"(a), "s"(m));
    return t;
}
// per half: a * m + c (mod 2^16), m wave-uniform
__device__ __forceinline__ int pk_mad(int a, int m, int c) {
    int t;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(t) : "v"(a), "s"(m), "v"(c));
    return t;
}
// per half: a * M + c (mod 2^16) for a small compile-time M: an inline constant instead of a wave-uniform register
template <int M>
__device__ __forceinline__ int pk_mad_k(int a, int c) {
    static_assert(M >= 0 && M <= 64, "inline integer constant");
    int t;
    asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(t) : "v"(a), "n"(M), "v"(c));
    return t;
}
__device__ __forceinline__ int clamp16(int x) { return x < -32768 ? -32768 : (x > 32767 ? 32767 : x); }

// ------------------------------------------------------------------ LDS image
struct WaveLds {
    short* D_col; short* C_col; short* D_row; short* R_row;   // per wave
    short* misc;          // per wave: 16 scan artefact constants, temp1[16], temp2[16]
    const char* table;    // per workgroup: NUC packed-pair score table (8 KB) / AA 27x32 bytes / BYTES {match, mismatch}
};
struct Best { int mx; int row; int col; };

// max of the first 8 entries of two packed border registers at once (scan_block.rs:1020-1022 for both borders): the two
// per-lane maxima share one register (a: low half, b: high half) for the reduction over lanes 0..3
__device__ __forceinline__ void first8_max2(int a, int b, int& ma, int& mb) {
    const s16x2 sa = as_s(a), sb = as_s(b);
    const int xa = vmax(a, as_i(s16x2{sa.y, sa.x})), xb = vmax(b, as_i(s16x2{sb.y, sb.x}));   // both halves = max(lo, hi)
    int m = __builtin_amdgcn_perm(xb, xa, 0x05040100);        // {xa.lo, xb.lo}
    m = vmax(m, __builtin_amdgcn_update_dpp(m, m, 0xB1, 0xf, 0xf, false));   // quad_perm:[1,0,3,2]
    m = vmax(m, __builtin_amdgcn_update_dpp(m, m, 0x4E, 0xf, 0xf, false));   // quad_perm:[2,3,0,1]
    const s16x2 r = as_s(__builtin_amdgcn_readlane(m, 0));
    ma = r.x; mb = r.y;
}
// loop-invariant per-lane / per-kernel values
struct FillConsts {
    int go2, ge2, ome2;       // splat(gap_open), splat(gap_extend), splat(open (-) extend)
    int g12;                  // {g, 2g}
    int ones;                 // 0x00010001
    int laneKG, lanem1KG;     // lane * 2g; (lane - 1) * 2g, except lane 0 which holds -32768 (no lane above: a candidate that never wins)
    int vconst;               // scan artefact constants of this lane's two cells (avx2.rs:315-338)
    int vconst_top;           // single-chunk columns: max(vconst, (cell + 1) g), i.e. with the MIN = 0 carry above the column folded in
    int gap_extend;
};

// The scan constants of a lane that owns cells 2 lane, 2 lane + 1 of a 128-cell chunk (every kernel's two-cells-per-lane code; `lane` is the
// lane inside the chunk -- the row lane of k_quad). One definition: the kernels and the device-side known-answer test of the lane
// primitives (k_lane_kat, ba_kernels.hip) build the same values.
__device__ __forceinline__ FillConsts make_fill_consts(int lane, int gap_open, int g) {
    FillConsts fc;
    fc.gap_extend = g;
    fc.go2 = splat(gap_open); fc.ge2 = splat(g); fc.ome2 = splat(clamp16(gap_open - g));   // (scalar arithmetic: stays in an SGPR)
    fc.g12 = pk(g, 2 * g);
    fc.ones = 0x00010001;
    fc.laneKG = lane * 2 * g; fc.lanem1KG = lane ? (lane - 1) * 2 * g : -32768;
    // zero shift-in artefacts of the reference's in-vector scan (avx2.rs:315-338): lanes 0..6 and 8..14 of each
    // 16-cell vector see a virtual 0 at distance k%8+1, lane 7 sees 12g, lane 15 none
    int v[2];
    for (int h = 0; h < 2; h++) {
        const int k = (2 * lane + h) & 15;
        const int mult = k == 15 ? 0 : (k == 7 ? 12 : (k & 7) + 1);
        v[h] = mult ? max(-32768, mult * g) : -32768;
    }
    fc.vconst = pk(v[0], v[1]);
    fc.vconst_top = pk(max(v[0], max(-32768, (2 * lane + 1) * g)), max(v[1], max(-32768, (2 * lane + 2) * g)));
    return fc;
}

// one pair's AAProfile image (layout in ba_params.h)
struct ProfileView {
    const signed char* pos_aa; const short* aa_pos; const short* goC; const short* clC; const short* goR;
    uint32_t P;
};
// special alignment modes of a rectangle (scan_block.rs:89): handled on the generic path only
enum : uint32_t { SP_LOCAL = 1, SP_FQS_ROW0 = 2, SP_FQE = 4 };
struct FqeOut { int M; int j; };   // FREE_QUERY_END_GAPS: D_max / argmax_j of vector lane (query length % 16)

// per-chunk substitution-score lookup state, set up once per rectangle from the chunk's two vector-axis bytes
template <int KIND> struct ScoreKey;
template <> struct ScoreKey<KIND_NUC> { int off; };          // byte offset of the (a, b) pair's entries in the packed-pair table (nuc_key_off)
template <> struct ScoreKey<KIND_AA> { int a, b; };          // column indices (0..31)
template <> struct ScoreKey<KIND_BYTES> { int a, b; };       // raw bytes
template <> struct ScoreKey<KIND_PROFILE> { int a, b; };     // query residues (0..31), used by right rectangles only

// The NUC packed-pair table (8 KB of LDS per workgroup): entry (a, b, c) = {score(c, a), score(c, b)} for a lane's two vector-axis bytes (a & 15,
// b & 15) and a column byte (c & 7), scores.rs:157,196,204-209, at byte offset ((a << 4 | b) << 5) + (c << 2): the column's term is byte-sized, so
// the multi-pair kernels add it by SDWA byte select (eight columns from two registers, one instruction per score address).
// (Tried in round 5: a bit permutation of the address chosen for the LDS banks -- the letters that matter are 1, 3, 4, 7, which row-major orders
// fold onto a few banks: SQ_LDS_BANK_CONFLICT per 50 k-pair launch 8.5 G cycles for [c][a][b], 14.4 G for this order, 2.9 G for the permutation --
// and config 3 took 169.9 ms with it against 165.9: the fill is bound by vector issue, the conflicts cost nothing, the longer key computation does.)
BA_HD constexpr uint32_t nuc_key_off(uint32_t a, uint32_t b) { return (((a & 15u) << 4) | (b & 15u)) << 5; }
constexpr uint32_t NUC_GAM_LO = 0x0C080400u, NUC_GAM_HI = 0x1C181410u;   // byte c & 7 -> (c & 7) * 4
BA_HD constexpr uint32_t nuc_col_off(uint32_t c) { return (uint32_t)(((((unsigned long long)NUC_GAM_HI << 32) | NUC_GAM_LO) >> ((c & 7u) * 8u)) & 0xffu); }
// every kernel fills its workgroup's table with this
__device__ __forceinline__ void nuc_table_fill(char* tab, const int8_t* matrix, int tid, int nthreads) {
    for (int e = tid; e < 8 * 16 * 16; e += nthreads) {
        const int crow = e >> 8, a = (e >> 4) & 15, b = e & 15;
        *(int*)(tab + nuc_key_off((uint32_t)a, (uint32_t)b) + nuc_col_off((uint32_t)crow)) = (matrix[crow * 16 + a] & 0xffff) | (int)((uint32_t)matrix[crow * 16 + b] << 16);
    }
}

template <int KIND>
__device__ __forceinline__ ScoreKey<KIND> make_key(int a, int b) {
    ScoreKey<KIND> k;
    if constexpr (KIND == KIND_NUC) k.off = (int)nuc_key_off((uint32_t)a, (uint32_t)b);
    else if constexpr (KIND == KIND_AA || KIND == KIND_PROFILE) { k.a = a & 31; k.b = b & 31; }
    else { k.a = a; k.b = b; }
    return k;
}
// packed {score(cb, a), score(cb, b)} for column byte cb (scores.rs:121-127, 204-209, 263-267)
template <int KIND>
__device__ __forceinline__ int fetch_score(const char* table, const ScoreKey<KIND>& k, int cb) {
    if constexpr (KIND == KIND_NUC) return *(const int*)(table + nuc_col_off((uint32_t)cb) + k.off);
    else if constexpr (KIND == KIND_AA) {
        const signed char* row = (const signed char*)table + cb * 32;
        return pk(row[k.a], row[k.b]);
    } else if constexpr (KIND == KIND_BYTES) {
        const signed char* t = (const signed char*)table;
        return pk(k.a == cb ? t[0] : t[1], k.b == cb ? t[0] : t[1]);
    } else return 0;   // PROFILE: scores come from the profile image, see profile_score
}
// NUC, multi-pair kernels: the step's eight column bytes become eight table offsets (nuc_col_off, a byte each) with two operations per four columns
// (v_perm looks the bytes up in an 8-entry table), the lane's four key offsets come from its eight bytes two at a time (both halves of a register
// at once), and every score address is ONE instruction: the pair's base (table + key, per step) plus byte j
__device__ __forceinline__ uint32_t nuc_col_offsets(uint32_t cb4) { return (uint32_t)__builtin_amdgcn_perm((int)NUC_GAM_HI, (int)NUC_GAM_LO, (int)(cb4 & 0x07070707u)); }
__device__ __forceinline__ uint32_t nuc_keys2(uint32_t w) {   // w = {a, b, a', b'} bytes -> {nuc_key_off(a, b), nuc_key_off(a', b')} as 16-bit halves
    return ((w << 9) & 0x1E001E00u) | ((w >> 3) & 0x01E001E0u);
}
__device__ __forceinline__ uint32_t add_word(uint32_t packed, uint32_t base, int which) {
    uint32_t r;
    if (which & 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "=v"(r) : "v"(packed), "v"(base));
    else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD" : "=v"(r) : "v"(packed), "v"(base));
    return r;
}
__device__ __forceinline__ uint32_t add_byte(uint32_t packed, uint32_t base, int which) {
    uint32_t r;
    switch (which & 3) {
    case 0: asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(packed), "v"(base)); break;
    case 1: asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(packed), "v"(base)); break;
    case 2: asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(packed), "v"(base)); break;
    default: asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(packed), "v"(base)); break;
    }
    return r;
}
typedef const __attribute__((address_space(3))) int* lds_cint_ptr;
__device__ __forceinline__ int lds_read_i32(uint32_t addr) { return *(lds_cint_ptr)(uintptr_t)addr; }   // (addr: the 32-bit LDS address, e.g. the low word of a generic pointer into LDS)
// unaligned (2-byte aligned) load of two consecutive i16
__device__ __forceinline__ int load_pair_i16(const short* p) { int v; __builtin_memcpy(&v, p, 4); return v; }

// ------------------------------------------------------------------ shift step with the borders in registers
// One 8-column shift step of a single-chunk block (<= 128 cells) whose four borders stay in VGPRs from step to step
// (scan_block.rs:147-246 with place_block 1083-1228 and the border moves 1003-1061 folded in). (Ad, Ac) is the border pair
// along the step's vector axis (D_col/C_col for a right step, D_row/R_row for a down step), (Pd, Pr) the orthogonal pair,
// which this step shifts by 8 entries and extends by the last cell of each new column. LDS is used for that shift only:
// the re-based pair is written to Pl (D) and Pl + PR_DIST (R), every column appends behind it, and one read at +8 entries
// yields the shifted registers. `sink`: >= 352 bytes of this wave's LDS that nothing reads (lets every lane store the
// column's last cell unpredicated; only the last lane's address is the real one).
struct FastOut { int mx, row, col; int act_max8, pas_max8, corner_new; };

// R11 of a single-chunk column, two cells per lane (avx2.rs:297-338 per 16-cell vector + the carry from vector to vector,
// scan_block.rs:1144-1150; the column's R starts from MIN = 0 above its first cell), from x = D11_open (scan_block.rs:1143): the step inside the
// register, a SCAN-lane max-plus scan on values re-based by lane * 2g, the lanes above applied, the zero-shift-in artefact with the MIN carry
// folded in (FillConsts::vconst_top). No clamp: lane l >= 1 receives pm[l-1] + (l-1) 2g >= R(lane l-1) >= -32768, lane 0 the filler -32768.
template <int SCAN>
__device__ __forceinline__ int fast_scan(int x, const FillConsts& fc) {
    const s16x2 t2 = as_s(adds(x, fc.ge2));
    int r = vmax(x, as_i(s16x2{t2.x, t2.x}));
    const int pm = prefix_max_lanes<SCAN>((int)as_s(r).y - fc.laneKG);
    const s16x2 cs = as_s(add_shr1(pm, fc.lanem1KG));
    return vmax(vmax(r, adds(as_i(s16x2{cs.x, cs.x}), fc.g12)), fc.vconst_top);
}

// LANES: 64 / 32 / 16 = the block is exactly 128 / 64 / 32 cells (lane count, activity tests, trace indices and the depth of
// the scan are then compile-time); 0 = any single-chunk size (nl_in lanes).
// SP (round 5): the instantiation of the special-mode kernels. rz2 = the relative zero in both halves for LOCAL_START (every cell's D is at least that,
// scan_block.rs:1134-1136), MIN otherwise (no effect); local: LOCAL_START -- with TRACE every trace word is followed by its cells' zero-mask word, as
// place_rect writes them. (FREE_QUERY_START_GAPS differs from the plain modes only in row 0 of the matrix: those steps are not taken here.)
template <int KIND, bool TRACE, bool XDROP, int LANES, int PR_DIST, bool SP = false>
__device__ __forceinline__ void fast_rect(const char* table, const FillConsts& fc, int& Ad, int& Ac, int& Pd, int& Pr, short* Pl, short* sink,
                                          int vec_a, int vec_b, unsigned long long colbytes, int nl_in, int corner, int off_add, int loc_thr,
                                          uint32_t* __restrict__ trace_out, FastOut& o, int rz2 = 0, bool local = false) {
    const int lane = lane_id();
    constexpr bool FULL128 = LANES == 64;
    constexpr int SCAN = LANES ? LANES : 64;
    const int nl = LANES ? LANES : nl_in;            // active lanes = block size / 2
    const bool active = FULL128 ? true : lane < nl;
    const int offa = splat(off_add);
    int d = adds(Ad, offa), c = adds(Ac, offa);      // just_offset (scan_block.rs:1003-1012)
    const int pd = adds(Pd, offa), pr = adds(Pr, offa);
    lds_fence();
    *(int*)(Pl + 2 * lane) = pd; *(int*)(Pl + PR_DIST + 2 * lane) = pr;   // (lanes beyond the block write slots nothing reads before it is rewritten)
    // D_corner for a following orthogonal step: the orthogonal border's entry 7, re-based (scan_block.rs:1042)
    o.corner_new = __builtin_amdgcn_readlane(pd, 3) >> 16;
    const ScoreKey<KIND> key = make_key<KIND>(vec_a, vec_b);
    int dmax = 0, tacc = 0, zacc = 0;
    int dcol[STEP];                                  // X-drop: D of every column, kept for the (lazy) location of the maximum
    const bool last_lane = is_lane(nl - 1);          // owns the last cell of every column
    short* last_base = last_lane ? Pl + 2 * nl : sink + lane;
    int sc_next = fetch_score<KIND>(table, key, (int)(colbytes & 0xff));
#pragma unroll
    for (int j = 0; j < STEP; j++) {
        const int sc = sc_next;
        if (j < STEP - 1) sc_next = fetch_score<KIND>(table, key, (int)((colbytes >> (8 * (j + 1))) & 0xff));
        // D00: previous column shifted down one cell (scan_block.rs:1125); only column 0 has a cell above the block
        int prev = wave_shr1_z(d);
        if (j == 0) prev = set_lane0(prev, (int)((uint32_t)corner << 16));
        const int d00 = __builtin_amdgcn_alignbit(d, prev, 16);
        int d11 = adds(d00, sc);
        if (SP) d11 = vmax(d11, rz2);
        const int copen = adds(d, fc.go2);
        const int cn = vmax(adds(c, fc.ge2), copen);
        d11 = vmax(d11, cn);
        const int x = adds(d11, fc.ome2);            // D11_open
        const int r = fast_scan<SCAN>(x, fc);        // (the in-lane step, the scan over the lanes, the artefact constant: see fast_scan)
        const int dn = vmax(d11, r);
        if (TRACE) {
            // the four flags of a cell (D != C, D != R, C != C_open, R != D_open): each is "left side greater", i.e. the sign of
            // a saturating difference; the sign bits of both halves go into the step's trace word with 32-bit logic ops
            // (nibble = nC | nR << 1 | nCo << 2 | nRo << 3; same-box A/B: +1.0 % over masking every difference and or-ing them)
            // (bit-field inserts take the sign bits and leave the rest of the differences behind as garbage below the nibble,
            // which one mask removes when the nibble joins the word: 8 instructions per column besides the four differences)
            const uint32_t sC = (uint32_t)subs(cn, dn), sR = (uint32_t)subs(r, dn), sCo = (uint32_t)subs(copen, cn), sRo = (uint32_t)subs(x, r);
            const uint32_t hi2 = bfi(0x80008000u, sRo, sCo >> 1), lo2 = bfi(0x80008000u, sR, sC >> 1);   // bits 15, 14 of each half
            const uint32_t nib = bfi(0xC000C000u, hi2, lo2 >> 2);                                        // bits 15..12: nRo, nCo, nR, nC
            tacc = (int)(((uint32_t)tacc >> 4) | (nib & 0xF000F000u));                                   // column j ends up in bits 4j .. 4j+3
            if (SP) zacc |= eq01(dn, rz2, fc.ones) << ((j & 3) * 4);                                     // zero mask (scan_block.rs:1184-1187): bit 0 of the cell's nibble position
            if ((j & 3) == 3) {   // (unpredicated: lanes beyond a small block write words that a later store covers, or the slot's slack)
                if (SP && local) { if (active) *(uint2*)(trace_out + 2u * (uint32_t)((j >> 2) * nl + lane)) = uint2{(uint32_t)tacc, (uint32_t)zacc}; }   // (twice as far: predicated)
                else trace_out[(uint32_t)((j >> 2) * nl + lane)] = (uint32_t)tacc;
                tacc = 0; zacc = 0;
            }
        }
        dmax = vmax(dmax, dn);
        dcol[j] = dn;
        d = dn; c = cn;
        // last cell of the column feeds the orthogonal border (scan_block.rs:1213-1214)
        last_base[j] = (short)(d >> 16); last_base[PR_DIST + j] = (short)(r >> 16);
    }
    lds_fence();
    Pd = *(const int*)(Pl + 2 * lane + STEP); Pr = *(const int*)(Pl + PR_DIST + 2 * lane + STEP);   // shift_and_offset (scan_block.rs:1040-1061)
    Ad = d; Ac = c;
    first8_max2(d, Pd, o.act_max8, o.pas_max8);
    if (XDROP) {
        // Rectangle maximum (every half of dmax is >= 0: D_max starts at MIN = 0). Its location -- among the cells equal to
        // the maximum: smallest row % 16, then largest column, then largest row (avx2.rs:271-274 with the last-writer-wins
        // bookkeeping of scan_block.rs:1198-1200) -- only matters if the step raises the best score (loc_thr = the
        // rectangle maximum that ties it), and is then found from the kept columns with lane masks on the scalar side.
        int m32 = max(dmax & 0xffff, (int)((uint32_t)dmax >> 16));
        if (!active) m32 = 0;
        const int M = max_lanes<SCAN>(m32);
        o.mx = M; o.row = 0; o.col = 0;
        if (M > loc_thr) {
            const unsigned long long act = FULL128 ? ~0ull : ((1ull << nl) - 1ull);
            const unsigned long long RL = __ballot((dmax & 0xffff) == M) & act, RH = __ballot((int)((uint32_t)dmax >> 16) == M) & act;
            auto fold8 = [](unsigned long long x) { uint32_t f = (uint32_t)x | (uint32_t)(x >> 32); f |= f >> 16; f |= f >> 8; return f & 0xffu; };
            const uint32_t fL = fold8(RL), fH = fold8(RH);   // bit c: a row 2 (c + 8 m) [+ 1] reaches the maximum
            const uint32_t kL = fL ? 2u * (uint32_t)__builtin_ctz(fL) : 99u, kH = fH ? 2u * (uint32_t)__builtin_ctz(fH) + 1u : 99u;
            const uint32_t k = min(kL, kH);                  // smallest row % 16 among the rows that reach the maximum
            const uint32_t h = k & 1u;
            const unsigned long long rows = (h ? RH : RL) & (0x0101010101010101ull << (k >> 1));
            bool found = false;
#pragma unroll
            for (int j = STEP - 1; j >= 0; j--) {            // the last column in which one of those rows holds the maximum; its largest row
                if (!found) {
                    const unsigned long long cm = (h ? __ballot((int)((uint32_t)dcol[j] >> 16) == M) : __ballot((dcol[j] & 0xffff) == M)) & rows;
                    if (cm) { o.col = j; o.row = 2 * (63 - __builtin_clzll(cm)) + (int)h; found = true; }
                }
            }
        }
    } else {
        int lm = max((int)as_s(dmax).x, (int)as_s(dmax).y);
        if (!active) lm = -32768;
        o.mx = max_lanes<SCAN>(lm); o.row = 0; o.col = 0;
    }
}

// ------------------------------------------------------------------ block fill
// Fills a width x height rectangle column by column (scan_block.rs:1083-1228; for a down shift the caller swaps the
// sequences exactly as the reference does). NCH = chunks of 128 cells per column; below 128 cells NCH = 1 and only
// height / 2 lanes are active. width is a multiple of 8.
// PDIR (KIND_PROFILE only): 1 = vectors along the query, one profile position per column (place_block_profile_right,
// scan_block.rs:612-783 with $right = true); 2 = vectors along the profile, one query residue per column.
// Row tiling of rectangles taller than the register budget (blocks of 4096 .. 32768 cells): the rectangle is filled tile by
// tile, each a place_rect call over TILE rows and all columns; a tile hands its last row (D and R per column) to the tile
// below through two scratch arrays, which is all the recurrence needs across the cut (scan_block.rs:1123-1150: D00 from the
// previous column, the R carry from the row above). The reference's 16-lane vectors never straddle a tile (TILE % 16 == 0).
struct TileCtx {
    int ch_base, nch_total;     // where this tile's chunks sit in the rectangle's trace layout
    bool first, last;           // first: the rows above are outside the rectangle; last: its last row feeds the real orthogonal border
    int corner0;                // !first: D of the cell above the tile in the column left of the rectangle (already re-based)
    const short* topD; const short* topR;   // !first: D and R of the row above, per column (the tile above wrote them; overwritten in place)
    bool break_armed;           // the whole rectangle's early-break condition (scan_block.rs:1216-1224)
    // FREE_QUERY_END_GAPS over tiles: per column, the running maximum of the tracked vector lane over the tiles so far (fqR) and the
    // best value with which a tracked vector tied or raised the maximum of everything above it in its column (fqT); see place_rect
    short* fqR; short* fqT;
};
// uniform value from memory this kernel also writes: a vector load (never the scalar cache), then broadcast
__device__ __forceinline__ int load_uniform_i16(const short* p) {
    uintptr_t a = (uintptr_t)p;
    asm volatile("" : "+v"(a));
    return uni((int)*(const short*)a);
}

// LOC: keep the per-row "last column at the running maximum" bookkeeping and resolve the location of the rectangle maximum
// (X-drop needs it when the step raises the best score; a speculative grow -- ba_driver.hpp -- goes without).
template <int NCH, int KIND, bool TRACE, bool XDROP, int PDIR = 0, bool LOC = XDROP>
__device__ __forceinline__ Best place_rect(const WaveLds& L, const FillConsts& fc, const uint8_t* __restrict__ seqV,
                                           const uint8_t* __restrict__ seqC, uint32_t lenV, uint32_t lenC, uint32_t start_i,
                                           uint32_t start_j, uint32_t width, uint32_t height, short* Dc, short* Cc, short* Dr,
                                           short* Rr, int corner, int rel_zero, int off_add, uint32_t* __restrict__ trace_out,
                                           unsigned long long& cells, unsigned long long* tacc_prof = nullptr,
                                           uint32_t sp = 0, FqeOut* fq = nullptr, const ProfileView* pv = nullptr, const TileCtx* tc = nullptr) {
    BA_TSTAMP(tp0);
    static_assert((KIND == KIND_PROFILE) == (PDIR != 0), "profile rectangles come with a direction");
    const int lane = lane_id();
    const int nl = NCH > 1 ? 64 : (int)(height >> 1);   // active lanes
    const bool active = lane < nl;
    Best res{0, 0, 0};                                   // MIN = 0 (avx2.rs:16)
    if (width == 0 || height == 0) return res;

    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int d[NCH], c[NCH], dmax[NCH], jlast[NCH], tacc[NCH];
    ScoreKey<KIND> key[NCH];
    int fqM = 0, fqJ = 0;            // SP_FQE: running max / last column of the tracked vector lane
    const int offa = splat(off_add);
    int ca[NCH], cb_[NCH];           // the chunk's two vector-axis bytes: all loads issued before any is used
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        ca[ch] = 0; cb_[ch] = 0;
        if (PDIR != 2 && active) { ca[ch] = seqV[start_i + ch * 128 + 2 * lane]; cb_[ch] = seqV[start_i + ch * 128 + 2 * lane + 1]; }
    }
    // profile, vectors along the profile: per-position gap costs of this lane's two cells, swapped as the reference
    // swaps them for its "down" orientation (scan_block.rs:671-682)
    int goCv[PDIR == 2 ? NCH : 1], goRv[PDIR == 2 ? NCH : 1], clRv[PDIR == 2 ? NCH : 1];
    if constexpr (PDIR == 2) {
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            const uint32_t row = start_i + ch * 128 + 2 * lane;
            goCv[ch] = goRv[ch] = clRv[ch] = 0;
            if (active) {
                goCv[ch] = adds(load_pair_i16(pv->goR + row), fc.ge2);
                goRv[ch] = load_pair_i16(pv->goC + row);
                clRv[ch] = load_pair_i16(pv->clC + row);
            }
        }
    }
    // profile, vectors along the query: this column's (and the next one's) per-position gap costs, wave-uniform
    int col_goC = 0, col_clC = 0, col_goR = 0, nxt_goC = 0, nxt_clC = 0, nxt_goR = 0;
    auto profile_col_gaps = [&](uint32_t idx, int& goC, int& clC, int& goR) {
        goC = splat(clamp16((int)pv->goC[idx] + fc.gap_extend)); clC = splat((int)pv->clC[idx]); goR = splat((int)pv->goR[idx]);
    };
    // packed scores of this lane's two cells in every chunk for one column of a profile rectangle
    auto profile_score = [&](uint32_t col, int cbyte, int ch) -> int {
        if constexpr (PDIR == 1) {
            const signed char* row = pv->pos_aa + (uint64_t)col * 32;
            return active ? pk(row[key[ch].a], row[key[ch].b]) : 0;
        } else {
            return active ? load_pair_i16(pv->aa_pos + (uint64_t)(cbyte & 31) * pv->P + start_i + ch * 128 + 2 * lane) : 0;
        }
    };
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        const int r0 = ch * 128 + 2 * lane;
        int dv = 0, cv = 0, a = ca[ch], b = cb_[ch];
        if (active) { dv = *(const int*)(Dc + r0); cv = *(const int*)(Cc + r0); }
        d[ch] = adds(dv, offa);                 // just_offset folded into the load (scan_block.rs:1003-1012)
        c[ch] = adds(cv, offa);
        dmax[ch] = 0; jlast[ch] = 0; tacc[ch] = 0;
        key[ch] = make_key<KIND>(a, b);
    }
    const bool break_armed = tc ? tc->break_armed : (!XDROP && !(sp & SP_FQE) && (start_i + height > lenV));
    const int NCHT = tc ? tc->nch_total : NCH, CHB = tc ? tc->ch_base : 0;   // trace layout of the whole rectangle
    const bool below = tc && !tc->first;                                      // rows above this tile belong to the rectangle
    int top_d_hold = below ? tc->corner0 : 0, top_d_next = 0, top_r_next = 0;
    if (below) { top_d_next = load_uniform_i16(tc->topD); top_r_next = load_uniform_i16(tc->topR); }
    const int rz2 = splat(rel_zero);
    // SP_LOCAL: every trace word is followed by the zero mask of its cells (scan_block.rs:1184-1187) -- bit 0 of the cell's nibble position; the
    // walk's window takes both with the same loads
    const uint32_t tmul = (TRACE && (sp & SP_LOCAL)) ? 2u : 1u;
    int zacc[NCH];
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) zacc[ch] = 0;
    int corner_cur = corner;
    int cvec = PDIR == 1 ? 0 : (int)seqC[start_j + (lane & 7)];   // 8 column bytes at a time, one per lane (lanes 0..7)
    // Profiles, blocks up to 256 cells: the scores (and per-position gap costs) of a whole group of 8 columns are fetched
    // from the pair's image at once -- one round of HBM latency per 8 columns instead of one per column. Vectors along
    // the query: the transposed table aa_pos holds the 8 positions of a residue contiguously (two 16-byte loads per lane
    // and chunk); vectors along the profile: one 4-byte load per column and chunk.
    constexpr bool PGROUP = PDIR != 0 && NCH <= 2;
    uint4 ga[PGROUP && PDIR == 1 ? NCH : 1], gb[PGROUP && PDIR == 1 ? NCH : 1];
    int gs[PGROUP && PDIR == 2 ? STEP : 1][PGROUP && PDIR == 2 ? NCH : 1];
    int gg_goC = 0, gg_clC = 0, gg_goR = 0;      // PDIR == 1: lane k holds the gap costs of the group's k-th column
    auto refill_group = [&](uint32_t j0) {
        if constexpr (PGROUP && PDIR == 1) {
            const uint32_t idx = start_j + j0;
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) {
                ga[ch] = uint4{0, 0, 0, 0}; gb[ch] = uint4{0, 0, 0, 0};
                if (active) {
                    __builtin_memcpy(&ga[ch], pv->aa_pos + (uint64_t)key[ch].a * pv->P + idx, 16);
                    __builtin_memcpy(&gb[ch], pv->aa_pos + (uint64_t)key[ch].b * pv->P + idx, 16);
                }
            }
            gg_goC = (int)pv->goC[idx + (lane & 7)]; gg_clC = (int)pv->clC[idx + (lane & 7)]; gg_goR = (int)pv->goR[idx + (lane & 7)];
        } else if constexpr (PGROUP && PDIR == 2) {
            cvec = (int)seqC[start_j + j0 + (lane & 7)];
#pragma unroll
            for (int k = 0; k < STEP; k++) {
                const int cbyte = __builtin_amdgcn_readlane(cvec, k);
#pragma unroll
                for (int ch = 0; ch < NCH; ch++)
                    gs[k][ch] = active ? load_pair_i16(pv->aa_pos + (uint64_t)(cbyte & 31) * pv->P + start_i + ch * 128 + 2 * lane) : 0;
            }
        }
    };
    int sc_next[NCH];
    if constexpr (!PGROUP) {
        const int cb0 = __builtin_amdgcn_readlane(cvec, 0);
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) sc_next[ch] = PDIR ? profile_score(start_j, cb0, ch) : fetch_score<KIND>(L.table, key[ch], cb0);
        if constexpr (PDIR == 1) profile_col_gaps(start_j, nxt_goC, nxt_clC, nxt_goR);
    }
    BA_TSTAMP(tp1);
    const bool last_lane = is_lane(nl - 1);          // owns the last cell of every column
    // kc: an integral_constant; >= 0 in the grouped profile mode = the column's index inside its group of 8
    auto column = [&](const uint32_t j, auto kc) -> bool {   // returns false when the fill stops early (scan_block.rs:1216-1224)
        constexpr int K = decltype(kc)::value;
        int sc[NCH];
        if constexpr (K >= 0) {
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) {
                if constexpr (PDIR == 1) {
                    const uint32_t wa = K < 2 ? ga[ch].x : (K < 4 ? ga[ch].y : (K < 6 ? ga[ch].z : ga[ch].w));
                    const uint32_t wb = K < 2 ? gb[ch].x : (K < 4 ? gb[ch].y : (K < 6 ? gb[ch].z : gb[ch].w));
                    sc[ch] = pk((int)(wa >> ((K & 1) * 16)), (int)(wb >> ((K & 1) * 16)));
                } else sc[ch] = gs[K][ch];
            }
            if constexpr (PDIR == 1) {
                col_goC = splat(clamp16(__builtin_amdgcn_readlane(gg_goC, K) + fc.gap_extend));
                col_clC = splat(__builtin_amdgcn_readlane(gg_clC, K));
                col_goR = splat(__builtin_amdgcn_readlane(gg_goR, K));
            }
        } else {
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) sc[ch] = sc_next[ch];
        }
        // scores of the next column are fetched while this one is computed
        if (K < 0 && PDIR != 1 && ((j + 1) & 7) == 0 && j + 1 < width) cvec = (int)seqC[start_j + j + 1 + (lane & 7)];
        if constexpr (K < 0 && PDIR == 1) { col_goC = nxt_goC; col_clC = nxt_clC; col_goR = nxt_goR; }
        if (K < 0) {
            const int cbn = __builtin_amdgcn_readlane(cvec, (int)((j + 1) & 7));
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) sc_next[ch] = PDIR ? profile_score(start_j + j + 1, cbn, ch) : fetch_score<KIND>(L.table, key[ch], cbn);
            if constexpr (PDIR == 1) profile_col_gaps(start_j + j + 1, nxt_goC, nxt_clC, nxt_goR);
        }
        // cell (0,0) -- or, with free query start gaps, every cell of row 0 -- starts from the relative zero
        // (scan_block.rs:1130-1136)
        const bool first_cell = ((j == 0 && start_i == 0 && start_j == 0 && !(sp & SP_LOCAL)) || ((sp & SP_FQS_ROW0) && start_i == 0));
        int up_d = (int)((uint32_t)corner_cur << 16);   // D of the cell above the chunk, previous column (hi half)
        corner_cur = 0;
        int carry_r = 0;                                 // R of the cell above the chunk, this column: MIN at the top
        if (below) {   // the tile above supplies both (its values for the next column are fetched before this column overwrites its own)
            up_d = (int)((uint32_t)top_d_hold << 16); carry_r = top_r_next; top_d_hold = top_d_next;
            if (j + 1 < width) { top_d_next = load_uniform_i16(tc->topD + j + 1); top_r_next = load_uniform_i16(tc->topR + j + 1); }
        }
        const int jp1 = splat((int)j + 1);
        int r_last = 0;
        int fq_cm = -32768, fq_rec = -32768;   // (SP_FQE over tiles, see below)
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            // D00: previous column shifted down one cell (scan_block.rs:1125); lane 0 takes the cell above the chunk
            int prev = wave_shr1_z(d[ch]);
            if (up_d != 0) prev = set_lane0(prev, up_d);
            if (NCH > 1) up_d = __builtin_amdgcn_readlane(d[ch], 63);
            const int d00 = __builtin_amdgcn_alignbit(d[ch], prev, 16);
            int d11 = adds(d00, sc[ch]);
            if (ch == 0 && first_cell) {                 // cell (0,0), scan_block.rs:1130-1132: lane 0, low half
                const int v0 = __builtin_amdgcn_readlane(d11, 0);
                d11 = set_lane0(d11, (v0 & (int)0xffff0000) | (rel_zero & 0xffff));
            }
            if (sp & SP_LOCAL) d11 = vmax(d11, rz2);   // a local alignment may start anywhere (scan_block.rs:1134-1136)
            const int copen = adds(d[ch], PDIR == 0 ? fc.go2 : (PDIR == 1 ? col_goC : goCv[ch]));
            const int cn = vmax(adds(c[ch], fc.ge2), copen);
            const int cend = PDIR == 1 ? adds(cn, col_clC) : cn;                       // C11_end (scan_block.rs:697-701)
            d11 = vmax(d11, cend);
            const int x = adds(d11, PDIR == 0 ? fc.ome2 : (PDIR == 1 ? col_goR : goRv[ch]));   // D11_open
            // R11: in-lane step, then the 64-lane scan on values re-based by lane * 2g
            const s16x2 t2 = as_s(adds(x, fc.ge2));
            int r = vmax(x, as_i(s16x2{t2.x, t2.x}));
            const int pm = wave_prefix_max((int)as_s(r).y - fc.laneKG);
            // what the cells above this lane contribute: lanes above in this chunk, or (lane*2g + carry) from above the
            // chunk, which is MIN = 0 at the top of the column
            int cin = add_shr1(pm, fc.lanem1KG);
            cin = max(max(cin, NCH > 1 ? fc.laneKG + carry_r : fc.laneKG), -32768);
            const s16x2 cs = as_s(cin);
            r = vmax(vmax(r, adds(as_i(s16x2{cs.x, cs.x}), fc.g12)), fc.vconst);
            if (NCH > 1) carry_r = (int)(short)(__builtin_amdgcn_readlane(r, 63) >> 16);
            const int rend = PDIR == 2 ? adds(r, clRv[ch]) : r;                        // R11_end (scan_block.rs:712-716)
            const int dn = vmax(d11, rend);
            if (TRACE) {
                const int nC = neq01(dn, cend, fc.ones), nR = neq01(dn, rend, fc.ones);
                const int nCo = neq01(cn, copen, fc.ones), nRo = neq01(r, x, fc.ones);
                // "R opened" (R == D_open; stored as "differs" like the other three) stays with the cell it was computed in; the reference shifts it to the cell
                // below (scan_block.rs:1179-1182) -- the traceback resolves it at the destination of the gap move instead,
                // which saves the lane shift here and the carry between chunks
                int nib = pk_mad_k<2>(nR, nC);
                nib = pk_mad_k<4>(nCo, nib);
                nib = pk_mad_k<8>(nRo, nib);
                tacc[ch] |= nib << ((j & 3) * 4);
                if (sp & SP_LOCAL) zacc[ch] |= eq01(dn, rz2, fc.ones) << ((j & 3) * 4);
            }
            if (sp & SP_FQE) {
                // D_max / argmax_j of vector lane k = len % 16 in the reference's visiting order (columns outer, vectors
                // inner; scan_block.rs:1189-1201): a running max over that lane's cells of this column, seeded with the
                // max of all earlier columns; a tracked vector records the column when it ties or raises it
                const int k = (int)(lenV & 15);
                const bool mine = active && (lane & 7) == (k >> 1);
                const int v = mine ? ((k & 1) ? (int)as_s(dn).y : (int)as_s(dn).x) : -32768;
                const int pm = wave_prefix_max(v);
                const uint32_t vec_base = start_i + ch * 128 + ((2 * lane) & ~15);
                if (tc) {
                    // Row tiles visit the cells tile by tile, not column by column: here only what this tile's part of the column
                    // says -- its maximum (fq_cm) and the best value with which one of its tracked vectors tied or raised everything
                    // above it in the tile (fq_rec) --; the columns are put together after the last tile (Aligner::run).
                    const bool rec = mine && vec_base + 16 > lenV && v == max(pm, fq_cm);
                    fq_rec = max(fq_rec, wave_max(rec ? v : -32768));
                    fq_cm = max(fq_cm, __builtin_amdgcn_readlane(pm, 63));
                } else {
                const bool hit = mine && vec_base + 16 > lenV && v == max(pm, fqM);
                if (__any(hit)) fqJ = (int)j;
                fqM = max(fqM, __builtin_amdgcn_readlane(pm, 63));
                }
            }
            dmax[ch] = vmax(dmax[ch], dn);
            if (LOC) {   // jlast = 1 + last column whose cell ties or raises its row's running max
                jlast[ch] = vmaxu(jlast[ch], pk_mul(eq01(dmax[ch], dn, fc.ones), jp1));
            }
            d[ch] = dn; c[ch] = cn;
            if (ch == NCH - 1) r_last = r;
        }
        if (TRACE && (j & 3) == 3) {
            if (active) {
#pragma unroll
                for (int ch = 0; ch < NCH; ch++) {
                    const uint32_t at = (((j >> 2) * NCHT + CHB + ch) * nl + lane) * tmul;
                    if (sp & SP_LOCAL) *(uint2*)(trace_out + at) = uint2{(uint32_t)tacc[ch], (uint32_t)zacc[ch]};
                    else trace_out[at] = (uint32_t)tacc[ch];
                }
            }
#pragma unroll
            for (int ch = 0; ch < NCH; ch++) { tacc[ch] = 0; zacc[ch] = 0; }
        }
        // last cell of the column feeds the orthogonal border (scan_block.rs:1213-1214)
        if (last_lane) { Dr[j] = (short)(d[NCH - 1] >> 16); Rr[j] = (short)(r_last >> 16); }
        if ((sp & SP_FQE) && tc) {   // this tile's share of column j into the per-column arrays (a record of the tile counts if it also beats the tiles above)
            const int Rb = tc->first ? -32768 : load_uniform_i16(tc->fqR + j), Tb = tc->first ? -32768 : load_uniform_i16(tc->fqT + j);
            if (last_lane) { tc->fqT[j] = (short)max(Tb, fq_rec >= Rb ? fq_rec : -32768); tc->fqR[j] = (short)max(Rb, fq_cm); }
        }
        cells += height;
        if (break_armed && start_j + j >= lenC) {   // scan_block.rs:1216-1224
            if (TRACE && (j & 3) != 3 && active) {
#pragma unroll
                for (int ch = 0; ch < NCH; ch++) {
                    const uint32_t at = (((j >> 2) * NCHT + CHB + ch) * nl + lane) * tmul;
                    if (sp & SP_LOCAL) *(uint2*)(trace_out + at) = uint2{(uint32_t)tacc[ch], (uint32_t)zacc[ch]};
                    else trace_out[at] = (uint32_t)tacc[ch];
                }
            }
            return false;
        }
        return true;
    };
    using NoGroup = std::integral_constant<int, -1>;
    if constexpr (PGROUP) {
        bool go = true;
        for (uint32_t j0 = 0; go && j0 < width; j0 += STEP) {   // (rectangle widths are multiples of 8)
            refill_group(j0);
#define BA_GCOL(K) if (go) go = column(j0 + K, std::integral_constant<int, K>{})
            BA_GCOL(0); BA_GCOL(1); BA_GCOL(2); BA_GCOL(3); BA_GCOL(4); BA_GCOL(5); BA_GCOL(6); BA_GCOL(7);
#undef BA_GCOL
        }
    } else {
        for (uint32_t j = 0; j < width; j++) { if (!column(j, NoGroup{})) break; }
    }
    BA_TSTAMP(tp2);
    if (fq) { fq->M = fqM; fq->j = fqJ; }
    // ---- write the vector-axis border back
    if (active) {
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) { *(int*)(Dc + ch * 128 + 2 * lane) = d[ch]; *(int*)(Cc + ch * 128 + 2 * lane) = c[ch]; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- rectangle max and (X-drop) its location: among cells equal to the max, smallest (row % 16), then
    // largest column, then largest row (avx2.rs:271-274 + scan_block.rs:1198-1200 last-writer-wins per lane)
    int lm = -32768;
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) lm = max(lm, max((int)as_s(dmax[ch]).x, (int)as_s(dmax[ch]).y));
    if (!active) lm = -32768;
    const int M = wave_max(lm);
    res.mx = M;
    if (LOC) {
        int kmin = 0x7fffffff;
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int v = h ? (int)as_s(dmax[ch]).y : (int)as_s(dmax[ch]).x;
                const int jl1 = h ? (jlast[ch] >> 16) & 0xffff : jlast[ch] & 0xffff;
                const int jl = jl1 ? jl1 - 1 : 0;
                const int row = ch * 128 + 2 * lane + h;
                const int k = ((row & 15) << 24) | ((4095 - jl) << 12) | (4095 - row);
                if (active && v == M) kmin = min(kmin, k);
            }
        }
        kmin = wave_min(kmin);
        res.col = 4095 - ((kmin >> 12) & 4095);
        res.row = 4095 - (kmin & 4095);
    }
    BA_TSTAMP(tp3);
    BA_TADD(tacc_prof, NCH == 1 ? 4 : 8, tp0, tp1);
    BA_TADD(tacc_prof, NCH == 1 ? 5 : 9, tp1, tp2);
    BA_TADD(tacc_prof, NCH == 1 ? 6 : 10, tp2, tp3);
    return res;
}


// ------------------------------------------------------------------ tall rectangles, eight cells per lane
// Rectangles of 512 / 1024 / 2048 rows (the grow steps that end every X-drop alignment of the 10 kbp configuration: 29 % of its
// cells) with EIGHT cells per lane: a column is cut into chunks of 512 cells, lane l of a chunk owning cells 8l .. 8l+7 as four
// packed registers. The gap scan along the column (avx2.rs:297-338 + scan_block.rs:1144-1150) is an in-lane chain over the four
// registers followed by ONE 64-lane DPP scan per 512 cells (place_rect: one per 128), with a scalar carry between chunks. Same
// recurrences, same trace words (a lane's four registers are four consecutive words of place_rect's layout: word ((col >> 2) *
// (rows / 128) + row / 128) * 64 + (row % 128) / 2), same results; sequence kinds only, no special modes (those stay with place_rect).
template <int NC8, int KIND, bool TRACE, bool XDROP, bool LOC = XDROP>
__device__ __forceinline__ Best place_rect8(const WaveLds& L, const FillConsts& fc, const uint8_t* __restrict__ seqV, const uint8_t* __restrict__ seqC,
                                            uint32_t lenV, uint32_t lenC, uint32_t start_i, uint32_t start_j, uint32_t width, short* Dc, short* Cc,
                                            short* Dr, short* Rr, int corner, int rel_zero, int off_add, uint32_t* __restrict__ trace_out, unsigned long long& cells) {
    static_assert(KIND != KIND_PROFILE, "sequence kinds only");
    constexpr uint32_t height = NC8 * 512u;
    constexpr int NCH = NC8 * 4;   // chunks of 128 cells in the trace layout
    const int lane = lane_id();
    Best res{0, 0, 0};
    if (width == 0) return res;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int g = fc.gap_extend;
    int laneKG = lane * 8 * g, lanem1KG = lane ? (lane - 1) * 8 * g : -32768;
    int G[4];
#pragma unroll
    for (int k = 0; k < 4; k++) G[k] = pk(max(-32768, (2 * k + 1) * g), max(-32768, (2 * k + 2) * g));
    // zero shift-in artefacts of the reference's in-vector scan (avx2.rs:315-338; see k_align): R of a cell is at least ((cell & 7) + 1) g -- for a
    // lane's first seven cells that is the wave-uniform G[k], for the eighth 12g in even lanes (vector cell 7) and nothing in odd ones (cell 15).
    // max(cs + G, G) = max(cs, 0) + G (saturating adds of non-positive constants compose): the carry that enters the lane is floored once per
    // column at {0, 4g | none} instead of a max with the artefact per register (round 5: 3 instructions fewer per 512-cell column)
    int w0 = pk(0, (lane & 1) ? -32768 : 4 * g);
    const int offa = splat(off_add);
    int d[NC8][4], c[NC8][4], dmax[NC8][4], jlast[NC8][4], tacc[NC8][4];
    ScoreKey<KIND> key[NC8][4];
    uint32_t vb[NC8][2];
#pragma unroll
    for (int c8 = 0; c8 < NC8; c8++) {   // all loads issued before any is used
        const uint32_t* vp = (const uint32_t*)(seqV + start_i + c8 * 512 + 8 * lane);   // (positions are multiples of 8, images 4-byte aligned)
        vb[c8][0] = vp[0]; vb[c8][1] = vp[1];
    }
#pragma unroll
    for (int c8 = 0; c8 < NC8; c8++) {
        const int4 dv = *(const int4*)(Dc + c8 * 512 + 8 * lane), cv = *(const int4*)(Cc + c8 * 512 + 8 * lane);
        d[c8][0] = adds(dv.x, offa); d[c8][1] = adds(dv.y, offa); d[c8][2] = adds(dv.z, offa); d[c8][3] = adds(dv.w, offa);   // just_offset (scan_block.rs:1003-1012)
        c[c8][0] = adds(cv.x, offa); c[c8][1] = adds(cv.y, offa); c[c8][2] = adds(cv.z, offa); c[c8][3] = adds(cv.w, offa);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            dmax[c8][k] = 0; jlast[c8][k] = 0; tacc[c8][k] = 0;
            const uint32_t w = vb[c8][k >> 1];
            key[c8][k] = make_key<KIND>((int)((w >> (16 * (k & 1))) & 0xffu), (int)((w >> (16 * (k & 1) + 8)) & 0xffu));
        }
    }
    // (scan_block.rs:1216-1224: a column at or beyond the end of the column sequence is the last one computed when the rectangle
    // reaches below the vector sequence's end -- known before the loop)
    if (!XDROP && start_i + height > lenV) width = min(width, (lenC > start_j ? lenC - start_j : 0u) + 1u);
    // Chunks in a skewed order: iteration t takes chunk c8 through column t - c8. A chunk needs from the chunk above it the R carry of
    // the same column and the last D of the previous column -- both produced one iteration earlier -- so the NC8 chunk computations of
    // an iteration are independent of each other and their dependent chains (in-lane chain, 64-lane scan) interleave; in column
    // order chunk c8 + 1 waits for chunk c8's scan. Same operations on the same values.
    // Column bytes: a window of 16 (lanes 0 .. 15), the 8 columns around the leading chunk's next column and the 8 before them.
    auto load_cols = [&](int base) { const int col = max(base + (lane & 15), 0); return (int)seqC[start_j + (uint32_t)col]; };
    int cbase = -8;
    int cvec = load_cols(cbase);
    int sc_next[NC8][4];
    {
        const int cb0 = __builtin_amdgcn_readlane(cvec, 8);   // column 0
#pragma unroll
        for (int c8 = 0; c8 < NC8; c8++)
#pragma unroll
            for (int k = 0; k < 4; k++) sc_next[c8][k] = fetch_score<KIND>(L.table, key[c8][k], cb0);
    }
    const bool last_lane = is_lane(63);
    const uint32_t tw_lane = (uint32_t)((lane >> 4) * 64 + (lane & 15) * 4);   // this lane's four words inside a 512-cell chunk of one column group
    int up_cap[NC8], carry_cap[NC8];   // chunk c8's last D before / last R after its column of the previous iteration: for chunk c8 + 1
#pragma unroll
    for (int c8 = 0; c8 < NC8; c8++) { up_cap[c8] = 0; carry_cap[c8] = 0; }
    const uint32_t iters = width + (uint32_t)NC8 - 1u;
    // (GUARD: the first and last NC8 - 1 iterations, where some chunk has no column; the iterations between run every chunk without a
    // branch in between -- one basic block, which is what lets the scheduler interleave the chunks)
    auto iteration = [&](uint32_t t, auto guard_c) {
        constexpr bool GUARD = decltype(guard_c)::value;
        // (the three per-lane constants of the scan are kept out of the allocator's list of cheap things to park in scratch memory: as
        // loop invariants they were, in some builds, reloaded in the middle of an iteration, behind every outstanding memory operation --
        // config 3 176 against 184 ms depending on unrelated code elsewhere in the kernel. The same marks in fast_rect / place_rect:
        // no gain there, and the score-only kernels lose 2.5 %)
        asm volatile("" : "+v"(laneKG), "+v"(lanem1KG), "+v"(w0));
        if (((t + 1) & 7) == 0) { cbase = (int)(t + 1) - 8; cvec = load_cols(cbase); }   // (t + 1 - c8 >= cbase for every chunk: NC8 <= 4)
        int d_last = 0, r_last = 0;
#pragma unroll
        for (int c8 = NC8 - 1; c8 >= 0; c8--) {   // (descending: a chunk reads what the chunk above it left in the previous iteration)
            const uint32_t jc = t - (uint32_t)c8;
            if (GUARD && jc >= width) continue;   // (also t < c8)
            int sc[4];
#pragma unroll
            for (int k = 0; k < 4; k++) sc[k] = sc_next[c8][k];
            {   // scores of this chunk's next column are fetched while this one is computed
                const int cbn = __builtin_amdgcn_readlane(cvec, (int)(jc + 1) - cbase);
#pragma unroll
                for (int k = 0; k < 4; k++) sc_next[c8][k] = fetch_score<KIND>(L.table, key[c8][k], cbn);
            }
            // D of the cell above the chunk, previous column (hi half): the corner for the first column of the top chunk
            const int up_d = c8 == 0 ? (jc == 0 ? (int)((uint32_t)corner << 16) : 0) : up_cap[c8 - 1];
            const int carry_r = c8 == 0 ? 0 : carry_cap[c8 - 1];   // R of the cell above the chunk, this column: MIN at the top
            const int jp1 = splat((int)jc + 1);
            // D00: previous column shifted down one cell (scan_block.rs:1125); lane 0 takes the cell above the chunk
            int prev = wave_shr1_z(d[c8][3]);
            prev = set_lane0(prev, up_d);          // (0 where there is no cell above: what the shift put there anyway; no branch)
            if (NC8 > 1 && c8 < NC8 - 1) up_cap[c8] = __builtin_amdgcn_readlane(d[c8][3], 63);
            int d00[4];
            d00[0] = __builtin_amdgcn_alignbit(d[c8][0], prev, 16);
#pragma unroll
            for (int k = 1; k < 4; k++) d00[k] = __builtin_amdgcn_alignbit(d[c8][k], d[c8][k - 1], 16);
            int d11[4], copen[4], cn[4], x[4], r[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                d11[k] = adds(d00[k], sc[k]);
                if ((GUARD || NC8 == 1) && c8 == 0 && k == 0 && jc == 0 && start_i == 0 && start_j == 0) {   // (NC8 > 1: column 0 of the top chunk is a guarded iteration)   // cell (0,0) starts from the relative zero (scan_block.rs:1130-1132): lane 0, low half
                    const int v0 = __builtin_amdgcn_readlane(d11[0], 0);
                    d11[0] = set_lane0(d11[0], (v0 & (int)0xffff0000) | (rel_zero & 0xffff));
                }
                copen[k] = adds(d[c8][k], fc.go2);
                cn[k] = vmax(adds(c[c8][k], fc.ge2), copen[k]);
                d11[k] = vmax(d11[k], cn[k]);
                x[k] = adds(d11[k], fc.ome2);                                      // D11_open
                const s16x2 t2 = as_s(adds(x[k], fc.ge2));
                r[k] = vmax(x[k], as_i(s16x2{t2.x, t2.x}));                        // inside the register
            }
#pragma unroll
            for (int k = 1; k < 4; k++) { const s16x2 pr = as_s(r[k - 1]); r[k] = vmax(r[k], adds(as_i(s16x2{pr.y, pr.y}), G[0])); }   // the chain over the lane's registers
            const int pm = wave_prefix_max((int)as_s(r[3]).y - laneKG);           // one 64-lane scan on values re-based by lane * 8g
            // what the cells above this lane contribute: lanes above in this chunk, or (lane * 8g + carry) from above the chunk,
            // which is MIN = 0 at the top of the column
            int cin = add_shr1(pm, lanem1KG);
            cin = max(max(cin, c8 > 0 ? laneKG + carry_r : laneKG), -32768);
            const s16x2 cs = as_s(cin);
            const s16x2 cf = as_s(vmax(as_i(s16x2{cs.x, cs.x}), w0));
            const int csp = as_i(cf), csl = as_i(s16x2{cf.x, cf.x});
#pragma unroll
            for (int k = 0; k < 4; k++) r[k] = vmax(r[k], adds(k < 3 ? csl : csp, G[k]));
            if (NC8 > 1 && c8 < NC8 - 1) carry_cap[c8] = (int)(short)(__builtin_amdgcn_readlane(r[3], 63) >> 16);
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int dn = vmax(d11[k], r[k]);
                if (TRACE) {   // the cell's four flags as sign bits of saturating differences (see fast_rect)
                    const uint32_t sC = (uint32_t)subs(cn[k], dn), sR = (uint32_t)subs(r[k], dn), sCo = (uint32_t)subs(copen[k], cn[k]), sRo = (uint32_t)subs(x[k], r[k]);
                    const uint32_t hi2 = bfi(0x80008000u, sRo, sCo >> 1), lo2 = bfi(0x80008000u, sR, sC >> 1);
                    const uint32_t nib = bfi(0xC000C000u, hi2, lo2 >> 2);
                    tacc[c8][k] = (int)(((uint32_t)tacc[c8][k] >> 4) | (nib & 0xF000F000u));
                }
                dmax[c8][k] = vmax(dmax[c8][k], dn);
                if (LOC) jlast[c8][k] = vmaxu(jlast[c8][k], pk_mul(eq01(dmax[c8][k], dn, fc.ones), jp1));   // 1 + last column whose cell ties or raises its row's running max
                d[c8][k] = dn; c[c8][k] = cn[k];
            }
            if (TRACE && ((jc & 3) == 3 || jc + 1 == width)) {
                // (a last, unfinished column group -- only after the early break: its columns so far sit in the high nibbles)
                const int sh = 4 * (3 - (int)(jc & 3));
                *(int4*)(trace_out + ((jc >> 2) * NCH + c8 * 4) * 64 + tw_lane) =
                    int4{(int)((uint32_t)tacc[c8][0] >> sh), (int)((uint32_t)tacc[c8][1] >> sh), (int)((uint32_t)tacc[c8][2] >> sh), (int)((uint32_t)tacc[c8][3] >> sh)};
                tacc[c8][0] = tacc[c8][1] = tacc[c8][2] = tacc[c8][3] = 0;
            }
            if (c8 == NC8 - 1) { d_last = d[NC8 - 1][3]; r_last = r[3]; }
        }
        // last cell of the column feeds the orthogonal border (scan_block.rs:1213-1214)
        const uint32_t jl = t - (uint32_t)(NC8 - 1);
        if (last_lane && (!GUARD || jl < width)) { Dr[jl] = (short)(d_last >> 16); Rr[jl] = (short)(r_last >> 16); }
    };
    {
        uint32_t t = 0;
        for (; t < (uint32_t)(NC8 - 1) && t < iters; t++) iteration(t, std::true_type{});
        for (; t < width; t++) iteration(t, std::false_type{});
        for (; t < iters; t++) iteration(t, std::true_type{});
    }
    cells += (unsigned long long)height * width;
    // ---- write the vector-axis border back
#pragma unroll
    for (int c8 = 0; c8 < NC8; c8++) {
        *(int4*)(Dc + c8 * 512 + 8 * lane) = int4{d[c8][0], d[c8][1], d[c8][2], d[c8][3]};
        *(int4*)(Cc + c8 * 512 + 8 * lane) = int4{c[c8][0], c[c8][1], c[c8][2], c[c8][3]};
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- rectangle max and (X-drop) its location: among cells equal to the max, smallest (row % 16), then largest column, then
    // largest row (avx2.rs:271-274 + scan_block.rs:1198-1200 last-writer-wins per lane)
    int lm = -32768;
#pragma unroll
    for (int c8 = 0; c8 < NC8; c8++)
#pragma unroll
        for (int k = 0; k < 4; k++) lm = max(lm, max((int)as_s(dmax[c8][k]).x, (int)as_s(dmax[c8][k]).y));
    const int M = wave_max(lm);
    res.mx = M;
    if (LOC) {
        int kmin = 0x7fffffff;
#pragma unroll
        for (int c8 = 0; c8 < NC8; c8++)
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int v = h ? (int)as_s(dmax[c8][k]).y : (int)as_s(dmax[c8][k]).x;
                    const int jl1 = h ? (jlast[c8][k] >> 16) & 0xffff : jlast[c8][k] & 0xffff;
                    const int jl = jl1 ? jl1 - 1 : 0;
                    const int row = c8 * 512 + 8 * lane + 2 * k + h;
                    const int kk = ((row & 15) << 24) | ((4095 - jl) << 12) | (4095 - row);
                    if (v == M) kmin = min(kmin, kk);
                }
        kmin = wave_min(kmin);
        res.col = 4095 - ((kmin >> 12) & 4095);
        res.row = 4095 - (kmin & 4095);
    }
    return res;
}

}  // namespace ba
