// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build/load this.
//
// CPU restatement (C++17) of block-aligner v0.5.1's adaptive-block affine-gap aligner, AVX2 (L = 16)
// semantics: driver (shift right/down, grow, shrink, X-drop), block fill, trace store, CIGAR traceback,
// scoring matrices, sequence-to-profile alignment.
//
// Follows (all paths relative to /root/reference):
//   src/scan_block.rs:94-595    align_core            -> Block::align_core
//   src/scan_block.rs:612-783   place_block_profile_* -> Block::place_block_profile
//   src/scan_block.rs:1003-1061 just_offset/prefix_max/suffix_max/shift_and_offset
//   src/scan_block.rs:1083-1228 place_block           -> Block::place_block
//   src/scan_block.rs:1252-1340 Allocated             -> Block members
//   src/scan_block.rs:1344-1692 Trace, cigar_core, blocks
//   src/scan_block.rs:1790-1884 PaddedBytes
//   src/scores.rs:40-338,452-715 matrices, Gaps, AAProfile
//   src/cigar.rs                Cigar / OpLen / Operation
//   src/lib.rs:109-111          percent_len
//
// PARITY STATUS: the reference is Rust and cannot be built in this image (no rustc/cargo), so this
// restatement is pinned by the reference's own known-answer tests only (scan_block.rs:1908-2230,
// avx2.rs:469-489, lib.rs:8-35; transcribed in tests/golden/reference_kats.json). Beyond those vectors
// parity with the Rust binary is unpinned.
//
// The SIMD layer is chosen at compile time: -DBA_ORACLE_SCALAR selects the scalar lane model,
// otherwise the AVX2-intrinsic layer is used.
#pragma once
#ifdef BA_ORACLE_SCALAR
#include "simd_scalar.hpp"
#else
#include "simd_avx2.hpp"
#endif

#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace ba_oracle {

// scan_block.rs:787-790
constexpr size_t STEP = 8;
constexpr int X_DROP_ITER = 2;
constexpr bool SHRINK = true;

[[noreturn]] static inline void ba_fail(const char* msg) { throw std::runtime_error(msg); }
#define BA_REQUIRE(cond, msg) do { if (!(cond)) ba_fail(msg); } while (0)

static inline int16_t clamp16(int32_t x) {  // scan_block.rs:1704-1706
    return (int16_t)(x < -32768 ? -32768 : (x > 32767 ? 32767 : x));
}
static inline uint8_t ascii_upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

// lib.rs:109-111 (f32 arithmetic, round half away from zero)
static inline size_t percent_len(size_t len, float p) {
    float x = std::round(p * (float)len);
    size_t v = (size_t)x;
    if (v < 32) v = 32;
    size_t pw = 1;
    while (pw < v) pw <<= 1;
    return pw < (size_t(1) << 14) ? pw : (size_t(1) << 14);
}

// ---------------------------------------------------------------- scoring (scores.rs)
struct Gaps { int8_t open, extend; };  // scores.rs:333-338

enum class MatKind { AA = 0, NUC = 1, BYTES = 2 };

struct AAMatrix {  // scores.rs:40-135
    static constexpr MatKind KIND = MatKind::AA;
    static constexpr uint8_t NUL = 'A' + 26;
    alignas(32) int8_t scores[27 * 32];
    AAMatrix() { std::memset(scores, 0x80, sizeof scores); }
    static AAMatrix simple(int8_t match, int8_t mismatch) {
        AAMatrix m;
        for (int i = 0; i < 26; i++)
            for (int j = 0; j < 26; j++) m.scores[i * 32 + j] = i == j ? match : mismatch;
        return m;
    }
    void set(uint8_t a, uint8_t b, int8_t s) {
        a = ascii_upper(a); b = ascii_upper(b);
        BA_REQUIRE(a >= 'A' && a <= 'Z' + 1 && b >= 'A' && b <= 'Z' + 1, "AAMatrix::set out of range");
        scores[(a - 'A') * 32 + (b - 'A')] = s;
        scores[(b - 'A') * 32 + (a - 'A')] = s;
    }
    int8_t get(uint8_t a, uint8_t b) const {
        a = ascii_upper(a); b = ascii_upper(b);
        BA_REQUIRE(a >= 'A' && a <= 'Z' + 1 && b >= 'A' && b <= 'Z' + 1, "AAMatrix::get out of range");
        return scores[(a - 'A') * 32 + (b - 'A')];
    }
    static uint8_t convert_char(uint8_t c) {
        c = ascii_upper(c);
        BA_REQUIRE(c >= 'A' && c <= NUL, "AAMatrix::convert_char out of range");
        return (uint8_t)(c - 'A');
    }
    V16 get_scores(uint8_t c, H16 v) const { return h_lookup2(scores + (size_t)c * 32, v); }  // scores.rs:121-127
};

struct NucMatrix {  // scores.rs:142-217
    static constexpr MatKind KIND = MatKind::NUC;
    static constexpr uint8_t NUL = 'Z';
    alignas(32) int8_t scores[8 * 16];
    NucMatrix() { std::memset(scores, 0x80, sizeof scores); }
    static NucMatrix simple(int8_t match, int8_t mismatch) {
        NucMatrix m;
        const uint8_t alpha[5] = {'A', 'T', 'C', 'G', 'N'};
        for (int i = 0; i < 5; i++)
            for (int j = 0; j < 5; j++)
                m.scores[(alpha[i] & 7) * 16 + (alpha[j] & 15)] = i == j ? match : mismatch;
        return m;
    }
    void set(uint8_t a, uint8_t b, int8_t s) {
        a = ascii_upper(a); b = ascii_upper(b);
        BA_REQUIRE(a >= 'A' && a <= 'Z' && b >= 'A' && b <= 'Z', "NucMatrix::set out of range");
        scores[(a & 7) * 16 + (b & 15)] = s;
        scores[(b & 7) * 16 + (a & 15)] = s;
    }
    int8_t get(uint8_t a, uint8_t b) const {
        a = ascii_upper(a); b = ascii_upper(b);
        return scores[(a & 7) * 16 + (b & 15)];
    }
    static uint8_t convert_char(uint8_t c) {
        c = ascii_upper(c);
        BA_REQUIRE(c >= 'A' && c <= NUL, "NucMatrix::convert_char out of range");
        return c;
    }
    V16 get_scores(uint8_t c, H16 v) const { return h_lookup1(scores + (size_t)(c & 7) * 16, v); }  // scores.rs:204-209
};

struct ByteMatrix {  // scores.rs:220-273
    static constexpr MatKind KIND = MatKind::BYTES;
    static constexpr uint8_t NUL = 0;
    int8_t match_score, mismatch_score;
    static ByteMatrix simple(int8_t match, int8_t mismatch) { return {match, mismatch}; }
    int8_t get(uint8_t a, uint8_t b) const { return a == b ? match_score : mismatch_score; }
    static uint8_t convert_char(uint8_t c) { return c; }
    V16 get_scores(uint8_t c, H16 v) const { return h_lookup_bytes(match_score, mismatch_score, c, v); }  // scores.rs:263-267
};

// scan_block.rs:1790-1884
struct PaddedBytes {
    std::vector<uint8_t> s;
    size_t len_ = 0;
    template <class M>
    static PaddedBytes from_bytes(const uint8_t* b, size_t n, size_t block_size) {
        PaddedBytes p;
        p.s.assign(1 + n + block_size, M::convert_char(M::NUL));
        for (size_t i = 0; i < n; i++) p.s[1 + i] = M::convert_char(b[i]);
        p.len_ = n;
        return p;
    }
    template <class M>
    static PaddedBytes from_bytes_rev(const uint8_t* b, size_t n, size_t block_size) {
        PaddedBytes p;
        p.s.assign(1 + n + block_size, M::convert_char(M::NUL));
        for (size_t i = 0; i < n; i++) p.s[1 + i] = M::convert_char(b[n - 1 - i]);
        p.len_ = n;
        return p;
    }
    uint8_t get(size_t i) const { return s[i]; }
    const uint8_t* ptr(size_t i) const { return s.data() + i; }
    size_t len() const { return len_; }
};

// scores.rs:452-715 (position-specific scoring matrix over A..Z)
struct AAProfile {
    static constexpr uint8_t NUL = 'A' + 26;
    std::vector<int16_t> aa_pos;   // [32][curr_len]
    std::vector<int8_t> pos_aa;    // [curr_len][32]
    int8_t gap_extend;
    std::vector<int16_t> pos_gap_open_C, pos_gap_close_C, pos_gap_open_R;
    size_t max_len, curr_len, str_len;

    AAProfile(size_t str_len_, size_t block_size, int8_t gap_extend_)
        : gap_extend(gap_extend_), max_len(str_len_ + block_size + 1), curr_len(max_len), str_len(str_len_) {
        aa_pos.assign(32 * max_len, (int16_t)-128);
        pos_aa.assign(max_len * 32, (int8_t)-128);
        pos_gap_open_C.assign(max_len, (int16_t)-128);
        pos_gap_close_C.assign(max_len, (int16_t)-128);
        pos_gap_open_R.assign(max_len, (int16_t)-128);
    }
    static AAProfile from_bytes(const uint8_t* b, size_t n, size_t block_size, int8_t match, int8_t mismatch,
                                int8_t gap_open_C, int8_t gap_close_C, int8_t gap_open_R, int8_t gap_extend) {
        AAProfile p(n, block_size, gap_extend);
        for (size_t i = 0; i < n; i++)
            for (uint8_t c = 'A'; c <= 'Z'; c++) p.set(i + 1, c, c == b[i] ? match : mismatch);
        for (size_t i = 0; i < n + 1; i++) {
            p.set_gap_open_C(i, gap_open_C);
            p.set_gap_close_C(i, gap_close_C);
            p.set_gap_open_R(i, gap_open_R);
        }
        return p;
    }
    size_t len() const { return str_len; }
    void clear(size_t str_len_, size_t block_size) {
        size_t cl = str_len_ + block_size + 1;
        BA_REQUIRE(cl <= max_len, "AAProfile::clear exceeds allocation");
        std::fill(aa_pos.begin(), aa_pos.begin() + 32 * cl, (int16_t)-128);
        std::fill(pos_aa.begin(), pos_aa.begin() + cl * 32, (int8_t)-128);
        std::fill(pos_gap_open_C.begin(), pos_gap_open_C.begin() + cl, (int16_t)-128);
        std::fill(pos_gap_close_C.begin(), pos_gap_close_C.begin() + cl, (int16_t)-128);
        std::fill(pos_gap_open_R.begin(), pos_gap_open_R.begin() + cl, (int16_t)-128);
        str_len = str_len_;
        curr_len = cl;
    }
    void set(size_t i, uint8_t b, int8_t score) {
        b = ascii_upper(b);
        BA_REQUIRE(b >= 'A' && b <= 'Z' + 1, "AAProfile::set out of range");
        pos_aa[i * 32 + (b - 'A')] = score;
        aa_pos[(size_t)(b - 'A') * curr_len + i] = score;
    }
    // scores.rs:677-714
    void set_all(const uint8_t* order, size_t order_len, const int8_t* scores, size_t scores_len,
                 unsigned left_shift, unsigned right_shift, bool rev) {
        BA_REQUIRE(order_len <= 32 && order_len > 0, "order too long");
        uint8_t o[32];
        for (int k = 0; k < 32; k++) o[k] = NUL - 'A';
        for (size_t k = 0; k < order_len; k++) {
            uint8_t b = ascii_upper(order[k]);
            BA_REQUIRE(b >= 'A' && b <= 'Z' + 1, "order byte out of range");
            o[k] = (uint8_t)(b - 'A');
        }
        BA_REQUIRE(scores_len / order_len == str_len, "scores length does not match profile length");
        size_t idx = 0;
        for (size_t n = 0; n < str_len; n++) {
            size_t i = rev ? str_len - n : 1 + n;
            for (size_t j = 0; j < order_len; j++) {
                int8_t sc = (int8_t)((int8_t)((uint8_t)scores[idx] << left_shift) >> right_shift);
                pos_aa[i * 32 + o[j]] = sc;
                aa_pos[(size_t)o[j] * curr_len + i] = sc;
                idx++;
            }
        }
    }
    void set_gap_open_C(size_t i, int8_t g) { BA_REQUIRE(g < 0, "Gap open cost must be negative!"); pos_gap_open_C[i] = g; }
    void set_gap_close_C(size_t i, int8_t g) { pos_gap_close_C[i] = g; }
    void set_gap_open_R(size_t i, int8_t g) { BA_REQUIRE(g < 0, "Gap open cost must be negative!"); pos_gap_open_R[i] = g; }
    void set_all_gap_open_C(int8_t g) { BA_REQUIRE(g < 0, "Gap open cost must be negative!"); std::fill(pos_gap_open_C.begin(), pos_gap_open_C.begin() + str_len + 1, (int16_t)g); }
    void set_all_gap_close_C(int8_t g) { std::fill(pos_gap_close_C.begin(), pos_gap_close_C.begin() + str_len + 1, (int16_t)g); }
    void set_all_gap_open_R(int8_t g) { BA_REQUIRE(g < 0, "Gap open cost must be negative!"); std::fill(pos_gap_open_R.begin(), pos_gap_open_R.begin() + str_len + 1, (int16_t)g); }
    int8_t get(size_t i, uint8_t b) const { b = ascii_upper(b); return pos_aa[i * 32 + (b - 'A')]; }
    int8_t get_gap_extend() const { return gap_extend; }

    V16 get_scores_pos(size_t i, H16 v) const { return h_lookup2(pos_aa.data() + i * 32, v); }             // scores.rs:596-602
    V16 get_scores_aa(size_t i, uint8_t c) const { return v_loadu(aa_pos.data() + (size_t)c * curr_len + i); }  // scores.rs:609-612
};

// ---------------------------------------------------------------- CIGAR (cigar.rs)
enum Operation : uint8_t { OP_SENTINEL = 0, OP_M = 1, OP_EQ = 2, OP_X = 3, OP_I = 4, OP_D = 5 };
struct OpLen { uint8_t op; size_t len; };

struct Cigar {  // stored reversed, element 0 is a sentinel (cigar.rs:42-95)
    std::vector<OpLen> s;
    size_t idx = 1;
    Cigar(size_t query_len, size_t reference_len) : s(query_len + reference_len + 5, OpLen{OP_SENTINEL, 0}) {}
    void clear(size_t query_len, size_t reference_len) {
        if (s.size() < query_len + reference_len + 5) s.resize(query_len + reference_len + 5);
        std::fill(s.begin(), s.begin() + query_len + reference_len + 5, OpLen{OP_SENTINEL, 0});
        idx = 1;
    }
    void add(uint8_t op) {
        idx += (op != s[idx - 1].op);
        s[idx - 1].op = op;
        s[idx - 1].len += 1;
    }
    size_t len() const { return idx - 1; }
    OpLen get(size_t i) const { return s[idx - 1 - i]; }
    std::string to_string() const {
        std::string out;
        for (size_t k = idx; k-- > 1;) {
            char c;
            switch (s[k].op) {
                case OP_M: c = 'M'; break;
                case OP_EQ: c = '='; break;
                case OP_X: c = 'X'; break;
                case OP_I: c = 'I'; break;
                case OP_D: c = 'D'; break;
                default: continue;
            }
            out += std::to_string(s[k].len);
            out += c;
        }
        return out;
    }
};

struct Rectangle { size_t row, col, width, height; };
struct AlignResult { int32_t score; size_t query_idx, reference_idx; };

// ---------------------------------------------------------------- Trace (scan_block.rs:1344-1692)
struct Trace {
    std::vector<int32_t> trace, trace2, zero_mask;
    std::vector<uint64_t> right;
    std::vector<uint32_t> block_start;
    std::vector<uint16_t> block_size;
    size_t trace_idx = 0, block_idx = 0, ckpt_trace_idx = 0, ckpt_block_idx = 0;
    size_t query_len = 0, reference_len = 0;
    bool local_start = false, free_query_start_gaps = false;

    void init(size_t qlen, size_t rlen, size_t max_size, bool local, bool fqs) {
        size_t len = qlen + rlen + 2;
        size_t n = (max_size / L) * (len + max_size * 2);
        trace.assign(n, 0);
        trace2.assign(n, 0);
        right.assign((len + 63) / 64, 0);
        block_start.assign(len * 2, 0);
        block_size.assign(len * 2, 0);
        if (local) zero_mask.assign(n, 0); else zero_mask.clear();
        trace_idx = block_idx = ckpt_trace_idx = ckpt_block_idx = 0;
        query_len = qlen; reference_len = rlen;
        local_start = local; free_query_start_gaps = fqs;
    }
    void clear(size_t qlen, size_t rlen) {
        std::fill(right.begin(), right.end(), 0);
        trace_idx = block_idx = ckpt_trace_idx = ckpt_block_idx = 0;
        query_len = qlen; reference_len = rlen;
    }
    void add_trace(int32_t t, int32_t t2) {
        if (trace_idx >= trace.size()) ba_fail("oracle: trace overflow");
        trace[trace_idx] = t; trace2[trace_idx] = t2; trace_idx++;
    }
    void add_zero_mask(int32_t m) { zero_mask[trace_idx] = m; }
    void add_block(size_t i, size_t j, size_t width, size_t height, bool r) {
        if (block_idx * 2 + 1 >= block_start.size()) ba_fail("oracle: block list overflow");
        block_start[block_idx * 2] = (uint32_t)i;
        block_start[block_idx * 2 + 1] = (uint32_t)j;
        block_size[block_idx * 2] = (uint16_t)height;
        block_size[block_idx * 2 + 1] = (uint16_t)width;
        size_t a = block_idx / 64, b = block_idx % 64;
        right[a] = (right[a] & ~(uint64_t(1) << b)) | ((uint64_t)r << b);
        block_idx++;
    }
    void add_trace_idx(size_t add) { trace_idx += add; }
    void save_ckpt() { ckpt_trace_idx = trace_idx; ckpt_block_idx = block_idx; }
    void restore_ckpt() { trace_idx = ckpt_trace_idx; block_idx = ckpt_block_idx; }

    std::vector<Rectangle> blocks() const {
        std::vector<Rectangle> res;
        for (size_t k = 0; k < block_idx; k++)
            res.push_back({block_start[2 * k], block_start[2 * k + 1], block_size[2 * k + 1], block_size[2 * k]});
        return res;
    }

    // One traceback transition. `table`: 0 = D, 1 = C, 2 = R. scan_block.rs:1532-1558
    struct Move { uint8_t op, di, dj, next; };
    static Move lut(bool right_blk, unsigned t, unsigned t2, unsigned table) {
        if (right_blk) {
            if (table == 1) return (t2 & 1) ? Move{OP_D, 0, 1, 0} : Move{OP_D, 0, 1, 1};
            if (table == 2) return (t2 & 2) ? Move{OP_I, 1, 0, 0} : Move{OP_I, 1, 0, 2};
            if (t == 0) return {OP_M, 1, 1, 0};
            if (t & 1) return (t2 & 1) ? Move{OP_D, 0, 1, 0} : Move{OP_D, 0, 1, 1};
            return (t2 & 2) ? Move{OP_I, 1, 0, 0} : Move{OP_I, 1, 0, 2};
        } else {
            if (table == 2) return (t2 & 1) ? Move{OP_I, 1, 0, 0} : Move{OP_I, 1, 0, 2};
            if (table == 1) return (t2 & 2) ? Move{OP_D, 0, 1, 0} : Move{OP_D, 0, 1, 1};
            if (t == 0) return {OP_M, 1, 1, 0};
            if (t & 1) return (t2 & 1) ? Move{OP_I, 1, 0, 0} : Move{OP_I, 1, 0, 2};
            return (t2 & 2) ? Move{OP_D, 0, 1, 0} : Move{OP_D, 0, 1, 1};
        }
    }

    // scan_block.rs:1482-1672
    void cigar_core(bool eq, size_t i, size_t j, const PaddedBytes* q, const PaddedBytes* r, Cigar& cigar) const {
        BA_REQUIRE(i <= query_len && j <= reference_len, "Traceback cigar end position must be in bounds!");
        if (eq) BA_REQUIRE(q && r, "cigar_eq needs both sequences");
        cigar.clear(i, j);
        size_t bidx = block_idx, tidx = trace_idx;
        unsigned table = 0;
        while (i > 0 || j > 0) {
            size_t bi, bj, bh, bw;
            bool right_blk;
            for (;;) {
                if (bidx == 0) ba_fail("oracle: traceback ran off the block list");
                bidx--;
                bi = block_start[bidx * 2]; bj = block_start[bidx * 2 + 1];
                bh = block_size[bidx * 2]; bw = block_size[bidx * 2 + 1];
                tidx -= bw * bh / L;
                if (i >= bi && j >= bj) { right_blk = (right[bidx / 64] >> (bidx % 64)) & 1; break; }
            }
            while (i >= bi && j >= bj && (i > 0 || j > 0)) {
                if (right_blk && free_query_start_gaps && i == 0) return;
                size_t ci = i - bi, cj = j - bj;
                size_t idx = right_blk ? tidx + ci / L + cj * (bh / L) : tidx + cj / L + ci * (bw / L);
                unsigned sh = (unsigned)(((right_blk ? ci : cj) % L) * 2);
                if (local_start && table == 0) {
                    if (((uint32_t)zero_mask[idx] >> sh) & 1) return;
                }
                unsigned t = ((uint32_t)trace[idx] >> sh) & 3;
                unsigned t2 = ((uint32_t)trace2[idx] >> sh) & 3;
                Move m = lut(right_blk, t, t2, table);
                uint8_t op = m.op;
                if (eq && op == OP_M) op = q->get(i) == r->get(j) ? OP_EQ : OP_X;
                i -= m.di; j -= m.dj; table = m.next;
                cigar.add(op);
            }
        }
    }
    void cigar(size_t i, size_t j, Cigar& c) const { cigar_core(false, i, j, nullptr, nullptr, c); }
    void cigar_eq(const PaddedBytes& q, const PaddedBytes& r, size_t i, size_t j, Cigar& c) const { cigar_core(true, i, j, &q, &r, c); }
};

// 32-byte aligned i16 scratch (scan_block.rs:1714-1783)
struct AlignedBuf {
    int16_t* p = nullptr;
    size_t n = 0;
    void alloc(size_t n_) {
        n = n_ < (size_t)L ? L : n_;
        p = (int16_t*)std::aligned_alloc(32, n * 2);
        std::memset(p, 0, n * 2);
    }
    ~AlignedBuf() { std::free(p); }
    void clear(size_t cnt) { for (size_t i = 0; i < cnt; i++) p[i] = MIN; }
};

enum Direction { DIR_RIGHT, DIR_DOWN, DIR_GROW };

struct Mode {  // the five const generics of Block (scan_block.rs:89)
    bool trace = false, x_drop = false, local_start = false, free_query_start_gaps = false, free_query_end_gaps = false;
};

// ---------------------------------------------------------------- Block
class Block {
  public:
    Mode mode;
    AlignResult res{0, 0, 0};
    Trace trace_;
    uint64_t cells_computed = 0;   // sum over every place_block call of columns iterated x height
    uint64_t steps = 0;
    size_t end_block_size = 0;

    Block(Mode m, size_t query_len, size_t reference_len, size_t max_size)
        : mode(m), alloc_qlen(query_len), alloc_rlen(reference_len), alloc_max(max_size) {
        BA_REQUIRE(max_size != 0 && (max_size & (max_size - 1)) == 0, "Block size must be a power of two!");
        if (m.trace) trace_.init(query_len, reference_len, max_size, m.local_start, m.free_query_start_gaps);
        else trace_.init(0, 0, 0, false, false);
        D_col.alloc(max_size); C_col.alloc(max_size); D_row.alloc(max_size); R_row.alloc(max_size);
        D_col_ckpt.alloc(max_size); C_col_ckpt.alloc(max_size); D_row_ckpt.alloc(max_size); R_row_ckpt.alloc(max_size);
        temp1.alloc(L); temp2.alloc(L);
    }

    // scan_block.rs:847-878
    template <class M>
    void align(const PaddedBytes& q, const PaddedBytes& r, const M& matrix, Gaps gaps, size_t smin, size_t smax, int32_t x_drop) {
        BA_REQUIRE(gaps.open < 0 && gaps.extend < 0, "Gap costs must be negative!");
        BA_REQUIRE(gaps.open < gaps.extend, "Gap open must cost more than gap extend!");
        size_t min_size = smin < (size_t)L ? L : smin, max_size = smax < (size_t)L ? L : smax;
        check_sizes(min_size, max_size, x_drop, q.len());
        clear(q.len(), r.len(), max_size);
        SeqCtx<M> ctx{&q, &r, &matrix, gaps};
        dispatch(ctx, q.len(), r.len(), min_size, max_size, x_drop);
    }
    // scan_block.rs:942-968
    void align_profile(const PaddedBytes& q, const AAProfile& p, size_t smin, size_t smax, int32_t x_drop) {
        BA_REQUIRE(p.get_gap_extend() < 0, "Gap extend cost must be negative!");
        size_t min_size = smin < (size_t)L ? L : smin, max_size = smax < (size_t)L ? L : smax;
        check_sizes(min_size, max_size, x_drop, q.len());
        clear(q.len(), p.len(), max_size);
        ProfCtx ctx{&q, &p};
        dispatch(ctx, q.len(), p.len(), min_size, max_size, x_drop);
    }
    // scan_block.rs:884-902 / 974-992; returns min_size reached or 0 for None
    template <class M>
    size_t align_exp(const PaddedBytes& q, const PaddedBytes& r, const M& matrix, Gaps gaps, size_t smin, size_t smax, int32_t x_drop, int32_t target) {
        size_t min_size = smin < (size_t)L ? L : smin, max_size = smax < (size_t)L ? L : smax;
        while (min_size <= max_size) {
            align(q, r, matrix, gaps, min_size, max_size, x_drop);
            if (res.score >= target) return min_size;
            min_size *= 2;
        }
        return 0;
    }
    size_t align_profile_exp(const PaddedBytes& q, const AAProfile& p, size_t smin, size_t smax, int32_t x_drop, int32_t target) {
        size_t min_size = smin < (size_t)L ? L : smin, max_size = smax < (size_t)L ? L : smax;
        while (min_size <= max_size) {
            align_profile(q, p, min_size, max_size, x_drop);
            if (res.score >= target) return min_size;
            min_size *= 2;
        }
        return 0;
    }
    const Trace& trace() const { BA_REQUIRE(mode.trace, "Block was created without TRACE"); return trace_; }

  private:
    size_t alloc_qlen, alloc_rlen, alloc_max;
    AlignedBuf D_col, C_col, D_row, R_row, D_col_ckpt, C_col_ckpt, D_row_ckpt, R_row_ckpt, temp1, temp2;

    template <class M> struct SeqCtx { const PaddedBytes* q; const PaddedBytes* r; const M* m; Gaps gaps; };
    struct ProfCtx { const PaddedBytes* q; const AAProfile* p; };

    void check_sizes(size_t min_size, size_t max_size, int32_t x_drop, size_t qlen) {
        BA_REQUIRE(min_size < 65535 && max_size < 65535, "Block sizes must be smaller than 2^16 - 1!");
        BA_REQUIRE((min_size & (min_size - 1)) == 0 && (max_size & (max_size - 1)) == 0, "Block sizes must be powers of two!");
        if (mode.x_drop) BA_REQUIRE(x_drop >= 0, "X-drop threshold amount must be nonnegative!");
        BA_REQUIRE(!mode.local_start || !mode.free_query_start_gaps, "Cannot set both LOCAL_START and FREE_QUERY_START_GAPS!");
        BA_REQUIRE(!mode.x_drop || !mode.free_query_end_gaps, "Cannot set both X_DROP and FREE_QUERY_END_GAPS!");
        BA_REQUIRE(!mode.free_query_end_gaps || min_size > qlen, "Min block size must be larger than the query length for FREE_QUERY_END_GAPS!");
    }
    // scan_block.rs:1322-1339
    void clear(size_t qlen, size_t rlen, size_t max_size) {
        BA_REQUIRE(qlen + rlen <= alloc_qlen + alloc_rlen, "sequence lengths exceed Block allocation");
        BA_REQUIRE(max_size <= alloc_max, "max block size exceeds Block allocation");
        trace_.clear(qlen, rlen);
        D_col.clear(max_size); C_col.clear(max_size); D_row.clear(max_size); R_row.clear(max_size);
        D_col_ckpt.clear(max_size); C_col_ckpt.clear(max_size); D_row_ckpt.clear(max_size); R_row_ckpt.clear(max_size);
        temp1.clear(L); temp2.clear(L);
        cells_computed = 0; steps = 0;
    }

    template <class Ctx>
    void dispatch(const Ctx& ctx, size_t qlen, size_t rlen, size_t min_size, size_t max_size, int32_t x_drop) {
        const bool special = mode.local_start || mode.free_query_start_gaps || mode.free_query_end_gaps;
#define BA_GO(T, X, S) align_core<T, X, S>(ctx, qlen, rlen, min_size, max_size, x_drop)
        if (special) {
            if (mode.trace) { if (mode.x_drop) BA_GO(true, true, true); else BA_GO(true, false, true); }
            else { if (mode.x_drop) BA_GO(false, true, true); else BA_GO(false, false, true); }
        } else {
            if (mode.trace) { if (mode.x_drop) BA_GO(true, true, false); else BA_GO(true, false, false); }
            else { if (mode.x_drop) BA_GO(false, true, false); else BA_GO(false, false, false); }
        }
#undef BA_GO
    }

    struct Best3 { V16 D_max, argmax_i, argmax_j; };

    // scan_block.rs:1003-1012
    static void just_offset(size_t block_size, int16_t* b1, int16_t* b2, V16 off_add) {
        for (size_t i = 0; i < block_size; i += L) {
            v_store(b1 + i, v_adds(v_load(b1 + i), off_add));
            v_store(b2 + i, v_adds(v_load(b2 + i), off_add));
        }
    }
    static int16_t prefix_max(const int16_t* b) { return v_prefix_hmax8(v_load(b)); }                       // 1020-1022
    static int16_t suffix_max(const int16_t* b, size_t n) { return v_suffix_hmax2(v_load(b + n - L)); }    // 1030-1032
    // scan_block.rs:1040-1061
    static V16 shift_and_offset(size_t block_size, int16_t* b1, int16_t* b2, const int16_t* t1, const int16_t* t2, V16 off_add) {
        V16 curr1 = v_adds(v_load(b1), off_add);
        V16 corner = v_set1(v_extract(curr1, STEP - 1));
        V16 curr2 = v_adds(v_load(b2), off_add);
        size_t i = 0;
        for (; i + L < block_size; i += L) {
            V16 next1 = v_adds(v_load(b1 + i + L), off_add);
            V16 next2 = v_adds(v_load(b2 + i + L), off_add);
            v_store(b1 + i, v_step(next1, curr1));
            v_store(b2 + i, v_step(next2, curr2));
            curr1 = next1; curr2 = next2;
        }
        v_store(b1 + block_size - L, v_step(v_load(t1), curr1));
        v_store(b2 + block_size - L, v_step(v_load(t2), curr2));
        return corner;
    }

    // ---- seq-seq fill: scan_block.rs:1083-1228. `query`/`reference` are in the function's own
    // orientation (swapped by the caller for Down), vectors run along `query`.
    template <bool TRACE, bool XDROP, bool SPECIAL, class M>
    Best3 place_block(const SeqCtx<M>& ctx, const PaddedBytes& query, const PaddedBytes& reference,
                      size_t start_i, size_t start_j, size_t width, size_t height,
                      int16_t* Dc, int16_t* Cc, int16_t* Dr, int16_t* Rr, V16 D_corner, int16_t relative_zero, bool right) {
        const bool LOCAL = SPECIAL && mode.local_start, FQS = SPECIAL && mode.free_query_start_gaps, FQE = SPECIAL && mode.free_query_end_gaps;
        const V16 gap_open = v_set1(ctx.gaps.open), gap_extend = v_set1(ctx.gaps.extend);
        const ScanConsts sc = v_scan_consts(gap_extend);
        const V16 open_minus_ext = v_subs(gap_open, gap_extend);
        Best3 b{v_set1(MIN), v_set1(0), v_set1(0)};
        if (width == 0 || height == 0) return b;

        for (size_t j = 0; j < width; j++) {
            V16 R01 = v_set1(MIN), D11 = v_set1(MIN), R11 = v_set1(MIN), prev_trace_R = v_set1(0);
            const uint8_t c = reference.get(start_j + j);
            for (size_t i = 0; i < height; i += L) {
                V16 D10 = v_load(Dc + i), C10 = v_load(Cc + i);
                V16 D00 = v_sl1(D10, D_corner);
                D_corner = D10;
                V16 scores = ctx.m->get_scores(c, h_loadu(query.ptr(start_i + i)));
                D11 = v_adds(D00, scores);
                if ((!LOCAL && start_i + i == 0 && start_j + j == 0) || (FQS && right && start_i + i == 0))
                    D11 = v_insert0(D11, relative_zero);
                if (LOCAL) D11 = v_max(D11, v_set1(relative_zero));

                V16 C11_open = v_adds(D10, gap_open);
                V16 C11 = v_max(v_adds(C10, gap_extend), C11_open);
                D11 = v_max(D11, C11);
                V16 D11_open = v_adds(D11, open_minus_ext);
                R11 = v_prefix_scan(D11_open, gap_extend, sc.lane);
                R11 = v_max(R11, v_adds(v_broadcasthi(R01), sc.gap_all));
                D11 = v_max(D11, R11);
                R01 = R11;

                if (TRACE) {
                    V16 tDC = v_cmpeq(D11, C11), tDR = v_cmpeq(D11, R11);
                    const V16 hi_bytes = v_set1((int16_t)0xFF00);
                    uint32_t t = v_movemask8(v_blend8(tDC, tDR, hi_bytes));
                    V16 tmpR = v_cmpeq(R11, D11_open);
                    V16 tR = v_sl1(tmpR, prev_trace_R);
                    uint32_t t2 = v_movemask8(v_blend8(v_cmpeq(C11, C11_open), tR, hi_bytes));
                    prev_trace_R = tmpR;
                    if (LOCAL) trace_.add_zero_mask((int32_t)v_movemask8(v_cmpeq(D11, v_set1(relative_zero))));
                    trace_.add_trace((int32_t)t, (int32_t)t2);
                }
                b.D_max = v_max(b.D_max, D11);
                if (XDROP || (FQE && start_i + i + L > query.len())) {
                    V16 mask = v_cmpeq(b.D_max, D11);
                    b.argmax_i = v_blend8(b.argmax_i, v_set1((int16_t)i), mask);
                    b.argmax_j = v_blend8(b.argmax_j, v_set1((int16_t)j), mask);
                }
                v_store(Dc + i, D11);
                v_store(Cc + i, C11);
            }
            D_corner = v_set1(MIN);
            Dr[j] = v_extract(D11, L - 1);
            Rr[j] = v_extract(R11, L - 1);
            cells_computed += height;
            if (!XDROP && !FQE && start_i + height > query.len() && start_j + j >= reference.len()) {
                if (TRACE) trace_.add_trace_idx((width - 1 - j) * (height / L));
                break;
            }
        }
        return b;
    }
    template <bool TRACE, bool XDROP, bool SPECIAL, class M>
    Best3 place_right(const SeqCtx<M>& ctx, size_t si, size_t sj, size_t w, size_t h, int16_t* Dc, int16_t* Cc, int16_t* Dr, int16_t* Rr, V16 corner, int16_t rz) {
        return place_block<TRACE, XDROP, SPECIAL>(ctx, *ctx.q, *ctx.r, si, sj, w, h, Dc, Cc, Dr, Rr, corner, rz, true);
    }
    template <bool TRACE, bool XDROP, bool SPECIAL, class M>
    Best3 place_down(const SeqCtx<M>& ctx, size_t si, size_t sj, size_t w, size_t h, int16_t* Dc, int16_t* Cc, int16_t* Dr, int16_t* Rr, V16 corner, int16_t rz) {
        return place_block<TRACE, XDROP, SPECIAL>(ctx, *ctx.r, *ctx.q, si, sj, w, h, Dc, Cc, Dr, Rr, corner, rz, false);
    }

    // ---- seq-profile fill: scan_block.rs:612-783. RIGHT: vectors along the query, one profile
    // position per column. !RIGHT: vectors along the profile, one query byte per "column".
    template <bool TRACE, bool XDROP, bool SPECIAL, bool RIGHT>
    Best3 place_block_profile(const ProfCtx& ctx, size_t start_i, size_t start_j, size_t width, size_t height,
                              int16_t* Dc, int16_t* Cc, int16_t* Dr, int16_t* Rr, V16 D_corner, int16_t relative_zero) {
        const bool LOCAL = SPECIAL && mode.local_start, FQS = SPECIAL && mode.free_query_start_gaps, FQE = SPECIAL && mode.free_query_end_gaps;
        const PaddedBytes& q = *ctx.q;
        const AAProfile& p = *ctx.p;
        const V16 gap_extend = v_set1(p.get_gap_extend());
        const ScanConsts sc = v_scan_consts(gap_extend);
        Best3 b{v_set1(MIN), v_set1(0), v_set1(0)};
        V16 gap_open_C = v_set1(MIN), gap_close_C = v_set1(MIN), gap_open_R = v_set1(MIN), gap_close_R = v_set1(MIN);
        if (width == 0 || height == 0) return b;
        // lengths in the function's own orientation (macro args `$query`, `$reference`)
        const size_t own_query_len = RIGHT ? q.len() : p.len();
        const size_t own_reference_len = RIGHT ? p.len() : q.len();

        for (size_t j = 0; j < width; j++) {
            V16 R01 = v_set1(MIN), D11 = v_set1(MIN), R11 = v_set1(MIN), prev_trace_R = v_set1(0);
            size_t idx = 0;
            if (RIGHT) {
                idx = start_j + j;
                gap_open_C = v_set1(p.pos_gap_open_C[idx]);
                gap_close_C = v_set1(p.pos_gap_close_C[idx]);
                gap_open_R = v_set1(p.pos_gap_open_R[idx]);
            }
            for (size_t i = 0; i < height; i += L) {
                V16 D10 = v_load(Dc + i), C10 = v_load(Cc + i);
                V16 D00 = v_sl1(D10, D_corner);
                D_corner = D10;
                if (!RIGHT) {
                    idx = start_i + i;
                    gap_open_C = v_loadu(p.pos_gap_open_R.data() + idx);
                    gap_open_R = v_loadu(p.pos_gap_open_C.data() + idx);
                    gap_close_R = v_loadu(p.pos_gap_close_C.data() + idx);
                }
                V16 scores = RIGHT ? p.get_scores_pos(idx, h_loadu(q.ptr(start_i + i)))
                                   : p.get_scores_aa(idx, q.get(start_j + j));
                D11 = v_adds(D00, scores);
                if ((!LOCAL && start_i + i == 0 && start_j + j == 0) || (FQS && RIGHT && start_i + i == 0))
                    D11 = v_insert0(D11, relative_zero);
                if (LOCAL) D11 = v_max(D11, v_set1(relative_zero));

                V16 C11_open = v_adds(D10, v_adds(gap_open_C, gap_extend));
                V16 C11 = v_max(v_adds(C10, gap_extend), C11_open);
                V16 C11_end = RIGHT ? v_adds(C11, gap_close_C) : C11;
                D11 = v_max(D11, C11_end);
                V16 D11_open = v_adds(D11, gap_open_R);
                R11 = v_prefix_scan(D11_open, gap_extend, sc.lane);
                R11 = v_max(R11, v_adds(v_broadcasthi(R01), sc.gap_all));
                V16 R11_end = RIGHT ? R11 : v_adds(R11, gap_close_R);
                D11 = v_max(D11, R11_end);
                R01 = R11;

                if (TRACE) {
                    V16 tDC = v_cmpeq(D11, C11_end), tDR = v_cmpeq(D11, R11_end);
                    const V16 hi_bytes = v_set1((int16_t)0xFF00);
                    uint32_t t = v_movemask8(v_blend8(tDC, tDR, hi_bytes));
                    V16 tmpR = v_cmpeq(R11, D11_open);
                    V16 tR = v_sl1(tmpR, prev_trace_R);
                    uint32_t t2 = v_movemask8(v_blend8(v_cmpeq(C11, C11_open), tR, hi_bytes));
                    prev_trace_R = tmpR;
                    if (LOCAL) trace_.add_zero_mask((int32_t)v_movemask8(v_cmpeq(D11, v_set1(relative_zero))));
                    trace_.add_trace((int32_t)t, (int32_t)t2);
                }
                b.D_max = v_max(b.D_max, D11);
                if (XDROP || (FQE && start_i + i + L > own_query_len)) {
                    V16 mask = v_cmpeq(b.D_max, D11);
                    b.argmax_i = v_blend8(b.argmax_i, v_set1((int16_t)i), mask);
                    b.argmax_j = v_blend8(b.argmax_j, v_set1((int16_t)j), mask);
                }
                v_store(Dc + i, D11);
                v_store(Cc + i, C11);
            }
            D_corner = v_set1(MIN);
            Dr[j] = v_extract(D11, L - 1);
            Rr[j] = v_extract(R11, L - 1);
            cells_computed += height;
            if (!XDROP && !FQE && start_i + height > own_query_len && start_j + j >= own_reference_len) {
                if (TRACE) trace_.add_trace_idx((width - 1 - j) * (height / L));
                break;
            }
        }
        return b;
    }
    template <bool TRACE, bool XDROP, bool SPECIAL>
    Best3 place_right(const ProfCtx& ctx, size_t si, size_t sj, size_t w, size_t h, int16_t* Dc, int16_t* Cc, int16_t* Dr, int16_t* Rr, V16 corner, int16_t rz) {
        return place_block_profile<TRACE, XDROP, SPECIAL, true>(ctx, si, sj, w, h, Dc, Cc, Dr, Rr, corner, rz);
    }
    template <bool TRACE, bool XDROP, bool SPECIAL>
    Best3 place_down(const ProfCtx& ctx, size_t si, size_t sj, size_t w, size_t h, int16_t* Dc, int16_t* Cc, int16_t* Dr, int16_t* Rr, V16 corner, int16_t rz) {
        return place_block_profile<TRACE, XDROP, SPECIAL, false>(ctx, si, sj, w, h, Dc, Cc, Dr, Rr, corner, rz);
    }

    void copy4(AlignedBuf& a, AlignedBuf& b, AlignedBuf& c, AlignedBuf& d,
               const AlignedBuf& sa, const AlignedBuf& sb, const AlignedBuf& sc, const AlignedBuf& sd, size_t n) {
        std::memcpy(a.p, sa.p, n * 2); std::memcpy(b.p, sb.p, n * 2);
        std::memcpy(c.p, sc.p, n * 2); std::memcpy(d.p, sd.p, n * 2);
    }

    // ---- driver: scan_block.rs:94-595
    template <bool TRACE, bool XDROP, bool SPECIAL, class Ctx>
    void align_core(const Ctx& ctx, size_t qlen, size_t rlen, size_t min_size, size_t max_size, int32_t x_drop) {
        const bool FQE = SPECIAL && mode.free_query_end_gaps;
        size_t si = 0, sj = 0;
        int32_t best_max = 0;
        size_t best_argmax_i = 0, best_argmax_j = 0;
        Direction prev_dir = DIR_GROW, dir = DIR_GROW;
        size_t prev_size = 0, block_size = min_size;
        int32_t off = 0, prev_off, off_max = 0;
        size_t y_drop_iter = 0;
        int x_drop_iter = 0;
        size_t i_ckpt = si, j_ckpt = sj;
        int32_t off_ckpt = 0;
        V16 D_corner = v_set1(MIN);

        for (;;) {
            steps++;
            prev_off = off;
            Best3 grow{v_set1(MIN), v_set1(0), v_set1(0)};
            Best3 cur;
            int16_t right_max, down_max;
            switch (dir) {
                case DIR_RIGHT: {
                    off = off_max;
                    V16 off_add = v_set1(clamp16(prev_off - off));
                    if (TRACE) trace_.add_block(si, sj + block_size - STEP, STEP, block_size, true);
                    just_offset(block_size, D_col.p, C_col.p, off_add);
                    cur = place_right<TRACE, XDROP, SPECIAL>(ctx, si, sj + block_size - STEP, STEP, block_size,
                                                            D_col.p, C_col.p, temp1.p, temp2.p,
                                                            prev_dir == DIR_DOWN ? v_adds(D_corner, off_add) : v_set1(MIN),
                                                            clamp16(-off + (int32_t)ZERO));
                    right_max = prefix_max(D_col.p);
                    D_corner = shift_and_offset(block_size, D_row.p, R_row.p, temp1.p, temp2.p, off_add);
                    down_max = prefix_max(D_row.p);
                    break;
                }
                case DIR_DOWN: {
                    off = off_max;
                    V16 off_add = v_set1(clamp16(prev_off - off));
                    if (TRACE) trace_.add_block(si + block_size - STEP, sj, block_size, STEP, false);
                    just_offset(block_size, D_row.p, R_row.p, off_add);
                    cur = place_down<TRACE, XDROP, SPECIAL>(ctx, sj, si + block_size - STEP, STEP, block_size,
                                                           D_row.p, R_row.p, temp1.p, temp2.p,
                                                           prev_dir == DIR_RIGHT ? v_adds(D_corner, off_add) : v_set1(MIN),
                                                           clamp16(-off + (int32_t)ZERO));
                    down_max = prefix_max(D_row.p);
                    D_corner = shift_and_offset(block_size, D_col.p, C_col.p, temp1.p, temp2.p, off_add);
                    right_max = prefix_max(D_col.p);
                    break;
                }
                default: {  // DIR_GROW
                    D_corner = v_set1(MIN);
                    size_t grow_step = block_size - prev_size;
                    if (TRACE) trace_.add_block(si + prev_size, sj, prev_size, grow_step, false);
                    Best3 g1 = place_down<TRACE, XDROP, SPECIAL>(ctx, sj, si + prev_size, grow_step, prev_size,
                                                                 D_row.p, R_row.p, D_col.p + prev_size, C_col.p + prev_size,
                                                                 v_set1(MIN), clamp16(-off + (int32_t)ZERO));
                    if (TRACE) trace_.add_block(si, sj + prev_size, grow_step, block_size, true);
                    cur = place_right<TRACE, XDROP, SPECIAL>(ctx, si, sj + prev_size, grow_step, block_size,
                                                            D_col.p, C_col.p, D_row.p + prev_size, R_row.p + prev_size,
                                                            v_set1(MIN), clamp16(-off + (int32_t)ZERO));
                    right_max = prefix_max(D_col.p);
                    down_max = prefix_max(D_row.p);
                    grow = g1;
                    copy4(D_col_ckpt, C_col_ckpt, D_row_ckpt, R_row_ckpt, D_col, C_col, D_row, R_row, block_size);
                    if (TRACE) trace_.save_ckpt();
                    break;
                }
            }

            prev_dir = dir;
            int16_t D_max_max = FQE ? v_extract(cur.D_max, (int)(qlen % L)) : v_hmax(cur.D_max);
            int16_t grow_max = v_hmax(grow.D_max);
            int16_t mx = D_max_max > grow_max ? D_max_max : grow_max;
            off_max = off + (int32_t)mx - (int32_t)ZERO;
            y_drop_iter++;
            bool grow_no_max = dir == DIR_GROW;

            if (off_max > best_max) {
                if (FQE) {
                    size_t idx_j = (size_t)(uint16_t)v_extract(cur.argmax_j, (int)(qlen % L));
                    best_argmax_i = qlen;
                    if (dir == DIR_RIGHT) best_argmax_j = sj + (block_size - STEP) + idx_j;
                    else if (dir == DIR_GROW) best_argmax_j = sj + prev_size + idx_j;
                    else ba_fail("oracle: FREE_QUERY_END_GAPS reached a Down step");
                }
                if (XDROP) {
                    int lane = v_hargmax(cur.D_max, D_max_max);
                    // lane == L only when the max came solely from the grow-down rect; the values read
                    // below are then unused (reference reads lane 16 via trailing_zeros()/2 == 16 too,
                    // which would be out of bounds there; it cannot happen on the Right/Down paths).
                    size_t idx_i = lane < L ? (size_t)(uint16_t)v_extract(cur.argmax_i, lane) : 0;
                    size_t idx_j = lane < L ? (size_t)(uint16_t)v_extract(cur.argmax_j, lane) : 0;
                    size_t r = idx_i + (size_t)lane, c = (block_size - STEP) + idx_j;
                    if (dir == DIR_RIGHT) { best_argmax_i = si + r; best_argmax_j = sj + c; }
                    else if (dir == DIR_DOWN) { best_argmax_i = si + c; best_argmax_j = sj + r; }
                    else if (D_max_max >= grow_max) { best_argmax_i = si + idx_i + lane; best_argmax_j = sj + prev_size + idx_j; }
                    else {
                        int l2 = v_hargmax(grow.D_max, grow_max);
                        size_t gi = (size_t)(uint16_t)v_extract(grow.argmax_i, l2), gj = (size_t)(uint16_t)v_extract(grow.argmax_j, l2);
                        best_argmax_i = si + prev_size + gj;
                        best_argmax_j = sj + gi + l2;
                    }
                }
                if (block_size < max_size) {
                    i_ckpt = si; j_ckpt = sj; off_ckpt = off;
                    copy4(D_col_ckpt, C_col_ckpt, D_row_ckpt, R_row_ckpt, D_col, C_col, D_row, R_row, block_size);
                    if (TRACE) trace_.save_ckpt();
                    grow_no_max = false;
                }
                best_max = off_max;
                y_drop_iter = 0;
            }

            if (XDROP) {
                if (off_max < best_max - x_drop) {
                    if (x_drop_iter < X_DROP_ITER - 1) x_drop_iter++;
                    else break;
                } else x_drop_iter = 0;
            }
            if (si + block_size > qlen && sj + block_size > rlen) break;
            if (sj + block_size > rlen) { si += STEP; dir = DIR_DOWN; continue; }
            if (si + block_size > qlen) { sj += STEP; dir = DIR_RIGHT; continue; }

            size_t next_size = block_size * 2;
            if (next_size <= max_size) {
                if (y_drop_iter > (block_size / STEP) - 1 || grow_no_max) {
                    prev_size = block_size;
                    block_size = next_size;
                    dir = DIR_GROW;
                    si = i_ckpt; sj = j_ckpt; off = off_ckpt;
                    copy4(D_col, C_col, D_row, R_row, D_col_ckpt, C_col_ckpt, D_row_ckpt, R_row_ckpt, prev_size);
                    if (TRACE) trace_.restore_ckpt();
                    y_drop_iter = 0;
                    continue;
                }
            }
            if (SHRINK && block_size > min_size && y_drop_iter == 0) {
                int16_t a = suffix_max(D_row.p, block_size), bb = suffix_max(D_col.p, block_size);
                int16_t shrink_max = a > bb ? a : bb;
                if (shrink_max >= mx) {
                    prev_dir = DIR_GROW;
                    block_size /= 2;
                    std::memmove(D_col.p, D_col.p + block_size, block_size * 2);
                    std::memmove(C_col.p, C_col.p + block_size, block_size * 2);
                    std::memmove(D_row.p, D_row.p + block_size, block_size * 2);
                    std::memmove(R_row.p, R_row.p + block_size, block_size * 2);
                    si += block_size; sj += block_size;
                    i_ckpt = si; j_ckpt = sj; off_ckpt = off;
                    copy4(D_col_ckpt, C_col_ckpt, D_row_ckpt, R_row_ckpt, D_col, C_col, D_row, R_row, block_size);
                    right_max = prefix_max(D_col.p);
                    down_max = prefix_max(D_row.p);
                    if (TRACE) trace_.save_ckpt();
                    y_drop_iter = 0;
                }
            }
            if (down_max > right_max) { si += STEP; dir = DIR_DOWN; }
            else { sj += STEP; dir = DIR_RIGHT; }
        }

        end_block_size = block_size;
        if (XDROP || FQE) {
            res = {best_max, best_argmax_i, best_argmax_j};
        } else {
            int32_t score;
            if (dir == DIR_DOWN) score = off + (int32_t)D_row.p[rlen - sj] - (int32_t)ZERO;
            else score = off + (int32_t)D_col.p[qlen - si] - (int32_t)ZERO;
            res = {score, qlen, rlen};
        }
    }
};

}  // namespace ba_oracle
