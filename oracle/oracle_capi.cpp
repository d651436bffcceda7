// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
// Plain C entry points over block_aligner_oracle.hpp for ctypes (tests/, smoke(), bench.py cpu_baseline).
#include "block_aligner_oracle.hpp"

#include <atomic>
#include <chrono>
#include <memory>
#include <thread>

using namespace ba_oracle;

static thread_local std::string g_err;

enum : uint32_t {
    F_TRACE = 1u << 0, F_XDROP = 1u << 1, F_LOCAL = 1u << 2, F_FQS = 1u << 3, F_FQE = 1u << 4, F_CIGAR_EQ = 1u << 5
};

static Mode mode_of(uint32_t flags) {
    Mode m;
    m.trace = flags & F_TRACE; m.x_drop = flags & F_XDROP; m.local_start = flags & F_LOCAL;
    m.free_query_start_gaps = flags & F_FQS; m.free_query_end_gaps = flags & F_FQE;
    return m;
}

namespace {
struct AnyMatrix {
    int kind;
    AAMatrix aa; NucMatrix nuc; ByteMatrix bytes{0, 0};
    AnyMatrix(int k, const int8_t* data) : kind(k) {
        if (k == 0) std::memcpy(aa.scores, data, sizeof aa.scores);
        else if (k == 1) std::memcpy(nuc.scores, data, sizeof nuc.scores);
        else bytes = ByteMatrix{data[0], data[1]};
    }
    PaddedBytes pad(const uint8_t* b, size_t n, size_t block) const {
        if (kind == 0) return PaddedBytes::from_bytes<AAMatrix>(b, n, block);
        if (kind == 1) return PaddedBytes::from_bytes<NucMatrix>(b, n, block);
        return PaddedBytes::from_bytes<ByteMatrix>(b, n, block);
    }
    void align(Block& blk, const PaddedBytes& q, const PaddedBytes& r, Gaps g, size_t mn, size_t mx, int32_t x) const {
        if (kind == 0) blk.align(q, r, aa, g, mn, mx, x);
        else if (kind == 1) blk.align(q, r, nuc, g, mn, mx, x);
        else blk.align(q, r, bytes, g, mn, mx, x);
    }
};
uint64_t surviving_cells(const Trace& t) {
    uint64_t s = 0;
    for (auto& b : t.blocks()) s += (uint64_t)b.width * b.height;
    return s;
}
}  // namespace

extern "C" {

const char* ba_oracle_backend(void) { return simd_backend_name(); }
const char* ba_oracle_last_error(void) { return g_err.c_str(); }

// Lane primitives, for pinning the lane model against the intrinsics (and against avx2.rs:469-489).
// op: 0 prefix_scan(a, g=b[0]) 1 sl1(a,b) 2 step(a,b) 3 broadcasthi(a) 4 hmax(a)->out[0] 5 prefix_hmax8 6 suffix_hmax2
//     7 hargmax(a, b[0]) 8 adds 9 subs 10 gap_all consts(g=b[0]) 11 movemask8(blend8(a,b,0xFF00)) -> out[0..1]
int ba_oracle_lane_op(int op, const int16_t* a, const int16_t* b, int16_t* out) {
    alignas(32) int16_t ta[L], tb[L], to[L] = {0};
    std::memcpy(ta, a, sizeof ta); std::memcpy(tb, b, sizeof tb);
    V16 A = v_load(ta), B = v_load(tb);
    switch (op) {
        case 0: { V16 g = v_set1(tb[0]); ScanConsts c = v_scan_consts(g); v_store(to, v_prefix_scan(A, g, c.lane)); break; }
        case 1: v_store(to, v_sl1(A, B)); break;
        case 2: v_store(to, v_step(A, B)); break;
        case 3: v_store(to, v_broadcasthi(A)); break;
        case 4: to[0] = v_hmax(A); break;
        case 5: to[0] = v_prefix_hmax8(A); break;
        case 6: to[0] = v_suffix_hmax2(A); break;
        case 7: to[0] = (int16_t)v_hargmax(A, tb[0]); break;
        case 8: v_store(to, v_adds(A, B)); break;
        case 9: v_store(to, v_subs(A, B)); break;
        case 10: { ScanConsts c = v_scan_consts(v_set1(tb[0])); v_store(to, c.gap_all); break; }
        case 11: { uint32_t m = v_movemask8(v_blend8(A, B, v_set1((int16_t)0xFF00))); to[0] = (int16_t)(m & 0xFFFF); to[1] = (int16_t)(m >> 16); break; }
        default: return 1;
    }
    std::memcpy(out, to, sizeof to);
    return 0;
}

size_t ba_oracle_percent_len(size_t len, float p) { return percent_len(len, p); }

// One seq-seq alignment. stats (optional, 4 x u64): cells computed, driver steps, final block size,
// surviving-rectangle cells (TRACE only).  Returns 0, or 1 with ba_oracle_last_error() set
// (the reference would panic/abort on the same precondition).
int ba_oracle_align(int kind, const int8_t* matrix, const uint8_t* q, size_t qlen, const uint8_t* r, size_t rlen,
                    int8_t gap_open, int8_t gap_extend, size_t min_size, size_t max_size, int32_t x_drop,
                    uint32_t flags, int32_t* score, uint64_t* qidx, uint64_t* ridx,
                    char* cigar_buf, size_t cigar_cap, uint64_t* stats) {
    try {
        AnyMatrix m(kind, matrix);
        size_t pad = max_size < (size_t)L ? (size_t)L : max_size;
        PaddedBytes pq = m.pad(q, qlen, pad), pr = m.pad(r, rlen, pad);
        Block blk(mode_of(flags), qlen, rlen, pad);
        m.align(blk, pq, pr, Gaps{gap_open, gap_extend}, min_size, max_size, x_drop);
        *score = blk.res.score; *qidx = blk.res.query_idx; *ridx = blk.res.reference_idx;
        if (cigar_buf && cigar_cap) cigar_buf[0] = 0;
        if ((flags & F_TRACE) && cigar_buf) {
            Cigar c(blk.res.query_idx, blk.res.reference_idx);
            if (flags & F_CIGAR_EQ) blk.trace().cigar_eq(pq, pr, blk.res.query_idx, blk.res.reference_idx, c);
            else blk.trace().cigar(blk.res.query_idx, blk.res.reference_idx, c);
            std::string s = c.to_string();
            if (s.size() + 1 > cigar_cap) { g_err = "cigar buffer too small"; return 1; }
            std::memcpy(cigar_buf, s.c_str(), s.size() + 1);
        }
        if (stats) {
            stats[0] = blk.cells_computed; stats[1] = blk.steps; stats[2] = blk.end_block_size;
            stats[3] = (flags & F_TRACE) ? surviving_cells(blk.trace()) : 0;
        }
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// Trace::blocks() of one seq-seq TRACE alignment (scan_block.rs:1676-1691): writes up to `cap` rectangles as
// {row, col, width, height} u64 quadruples in fill order; *count = their number.
int ba_oracle_align_blocks(int kind, const int8_t* matrix, const uint8_t* q, size_t qlen, const uint8_t* r, size_t rlen,
                           int8_t gap_open, int8_t gap_extend, size_t min_size, size_t max_size, int32_t x_drop,
                           uint32_t flags, uint64_t* rects, size_t cap, uint64_t* count) {
    try {
        AnyMatrix m(kind, matrix);
        size_t pad = max_size < (size_t)L ? (size_t)L : max_size;
        PaddedBytes pq = m.pad(q, qlen, pad), pr = m.pad(r, rlen, pad);
        Block blk(mode_of(flags | F_TRACE), qlen, rlen, pad);
        m.align(blk, pq, pr, Gaps{gap_open, gap_extend}, min_size, max_size, x_drop);
        const std::vector<Rectangle> bl = blk.trace().blocks();
        *count = bl.size();
        for (size_t k = 0; k < bl.size() && k < cap; k++) {
            rects[4 * k] = bl[k].row; rects[4 * k + 1] = bl[k].col; rects[4 * k + 2] = bl[k].width; rects[4 * k + 3] = bl[k].height;
        }
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// Exponential search on the min block size (scan_block.rs:884-902). *reached = min size that met the target, or 0.
int ba_oracle_align_exp(int kind, const int8_t* matrix, const uint8_t* q, size_t qlen, const uint8_t* r, size_t rlen,
                        int8_t gap_open, int8_t gap_extend, size_t min_size, size_t max_size, int32_t x_drop,
                        int32_t target, uint32_t flags, int32_t* score, uint64_t* qidx, uint64_t* ridx, uint64_t* reached) {
    try {
        AnyMatrix m(kind, matrix);
        size_t pad = max_size < (size_t)L ? (size_t)L : max_size;
        PaddedBytes pq = m.pad(q, qlen, pad), pr = m.pad(r, rlen, pad);
        Block blk(mode_of(flags), qlen, rlen, pad);
        Gaps g{gap_open, gap_extend};
        size_t got;
        if (kind == 0) got = blk.align_exp(pq, pr, m.aa, g, min_size, max_size, x_drop, target);
        else if (kind == 1) got = blk.align_exp(pq, pr, m.nuc, g, min_size, max_size, x_drop, target);
        else got = blk.align_exp(pq, pr, m.bytes, g, min_size, max_size, x_drop, target);
        *score = blk.res.score; *qidx = blk.res.query_idx; *ridx = blk.res.reference_idx; *reached = got;
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// One seq-profile alignment. pos_aa: (plen+1) x 32 i8 rows (row 0 = padding column), gap arrays: plen+1 i8 each.
int ba_oracle_align_profile(const uint8_t* q, size_t qlen, size_t plen, const int8_t* pos_aa,
                            const int8_t* gap_open_C, const int8_t* gap_close_C, const int8_t* gap_open_R, int8_t gap_extend,
                            size_t min_size, size_t max_size, int32_t x_drop, uint32_t flags,
                            int32_t* score, uint64_t* qidx, uint64_t* ridx, char* cigar_buf, size_t cigar_cap, uint64_t* stats) {
    try {
        size_t pad = max_size < (size_t)L ? (size_t)L : max_size;
        PaddedBytes pq = PaddedBytes::from_bytes<AAMatrix>(q, qlen, pad);
        AAProfile p(plen, pad, gap_extend);
        for (size_t i = 0; i <= plen; i++) {
            for (int b = 0; b < 32; b++) {
                p.pos_aa[i * 32 + b] = pos_aa[i * 32 + b];
                p.aa_pos[(size_t)b * p.curr_len + i] = pos_aa[i * 32 + b];
            }
            p.pos_gap_open_C[i] = gap_open_C[i];
            p.pos_gap_close_C[i] = gap_close_C[i];
            p.pos_gap_open_R[i] = gap_open_R[i];
        }
        Block blk(mode_of(flags), qlen, plen, pad);
        blk.align_profile(pq, p, min_size, max_size, x_drop);
        *score = blk.res.score; *qidx = blk.res.query_idx; *ridx = blk.res.reference_idx;
        if (cigar_buf && cigar_cap) cigar_buf[0] = 0;
        if ((flags & F_TRACE) && cigar_buf) {
            Cigar c(blk.res.query_idx, blk.res.reference_idx);
            blk.trace().cigar(blk.res.query_idx, blk.res.reference_idx, c);
            std::string s = c.to_string();
            if (s.size() + 1 > cigar_cap) { g_err = "cigar buffer too small"; return 1; }
            std::memcpy(cigar_buf, s.c_str(), s.size() + 1);
        }
        if (stats) {
            stats[0] = blk.cells_computed; stats[1] = blk.steps; stats[2] = blk.end_block_size;
            stats[3] = (flags & F_TRACE) ? surviving_cells(blk.trace()) : 0;
        }
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

// Batch of seq-seq alignments over a packed byte pool (raw, unpadded, unconverted bytes), for the parity
// tests and for bench.py's cpu_baseline. One Block per thread, reused across pairs (examples/profile.rs
// style); padding/conversion of the inputs happens before the timed region.
//   cig_ops: packed (len << 4 | op) u32 runs in alignment order, pair p at [cig_off[p], cig_off[p] + cig_len[p]);
//            capacity for pair p is q_len[p] + r_len[p] + 1 runs starting at cig_off[p] (caller-computed).
// Outputs may be NULL when not wanted. seconds = wall time of the timed region, cells = sum of computed cells.
int ba_oracle_batch_align(int kind, const int8_t* matrix, const uint8_t* pool,
                          const uint64_t* q_off, const uint32_t* q_len, const uint64_t* r_off, const uint32_t* r_len, size_t n,
                          int8_t gap_open, int8_t gap_extend, size_t min_size, size_t max_size, int32_t x_drop, uint32_t flags,
                          int n_threads, int32_t* scores, uint32_t* qidx, uint32_t* ridx,
                          uint32_t* cig_ops, const uint64_t* cig_off, uint32_t* cig_len,
                          uint64_t* cells_out, double* seconds_out) {
    try {
        AnyMatrix m(kind, matrix);
        const size_t pad = max_size < (size_t)L ? (size_t)L : max_size;
        std::vector<PaddedBytes> pq(n), pr(n);
        size_t max_q = 0, max_r = 0;
        for (size_t p = 0; p < n; p++) {
            pq[p] = m.pad(pool + q_off[p], q_len[p], pad);
            pr[p] = m.pad(pool + r_off[p], r_len[p], pad);
            if (q_len[p] > max_q) max_q = q_len[p];
            if (r_len[p] > max_r) max_r = r_len[p];
        }
        if (n_threads < 1) n_threads = 1;
        // one Block per thread, constructed (and so first-touched) by the thread that uses it
        std::vector<std::unique_ptr<Block>> blocks(n_threads);
        std::atomic<size_t> next{0};
        std::atomic<uint64_t> cells{0};
        std::atomic<bool> failed{false};
        std::string err;
        const Gaps g{gap_open, gap_extend};
        auto worker = [&](int t) {
            if (!blocks[t]) blocks[t].reset(new Block(mode_of(flags), max_q, max_r, pad));
            Block& blk = *blocks[t];
            Cigar cg((flags & F_TRACE) ? max_q : 0, (flags & F_TRACE) ? max_r : 0);
            uint64_t local_cells = 0;
            try {
                for (;;) {
                    size_t base = next.fetch_add(16);
                    if (base >= n) break;
                    for (size_t p = base; p < n && p < base + 16; p++) {
                        m.align(blk, pq[p], pr[p], g, min_size, max_size, x_drop);
                        local_cells += blk.cells_computed;
                        if (scores) scores[p] = blk.res.score;
                        if (qidx) qidx[p] = (uint32_t)blk.res.query_idx;
                        if (ridx) ridx[p] = (uint32_t)blk.res.reference_idx;
                        if ((flags & F_TRACE) && cig_len) {
                            if (flags & F_CIGAR_EQ) blk.trace().cigar_eq(pq[p], pr[p], blk.res.query_idx, blk.res.reference_idx, cg);
                            else blk.trace().cigar(blk.res.query_idx, blk.res.reference_idx, cg);
                            size_t k = cg.len();
                            cig_len[p] = (uint32_t)k;
                            if (cig_ops) for (size_t e = 0; e < k; e++) {
                                OpLen o = cg.get(e);
                                cig_ops[cig_off[p] + e] = (uint32_t)((o.len << 4) | o.op);
                            }
                        }
                    }
                }
            } catch (const std::exception& e) {
                if (!failed.exchange(true)) err = e.what();
            }
            cells += local_cells;
        };
        {   // allocate before the clock starts
            std::vector<std::thread> th;
            for (int t = 0; t < n_threads; t++) th.emplace_back([&, t] { blocks[t].reset(new Block(mode_of(flags), max_q, max_r, pad)); });
            for (auto& x : th) x.join();
        }
        auto t0 = std::chrono::steady_clock::now();
        if (n_threads == 1) worker(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < n_threads; t++) th.emplace_back(worker, t);
            for (auto& x : th) x.join();
        }
        auto t1 = std::chrono::steady_clock::now();
        if (failed) { g_err = err; return 1; }
        if (cells_out) *cells_out = cells.load();
        if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}


// Batch of seq-profile alignments (Block::align_profile, scan_block.rs:942-968), threaded like ba_oracle_batch_align: bench.py's
// check of every pair of the PSSM configuration. Profile p occupies rows [p_off[p], p_off[p] + p_len[p] + 1) of pos_aa_all
// ((len + 1) x 32 i8 rows, row 0 = the padding column) and of the three gap arrays. cells_pp: computed cells per pair (optional).
int ba_oracle_batch_align_profile(const uint8_t* pool, const uint64_t* q_off, const uint32_t* q_len, size_t n,
                                  const int8_t* pos_aa_all, const int8_t* goC_all, const int8_t* gcC_all, const int8_t* goR_all,
                                  const uint64_t* p_off, const uint32_t* p_len, int8_t gap_extend,
                                  size_t min_size, size_t max_size, int32_t x_drop, uint32_t flags, int n_threads,
                                  int32_t* scores, uint32_t* qidx, uint32_t* ridx, uint32_t* cig_ops, const uint64_t* cig_off, uint32_t* cig_len,
                                  uint64_t* cells_pp, double* seconds_out) {
    try {
        const size_t pad = max_size < (size_t)L ? (size_t)L : max_size;
        size_t max_q = 0, max_r = 0;
        for (size_t p = 0; p < n; p++) { if (q_len[p] > max_q) max_q = q_len[p]; if (p_len[p] > max_r) max_r = p_len[p]; }
        if (n_threads < 1) n_threads = 1;
        std::atomic<size_t> next{0};
        std::atomic<bool> failed{false};
        std::string err;
        auto worker = [&](int) {
            try {
                Block blk(mode_of(flags), max_q, max_r, pad);
                Cigar cg((flags & F_TRACE) ? max_q : 0, (flags & F_TRACE) ? max_r : 0);
                for (;;) {
                    const size_t base = next.fetch_add(16);
                    if (base >= n) break;
                    for (size_t p = base; p < n && p < base + 16; p++) {
                        PaddedBytes pq = PaddedBytes::from_bytes<AAMatrix>(pool + q_off[p], q_len[p], pad);
                        const size_t plen = p_len[p];
                        AAProfile pr(plen, pad, gap_extend);
                        for (size_t i = 0; i <= plen; i++) {
                            const size_t row = p_off[p] + i;
                            for (int b = 0; b < 32; b++) {
                                pr.pos_aa[i * 32 + b] = pos_aa_all[row * 32 + b];
                                pr.aa_pos[(size_t)b * pr.curr_len + i] = pos_aa_all[row * 32 + b];
                            }
                            pr.pos_gap_open_C[i] = goC_all[row]; pr.pos_gap_close_C[i] = gcC_all[row]; pr.pos_gap_open_R[i] = goR_all[row];
                        }
                        blk.align_profile(pq, pr, min_size, max_size, x_drop);
                        if (scores) scores[p] = blk.res.score;
                        if (qidx) qidx[p] = (uint32_t)blk.res.query_idx;
                        if (ridx) ridx[p] = (uint32_t)blk.res.reference_idx;
                        if (cells_pp) cells_pp[p] = blk.cells_computed;
                        if ((flags & F_TRACE) && cig_len) {
                            blk.trace().cigar(blk.res.query_idx, blk.res.reference_idx, cg);
                            const size_t k = cg.len();
                            cig_len[p] = (uint32_t)k;
                            if (cig_ops) for (size_t e = 0; e < k; e++) { OpLen o = cg.get(e); cig_ops[cig_off[p] + e] = (uint32_t)((o.len << 4) | o.op); }
                        }
                    }
                }
            } catch (const std::exception& e) {
                if (!failed.exchange(true)) err = e.what();
            }
        };
        auto t0 = std::chrono::steady_clock::now();
        if (n_threads == 1) worker(0);
        else {
            std::vector<std::thread> th;
            for (int t = 0; t < n_threads; t++) th.emplace_back(worker, t);
            for (auto& x : th) x.join();
        }
        auto t1 = std::chrono::steady_clock::now();
        if (failed) { g_err = err; return 1; }
        if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}

}  // extern "C"
