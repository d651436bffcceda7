// block_aligner_amd — structures shared by the host library and the gfx950 kernels (plain C++, no device code).
#pragma once
#include <stdint.h>

namespace ba {

constexpr int ZERO = 1 << 14;     // avx2.rs:15
constexpr int STEP = 8;           // scan_block.rs:787
constexpr int KIND_AA = 0, KIND_NUC = 1, KIND_BYTES = 2, KIND_PROFILE = 3;   // PROFILE: the "reference" is a position-specific AAProfile
constexpr int DIR_RIGHT = 0, DIR_DOWN = 1, DIR_GROW = 2;

enum : uint32_t {
    F_TRACE = 1u << 0, F_XDROP = 1u << 1, F_LOCAL = 1u << 2, F_FQS = 1u << 3, F_FQE = 1u << 4, F_CIGAR_EQ = 1u << 5
};
enum : uint32_t { ST_OK = 0, ST_TRACE_OVERFLOW = 1, ST_BLOCKS_OVERFLOW = 2, ST_CIGAR_OVERFLOW = 4, ST_TRACEBACK_LOST = 8, ST_WATCHDOG = 16, ST_SLOT_TIMEOUT = 32, ST_MODE = 64,
                  ST_CLASS_OVERFLOW = 128 /* internal (round 6): the pair's block wants to grow past the launch's block class -- the host runs it again in the row-tiled class, batch_wait */ };

struct BlockRec {   // one computed rectangle (scan_block.rs:1428-1443), 16 bytes
    uint32_t i, j;
    uint16_t h, w;        // h = rows, w = columns of the rectangle
    uint32_t trace_base;  // dword index into this slot's trace arena; bit 31 = "right" (vectors along rows)
};

struct SlotInfo { uint32_t pair, nblocks, end_i, end_j; };

// Device image of one AAProfile (scores.rs:452-468) inside the sequence pool, P = profile_positions(len, max_size):
//   int8  pos_aa[P][32]   | int16 aa_pos[32][P] | int16 gap_open_C[P] | gap_close_C[P] | gap_open_R[P]
// positions beyond the profile hold the reference's defaults (-128), so block reads past the end need no bounds checks
#if defined(__HIPCC__)
#define BA_HD __host__ __device__
#else
#define BA_HD
#endif
BA_HD inline uint32_t profile_positions(uint32_t len, uint32_t max_size) { return (len + max_size + 17u) & ~1u; }
BA_HD inline uint64_t profile_image_bytes(uint32_t len, uint32_t max_size) { return (uint64_t)profile_positions(len, max_size) * (32 + 64 + 6); }

// A pair in flight between kernels (small-block batches, ba_quad.hpp): the driver's state at the top of its loop
// (scan_block.rs:123-130's locals) with the four 32-cell borders and the checkpoint. 64 + 2 x 256 bytes.
struct PairCont {
    uint32_t pair, si, sj; int32_t dir, prev_dir, off, off_max, best_max;
    uint32_t y_drop_iter; int32_t x_drop_iter, D_corner; uint32_t best_i, best_j, ck_i, ck_j; int32_t ck_off;
    unsigned long long cells; uint32_t step_budget;
    uint32_t trace_top, nblocks, ck_trace_top, ck_nblocks, status;   // TRACE: the pair's trace stack (it lives in the pair's own region)
    uint32_t borders[4][16];   // D_col, C_col, D_row, R_row: lane l holds cells 2l, 2l+1 (packed i16)
    uint32_t ckpt[4][16];
};
static_assert(sizeof(PairCont) == 96 + 512, "PairCont layout");

struct BatchParams {
    // inputs: pool holds PaddedBytes images: [NULL] + converted bytes + NULL x pad (scan_block.rs:1790-1812)
    const uint8_t* pool;
    const uint64_t* q_off; const uint32_t* q_len;
    const uint64_t* r_off; const uint32_t* r_len;
    uint32_t n;
    int32_t gap_open, gap_extend;
    uint32_t min_size, max_size;
    int32_t x_drop;
    uint32_t flags;
    const int8_t* matrix;      // AA 27x32, NUC 8x16, BYTES {match, mismatch}
    // outputs
    int32_t* score; uint32_t* query_idx; uint32_t* reference_idx;
    uint32_t* cig_ops; const uint64_t* cig_off; uint32_t* cig_start; uint32_t* cig_len;   // runs (len << 4 | op)
    unsigned long long* cells;   // per pair: computed DP cells, may be null
    uint32_t* status;            // per pair error bits
    uint32_t* nblocks_out;       // per pair: rectangles on the trace stack at the end (may be null)
    uint32_t* slot_out;          // per pair: trace slot that holds its stack (for a later k_traceback; may be null)
    uint32_t* trace_words_out;   // per pair: trace words on the stack at the end = surviving-rectangle cells / 8 (may be null)
    // scratch, one region per resident wave
    uint32_t* trace_arena; uint64_t trace_stride;     // dwords per slot
    BlockRec* blocks; uint64_t blocks_stride;         // records per slot
    // pair-slot batches (the small-block TRACE pipeline): every pair owns a region for the whole batch, pair p's trace words at
    // trace_arena + trace_off[p] (capacity trace_off[p + 1] - trace_off[p]) and its records at blocks + blocks_off[p]; slot = pair
    const uint64_t* trace_off; const uint64_t* blocks_off;
    short* ckpt;                 // per fill wave: 4 x max_size i16 (best-so-far borders, scan_block.rs:406-427)
    short* big;                  // block sizes above 2048 only: per fill wave big_wave_shorts(max_size) i16 -- the four live borders
                                 // (too large for LDS) and the two row hand-off arrays of the tiled fill (ba_device.hpp TileCtx)
    uint32_t* work_counter;
    unsigned long long* prof;    // development (-DBA_TIMING builds): per-phase cycle sums, 32 slots
    // in-launch hand-off of finished trace stacks from fill waves to traceback lanes (TRACE batches)
    uint32_t tb_stride;          // 0: fill waves walk their own tracebacks. s > 0: wave 0 of every s-th workgroup walks tracebacks
    uint32_t slots_per_wave;     // trace arena slots owned by each fill wave (a slot is busy until its traceback is done)
    uint32_t n_slots;
    uint32_t tb_reserve;         // the last tb_reserve hand-offs are left to fill waves that ran out of pairs (one lane each: see traceback_consumer)
    uint32_t tb_qmask;           // ring size - 1 (power of two >= max(traceback lanes, n_slots): live claims never share a position)
    uint32_t* tb_queue;          // ring entries: slot + 1, or 0x80000000 | pair for a pair without a trace stack; 0 = empty
    uint32_t* tb_ctrl;           // [0] tail (next entry to produce), [32] head (next entry to claim); separate cache lines
    uint32_t* slot_free;         // per slot: 1 = free, 0 = owned by a fill wave or a pending traceback
    SlotInfo* slot_info;         // per slot: what the traceback lane needs
    // small-block batches (min block 32): k_quad starts every pair and runs its plain shift steps at 32 cells; a pair that needs
    // anything else leaves as a PairCont record (cont_out, indexed by the pair's position in the batch; cont_out_flag[p] = 1), a
    // pair k_quad cannot start gets flag 2, a finished one stays 0; every flagged pair is also appended to the queue cq_*, which
    // per-pair kernels launched with cont_mode = 2 drain -- one of them while k_quad is still running.
    uint32_t cont_mode;          // k_align: 0 plain, 2 take pairs from the queue (resume from cont_in / from scratch).
                                 // k_walk: 0 every pair, 1 only the pairs with cont_in_flag 0 (finished by k_quad)
    const PairCont* cont_in; const uint32_t* cont_in_flag;
    PairCont* cont_out; uint32_t* cont_out_flag;
    uint32_t* cq_queue;          // n entries: 1 + 2 * pair + (1: run from scratch), 0 = not yet written
    uint32_t* cq_ctrl;           // [0] tail (entries appended), [16] head (tickets handed out), [20] k_quad is on the device, [32] producer waves done, [48] a ticket starved
    uint32_t cq_producers;       // k_quad waves of the launch
    uint32_t cq_side;            // 1: this per-pair launch runs beside k_quad (it may stop waiting; the launch after k_quad drains the queue)
    uint32_t inline_len2;        // pair-slot batches, per-pair kernel: pairs with |q| + |r| >= this walk their paths at once (lane 0), shorter ones leave them to k_walk
    uint32_t ckpt_wave0;         // this launch's first wave in the checkpoint arena (two per-pair kernels of one batch run side by side)
    uint32_t mq_drain;           // k_multi: the last mq_drain pairs of the batch are not taken into slots (four pairs per wave, each four times as long in
                                 // flight) but one at a time by waves whose slots have emptied, and run on all lanes to their end: a finer ragged end
    uint32_t* mq_donate;         // k_multi: one word per wave of the launch (blockIdx * 8 + wave), bit s = slot s of that wave is offered to waves that have run out
                                 // of work (end of the batch), followed by two counters: [waves] fill waves that will offer nothing more, [waves + 32] offers outstanding; null = no donation
    uint32_t mq_waves;           // k_multi: waves of the launch (workgroups x 8): the number of words in mq_donate
    uint32_t walk_wave_n;        // k_walk: the first walk_wave_n pairs of the batch order are walked one to a wave (walk_wave), the others one to a lane
    uint32_t work_chunk;         // pairs (records) a wave takes per atomic on the work counter (one counter serves ~90 atomics / us)
    uint32_t sm_excl_first;      // k_small: this launch's exclusive pairs are [sm_excl_first, sm_excl_n) of the batch order (the longest of them run in a launch of their own, beside the main one)
    uint32_t sm_excl_n;          // k_small: the first sm_excl_n pairs of the batch order (its longest) run one to a wave on all lanes, from start to end (work_counter[2])
    // single-pair traceback request (k_traceback): end position
    uint32_t tb_i, tb_j, tb_nblocks, tb_slot;
};


// blocks of 4096 .. 32768 cells: borders in an L2-resident arena, rectangles filled in row tiles of BIG_TILE cells
constexpr uint32_t BIG_TILE = 2048;
BA_HD constexpr uint64_t big_array_shorts(uint32_t max_size) { return (uint64_t)max_size + 64; }
BA_HD constexpr uint64_t big_wave_shorts(uint32_t max_size) { return 8 * big_array_shorts(max_size); }   // 4 borders, 2 row hand-off arrays, 2 per-column arrays of the FREE_QUERY_END_GAPS bookkeeping

constexpr uint32_t MQ_B_HOST = 128;   // k_multi: block size of a slot (ba_driver.hpp MQ_B)
// k_multi: per wave and slot two state buffers and a record in the `big` arena (ba_multi.hpp)
constexpr uint32_t MQ_BUF_BYTES = 4 * 256 + 64, MQ_SLOT_BYTES = 2 * MQ_BUF_BYTES + 128, MQ_WAVE_BYTES = 4 * MQ_SLOT_BYTES;
// k_multi: while the slots run, the same buffers live in the wave's LDS region (the solo borders' space, which the slots do not need):
// 4 slots x 2 buffers x 1 KB of borders, then 8 x 9 scalars
constexpr uint32_t MQ_LDS_SCALARS = 8192, MQ_LSC_INTS = 28, MQ_LDS_BYTES = 8192 + 4 * MQ_LSC_INTS * 4;   // per slot: 2 x 9 scalars, 2 free, 8 constants of the slot's pair (sequence addresses and lengths, trace slot)
// k_small (ba_small.hpp): sixteen pairs per wave while the block is 32 cells -- slots of 4 lanes x 8 cells. Per wave and slot in the `big`
// arena: two state buffers (4 border arrays x 4 lanes x 16 bytes + 8 scalars) and a 128-byte record; while the slots run the buffers live in
// the wave's LDS region (16 slots x 2 buffers x 256 bytes of borders, then 16 x 2 x 8 scalars)
constexpr uint32_t SM_B_HOST = 32, SM_SLOTS = 16;
constexpr uint32_t SM_BUF_BYTES = 4 * 64 + 32, SM_SLOT_BYTES = 2 * SM_BUF_BYTES + 128, SM_WAVE_BYTES = SM_SLOTS * SM_SLOT_BYTES;
constexpr uint32_t SM_LDS_SCALARS = 8192, SM_LDS_BYTES = 8192 + SM_SLOTS * 2 * 32;
#ifndef BA_WAVES_PER_WG
#define BA_WAVES_PER_WG 8   // (development: a kernel object built with 6 runs three waves per SIMD under a host built with 8 -- the arenas are only larger than needed)
#endif
constexpr int WAVES_PER_WG = BA_WAVES_PER_WG;   // independent waves per workgroup; they share the read-only score table in LDS
constexpr int MQ_GEOM_WPW = 4;   // k_multi's geometries of three / two waves per SIMD (round 6): workgroups of one wave per SIMD, three / two of them per CU

// LDS layout: [score table (per workgroup)] [wave 0: 4 borders + misc] [wave 1: ...] ...
BA_HD constexpr uint32_t lds_array_bytes_h(uint32_t max_size) { return max_size * 2 + 32; }
BA_HD constexpr uint32_t lds_wave_bytes_h(uint32_t max_size) { return 4 * lds_array_bytes_h(max_size) + 128; }
BA_HD constexpr uint32_t lds_table_bytes_h(int kind) { return kind == KIND_NUC ? 8192 : 896; }
BA_HD constexpr uint32_t lds_wg_bytes_h(int kind, uint32_t max_size) { return lds_table_bytes_h(kind) + WAVES_PER_WG * lds_wave_bytes_h(max_size); }
BA_HD constexpr uint32_t mq_wave_bytes_h(uint32_t max_size) { return lds_wave_bytes_h(max_size) > MQ_LDS_BYTES ? lds_wave_bytes_h(max_size) : MQ_LDS_BYTES; }   // k_multi
BA_HD constexpr uint32_t mq_wg_bytes_h(int kind, uint32_t max_size, uint32_t wpw = WAVES_PER_WG) { return lds_table_bytes_h(kind) + wpw * mq_wave_bytes_h(max_size); }
constexpr uint32_t SM_PROF_STAGE = 16 * 256;   // k_small, profile batches: per wave, 8 rows x 32 bytes of pos_aa per slot (the columns of a right step; behind the wave's region)
BA_HD constexpr uint32_t sm_wave_bytes_h(uint32_t max_size, int kind = 0) {   // k_small
    return (lds_wave_bytes_h(max_size) > SM_LDS_BYTES ? lds_wave_bytes_h(max_size) : SM_LDS_BYTES) + (kind == KIND_PROFILE ? SM_PROF_STAGE : 0u);
}
BA_HD constexpr uint32_t sm_wg_bytes_h(int kind, uint32_t max_size) { return lds_table_bytes_h(kind) + WAVES_PER_WG * sm_wave_bytes_h(max_size, kind); }
// TRACE batches: one more region behind the waves' for the workgroup's traceback wave (ba_driver.hpp tb_step): per lane
// a 76-byte record (10 trace words + 16 query + 16 reference bytes; 19 dwords: conflict-free) and the 128-byte move table
constexpr uint32_t TB_LANE_BYTES = 76, TB_LUT_BYTES = 128, TB_LDS_BYTES = 5120;
constexpr uint32_t TB_LANE_BYTES_L2 = 100;   // k_multi's traceback waves (16 trace words per window): their records sit in the wave's own LDS region   // table first, then the records (a helper fill wave uses one)
// LOCAL_START batches (round 5): the window also holds the cells' zero-mask bits -- 20 interleaved trace / mask words of a per-pair rectangle,
// or a slot rectangle's 16 trace + 4 mask words --, then the sequence windows
constexpr uint32_t TB_LANE_BYTES_LOC = 116, TB_LDS_BYTES_LOC = 7680;
static_assert(TB_LUT_BYTES + 64 * TB_LANE_BYTES_LOC <= TB_LDS_BYTES_LOC, "k_walk LDS (LOCAL_START)");
constexpr uint32_t TB_LDS_BYTES_L2 = 6784;   // k_walk over a k_small batch: the move table + 64 records of TB_LANE_BYTES_L2
static_assert(TB_LUT_BYTES + 64 * TB_LANE_BYTES_L2 <= TB_LDS_BYTES_L2, "k_walk LDS (slot rectangles)");
static_assert(MQ_LDS_BYTES % 16 == 0 && TB_LUT_BYTES + 64 * TB_LANE_BYTES_L2 <= MQ_LDS_BYTES, "k_multi LDS");
// round 6: behind the records the tables of tb_diag (ba_driver.hpp: runs of diagonal moves at once), 8-byte aligned
constexpr uint32_t TB_DIAG_LUT_L2 = 256, TB_DIAG_LUT_STD = 64;   // (L2: F 128 + G 64 + C 64 bytes)
static_assert(TB_LUT_BYTES + 64 * TB_LANE_BYTES + TB_DIAG_LUT_STD <= TB_LDS_BYTES && TB_LUT_BYTES + TB_LANE_BYTES + 8 + TB_DIAG_LUT_STD <= lds_wave_bytes_h(128), "traceback LDS regions");
static_assert((TB_LUT_BYTES + 64 * TB_LANE_BYTES) % 8 == 0 && (TB_LUT_BYTES + 64 * TB_LANE_BYTES_L2) % 8 == 0, "tb_diag's tables are read 8 bytes at a time");
static_assert(TB_LUT_BYTES + 64 * TB_LANE_BYTES_L2 + TB_DIAG_LUT_L2 <= TB_LDS_BYTES_L2 && TB_LUT_BYTES + 64 * TB_LANE_BYTES_L2 + TB_DIAG_LUT_L2 <= MQ_LDS_BYTES, "tb_diag's tables (slot rectangles)");
// the special-mode instantiations of k_multi / k_small put TB_LANE_BYTES_LOC records into a wave's own region
static_assert(TB_LUT_BYTES + 64 * TB_LANE_BYTES_LOC <= MQ_LDS_BYTES && TB_LUT_BYTES + 64 * TB_LANE_BYTES_LOC <= SM_LDS_BYTES, "special-mode walk records in a wave's region");

}  // namespace ba
