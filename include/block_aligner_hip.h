/*
 * block_aligner_hip.h — C ABI of the MI355X (gfx950) backend for block-aligner.
 *
 * Part 1 is, symbol for symbol, the reference's C API (/root/reference/c/block_aligner.h, generated from
 * /root/reference/src/ffi.rs): a program written against that header links against libblock_aligner_hip.so
 * unchanged (see tests/c_abi/abi_check.c, a gcc-compiled caller in the style of /root/reference/c/example.c). Every alignment is
 * executed by the HIP kernels in block_aligner_amd/csrc; there is no CPU fallback — if no gfx950 device or
 * HIP runtime is usable the call aborts with a message, like the reference's panic=abort.
 *
 * Part 2 adds what the reference's FFI lacks for the hot path named in BASELINE.json: nucleotide / byte
 * matrices (ffi.rs:5 "do not have bindings yet") and a batch launcher that aligns many independent pairs in
 * one kernel launch (one wavefront per pair). A Rust `simd_hip` backend binds exactly these.
 *
 * Error behaviour: Part 1 functions keep the reference contract (no error codes; a violated precondition
 * aborts the process with the reference's assert message). Part 2 functions return 0 on success and a
 * non-zero code otherwise, with ba_last_error() giving the message.
 */
#ifndef BLOCK_ALIGNER_HIP_H
#define BLOCK_ALIGNER_HIP_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------------ */
/* Part 1 — the reference C API                                                                           */
/* ------------------------------------------------------------------------------------------------------ */

/* cigar.rs:10-31, c/block_aligner.h:17-57: `enum Operation` with a one-byte representation */
enum Operation
#ifdef __cplusplus
    : uint8_t
#endif
{ Sentinel = 0, M = 1, Eq = 2, X = 3, I = 4, D = 5 };
#ifndef __cplusplus
typedef uint8_t Operation;
#endif

typedef struct AAMatrix AAMatrix;       /* scores.rs:40-44: 27 x 32 int8, 32-byte aligned, 864 bytes */
typedef struct NucMatrix NucMatrix;     /* scores.rs:142-146: 8 x 16 int8, 32-byte aligned, 128 bytes */
typedef struct AAProfile AAProfile;     /* opaque */
typedef struct Cigar Cigar;             /* opaque */
typedef struct PaddedBytes PaddedBytes; /* opaque */

typedef struct OpLen { Operation op; uintptr_t len; } OpLen;                                   /* cigar.rs:34-39 */
typedef void* BlockHandle;                                                                      /* ffi.rs:15 */
typedef struct Gaps { int8_t open; int8_t extend; } Gaps;                                       /* scores.rs:333-338 */
typedef struct SizeRange { uintptr_t min; uintptr_t max; } SizeRange;                           /* ffi.rs:18-23 */
typedef struct AlignResult { int32_t score; uintptr_t query_idx; uintptr_t reference_idx; } AlignResult; /* scan_block.rs:1887-1893 */
typedef struct ByteMatrix { int8_t match_score; int8_t mismatch_score; } ByteMatrix;           /* scores.rs:220-225 */

/* data symbols (scores.rs:275-311, c/block_aligner.h:140-162); C callers take their address */
extern const struct NucMatrix NW1;
extern const struct AAMatrix BLOSUM45, BLOSUM50, BLOSUM62, BLOSUM80, BLOSUM90;
extern const struct AAMatrix PAM100, PAM120, PAM160, PAM200, PAM250;
extern const struct ByteMatrix BYTES1;

/* AAMatrix — ffi.rs:31-48 */
struct AAMatrix* block_new_simple_aamatrix(int8_t match_score, int8_t mismatch_score);
void block_set_aamatrix(struct AAMatrix* matrix, uint8_t a, uint8_t b, int8_t score);
void block_free_aamatrix(struct AAMatrix* matrix);

/* AAProfile — ffi.rs:60-195 */
struct AAProfile* block_new_aaprofile(uintptr_t str_len, uintptr_t block_size, int8_t gap_extend);
uintptr_t block_len_aaprofile(const struct AAProfile* profile);
void block_clear_aaprofile(struct AAProfile* profile, uintptr_t str_len, uintptr_t block_size);
void block_set_aaprofile(struct AAProfile* profile, uintptr_t i, uint8_t b, int8_t score);
void block_set_all_aaprofile(struct AAProfile* profile, const uint8_t* order, uintptr_t order_len, const int8_t* scores,
                             uintptr_t scores_len, uintptr_t left_shift, uintptr_t right_shift);
void block_set_all_rev_aaprofile(struct AAProfile* profile, const uint8_t* order, uintptr_t order_len, const int8_t* scores,
                                 uintptr_t scores_len, uintptr_t left_shift, uintptr_t right_shift);
void block_set_gap_open_C_aaprofile(struct AAProfile* profile, uintptr_t i, int8_t gap);
void block_set_gap_close_C_aaprofile(struct AAProfile* profile, uintptr_t i, int8_t gap);
void block_set_gap_open_R_aaprofile(struct AAProfile* profile, uintptr_t i, int8_t gap);
void block_set_all_gap_open_C_aaprofile(struct AAProfile* profile, int8_t gap);
void block_set_all_gap_close_C_aaprofile(struct AAProfile* profile, int8_t gap);
void block_set_all_gap_open_R_aaprofile(struct AAProfile* profile, int8_t gap);
int8_t block_get_aaprofile(const struct AAProfile* profile, uintptr_t i, uint8_t b);
int8_t block_get_gap_extend_aaprofile(const struct AAProfile* profile);
void block_free_aaprofile(struct AAProfile* profile);

/* Cigar — ffi.rs:201-225 */
struct Cigar* block_new_cigar(uintptr_t query_len, uintptr_t reference_len);
struct OpLen block_get_cigar(const struct Cigar* cigar, uintptr_t i);
uintptr_t block_len_cigar(const struct Cigar* cigar);
void block_free_cigar(struct Cigar* cigar);

/* PaddedBytes (amino acids) — ffi.rs:231-257 */
struct PaddedBytes* block_new_padded_aa(uintptr_t len, uintptr_t max_size);
void block_set_bytes_padded_aa(struct PaddedBytes* padded, const uint8_t* s, uintptr_t len, uintptr_t max_size);
void block_set_bytes_rev_padded_aa(struct PaddedBytes* padded, const uint8_t* s, uintptr_t len, uintptr_t max_size);
void block_free_padded_aa(struct PaddedBytes* padded);

/* Block<TRACE, X_DROP> over AAMatrix / AAProfile — ffi.rs:262-403 (gen_functions! x 4) */
#define BA_DECLARE_BLOCK_FNS(S, CIG, CIGEQ)                                                                                   \
    BlockHandle block_new_##S(uintptr_t query_len, uintptr_t reference_len, uintptr_t max_size);                              \
    void block_align_##S(BlockHandle b, const struct PaddedBytes* q, const struct PaddedBytes* r, const struct AAMatrix* m,   \
                         struct Gaps g, struct SizeRange s, int32_t x);                                                       \
    void block_align_profile_##S(BlockHandle b, const struct PaddedBytes* q, const struct AAProfile* r, struct SizeRange s,   \
                                 int32_t x);                                                                                  \
    struct AlignResult block_res_##S(BlockHandle b);                                                                          \
    void CIG(BlockHandle b, uintptr_t query_idx, uintptr_t reference_idx, struct Cigar* cigar);                               \
    void CIGEQ(BlockHandle b, const struct PaddedBytes* q, const struct PaddedBytes* r, uintptr_t query_idx,                  \
               uintptr_t reference_idx, struct Cigar* cigar);                                                                 \
    void block_free_##S(BlockHandle b);

BA_DECLARE_BLOCK_FNS(aa, _block_cigar_aa, _block_cigar_eq_aa)                                   /* ffi.rs:333-349 */
BA_DECLARE_BLOCK_FNS(aa_xdrop, _block_cigar_aa_xdrop, _block_cigar_eq_aa_xdrop)                 /* ffi.rs:351-367 */
BA_DECLARE_BLOCK_FNS(aa_trace, block_cigar_aa_trace, block_cigar_eq_aa_trace)                   /* ffi.rs:369-385 */
BA_DECLARE_BLOCK_FNS(aa_trace_xdrop, block_cigar_aa_trace_xdrop, block_cigar_eq_aa_trace_xdrop) /* ffi.rs:387-403 */

/* ------------------------------------------------------------------------------------------------------ */
/* Part 2 — extensions for the MI355X hot path                                                            */
/* ------------------------------------------------------------------------------------------------------ */

/* thread-local message for the last failing Part 2 call */
const char* ba_last_error(void);
/* 1 for the development build of the library (lib/libblock_aligner_hip_dev.so: reads the BA_* switches, tests and tools only), 0 for the
 * release library, which reads no environment variables. */
int ba_dev_build(void);
/* Hash of the kernel sources this library was built from (tools/kernel_hash.py at build time): equal for the release and the
 * development library of one build. */
const char* ba_build_id(void);
/* Page-locked host memory for result buffers (device-to-host copies into it run at PCIe speed; into pageable memory at a fraction). */
void* ba_host_alloc(uint64_t bytes);
void ba_host_free(void* p);
/* number of usable HIP devices (0 if the runtime is unusable); ba_set_device selects the one later calls OF THE CALLING
 * THREAD use (the selection is per host thread, so one thread can drive each GPU) */
int ba_device_count(void);
int ba_set_device(int device);
/* free / total bytes of the selected device's memory (hipMemGetInfo) */
int ba_device_memory(uint64_t* free_bytes, uint64_t* total_bytes);
/* lib.rs:109-111 */
uintptr_t block_percent_len(uintptr_t len, float p);

/* Nucleotide / byte matrices and padded strings (the reference has Rust API only: scores.rs:142-273,
 * scan_block.rs:1798-1822 instantiated with NucMatrix / ByteMatrix). */
struct NucMatrix* block_new_simple_nucmatrix(int8_t match_score, int8_t mismatch_score);
void block_set_nucmatrix(struct NucMatrix* matrix, uint8_t a, uint8_t b, int8_t score);
void block_free_nucmatrix(struct NucMatrix* matrix);
struct PaddedBytes* block_new_padded_nuc(uintptr_t len, uintptr_t max_size);
void block_set_bytes_padded_nuc(struct PaddedBytes* padded, const uint8_t* s, uintptr_t len, uintptr_t max_size);
void block_set_bytes_rev_padded_nuc(struct PaddedBytes* padded, const uint8_t* s, uintptr_t len, uintptr_t max_size);
void block_free_padded_nuc(struct PaddedBytes* padded);
struct PaddedBytes* block_new_padded_bytes(uintptr_t len, uintptr_t max_size);
void block_set_bytes_padded_bytes(struct PaddedBytes* padded, const uint8_t* s, uintptr_t len, uintptr_t max_size);
void block_free_padded_bytes(struct PaddedBytes* padded);

/* mode bits of Block<TRACE, X_DROP, LOCAL_START, FREE_QUERY_START_GAPS, FREE_QUERY_END_GAPS> (scan_block.rs:89) */
enum {
    BA_TRACE = 1u << 0,
    BA_X_DROP = 1u << 1,
    BA_LOCAL_START = 1u << 2,
    BA_FREE_QUERY_START_GAPS = 1u << 3,
    BA_FREE_QUERY_END_GAPS = 1u << 4,
    BA_CIGAR_EQ = 1u << 5 /* batch only: emit =/X instead of M (Trace::cigar_eq, scan_block.rs:1478-1480) */
};
enum { BA_KIND_AA = 0, BA_KIND_NUC = 1, BA_KIND_BYTES = 2 };

/* Generic per-pair block: Block::<mode>::new / align::<matrix kind> / res / trace().cigar[_eq]
 * (scan_block.rs:798-805, 847-878, 1235-1244, 1469-1480). `matrix` points at an AAMatrix, NucMatrix or
 * ByteMatrix according to `kind`; q and r must have been built for the same kind. */
BlockHandle block_new_generic(uint32_t mode, uintptr_t query_len, uintptr_t reference_len, uintptr_t max_size);
void block_align_generic(BlockHandle b, int kind, const struct PaddedBytes* q, const struct PaddedBytes* r, const void* matrix,
                         struct Gaps g, struct SizeRange s, int32_t x);
struct AlignResult block_res_generic(BlockHandle b);
/* Block::<mode>::align_profile (scan_block.rs:942-968): q must be an AA PaddedBytes; the gap costs come from the profile. */
void block_align_profile_generic(BlockHandle b, const struct PaddedBytes* q, const struct AAProfile* profile, struct SizeRange s,
                                 int32_t x);
/* The same two calls for a caller that keeps its own PaddedBytes (the Rust crate behind the `simd_hip` feature, rust/src/): the padded
 * image as the reference stores it -- [NULL] + converted bytes + NULL x block_size (src/scan_block.rs:1790-1812) -- passed by pointer
 * (q_s[0] is the NULL pad; q_len the sequence length). */
void block_align_padded_generic(BlockHandle b, int kind, const uint8_t* q_s, uintptr_t q_len, const uint8_t* r_s, uintptr_t r_len,
                                const void* matrix, struct Gaps g, struct SizeRange s, int32_t x);
void block_align_profile_padded_generic(BlockHandle b, const uint8_t* q_s, uintptr_t q_len, const struct AAProfile* profile,
                                        struct SizeRange s, int32_t x);
void block_cigar_generic(BlockHandle b, uintptr_t query_idx, uintptr_t reference_idx, struct Cigar* cigar);
void block_cigar_eq_generic(BlockHandle b, const struct PaddedBytes* q, const struct PaddedBytes* r, uintptr_t query_idx,
                            uintptr_t reference_idx, struct Cigar* cigar);
void block_free_generic(BlockHandle b);
/* Trace::blocks() (scan_block.rs:1676-1691): the rectangles computed for the last alignment, in fill order. Returns their
 * number; writes at most `capacity` of them (out may be NULL to query the count). */
struct Rectangle { uintptr_t row, col, width, height; };
uintptr_t block_trace_blocks_generic(BlockHandle b, struct Rectangle* out, uintptr_t capacity);

/* ---- batch launcher: many independent pairs, one persistent kernel launch, one wavefront per pair.
 *
 * `pool` holds the raw (unpadded, unconverted) sequence bytes; pair p is query pool[q_off[p] .. +q_len[p]) against
 * reference pool[r_off[p] .. +r_len[p]). The library builds the PaddedBytes images (scan_block.rs:1798-1812: convert_char,
 * NULL pads; on the device when the pairs come out of one dense buffer, a byte outside the alphabet is an error) and keeps
 * them, the matrix and all scratch resident in device memory for the life of the batch object, so ba_batch_run can
 * be timed with inputs already in HBM.
 *
 * Results per pair: score / query_idx / reference_idx as AlignResult (scan_block.rs:567-592); computed DP cells
 * (sum over every block-fill call of columns iterated x height: the GCUPS numerator); CIGAR runs when BA_TRACE is set,
 * each run packed as (len << 4 | Operation), in alignment order. */
typedef struct BaBatch BaBatch;

BaBatch* ba_batch_create(int kind, const void* matrix, struct Gaps gaps, struct SizeRange size, int32_t x_drop, uint32_t mode,
                         const uint8_t* pool, const uint64_t* q_off, const uint32_t* q_len, const uint64_t* r_off,
                         const uint32_t* r_len, uintptr_t n_pairs);
/* Bulk form of block_set_aaprofile / block_set_gap_*_aaprofile for callers that hold the profile as arrays: copies the
 * first `positions` rows of pos_aa ([position][32], column = byte - 'A') and entries of the three gap arrays. */
int ba_aaprofile_set_raw(struct AAProfile* profile, const int8_t* pos_aa, const int8_t* gap_open_C, const int8_t* gap_close_C,
                         const int8_t* gap_open_R, uintptr_t positions);

/* Sequence-to-profile batch (Block::align_profile over many pairs; examples/pssm_bench.rs:86-103): pair p aligns the
 * amino-acid query pool[q_off[p] .. +q_len[p]) to *profiles[p]. Every profile must have been created with a block size
 * >= size.max and share one gap_extend. The profiles are copied to the device; the caller keeps ownership.
 * BA_CIGAR_EQ is rejected (there is no second sequence to compare with). */
BaBatch* ba_batch_create_profile(const struct AAProfile* const* profiles, struct SizeRange size, int32_t x_drop, uint32_t mode,
                                 const uint8_t* pool, const uint64_t* q_off, const uint32_t* q_len, uintptr_t n_pairs);
/* Replace the pairs of an existing batch and keep its device buffers (the trace arena above all, whose allocation
 * dominates the set-up time): same matrix, gaps, block range and modes. The new set must fit what the batch was created
 * with: no more pairs, no more sequence bytes in total, no pair longer than the longest original one. */
int ba_batch_reload(BaBatch* batch, const uint8_t* pool, const uint64_t* q_off, const uint32_t* q_len, const uint64_t* r_off,
                    const uint32_t* r_len, uintptr_t n_pairs);
int ba_batch_reload_profile(BaBatch* batch, const struct AAProfile* const* profiles, const uint8_t* pool, const uint64_t* q_off,
                            const uint32_t* q_len, uintptr_t n_pairs);
/* Launch on the batch's stream and wait. kernel_ms (optional) = HIP-event time of the alignment kernel alone. */
int ba_batch_run(BaBatch* batch, float* kernel_ms);
/* The two halves of ba_batch_run: enqueue on the batch's own stream and return / wait for it. Launches of different
 * batches overlap on the device; results, cigars and reload are valid after the wait.
 * At most TWO launches in flight per device if they are BA_TRACE batches of more than a few pairs per resident wave: such a launch's waves wait for each other
 * (the traceback waves for the fill waves' hand-offs, k_multi's idle waves for its last fill wave), so it must become fully resident to end. The first launch
 * on an idle device is; a second one follows it; three or more may each hold a part of the device and keep each other's remaining workgroups out
 * (ba_sized_batch_run orders its ranges' launches accordingly). Score-only batches have no such waits. */
int ba_batch_launch(BaBatch* batch);
int ba_batch_wait(BaBatch* batch, float* kernel_ms);
/* Upper bound on a ba_batch_wait / ba_batch_run (milliseconds; default 600000; 0 = none): the kernels of a launch wait for each other without a
 * give-up, so the host bounds the wait -- past it the call fails (ba_last_error) instead of never returning. Process-wide. A timeout is not a failed
 * launch: the batch stays in flight (results and a relaunch are refused; a later ba_batch_wait may still succeed -- raise the limit, or pass 0, for
 * launches that legitimately take longer), and ba_batch_destroy blocks until the device has let go of the batch's memory. */
void ba_set_wait_limit_ms(uint64_t ms);
/* Copy results to host arrays of n_pairs elements; any pointer may be NULL. status: 0 = ok, else BA_ST_* bits. */
int ba_batch_results(BaBatch* batch, int32_t* score, uint32_t* query_idx, uint32_t* reference_idx, uint64_t* cells,
                     uint32_t* cigar_len, uint32_t* status);
/* CIGAR runs of all pairs, concatenated in pair order (pair p occupies cigar_len[p] entries after the pairs before it).
 * `capacity` = number of uint32 entries available in `runs`; fails if too small. */
int ba_batch_cigars(BaBatch* batch, uint32_t* runs, uint64_t capacity);
/* Optional, between ba_batch_launch and ba_batch_wait: gather the CIGAR runs on the device right behind the alignment kernels -- into
 * `pinned_out` (memory from ba_host_alloc, capacity in runs; after ba_batch_wait the runs are in host memory and ba_batch_cigars on the same
 * pointer copies nothing), or with pinned_out = NULL into a device buffer (ba_batch_cigars is then one device-to-host copy). Either way no
 * kernel or copy has to find room beside another batch's launch afterwards. */
int ba_batch_compact_cigars(BaBatch* batch, uint32_t* pinned_out, uint64_t pinned_capacity);
/* TRACE batches: per pair, the sum of width x height over the rectangles left on its trace stack (Trace::blocks(),
 * scan_block.rs:1676-1691; the numerator of the reference's "DP fraction", examples/uc_accuracy.rs:88-89). */
int ba_batch_surviving_cells(BaBatch* batch, uint64_t* cells);
/* Facts about the launch: out[0] grid (resident waves), [1] LDS bytes per wave, [2] trace arena bytes, [3] padded pool bytes */
int ba_batch_info(BaBatch* batch, uint64_t out[4]);
/* Which fill kernel the batch's launches use: 0 the per-pair kernel (k_align), 1 four pairs per wave at 128 cells (k_multi), 2 the round-2/3
 * small-block pipeline (k_quad + queue; profile batches), 3 sixteen pairs per wave at 32 cells (k_small). -1 for a null batch. */
int ba_batch_kernel(BaBatch* batch);
/* k_multi batches: the launch geometry chosen for the batch size -- 0: eight-wave workgroups at four waves per SIMD (batches of many rounds); 3 / 2: four-wave
 * workgroups at three / two waves per SIMD (DNA, block classes 512 and 1024: batches whose pairs fill that many waves' slots about once). -1 for a null batch. */
int ba_batch_geometry(BaBatch* batch);
/* X-drop + BA_TRACE batches: cells of the last run's speculative, untraced rectangles (the chain of grows that closes an X-drop alignment
 * can lie on no path: filled without trace flags and location bookkeeping) -- a part of the computed cells that needed 14 instead of 20
 * int16 operations per cell (bench.py: roofline.ops_required). */
int ba_batch_spec_cells(BaBatch* batch, uint64_t* cells);
/* Large TRACE batches size their trace slots for the expected stack, not for the reference's worst case (Trace::new,
 * scan_block.rs:1363-1366); pairs that outgrow a slot are re-run with full-size slots inside ba_batch_run / ba_batch_wait. (Large: from 4096 pairs, or
 * from 256 pairs of 10 kbp and more.) Likewise a batch whose block range starts at 128 .. 1024 cells and ends above 2048 is launched in the 2048-cell class;
 * pairs whose block wants to grow past 2048 cells are re-run in the row-tiled class, and a batch of which more than an eighth did is launched in the
 * row-tiled class from its next run on.
 * Number of pairs the last run re-ran (results are identical either way; -1 for a null batch). */
int ba_batch_retried(BaBatch* batch);
void ba_batch_destroy(BaBatch* batch);

/* ---- every pair with its own block range. The reference's callers choose the range per pair -- percent_len(max(|q|, |r|), 0.01) ..=
 * percent_len(max(|q|, |r|), p) in /root/reference/examples/nanopore_bench_global.rs:144-171, Block::align(..., min..=max, x) in
 * src/scan_block.rs:847 --, while ba_batch_create takes one range for the whole batch. These calls bin the pairs by (min, max), align every bin as
 * a batch of its own (same kernels, same results as ba_batch_create with that range) and give results and CIGAR runs back in the caller's order. */
typedef struct BaSizedBatch BaSizedBatch;
BaSizedBatch* ba_sized_batch_create(int kind, const void* matrix, struct Gaps gaps, const struct SizeRange* size_per_pair, int32_t x_drop, uint32_t mode,
                                    const uint8_t* pool, const uint64_t* q_off, const uint32_t* q_len, const uint64_t* r_off, const uint32_t* r_len,
                                    uintptr_t n_pairs);
/* ... the range of pair p = block_percent_len(max(|q|, |r|), min_percent) .. block_percent_len(max(|q|, |r|), max_percent) (lib.rs:109-111) */
BaSizedBatch* ba_sized_batch_create_percent(int kind, const void* matrix, struct Gaps gaps, float min_percent, float max_percent, int32_t x_drop, uint32_t mode,
                                            const uint8_t* pool, const uint64_t* q_off, const uint32_t* q_len, const uint64_t* r_off, const uint32_t* r_len,
                                            uintptr_t n_pairs);
/* The bins are launched together (each on its own stream, sharing the device) and waited for. kernel_ms (optional): host wall-clock milliseconds from the
 * first launch to the last completion -- overflow re-runs and the host's work between the waits included; NOT a sum of HIP-event times: the bins overlap
 * (per-bin event times of the same run: ba_sized_batch_classes). */
int ba_sized_batch_run(BaSizedBatch* batch, float* kernel_ms);
int ba_sized_batch_results(BaSizedBatch* batch, int32_t* score, uint32_t* query_idx, uint32_t* reference_idx, uint64_t* cells, uint32_t* cigar_len,
                           uint32_t* status);
int ba_sized_batch_cigars(BaSizedBatch* batch, uint32_t* runs, uint64_t capacity);
/* the bins: their ranges, pair counts, fill kernels (ba_batch_kernel) and kernel times of the last run; any pointer may be NULL; returns the number of bins */
int ba_sized_batch_classes(BaSizedBatch* batch, struct SizeRange* ranges, uint64_t* counts, int32_t* kernels, float* kernel_ms, int capacity);
void ba_sized_batch_destroy(BaSizedBatch* batch);

/* ---- one batch over several GPUs of a node (pairs are independent, so the batch shards without any exchange
 * step). The pair list is cut into contiguous cost-balanced slices (cost = |q| + |r|), one per entry of `devices` (an
 * entry may repeat a device); every slice is a batch of its own, built by its own host thread and launched on its own
 * stream. Results and CIGAR runs come back in the caller's pair order, exactly as from a single ba_batch_*. */
typedef struct BaMultiBatch BaMultiBatch;
BaMultiBatch* ba_multibatch_create(int kind, const void* matrix, struct Gaps gaps, struct SizeRange size, int32_t x_drop, uint32_t mode,
                                   const uint8_t* pool, const uint64_t* q_off, const uint32_t* q_len, const uint64_t* r_off,
                                   const uint32_t* r_len, uintptr_t n_pairs, const int* devices, int n_devices);
/* Launch on every device, then wait for all. kernel_ms (optional) = the longest device's kernel time. */
int ba_multibatch_run(BaMultiBatch* batch, float* kernel_ms);
int ba_multibatch_results(BaMultiBatch* batch, int32_t* score, uint32_t* query_idx, uint32_t* reference_idx, uint64_t* cells,
                          uint32_t* cigar_len, uint32_t* status);
int ba_multibatch_cigars(BaMultiBatch* batch, uint32_t* runs, uint64_t capacity);
/* slice boundaries: bounds[k] .. bounds[k + 1] are the pairs of devices[k]; returns the number of slices */
int ba_multibatch_parts(BaMultiBatch* batch, uint64_t* bounds, int capacity);
/* kernel time (ms, HIP events on the slice's own stream) of every slice in the last ba_multibatch_run; returns the number of slices */
int ba_multibatch_kernel_ms(BaMultiBatch* batch, float* ms, int capacity);
void ba_multibatch_destroy(BaMultiBatch* batch);
/* The slicing rule on its own (no device needed): bounds[0 .. parts] for contiguous slices of near-equal summed |q| + |r|. */
int ba_shard_slices(const uint32_t* q_len, const uint32_t* r_len, uintptr_t n_pairs, int parts, uint64_t* bounds);

enum { BA_ST_TRACE_OVERFLOW = 1, BA_ST_BLOCKS_OVERFLOW = 2, BA_ST_CIGAR_OVERFLOW = 4, BA_ST_TRACEBACK_LOST = 8, BA_ST_WATCHDOG = 16,
       BA_ST_SLOT_TIMEOUT = 32 /* never reported since round 4 (a fill wave that waits for a trace slot walks pending tracebacks itself); kept for ABI stability */,
       BA_ST_MODE = 64 /* FREE_QUERY_END_GAPS reached a down step: the reference panics there */ };
/* (bit 128 is the library's own: a pair of a block range that ends above 2048 cells wanted to grow past the 2048-cell class its batch was launched in -- ba_batch_wait
 * runs such pairs again in the row-tiled class before it returns, so the bit is never reported) */

/* One-shot convenience over create/run/results/cigars/destroy. */
int block_batch_align(int kind, const void* matrix, struct Gaps gaps, struct SizeRange size, int32_t x_drop, uint32_t mode,
                      const uint8_t* pool, const uint64_t* q_off, const uint32_t* q_len, const uint64_t* r_off,
                      const uint32_t* r_len, uintptr_t n_pairs, struct AlignResult* results, uint32_t* cigar_runs,
                      uint64_t cigar_capacity, uint32_t* cigar_len);

/* Block::align_exp / align_profile_exp over a batch (scan_block.rs:884-902, 974-992): every pair is aligned with the min
 * block size size.min; the pairs whose score stays below target_score go through another kernel pass with the min size
 * doubled, until they reach the target or the min size would exceed size.max. reached_min[p] = the min block size at
 * which pair p reached the target, 0 if it never did (results[p] is then that of the last attempt, as in the reference).
 * Scores and end positions only: BA_TRACE / BA_CIGAR_EQ are ignored (trace the finished pairs with ba_batch_create). */
int block_batch_align_exp(int kind, const void* matrix, struct Gaps gaps, struct SizeRange size, int32_t x_drop, int32_t target_score,
                          uint32_t mode, const uint8_t* pool, const uint64_t* q_off, const uint32_t* q_len, const uint64_t* r_off,
                          const uint32_t* r_len, uintptr_t n_pairs, struct AlignResult* results, uintptr_t* reached_min);
int block_batch_align_profile_exp(const struct AAProfile* const* profiles, struct SizeRange size, int32_t x_drop, int32_t target_score,
                                  uint32_t mode, const uint8_t* pool, const uint64_t* q_off, const uint32_t* q_len, uintptr_t n_pairs,
                                  struct AlignResult* results, uintptr_t* reached_min);

#ifdef __cplusplus
}
#endif
#endif /* BLOCK_ALIGNER_HIP_H */
