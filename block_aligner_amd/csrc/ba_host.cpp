// block_aligner_amd — host side of the C ABI declared in include/block_aligner_hip.h.
// Owns device memory, builds PaddedBytes images, launches the gfx950 kernels (ba_kernels.hip) and mirrors the
// reference's ffi.rs objects (PaddedBytes, AAMatrix, AAProfile, Cigar, Block handles). No alignment arithmetic is
// done on the host: without a usable HIP device every align call fails (batch API: error code; reference API: abort).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <functional>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include <atomic>
#include <chrono>
#include <thread>

#include "../../include/block_aligner_hip.h"
#include "aa_matrices.inc"
#include "ba_params.h"

using ba::BatchParams;
using ba::BlockRec;

// ------------------------------------------------------------------ kernel entry points (one TU per kind x class)
#define BA_DECL(K, P)                                                                                                 \
    extern "C" hipError_t ba_launch_k##K##_p##P(int, int, unsigned, unsigned, hipStream_t, const BatchParams*);       \
    extern "C" hipError_t ba_occupancy_k##K##_p##P(int, int, unsigned, int*);                                         \
    extern "C" hipError_t ba_launch_s_k##K##_p##P(int, int, unsigned, unsigned, hipStream_t, const BatchParams*);     \
    extern "C" hipError_t ba_occupancy_s_k##K##_p##P(int, int, unsigned, int*);
#define BA_DECL_KIND(K) BA_DECL(K, 1) BA_DECL(K, 2) BA_DECL(K, 4) BA_DECL(K, 8) BA_DECL(K, 16)
BA_DECL_KIND(0) BA_DECL_KIND(1) BA_DECL_KIND(2) BA_DECL_KIND(3)
#define BA_DECL_BIG(K)                                                                                                  \
    extern "C" hipError_t ba_launch_big_k##K##_p32(int, int, unsigned, unsigned, hipStream_t, const BatchParams*);    \
    extern "C" hipError_t ba_occupancy_big_k##K##_p32(int, int, unsigned, int*);
BA_DECL_BIG(0) BA_DECL_BIG(1) BA_DECL_BIG(2) BA_DECL_BIG(3)
#define BA_DECL_BIGS(K)                                                                                                 \
    extern "C" hipError_t ba_launch_bigs_k##K##_p32(int, int, unsigned, unsigned, hipStream_t, const BatchParams*);   \
    extern "C" hipError_t ba_occupancy_bigs_k##K##_p32(int, int, unsigned, int*);
BA_DECL_BIGS(0) BA_DECL_BIGS(1) BA_DECL_BIGS(2) BA_DECL_BIGS(3)
extern "C" hipError_t ba_launch_compact_cigars(hipStream_t, const uint32_t*, const uint64_t*, const uint32_t*, const uint64_t*, uint32_t*, uint32_t);
extern "C" hipError_t ba_launch_cigar_offsets_and_compact(hipStream_t, const uint32_t*, const uint64_t*, const uint32_t*, const uint32_t*, uint64_t*, uint32_t*,
                                                          unsigned long long*, unsigned long long, uint32_t);
extern "C" hipError_t ba_launch_traceback(hipStream_t, const BatchParams*);
extern "C" hipError_t ba_launch_lane_kat(hipStream_t, int, const short*, short*, int, unsigned);
extern "C" hipError_t ba_launch_walk(hipStream_t, const BatchParams*, uint32_t grid);
extern "C" hipError_t ba_launch_merge_retry(hipStream_t, const uint32_t*, uint32_t, const BatchParams*, const BatchParams*, const uint32_t*, uint32_t*);
extern "C" hipError_t ba_launch_pack_sequences(hipStream_t, int, const uint8_t*, const uint64_t*, const uint64_t*, const uint64_t*, const uint32_t*,
                                               const uint64_t*, const uint32_t*, uint8_t*, uint32_t, uint32_t, unsigned long long*);

typedef hipError_t (*LaunchFn)(int, int, unsigned, unsigned, hipStream_t, const BatchParams*);
typedef hipError_t (*OccFn)(int, int, unsigned, int*);
#define BA_ROW(K) {ba_launch_k##K##_p1, ba_launch_k##K##_p2, ba_launch_k##K##_p4, ba_launch_k##K##_p8, ba_launch_k##K##_p16}
#define BA_OROW(K) {ba_occupancy_k##K##_p1, ba_occupancy_k##K##_p2, ba_occupancy_k##K##_p4, ba_occupancy_k##K##_p8, ba_occupancy_k##K##_p16}
#define BA_SROW(K) {ba_launch_s_k##K##_p1, ba_launch_s_k##K##_p2, ba_launch_s_k##K##_p4, ba_launch_s_k##K##_p8, ba_launch_s_k##K##_p16}
#define BA_SOROW(K) {ba_occupancy_s_k##K##_p1, ba_occupancy_s_k##K##_p2, ba_occupancy_s_k##K##_p4, ba_occupancy_s_k##K##_p8, ba_occupancy_s_k##K##_p16}
// [special modes?][kind][block class]
static const LaunchFn g_launch[2][4][5] = {{BA_ROW(0), BA_ROW(1), BA_ROW(2), BA_ROW(3)}, {BA_SROW(0), BA_SROW(1), BA_SROW(2), BA_SROW(3)}};
static const OccFn g_occ[2][4][5] = {{BA_OROW(0), BA_OROW(1), BA_OROW(2), BA_OROW(3)}, {BA_SOROW(0), BA_SOROW(1), BA_SOROW(2), BA_SOROW(3)}};
// k_multi (ba_multi.hpp): four pairs per wave at 128 cells; sequence kinds only
#define BA_DECL_M(K, P)                                                                                               \
    extern "C" hipError_t ba_launch_m_k##K##_p##P(int, int, unsigned, unsigned, hipStream_t, const BatchParams*);     \
    extern "C" hipError_t ba_occupancy_m_k##K##_p##P(int, int, unsigned, int*);
#define BA_DECL_M_KIND(K) BA_DECL_M(K, 1) BA_DECL_M(K, 2) BA_DECL_M(K, 4) BA_DECL_M(K, 8) BA_DECL_M(K, 16)
BA_DECL_M_KIND(0) BA_DECL_M_KIND(1) BA_DECL_M_KIND(2)
#define BA_MROW(K) {ba_launch_m_k##K##_p1, ba_launch_m_k##K##_p2, ba_launch_m_k##K##_p4, ba_launch_m_k##K##_p8, ba_launch_m_k##K##_p16}
#define BA_MOROW(K) {ba_occupancy_m_k##K##_p1, ba_occupancy_m_k##K##_p2, ba_occupancy_m_k##K##_p4, ba_occupancy_m_k##K##_p8, ba_occupancy_m_k##K##_p16}
static const LaunchFn g_launch_m[3][5] = {BA_MROW(0), BA_MROW(1), BA_MROW(2)};
static const OccFn g_occ_m[3][5] = {BA_MOROW(0), BA_MOROW(1), BA_MOROW(2)};
// ... with slots of 256 cells, two pairs per wave (round 6: DNA batches that start at 256 cells -- percent_len of reads above 12.8 kbp --, block classes 512 .. 2048)
extern "C" hipError_t ba_launch_m256_k1_p4(int, int, unsigned, unsigned, hipStream_t, const BatchParams*);
extern "C" hipError_t ba_launch_m256_k1_p8(int, int, unsigned, unsigned, hipStream_t, const BatchParams*);
extern "C" hipError_t ba_launch_m256_k1_p16(int, int, unsigned, unsigned, hipStream_t, const BatchParams*);
extern "C" hipError_t ba_occupancy_m256_k1_p4(int, int, unsigned, int*);
extern "C" hipError_t ba_occupancy_m256_k1_p8(int, int, unsigned, int*);
extern "C" hipError_t ba_occupancy_m256_k1_p16(int, int, unsigned, int*);
extern "C" hipError_t ba_launch_mg3_k1_p4(int, int, unsigned, unsigned, hipStream_t, const BatchParams*);
extern "C" hipError_t ba_launch_mg3_k1_p8(int, int, unsigned, unsigned, hipStream_t, const BatchParams*);
extern "C" hipError_t ba_launch_mg2_k1_p4(int, int, unsigned, unsigned, hipStream_t, const BatchParams*);
extern "C" hipError_t ba_launch_mg2_k1_p8(int, int, unsigned, unsigned, hipStream_t, const BatchParams*);
extern "C" hipError_t ba_occupancy_mg3_k1_p4(int, int, unsigned, int*);
extern "C" hipError_t ba_occupancy_mg3_k1_p8(int, int, unsigned, int*);
extern "C" hipError_t ba_occupancy_mg2_k1_p4(int, int, unsigned, int*);
extern "C" hipError_t ba_occupancy_mg2_k1_p8(int, int, unsigned, int*);
// k_multi in four-wave workgroups at three / two waves per SIMD (DNA, block classes 512 and 1024): [waves per SIMD - 2][block class]
static const LaunchFn g_launch_mg[2][5] = {{nullptr, nullptr, ba_launch_mg2_k1_p4, ba_launch_mg2_k1_p8, nullptr}, {nullptr, nullptr, ba_launch_mg3_k1_p4, ba_launch_mg3_k1_p8, nullptr}};
static const OccFn g_occ_mg[2][5] = {{nullptr, nullptr, ba_occupancy_mg2_k1_p4, ba_occupancy_mg2_k1_p8, nullptr}, {nullptr, nullptr, ba_occupancy_mg3_k1_p4, ba_occupancy_mg3_k1_p8, nullptr}};
extern "C" hipError_t ba_launch_m512_k1_p8(int, int, unsigned, unsigned, hipStream_t, const BatchParams*);
extern "C" hipError_t ba_launch_m512_k1_p16(int, int, unsigned, unsigned, hipStream_t, const BatchParams*);
extern "C" hipError_t ba_occupancy_m512_k1_p8(int, int, unsigned, int*);
extern "C" hipError_t ba_occupancy_m512_k1_p16(int, int, unsigned, int*);
static const LaunchFn g_launch_m512[5] = {nullptr, nullptr, nullptr, ba_launch_m512_k1_p8, ba_launch_m512_k1_p16};   // k_multi with one slot of 512 cells per wave: [block class]
static const OccFn g_occ_m512[5] = {nullptr, nullptr, nullptr, ba_occupancy_m512_k1_p8, ba_occupancy_m512_k1_p16};
static const LaunchFn g_launch_m256[5] = {nullptr, nullptr, ba_launch_m256_k1_p4, ba_launch_m256_k1_p8, ba_launch_m256_k1_p16};   // [block class]
static const OccFn g_occ_m256[5] = {nullptr, nullptr, ba_occupancy_m256_k1_p4, ba_occupancy_m256_k1_p8, ba_occupancy_m256_k1_p16};
// ... and its LOCAL_START / FREE_QUERY_START_GAPS instantiations (the batch's flags choose)
#define BA_DECL_MS(K, P)                                                                                              \
    extern "C" hipError_t ba_launch_ms_k##K##_p##P(int, int, unsigned, unsigned, hipStream_t, const BatchParams*);    \
    extern "C" hipError_t ba_occupancy_ms_k##K##_p##P(int, int, unsigned, int*);
#define BA_DECL_MS_KIND(K) BA_DECL_MS(K, 1) BA_DECL_MS(K, 2) BA_DECL_MS(K, 4) BA_DECL_MS(K, 8) BA_DECL_MS(K, 16)
BA_DECL_MS_KIND(0) BA_DECL_MS_KIND(1) BA_DECL_MS_KIND(2)
#define BA_MSROW(K) {ba_launch_ms_k##K##_p1, ba_launch_ms_k##K##_p2, ba_launch_ms_k##K##_p4, ba_launch_ms_k##K##_p8, ba_launch_ms_k##K##_p16}
#define BA_MSOROW(K) {ba_occupancy_ms_k##K##_p1, ba_occupancy_ms_k##K##_p2, ba_occupancy_ms_k##K##_p4, ba_occupancy_ms_k##K##_p8, ba_occupancy_ms_k##K##_p16}
static const LaunchFn g_launch_ms[3][5] = {BA_MSROW(0), BA_MSROW(1), BA_MSROW(2)};
static const OccFn g_occ_ms[3][5] = {BA_MSOROW(0), BA_MSOROW(1), BA_MSOROW(2)};
// k_small (ba_small.hpp): sixteen pairs per wave at 32 cells; all four kinds (round 5: sequence-to-profile slots), block classes up to 1024 cells
#define BA_DECL_SM(K, P)                                                                                              \
    extern "C" hipError_t ba_launch_sm_k##K##_p##P(int, int, unsigned, unsigned, hipStream_t, const BatchParams*);    \
    extern "C" hipError_t ba_occupancy_sm_k##K##_p##P(int, int, unsigned, int*);
#define BA_DECL_SM_KIND(K) BA_DECL_SM(K, 1) BA_DECL_SM(K, 2) BA_DECL_SM(K, 4) BA_DECL_SM(K, 8)
BA_DECL_SM_KIND(0) BA_DECL_SM_KIND(1) BA_DECL_SM_KIND(2) BA_DECL_SM_KIND(3)
#define BA_SMROW(K) {ba_launch_sm_k##K##_p1, ba_launch_sm_k##K##_p2, ba_launch_sm_k##K##_p4, ba_launch_sm_k##K##_p8}
#define BA_SMOROW(K) {ba_occupancy_sm_k##K##_p1, ba_occupancy_sm_k##K##_p2, ba_occupancy_sm_k##K##_p4, ba_occupancy_sm_k##K##_p8}
static const LaunchFn g_launch_sm[4][4] = {BA_SMROW(0), BA_SMROW(1), BA_SMROW(2), BA_SMROW(3)};
static const OccFn g_occ_sm[4][4] = {BA_SMOROW(0), BA_SMOROW(1), BA_SMOROW(2), BA_SMOROW(3)};
// ... and its LOCAL_START / FREE_QUERY_START_GAPS instantiations (sequence kinds; the batch's flags choose)
#define BA_DECL_SMS(K, P)                                                                                             \
    extern "C" hipError_t ba_launch_sms_k##K##_p##P(int, int, unsigned, unsigned, hipStream_t, const BatchParams*);   \
    extern "C" hipError_t ba_occupancy_sms_k##K##_p##P(int, int, unsigned, int*);
#define BA_DECL_SMS_KIND(K) BA_DECL_SMS(K, 1) BA_DECL_SMS(K, 2) BA_DECL_SMS(K, 4) BA_DECL_SMS(K, 8)
BA_DECL_SMS_KIND(0) BA_DECL_SMS_KIND(1) BA_DECL_SMS_KIND(2)
#define BA_SMSROW(K) {ba_launch_sms_k##K##_p1, ba_launch_sms_k##K##_p2, ba_launch_sms_k##K##_p4, ba_launch_sms_k##K##_p8}
#define BA_SMSOROW(K) {ba_occupancy_sms_k##K##_p1, ba_occupancy_sms_k##K##_p2, ba_occupancy_sms_k##K##_p4, ba_occupancy_sms_k##K##_p8}
static const LaunchFn g_launch_sms[3][4] = {BA_SMSROW(0), BA_SMSROW(1), BA_SMSROW(2)};
static const OccFn g_occ_sms[3][4] = {BA_SMSOROW(0), BA_SMSOROW(1), BA_SMSOROW(2)};
extern "C" hipError_t ba_launch_walk_l2(hipStream_t, const BatchParams*, uint32_t grid);
extern "C" hipError_t ba_launch_walk_loc(hipStream_t, const BatchParams*, uint32_t grid);
typedef hipError_t (*QuadFn)(int, int, unsigned, hipStream_t, const BatchParams*);
extern "C" hipError_t ba_launch_quad_k0(int, int, unsigned, hipStream_t, const BatchParams*);
extern "C" hipError_t ba_launch_quad_k1(int, int, unsigned, hipStream_t, const BatchParams*);
extern "C" hipError_t ba_launch_quad_k2(int, int, unsigned, hipStream_t, const BatchParams*);
extern "C" hipError_t ba_launch_quad_k3(int, int, unsigned, hipStream_t, const BatchParams*);
extern "C" hipError_t ba_quad_grid_k0(int, int, unsigned*);
extern "C" hipError_t ba_quad_grid_k1(int, int, unsigned*);
extern "C" hipError_t ba_quad_grid_k2(int, int, unsigned*);
extern "C" hipError_t ba_quad_grid_k3(int, int, unsigned*);
typedef hipError_t (*QuadGridFn)(int, int, unsigned*);
static const QuadGridFn g_quad_grid[4] = {ba_quad_grid_k0, ba_quad_grid_k1, ba_quad_grid_k2, ba_quad_grid_k3};
static const QuadFn g_launch_quad[4] = {ba_launch_quad_k0, ba_launch_quad_k1, ba_launch_quad_k2, ba_launch_quad_k3};
// [special modes?][kind]
static const LaunchFn g_launch_big[2][4] = {{ba_launch_big_k0_p32, ba_launch_big_k1_p32, ba_launch_big_k2_p32, ba_launch_big_k3_p32},
                                            {ba_launch_bigs_k0_p32, ba_launch_bigs_k1_p32, ba_launch_bigs_k2_p32, ba_launch_bigs_k3_p32}};
static const OccFn g_occ_big[2][4] = {{ba_occupancy_big_k0_p32, ba_occupancy_big_k1_p32, ba_occupancy_big_k2_p32, ba_occupancy_big_k3_p32},
                                      {ba_occupancy_bigs_k0_p32, ba_occupancy_bigs_k1_p32, ba_occupancy_bigs_k2_p32, ba_occupancy_bigs_k3_p32}};
constexpr int BA_PCLASS_BIG = 5;
static inline int special_of(uint32_t mode) { return (mode & (BA_LOCAL_START | BA_FREE_QUERY_START_GAPS | BA_FREE_QUERY_END_GAPS)) ? 1 : 0; }
constexpr int BA_KIND_PROFILE_ = ba::KIND_PROFILE;   // batches whose "reference" is an AAProfile (sequence bytes: AA alphabet)

// ------------------------------------------------------------------ development switches
// The release library reads NO environment variables: its results and its launch geometry depend on the call's arguments only.
// Built with -DBA_DEV (lib/libblock_aligner_hip_dev.so: the tests that force a code path on small inputs, the measurement
// scripts under tools/) the switches below exist; they are listed in README.md.
#ifdef BA_DEV
#define dev_env(name) getenv(name)
#else
#define dev_env(name) ((const char*)nullptr)
#endif
#ifndef BA_BUILD_ID
#define BA_BUILD_ID "unknown"
#endif
extern "C" const char* ba_build_id(void) { return BA_BUILD_ID; }   // tools/kernel_hash.py over the kernel sources at build time
extern "C" int ba_dev_build(void) {
#ifdef BA_DEV
    return 1;
#else
    return 0;
#endif
}

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return 1;
}
[[noreturn]] static void die(const char* fmt, ...) {   // reference contract: assert! -> panic = abort (Cargo.toml:41)
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    fprintf(stderr, "block_aligner_hip: %s\n", buf);
    abort();
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail("%s failed: %s", #expr, hipGetErrorString(e_)); } while (0)

static thread_local int g_device = -1;   // per host thread: one thread may drive each GPU (ba_set_device; BaMultiBatch does)
static int ensure_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail("no usable HIP device (hipGetDeviceCount: %s); block_aligner_hip has no CPU fallback", hipGetErrorString(e));
    if (g_device < 0) g_device = 0;
    HIP_TRY(hipSetDevice(g_device));
    return 0;
}

// ------------------------------------------------------------------ matrices (layouts fixed by the reference ABI)
struct AAMatrix { alignas(32) int8_t scores[27 * 32]; };
struct NucMatrix { alignas(32) int8_t scores[8 * 16]; };

static constexpr AAMatrix expand_tri(const signed char* tri) {
    AAMatrix m{};
    for (int k = 0; k < 27 * 32; k++) m.scores[k] = -128;
    int idx = 0;
    const char* letters = BA_AA_LETTERS;
    for (int a = 0; a < BA_AA_NLETTERS; a++)
        for (int b = 0; b <= a; b++) {
            const int ia = letters[a] - 'A', ib = letters[b] - 'A';
            m.scores[ia * 32 + ib] = tri[idx];
            m.scores[ib * 32 + ia] = tri[idx];
            idx++;
        }
    return m;
}
static constexpr NucMatrix nuc_simple(int8_t match, int8_t mismatch) {   // scores.rs:150-164
    NucMatrix m{};
    for (int k = 0; k < 8 * 16; k++) m.scores[k] = -128;
    const char alpha[5] = {'A', 'T', 'C', 'G', 'N'};
    for (int i = 0; i < 5; i++)
        for (int j = 0; j < 5; j++) m.scores[(alpha[i] & 7) * 16 + (alpha[j] & 15)] = i == j ? match : mismatch;
    return m;
}
extern "C" {
extern const NucMatrix NW1 = nuc_simple(1, -1);
extern const AAMatrix BLOSUM45 = expand_tri(BA_TRI_BLOSUM45);
extern const AAMatrix BLOSUM50 = expand_tri(BA_TRI_BLOSUM50);
extern const AAMatrix BLOSUM62 = expand_tri(BA_TRI_BLOSUM62);
extern const AAMatrix BLOSUM80 = expand_tri(BA_TRI_BLOSUM80);
extern const AAMatrix BLOSUM90 = expand_tri(BA_TRI_BLOSUM90);
extern const AAMatrix PAM100 = expand_tri(BA_TRI_PAM100);
extern const AAMatrix PAM120 = expand_tri(BA_TRI_PAM120);
extern const AAMatrix PAM160 = expand_tri(BA_TRI_PAM160);
extern const AAMatrix PAM200 = expand_tri(BA_TRI_PAM200);
extern const AAMatrix PAM250 = expand_tri(BA_TRI_PAM250);
extern const ByteMatrix BYTES1 = {1, -1};
}

static inline uint8_t upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }
static inline int seq_kind(int kind) { return kind == BA_KIND_PROFILE_ ? BA_KIND_AA : kind; }
static inline uint8_t null_byte(int kind) { kind = seq_kind(kind); return kind == BA_KIND_AA ? 26 : (kind == BA_KIND_NUC ? 'Z' : 0); }
// convert_char of scores.rs:130-134 / 212-216 / 270-272; returns false on a byte the reference would assert on
static inline bool convert_char(int kind, uint8_t c, uint8_t* out) {
    kind = seq_kind(kind);
    if (kind == BA_KIND_BYTES) { *out = c; return true; }
    c = upper(c);
    if (kind == BA_KIND_AA) { if (c < 'A' || c > 'A' + 26) return false; *out = (uint8_t)(c - 'A'); return true; }
    if (c < 'A' || c > 'Z') return false;
    *out = c;
    return true;
}

// ------------------------------------------------------------------ host mirrors of the reference's opaque objects
struct PaddedBytes {   // scan_block.rs:1790-1884
    std::vector<uint8_t> s;
    size_t len;
    int kind;
};
struct Cigar {         // cigar.rs:42-145; runs kept in alignment order
    std::vector<OpLen> ops;
    size_t capacity;   // query_len + reference_len + 5 (cigar.rs:50)
};
struct AAProfile {     // scores.rs:452-468
    std::vector<int8_t> pos_aa;                    // [curr_len][32]
    std::vector<int16_t> gap_open_C, gap_close_C, gap_open_R;
    int8_t gap_extend;
    size_t max_len, curr_len, str_len;
};

// ------------------------------------------------------------------ device batch
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool owned = true;
    void view(void* q, size_t n) { free_(); p = q; bytes = n; owned = false; }   // a window into another buffer
    int alloc(size_t n) {
        free_();
        owned = true;
        if (n == 0) n = 4;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) { p = nullptr; return fail("hipMalloc(%zu bytes) failed: %s", n, hipGetErrorString(e)); }
        bytes = n;
        return 0;
    }
    void free_() { if (p && owned) (void)hipFree(p); p = nullptr; bytes = 0; }
    ~DevBuf() { free_(); }
    template <class T> T* as() const { return (T*)p; }
};

struct BaBatch {
    int device = 0;
    hipStream_t stream = nullptr, stream2 = nullptr;   // stream2: pair-slot small-block batches (see batch_launch)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_fork = nullptr, ev_join = nullptr, ev_l0 = nullptr, ev_m1 = nullptr;   // ev_l0 / ev_m1: batch_retry
    int kind = 0; uint32_t mode = 0;
    uint32_t n = 0, min_size = 0, max_size = 0, pclass = 0;
    int gap_open = 0, gap_extend = 0, x_drop = 0;
    uint32_t grid = 0, lds = 0, slots = 0;   // grid = workgroups of WAVES_PER_WG waves; slots = resident waves
    uint64_t trace_stride = 0, blocks_stride = 0, cig_total = 0, pool_bytes = 0;
    uint64_t trace_full = 0;    // trace words per slot by the reference's worst-case bound (Trace::new)
    bool adaptive = false;      // trace_stride < trace_full: pairs that overflow their slot are re-run by batch_wait
    bool opt_class = false;     // round 6: max_size belongs to the row-tiled class, the launch is the 2048-cell class's; pairs whose block wants past 2048 cells are re-run by batch_wait
    bool bet_lost = false;      // ... and the last run re-ran more than an eighth of its pairs: the next launch is the row-tiled class's (batch_drop_class_bet)
    uint64_t plan_fixed = 0, plan_maxlen2 = 0, plan_avg = 0;   // batch_plan's arguments, for that second plan
    uint32_t retried = 0;       // pairs the last run had to re-run with full-size slots
    uint64_t cap_n = 0, cap_pool = 0, cap_cig = 0, cap_maxlen2 = 0;   // what the device buffers were sized for (ba_batch_reload)
    DevBuf pool, q_off, q_len, r_off, r_len, matrix, score, qidx, ridx, cig_ops, cig_off, cig_len, cells, status, nblocks, pair_slot, trace_words, trace, blocks, ckpt, big, counter,
           tb_queue, tb_ctrl, slot_free, slot_info, prof, params_dev, donate;
    std::vector<uint32_t> h_order;   // device order -> caller's pair index (empty: identical); see Packed::order
    uint32_t tb_stride = 0, n_fill_waves = 0, slots_per_wave = 1, tb_qsize = 1, tb_reserve = 0;
    std::vector<uint64_t> h_q_off, h_r_off;   // padded offsets (host copy, for the per-handle traceback)
    bool ran = false, in_flight = false;
    uint32_t work_chunk = 1;    // pairs a wave takes per work-counter atomic (short pairs outrun one counter's ~90 atomics / us)
    uint32_t mq_drain = 0;      // k_multi: pairs at the end of the batch that are run one at a time (BatchParams::mq_drain)
    bool multi = false;         // the batch starts at 128 cells: four pairs per wave while a pair's block is 128 cells (ba_multi.hpp)
    uint32_t multi_b = 128;     // ... or at 256 cells (round 6): two pairs per wave, slots of 32 lanes
    uint32_t geom = 0;          // k_multi, round 6: 0 = workgroups of WAVES_PER_WG waves at four waves per SIMD; 3 / 2 = four-wave workgroups at that many waves per SIMD (batch_build)
    uint32_t wpw = ba::WAVES_PER_WG;   // waves per workgroup of the batch's launch (MQ_GEOM_WPW with geom)
    uint32_t walk_wave_n = 0;   // k_walk: the first walk_wave_n pairs of the batch order are walked one to a wave (plan_walks)
    uint32_t sm_excl_n = 0;     // k_small: the batch's longest pairs, run one to a wave (plan_exclusive)
    uint32_t sm_side_n = 0;     // ... of which the first sm_side_n run in a launch of their own beside the main one (TRACE batches: batch_launch)
    bool small = false;         // small-block batch: sixteen pairs per wave while their block is 32 cells, everything else by the same wave (ba_small.hpp)
    bool quad = false;          // small-block batch: pairs run 4 per wave while their block is 32 cells (ba_quad.hpp)
    DevBuf contA, cont_n;       // the PairCont records of the pairs k_quad hands to the per-pair kernel, and the per-pair flags
    DevBuf cq_queue, cq_ctrl;   // the queue those pairs travel through (ba_params.h)
    uint32_t quad_grid = 0, cq_grid = 0;   // workgroups of k_quad / of the per-pair kernel that runs beside it
    // pair-slot batch (a small-block batch with TRACE): every pair owns a region of the trace / record arenas for the whole
    // batch (offsets below, n + 1 entries each), the fill kernels only stack, and k_walk does all tracebacks at the end
    bool pipe = false;
    DevBuf trace_off, blocks_off;
    uint64_t pipe_words = 0, pipe_recs = 0;   // arena capacities (ba_batch_reload re-cuts them)
    // CIGAR runs gathered on the device right behind the alignment kernels (ba_batch_compact_cigars): the later ba_batch_cigars is then one
    // device-to-host copy -- no kernel that would have to wait for room beside another batch's persistent launch
    DevBuf compact, compact_off, compact_total, dev_of;
    uint64_t compact_cap = 0, compact_used_cap = 0; bool compacted = false; uint32_t* compact_host = nullptr;
    unsigned long long* h_total = nullptr;   // page-locked mailbox the gather writes its total to (read without a copy)
    bool handle_mode = false;   // the device state of one Block handle: one pair per launch, CIGARs only on request (k_traceback)
    DevBuf hblk, rblk;          // handle mode: everything uploaded per align / everything read back, one buffer each
    BatchParams params() const {
        BatchParams bp{};
        bp.pool = pool.as<uint8_t>();
        bp.q_off = q_off.as<uint64_t>(); bp.q_len = q_len.as<uint32_t>();
        bp.r_off = r_off.as<uint64_t>(); bp.r_len = r_len.as<uint32_t>();
        bp.n = n; bp.gap_open = gap_open; bp.gap_extend = gap_extend;
        bp.min_size = min_size; bp.max_size = max_size; bp.x_drop = x_drop; bp.flags = mode | (dev_env("BA_NO_FAST") ? 0x100u : 0u) | ((dev_env("BA_SKIP_WALK") || dev_env("BA_NO_TRACEBACK")) ? 0x200u : 0u) | ((handle_mode || dev_env("BA_NO_SPEC")) ? 0x400u : 0u) | (dev_env("BA_NO_TB_WAVES") ? 0x800u : 0u) | (dev_env("BA_NO_REFILL") ? 0x1000u : 0u) | (dev_env("BA_NO_STEAL") ? 0x2000u : 0u);   // (0x400: no speculative grows -- a handle's trace may be walked from any cell)
        bp.matrix = matrix.as<int8_t>();
        bp.score = score.as<int32_t>(); bp.query_idx = qidx.as<uint32_t>(); bp.reference_idx = ridx.as<uint32_t>();
        bp.cig_ops = ((mode & BA_TRACE) && !handle_mode && !dev_env("BA_NO_TRACEBACK")) ? cig_ops.as<uint32_t>() : nullptr;   // env: development switch
        bp.cig_off = cig_off.as<uint64_t>(); bp.cig_start = nullptr; bp.cig_len = cig_len.as<uint32_t>();
        bp.cells = cells.as<unsigned long long>(); bp.status = status.as<uint32_t>(); bp.nblocks_out = nblocks.as<uint32_t>(); bp.slot_out = pair_slot.as<uint32_t>(); bp.trace_words_out = trace_words.as<uint32_t>();
        bp.walk_wave_n = walk_wave_n;
        bp.trace_arena = trace.as<uint32_t>(); bp.trace_stride = trace_stride;
        bp.blocks = blocks.as<BlockRec>(); bp.blocks_stride = blocks_stride;
        bp.trace_off = pipe ? trace_off.as<uint64_t>() : nullptr; bp.blocks_off = pipe ? blocks_off.as<uint64_t>() : nullptr;
        bp.ckpt = ckpt.as<short>(); bp.big = big.as<short>();
        bp.tb_stride = tb_stride; bp.slots_per_wave = slots_per_wave; bp.n_slots = slots; bp.tb_reserve = tb_reserve;
        bp.tb_qmask = tb_qsize - 1;
        bp.tb_queue = tb_queue.as<uint32_t>(); bp.tb_ctrl = tb_ctrl.as<uint32_t>();
        bp.slot_free = slot_free.as<uint32_t>(); bp.slot_info = slot_info.as<ba::SlotInfo>();
        bp.work_counter = counter.as<uint32_t>();
        bp.work_chunk = work_chunk;
        bp.mq_drain = mq_drain;
        bp.mq_waves = grid * wpw;
        bp.mq_donate = (multi && (mode & BA_TRACE) && !special_of(mode) && donate.p && !dev_env("BA_NO_DONATE")) ? donate.as<uint32_t>() : nullptr;   // (the special modes: no slot donation)   // (the score-only kernels are compiled without the end-of-batch code)
        bp.sm_excl_n = small ? sm_excl_n : 0; bp.sm_excl_first = 0;
        bp.prof = prof.as<unsigned long long>();
        return bp;
    }
    ~BaBatch() {
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (h_total) (void)hipHostFree(h_total);
        if (ev_l0) (void)hipEventDestroy(ev_l0);
        if (ev_m1) (void)hipEventDestroy(ev_m1);
        if (stream2) (void)hipStreamDestroy(stream2);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

static int check_align_params(bool profile, Gaps g, size_t min_size, size_t max_size, int32_t x_drop, uint32_t mode, std::string* why) {
    // scan_block.rs:847-862 (align) / 942-956 (align_profile: only the extend cost, which lives in the profile)
    if (profile) { if (!(g.extend < 0)) { *why = "Gap extend cost must be negative!"; return 1; } }
    else {
        if (!(g.open < 0 && g.extend < 0)) { *why = "Gap costs must be negative!"; return 1; }
        if (!(g.open < g.extend)) { *why = "Gap open must cost more than gap extend!"; return 1; }
    }
    if (!(min_size < 65535 && max_size < 65535)) { *why = "Block sizes must be smaller than 2^16 - 1!"; return 1; }
    if ((min_size & (min_size - 1)) || (max_size & (max_size - 1))) { *why = "Block sizes must be powers of two!"; return 1; }
    if ((mode & BA_X_DROP) && x_drop < 0) { *why = "X-drop threshold amount must be nonnegative!"; return 1; }
    if ((mode & BA_LOCAL_START) && (mode & BA_FREE_QUERY_START_GAPS)) { *why = "Cannot set both LOCAL_START and FREE_QUERY_START_GAPS!"; return 1; }
    if ((mode & BA_X_DROP) && (mode & BA_FREE_QUERY_END_GAPS)) { *why = "Cannot set both X_DROP and FREE_QUERY_END_GAPS!"; return 1; }
    return 0;
}

constexpr size_t BA_MAX_BLOCK = 32768;   // the reference takes any power of two below 65535 (scan_block.rs:855)
static int pclass_of(size_t max_size) {   // index into {1,2,4,8,16} packed registers per lane; 5 = tiled (blocks above 2048 cells)
    if (max_size <= 128) return 0;
    if (max_size == 256) return 1;
    if (max_size == 512) return 2;
    if (max_size == 1024) return 3;
    if (max_size == 2048) return 4;
    if (max_size == 4096 || max_size == 8192 || max_size == 16384 || max_size == 32768) return BA_PCLASS_BIG;
    return -1;
}
static inline uint32_t lds_class_cells(int pc) { return pc == BA_PCLASS_BIG ? 128u : 128u << pc; }   // LDS layout class of a kernel class

// Build a batch from already-converted byte ranges. `get(p, which, &ptr, &len)` yields pair p's query (0) / reference (1).
// AAProfile -> device image (layout in ba_params.h). `P` positions, defaults (-128) beyond the profile's own length.
static void profile_image(const AAProfile* pr, uint32_t P, uint8_t* dst) {
    int8_t* pos_aa = (int8_t*)dst;
    int16_t* aa_pos = (int16_t*)(dst + (size_t)P * 32);
    int16_t* goC = aa_pos + (size_t)P * 32; int16_t* clC = goC + P; int16_t* goR = clC + P;
    const size_t have = std::min<size_t>(P, pr->curr_len);
    memset(pos_aa, 0x80, (size_t)P * 32);
    memcpy(pos_aa, pr->pos_aa.data(), have * 32);
    for (size_t c = 0; c < 32; c++)
        for (size_t i = 0; i < P; i++) aa_pos[c * P + i] = (int16_t)pos_aa[i * 32 + c];    // scores.rs:552-553 keeps both orders in sync
    for (size_t i = 0; i < P; i++) {
        goC[i] = i < have ? pr->gap_open_C[i] : (int16_t)-128;
        clC[i] = i < have ? pr->gap_close_C[i] : (int16_t)-128;
        goR[i] = i < have ? pr->gap_open_R[i] : (int16_t)-128;
    }
}
struct NoProfiles { const AAProfile* operator()(size_t) const { return nullptr; } };

constexpr size_t POOL_SLACK = 256;
struct Packed {   // host-side packing of a set of pairs: padded images + the per-pair arrays the kernels read
    std::vector<uint64_t> qo, ro, cig_off;
    std::vector<uint32_t> ql, rl;
    std::vector<uint8_t> image;
    uint64_t total = 0, maxlen2 = 0, cig_total = 0;
    // Sequences that come out of one contiguous host buffer are not padded on the host: the raw bytes [raw, raw + raw_bytes)
    // go to the device as they are and k_pack_sequences builds the images there (raw_qo / raw_ro: offsets into raw).
    bool on_device = false;
    const uint8_t* raw = nullptr; uint64_t raw_bytes = 0; uint32_t pad = 0;
    std::vector<uint64_t> raw_qo, raw_ro;
    // Device order: entry s of every array above describes the caller's pair order[s] (empty = the caller's order). The
    // waves take pairs in device order, which is longest first (greedy longest-processing-time): no wave starts a long pair
    // when the rest of the batch is done. Lognormal protein lengths (22..8881, SURVEY 8d's config 4): +16 % score only,
    // +31 % with traceback against the caller's order (tools/lpt_experiment.py). Results are handed back in the caller's order.
    std::vector<uint32_t> order;
    size_t who(size_t s) const { return order.empty() ? s : order[s]; }
};
template <class GetSeq, class GetProfile, class Lap>
static int pack_pairs_in_order(int kind, Gaps gaps, size_t min_size, size_t max_size, uint32_t mode, size_t n, bool already_converted,
                               GetSeq get, GetProfile getp, Packed& P, Lap lap) {
    const bool profile = kind == BA_KIND_PROFILE_;
    // ---- PaddedBytes images: [NULL] + bytes + NULL x (max_size + 16)
    const size_t pad = max_size + 16;
    std::vector<uint64_t>& qo = P.qo; std::vector<uint64_t>& ro = P.ro;
    qo.resize(n); ro.resize(n);
    std::vector<uint32_t>& ql = P.ql; std::vector<uint32_t>& rl = P.rl;
    ql.resize(n); rl.resize(n);
    uint64_t total = 0, maxlen2 = 0, cig_total = 0;
    std::vector<uint64_t>& cig_off = P.cig_off;
    cig_off.resize(n + 1);
    const uint8_t* lo = nullptr; const uint8_t* hi = nullptr; uint64_t sum_len = 0;   // extent of the caller's bytes
    auto span = [&](const uint8_t* ptr, size_t len) {
        if (!lo || ptr < lo) lo = ptr;
        if (!hi || ptr + len > hi) hi = ptr + len;
        sum_len += len;
    };
    for (size_t p = 0; p < n; p++) {
        const uint8_t* ptr; size_t len;
        get(p, 0, &ptr, &len);
        if (len > 0x3fffffffu) { fail("sequence too long"); return 1; }
        span(ptr, len);
        ql[p] = (uint32_t)len; qo[p] = total; total += (1 + len + pad + 3) & ~(size_t)3;   // images start 4-byte aligned
        if ((mode & BA_FREE_QUERY_END_GAPS) && !(min_size > len)) {   // scan_block.rs:860-862
            fail("pair %zu: Min block size must be larger than the query length for FREE_QUERY_END_GAPS!", P.who(p)); return 1;
        }
        if (profile) {
            const AAProfile* pr = getp(p);
            if (!pr) { fail("pair %zu: null profile", P.who(p)); return 1; }
            if (pr->gap_extend != gaps.extend) { fail("pair %zu: profile gap_extend %d differs from the batch's %d", P.who(p), pr->gap_extend, gaps.extend); return 1; }
            len = pr->str_len;
            if (len > 0x3fffffffu) { fail("profile too long"); return 1; }
            rl[p] = (uint32_t)len; ro[p] = total; total += ba::profile_image_bytes((uint32_t)len, (uint32_t)max_size);
        } else {
            get(p, 1, &ptr, &len);
            if (len > 0x3fffffffu) { fail("sequence too long"); return 1; }
            span(ptr, len);
            rl[p] = (uint32_t)len; ro[p] = total; total += (1 + len + pad + 3) & ~(size_t)3;
        }
        maxlen2 = std::max<uint64_t>(maxlen2, (uint64_t)ql[p] + rl[p] + 2);
        cig_off[p] = cig_total;
        cig_total += (uint64_t)ql[p] + rl[p] + 1;
    }
    cig_off[n] = cig_total;
    total += POOL_SLACK;   // behind the last image: the kernels read whole 128-byte rows of a block unpredicated (prefetch_seq)
    lap("offsets");
    P.total = total; P.maxlen2 = maxlen2; P.cig_total = cig_total; P.pad = (uint32_t)pad;
    // One dense host buffer (the pooled batch calls): ship it as it is, pad and convert on the device. Scattered or
    // overlapping sources (PaddedBytes handles, a reference shared by many pairs) and profile batches are packed here.
    if (!profile && !already_converted && n >= 256 && n < (1u << 30) && (uint64_t)(hi - lo) <= 2 * sum_len + 4096 && !dev_env("BA_HOST_PACK")) {
        P.on_device = true; P.raw = lo; P.raw_bytes = (uint64_t)(hi - lo);
        P.raw_qo.resize(n); P.raw_ro.resize(n);
        for (size_t p = 0; p < n; p++) {
            const uint8_t* ptr; size_t len;
            get(p, 0, &ptr, &len); P.raw_qo[p] = (uint64_t)(ptr - lo);
            get(p, 1, &ptr, &len); P.raw_ro[p] = (uint64_t)(ptr - lo);
        }
        lap("raw offsets");
        return 0;
    }
    std::vector<uint8_t>& image = P.image;
    image.assign(total, null_byte(kind));
    lap("image allocation + padding");
    {   // convert / copy the sequences into their padded images: independent per pair, spread over host threads
        unsigned nthreads = std::thread::hardware_concurrency();
        if (nthreads > 16) nthreads = 16;
        if (nthreads < 1 || n < 4096) nthreads = 1;
        std::vector<size_t> bad(nthreads, (size_t)-1);
        std::vector<uint8_t> bad_byte(nthreads, 0);
        auto work = [&](unsigned t) {
            for (size_t p = (size_t)n * t / nthreads; p < (size_t)n * (t + 1) / nthreads; p++) {
                if (profile) profile_image(getp(p), ba::profile_positions(rl[p], (uint32_t)max_size), image.data() + ro[p]);
                for (int w = 0; w < (profile ? 1 : 2); w++) {
                    const uint8_t* ptr; size_t len;
                    get(p, w, &ptr, &len);
                    uint8_t* dst = image.data() + (w ? ro[p] : qo[p]) + 1;
                    if (already_converted) memcpy(dst, ptr, len);
                    else for (size_t k = 0; k < len; k++) {
                        if (!convert_char(kind, ptr[k], dst + k)) { if (bad[t] == (size_t)-1) { bad[t] = p; bad_byte[t] = ptr[k]; } break; }
                    }
                }
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nthreads; t++) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
        for (unsigned t = 0; t < nthreads; t++)
            if (bad[t] != (size_t)-1) { fail("pair %zu: byte 0x%02x is outside the matrix alphabet", P.who(bad[t]), bad_byte[t]); return 1; }
    }
    lap("image fill");
    P.total = total; P.maxlen2 = maxlen2; P.cig_total = cig_total;
    return 0;
}

template <class GetSeq, class GetProfile, class Lap>
static int pack_pairs(int kind, Gaps gaps, size_t min_size, size_t max_size, uint32_t mode, size_t n, bool already_converted,
                      GetSeq get, GetProfile getp, Packed& P, Lap lap) {
    std::vector<uint64_t> cost(n);   // cells to fill ~ (|q| + |r|) x block size
    for (size_t p = 0; p < n; p++) {
        const uint8_t* ptr; size_t len;
        get(p, 0, &ptr, &len); cost[p] = len;
        if (kind == BA_KIND_PROFILE_) { const AAProfile* pr = getp(p); cost[p] += pr ? pr->str_len : 0; }
        else { get(p, 1, &ptr, &len); cost[p] += len; }
    }
    P.order.resize(n);
    for (size_t p = 0; p < n; p++) P.order[p] = (uint32_t)p;
    std::stable_sort(P.order.begin(), P.order.end(), [&](uint32_t a, uint32_t c) { return cost[a] > cost[c]; });
    bool identity = true;
    for (size_t p = 0; p < n && identity; p++) identity = P.order[p] == p;
    if (identity || dev_env("BA_CALLER_ORDER")) P.order.clear();
    lap("longest-first order");
    const std::vector<uint32_t>& order = P.order;
    if (order.empty()) return pack_pairs_in_order(kind, gaps, min_size, max_size, mode, n, already_converted, get, getp, P, lap);
    return pack_pairs_in_order(kind, gaps, min_size, max_size, mode, n, already_converted,
                               [&](size_t s, int w, const uint8_t** ptr, size_t* len) { get(order[s], w, ptr, len); },
                               [&](size_t s) { return getp(order[s]); }, P, lap);
}

// Pair-slot batches: pair p's region of the trace arena holds its expected stack -- every step at the minimum block size, one
// grow sequence to the maximum, a few steps there -- times a margin (pct), never more than the reference's bound for the pair
// (Trace::new, scan_block.rs:1363-1366); its record list one entry per 4 residues. Pairs that outgrow their region report
// BA_ST_TRACE_OVERFLOW and are re-run by batch_wait. Returns the arena sizes in words / records.
static void pipe_regions(const BaBatch* b, const uint32_t* ql, const uint32_t* rl, size_t n, uint64_t pct, std::vector<uint64_t>& toff,
                         std::vector<uint64_t>& boff) {
    toff.resize(n + 1); boff.resize(n + 1);
    const uint64_t mx = b->max_size, mn = b->min_size;
    const uint64_t zm = (b->mode & BA_LOCAL_START) ? 2 : 1;   // (LOCAL_START: the zero masks, see batch_plan)
    // (k_small: the arena starts with the waves' sinks -- 64 lanes x 16 words each, where the slots without a step put their trace stores;
    // sized for any launch geometry: 4 workgroups of 8 waves per CU on up to 512 CUs)
    uint64_t t = b->small ? 512ull * 32 * 64 * 16 + 64 : 0, r = 0;
    t = (t + 15) & ~15ull;
    for (size_t p = 0; p < n; p++) {
        const uint64_t len2 = (uint64_t)ql[p] + rl[p] + 2;
        const uint64_t full = (mx / 16) * (len2 + 2 * mx) * 2 * zm + 64;
        const uint64_t want = (len2 * mn / 8 + mx * mx / 8 + 16 * mx) * zm * pct / 100 + 4096;
        toff[p] = t; t += (std::min(want, full) + 15) & ~15ull;
        boff[p] = r; r += len2 / 4 + 64;
    }
    toff[n] = t; boff[n] = r;
}
// Cut the arenas for the pairs (ql, rl). cap_words / cap_recs = 0: choose the margin by the free device memory (the caller
// allocates pipe_words / pipe_recs afterwards); otherwise the regions must fit the existing arenas. Returns 1 if they cannot.
// Device memory a batch may plan with: what is free, or -- while ba_sized_batch_create builds one batch per block range -- this batch's share of it
// (the ranges' batches are alive together and launched one after the other: each sized as if it were alone, the first ones took everything)
static thread_local uint64_t g_mem_cap = ~0ull;
static void mem_info(size_t* free_b, size_t* total_b) {
    *free_b = 0; *total_b = 0;
    (void)hipMemGetInfo(free_b, total_b);
    if ((uint64_t)*free_b > g_mem_cap) *free_b = (size_t)g_mem_cap;
}
static int pipe_cut(BaBatch* b, const uint32_t* ql, const uint32_t* rl, size_t n, uint64_t fixed_bytes, std::vector<uint64_t>& toff, std::vector<uint64_t>& boff) {
    size_t free_b = 0, total_b = 0;
    mem_info(&free_b, &total_b);
    uint64_t forced = 0;   // (development / test switch, as for the ring slots)
    if (const char* env = dev_env("BA_TRACE_MARGIN_PCT")) { int v = atoi(env); if (v > 0) forced = (uint64_t)v; }
    // (LOCAL_START: pairs that start in unrelated sequence sit at the maximum block size until the alignment is found -- 1 kbp pairs behind 100..300
    // unrelated bases, 50 k pairs: 2.2 % of them outgrew 175 % and were run again, 0.15 % outgrow 300 %)
    const bool local = b->mode & BA_LOCAL_START;
    for (uint64_t pct : {local ? 300ull : 175ull, local ? 175ull : 140ull, 110ull}) {
        if (forced) pct = forced;
        pipe_regions(b, ql, rl, n, pct, toff, boff);
        if (b->pipe_words) { if (toff[n] <= b->pipe_words && boff[n] <= b->pipe_recs) return 0; }
        else if (toff[n] * 4 + boff[n] * sizeof(BlockRec) + fixed_bytes + (1ull << 30) <= (uint64_t)free_b * 85 / 100) return 0;
    }
    return 1;
}

// Launch geometry and scratch sizes of a batch whose inputs are known: workgroups, LDS, trace slot size, traceback waves,
// slots per wave, hand-off ring. fixed_bytes = device memory the batch needs besides its per-wave scratch.
static int batch_plan(BaBatch* b, size_t n, uint64_t fixed_bytes, uint64_t maxlen2, bool full_trace, uint64_t avg_len2 = ~0ull) {
    const int kind = b->kind; const uint32_t mode = b->mode; const int pc = (int)b->pclass; const size_t max_size = b->max_size;
    const bool trace = mode & BA_TRACE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, b->device) != hipSuccess) { fail("hipGetDeviceProperties failed"); return 1; }
    // (sized by the block class 128 << pc, as the kernels lay it out, + the traceback wave's windows)
    // (k_multi's traceback waves keep their records in their own wave's region: no extra space)
    b->lds = b->multi ? ba::mq_wg_bytes_h(kind, lds_class_cells(pc), b->wpw) : (b->small ? ba::sm_wg_bytes_h(kind, lds_class_cells(pc)) : ba::lds_wg_bytes_h(kind, lds_class_cells(pc)) + (trace ? (special_of(mode) ? ba::TB_LDS_BYTES_LOC : ba::TB_LDS_BYTES) : 0u));   // (the special modes' walk records also hold zero-mask bits)
    if (b->lds > 160 * 1024) { fail("block size %zu needs %u bytes of LDS per workgroup", max_size, b->lds); return 1; }
    if (b->lds > 64 * 1024) {
        // handled per kernel in the launcher TU (hipFuncSetAttribute) -- see ba_kernels.hip
    }
    int per_cu = 0;
    const OccFn occ = pc == BA_PCLASS_BIG ? g_occ_big[special_of(mode)][kind] : (b->multi ? (b->geom ? g_occ_mg[b->geom - 2][pc] : b->multi_b == 512 ? g_occ_m512[pc] : b->multi_b == 256 ? g_occ_m256[pc] : special_of(mode) ? g_occ_ms[kind][pc] : g_occ_m[kind][pc]) : (b->small ? (special_of(mode) ? g_occ_sms[kind][pc] : g_occ_sm[kind][pc]) : g_occ[special_of(mode)][kind][pc]));
    if (occ(trace, (mode & BA_X_DROP) != 0, b->lds, &per_cu) != hipSuccess || per_cu <= 0) {
        fail("occupancy query failed for kind %d class %d (lds %u)", kind, pc, b->lds); return 1;
    }
    if (per_cu * b->wpw > 32) per_cu = 32 / b->wpw;
    if (b->geom && per_cu > (int)b->geom) per_cu = (int)b->geom;   // (a score-only instantiation needs fewer registers than its geometry allows: still that many waves per SIMD)
    if (const char* env = dev_env("BA_WGS_PER_CU")) { int v = atoi(env); if (v > 0) per_cu = v; }
    uint64_t grid = (uint64_t)prop.multiProcessorCount * per_cu;
    // (k_multi, score-only batches, round 6: no more waves than the batch fills with four pairs each -- a wave that finds fewer goes through them one
    // at a time, solo. 10 kbp pairs, two waves per SIMD: 8 k pairs 16.4 -> 12.1 ms, 7 k 18.3 -> 16.1; three: 11 k 20.0 -> 18.2. With traceback the spare
    // waves are the ones that walk: 8 k 23.9 -> 25.2 ms, 11 k 27.9 -> 28.1 -- not cut.)
    const uint64_t per_wg = (uint64_t)b->wpw * (b->small ? ba::SM_SLOTS : ((b->multi && b->geom && !trace) ? 512u / b->multi_b : 1u));   // (the eight-wave workgroups: 16 k pairs 22.9 -> 24.2 ms, not cut)
    const uint64_t need = (n + per_wg - 1) / per_wg;
    if (grid > need) grid = need;
    if (const char* env = dev_env("BA_GRID")) { int v = atoi(env); if (v > 0 && (uint64_t)v < grid) grid = (uint64_t)v; }   // (development: fewer workgroups)
    // trace stack capacity per slot: same bound as Trace::new (scan_block.rs:1363-1366), in 32-bit words
    // (LOCAL_START keeps a zero-mask word behind every trace word: x2)
    const uint64_t zm = (mode & BA_LOCAL_START) ? 2 : 1;
    b->trace_full = trace ? (uint64_t)(max_size / 16) * (maxlen2 + 2 * max_size) * 2 * zm + 64 : 0;   // (+ 64 words of slack: unpredicated stores of the fast path)
    b->trace_stride = b->trace_full;
    // That bound assumes the block sits at its maximum size for the whole alignment (11.9 MB per 10 kbp pair at 1024, of
    // which config 3 uses 1.8 MB). Large batches get slots sized for the expected stack instead -- every step at the minimum
    // size, one full grow sequence to the maximum, a few steps there -- times a margin; the few pairs that outgrow
    // their slot report BA_ST_TRACE_OVERFLOW on the device and are re-run with full-size slots by batch_wait.
    b->adaptive = false;
    // (... and the batches of a sized batch's block ranges, which share the device memory: full-size slots of long pairs at large blocks -- 140 MB
    // each -- left a range's batch a few dozen resident waves)
    // (round 6: ... and batches of long pairs -- a full-size slot of a 32 kbp pair at 4096 cells is 140 MB, 2500 of them left room for 1816 resident waves)
    if (trace && !full_trace && !dev_env("BA_FULL_TRACE_SLOTS") && (n >= 4096 || (maxlen2 >= 20000 && n >= 256) || g_mem_cap != ~0ull || dev_env("BA_ADAPTIVE_TRACE"))) {
        const uint64_t est = (maxlen2 * b->min_size / 8 + (uint64_t)max_size * max_size / 8 + 16ull * max_size) * zm;
        // margin over the expected stack, in percent (development / test switch: BA_TRACE_MARGIN_PCT). LOCAL_START, short pairs: see pipe_cut (long
        // pairs: the flanks are a small part of the stack, and a slot of 3 x 2 x the plain size halves the number of resident waves)
        uint64_t pct = ((mode & BA_LOCAL_START) && maxlen2 <= 4096) ? 500 : 175;   // (round 6: 500 -- at 300 the bench's LOCAL_START line re-ran 33 of its 50 000 pairs in a second launch; these slots are ~1 MB)
        // (k_multi, round 5: nine smaller slots per wave instead of eight -- config 3, same box: 8 x 175 % 163.1 ms / 104.9 GB, 9 x 150 % 161.5 / 101.2,
        // 10 x 125 % 161.9 / 93.8, 9 x 125 % 161.6 / 84.4; its stacks reach 93 % of the expected size, and a pair that outgrows its slot is run again)
        if (b->multi && !(mode & BA_LOCAL_START)) pct = 125;
        if (const char* env = dev_env("BA_TRACE_MARGIN_PCT")) { int v = atoi(env); if (v > 0) pct = (uint64_t)v; }
        const uint64_t want = est * pct / 100 + 4096;
        if (want < b->trace_full) { b->trace_stride = (want + 15) & ~15ull; b->adaptive = true; }   // (16-word multiples: LOCAL_START stores word pairs)
    }
    b->blocks_stride = trace ? maxlen2 : 0;
    if (b->trace_stride >= (1ull << 30)) { fail("trace stack of %llu words per pair exceeds the 2^30 limit", (unsigned long long)b->trace_stride); return 1; }
    if (trace) {
        // very long pairs: a trace slot can be hundreds of MB, so fewer waves may be resident than the chip could hold --
        // shrink the launch until one slot per wave fits in device memory (a long pair keeps its wave busy for long anyway)
        size_t free_b = 0, total_b = 0;
        mem_info(&free_b, &total_b);
        const uint64_t per_slot = b->trace_stride * 4 + b->blocks_stride * sizeof(BlockRec);
        const uint64_t fixed = fixed_bytes + (1ull << 30);
        const uint64_t budget = free_b * 9 / 10 > fixed ? free_b * 9 / 10 - fixed : 0;
        const uint64_t max_waves = per_slot ? budget / per_slot : ~0ull;
        if (max_waves < b->wpw) { fail("device memory: one workgroup's trace slots need %llu MB, %llu MB are free", (unsigned long long)(per_slot * b->wpw >> 20), (unsigned long long)(budget >> 20)); return 1; }
        if (grid * b->wpw > max_waves) grid = max_waves / b->wpw;
    }
    if (b->small) grid = std::min<uint64_t>(grid, 512ull * 32 / b->wpw);   // (the trace arena's sink area holds 512 x 32 waves: pipe_regions)
    b->grid = (uint32_t)grid;
    b->cq_grid = 0;
    if (b->quad) {   // the per-pair kernel beside k_quad: one workgroup per CU, so that k_quad always finds room next to it
        b->cq_grid = (uint32_t)std::min<uint64_t>(grid, (uint64_t)prop.multiProcessorCount);
        if (const char* env = dev_env("BA_CQ_GRID")) { int v = atoi(env); if (v > 0) b->cq_grid = (uint32_t)std::min<uint64_t>(grid, (uint64_t)v); }
    }
    // TRACE batches big enough to keep them busy get dedicated traceback waves (ba_driver.hpp traceback_consumer)
    // and several trace slots per fill wave, so a wave can start its next pair while earlier ones are being walked.
    b->tb_stride = 0; b->slots_per_wave = 1;
    b->n_fill_waves = b->grid * b->wpw;
    // Short pairs: a walk is a few hundred dependent steps, cheaper done at once by the fill wave's lane 0 than handed to a
    // traceback lane (protein pairs of ~300 residues, block 32..256: 202 vs 99 GCUPS; 1 kbp DNA pairs already prefer the hand-off).
    const bool short_pairs = kind != BA_KIND_PROFILE_ && avg_len2 <= 1024 && !dev_env("BA_FORCE_TB");
    {   // several short pairs per work-counter atomic; long pairs one by one (a chunk of long pairs would lengthen the launch's ragged end)
        const uint64_t waves = grid * b->wpw;
        const uint64_t by_len = avg_len2 != ~0ull ? 16384 / (avg_len2 + 1) : 1, by_n = waves ? n / (waves * 8) : 1;
        b->work_chunk = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(8, std::min(by_len, by_n)));
        if (b->small) b->work_chunk = 4;   // (sixteen slots refill one by one; pairs come longest first: a long chunk would queue the longest pairs on one wave)
        // (round 5: eight once the batch gives every slot two pairs and more -- 200 k 1 kbp pairs 8.15 -> 8.05 ms, 400 k protein pairs 3.74 -> 3.66; 2: 8.34 / 4.12; 16: 8.17 / 3.72)
        if (b->small && n >= 2 * waves * ba::SM_SLOTS) b->work_chunk = 8;
        if (const char* env = dev_env("BA_WORK_CHUNK")) { int v = atoi(env); if (v > 0) b->work_chunk = (uint32_t)v; }   // (development)
    }
    // Few pairs: while the batch gives a resident wave no more than about two pairs, the wave walks each path itself with all its lanes
    // (walk_wave: ~0.2 - 0.35 us per cell) instead of handing it to a traceback lane (64 walks in lockstep: 0.55 - 0.9 us per cell, the
    // better use of the machine once every wave has pairs waiting) -- 10 kbp pairs, block 128..1024, same box: 256 pairs 22.7 -> 12.6 ms,
    // 2048: 25.3 -> 13.2, 4096 (one per wave): 26.3 -> 19.3, 8192: 35.8 -> 33.5, 12000: 40.3 -> 48.1 (kept on the lanes).
    // (one pair per wave at most: in every block class; two: measured for the classes with large LDS regions only)
    uint64_t few_x = lds_class_cells(pc) >= 512 ? 2u : 1u;
    if (const char* env = dev_env("BA_FEW_PAIRS_X")) { int v = atoi(env); if (v >= 0) few_x = (uint64_t)v; }   // (development)
    const bool few_pairs = !special_of(mode) && n <= few_x * (uint64_t)grid * b->wpw && !dev_env("BA_FORCE_TB");
    if (trace && (b->grid >= 32 || (dev_env("BA_FORCE_TB") && b->grid >= 2)) && !short_pairs && !few_pairs && !b->pipe && !dev_env("BA_INLINE_TRACEBACK")) {
        // one traceback wave per 4 workgroups = per 31 fill waves: at config 3 one per 5 ties and one per 6 is
        // 3.5 % slower, so 4 leaves a margin for workloads with more traceback per filled cell. (Workgroup b runs on XCD
        // b % 8, so the traceback waves sit on XCDs 0 and 4 only; measured against stride 3 / 5 -- all XCDs -- this makes
        // no difference now that a walk runs out of LDS.)
        // (the traceback work per filled cell grows as the block shrinks: with blocks below 128 cells one wave per 3
        // workgroups -- 1 kbp DNA, block 32..256: 360 GCUPS against 285 at one per 4)
        uint32_t stride = b->grid >= 32 ? (b->min_size >= 128 ? 4 : 3) : 2;
        // (k_multi fills faster -- four pairs per wave, eight cells per lane -- and its walks are what it waits for: one traceback wave
        // per 2 workgroups. Config 3, same box: per 4: 217 ms, per 3: 200.4, per 2: 198.9)
        if (b->multi && b->grid >= 32) stride = 2;
        // (round 6: the lanes' walk takes runs of diagonal moves at once -- ~45 % fewer instructions per rectangle crossed -- and one traceback wave per
        // 3 workgroups keeps up: 85 more fill waves. Same box, per 2 / 3 / 4 / 6: 100 k pairs 159.2 / 158.3 / 159.8 / 188.8 ms, 50 k 90.2 / 89.1 / 88.6 / 97.1,
        // 12.5 k 36.4 / 34.2 / 33.9 / 34.1. The special modes' walks are the round-5 ones: per 2 as before.)
        if (b->multi && b->grid >= 32 && !special_of(mode)) stride = 3;
        // (the four-wave workgroups of round 6's geometries: the same share of the waves. 12 k pairs at three waves per SIMD, one per 4 / 6 / 8 workgroups:
        // 33.3 / 29.0 / 30.3 ms; 8 k pairs at two: 3 / 4 / 6: 28.3 / 27.4 / 23.3)
        if (b->multi && b->geom && b->grid >= 32) stride = 6;
        if (const char* env = dev_env("BA_TB_STRIDE")) { int v = atoi(env); if (v > 0) stride = (uint32_t)v; }
        b->tb_stride = stride;
        b->n_fill_waves = b->grid * b->wpw - (b->grid + stride - 1) / stride;
        size_t free_b = 0, total_b = 0;
        mem_info(&free_b, &total_b);
        const uint64_t per_slot = b->trace_stride * 4 + b->blocks_stride * sizeof(BlockRec);
        const uint64_t fixed = fixed_bytes + (1ull << 30);
        uint32_t spw = 4;   // one being filled + three pending walks per fill wave, HBM permitting (188 GB at config 3; 3 slots: -1 %, 2: -17 %, 5: no gain)
        const uint32_t mslots = b->multi ? 512u / b->multi_b : 0u;   // k_multi: pairs a wave fills at once (four slots of 128 cells, or two of 256)
        if (b->multi) spw = mslots + 5;   // four pairs being filled + pending walks (round 4: 8 instead of 10 -- 105 instead of 131 GB at config 3 for -0.7 %; round 5: 9, smaller -- see the margin above; 7: 190 ms, 6: 208)
        if (const char* env = dev_env("BA_SLOTS_PER_WAVE")) { int v = atoi(env); if (v > 0) spw = (uint32_t)v; }
        while (spw > 1 && fixed + per_slot * spw * b->n_fill_waves > free_b * 9 / 10) spw--;
        if (b->multi && spw < mslots + 2) return fail("device memory: the multi-pair kernel needs two trace slots per wave beside its pairs'");   // (batch_build falls back to the per-pair kernel)
        b->slots_per_wave = spw;
        // The last hand-offs of the batch go to fill waves that have run out of pairs (one walking lane per wave, on
        // SIMDs with nothing else left to do): a walk alone is much shorter than one among 40 in lockstep, and the
        // batch ends one walk after its last fill. Half of the fill waves (measured at config 3 with 3968 of them: flat between
        // 1000 and 2500, 0.6 % slower at 3000, more below 500); fewer than all of them, so the fill waves can never all be
        // waiting for trace slots whose walks are reserved for helpers that do not exist yet.
        b->tb_reserve = b->n_fill_waves / 2;
        // (k_multi, round 5: the emptied waves' whole-wave walks take runs of diagonal moves at once -- one per fill wave; config 3, same box:
        // 164.7 -> 163.3 ms, twice as many: 169.0; 50 k pairs 92.1 -> 91.0, 25 k 55.4 -> 53.5; at 16 k pairs, four per wave, half stays better: 42.7 / 43.1)
        // (why all of them is safe here although the rule above says "fewer than all": a k_multi fill wave only waits when every one of its trace
        // slots is taken -- four by its live pairs, the other spw - 4 >= 2 by pending walks --, so if EVERY fill wave waited there would be at
        // least 2 n_fill_waves hand-offs pending, more than the n_fill_waves reserved ones: some of them are the dedicated lanes', which are
        // resident and walking; and a wave that waits long walks a pending hand-off itself, traceback_help_one. Enforced below: spw >= 6.)
        // (round 6, the cheaper lane walk: one per fill wave at every batch size -- 12.5 k pairs 34.2 -> 33.6 ms, 25 k 49.4 -> 49.2, 100 k 158.3 -> 157.9)
        if (b->multi && (n >= 5ull * b->n_fill_waves || !special_of(mode)) && spw >= mslots + 2) b->tb_reserve = b->n_fill_waves;
        // with fewer than three trace slots per wave a fill wave soon waits for the walk of its previous pair: leave
        // less of the batch to walkers that only exist once the first wave has run out of pairs
        if (spw < 3) b->tb_reserve = b->n_fill_waves / 4;
        if (const char* env = dev_env("BA_TB_RESERVE")) b->tb_reserve = (uint32_t)std::max(0, atoi(env));
        // (whatever the switches say: the reserved hand-offs stay below what the fill waves' spare slots can hold pending)
        if (b->multi) b->tb_reserve = std::min<uint32_t>(b->tb_reserve, b->n_fill_waves * (spw > mslots ? spw - mslots : 1) - 1);
    }
    // (round 4: a quarter of the fill waves instead of all of them -- since waves that run out of pairs take over other waves' slots at the
    // end of the batch, fewer pairs need to be kept out of the slots: config 3 178.9 -> 176.7 ms, 25 k pairs 60.5 -> 58.0 ms; 0: the same)
    b->mq_drain = b->multi ? ((dev_env("BA_NO_DONATE") || !trace || special_of(mode)) ? b->n_fill_waves : b->n_fill_waves / 4) : 0;
    if (const char* env = dev_env("BA_MQ_DRAIN")) b->mq_drain = (uint32_t)std::max(0, atoi(env));
    if (b->multi && b->slots_per_wave < 512u / b->multi_b) b->slots_per_wave = 512u / b->multi_b;   // (without the hand-off ring: one trace slot per slot of the wave)
    b->slots = b->n_fill_waves * b->slots_per_wave;
    if (b->pipe) b->slots = (uint32_t)n;   // (slot = pair: slot_info holds one entry per pair for k_walk)
    {
        const uint64_t lanes = b->tb_stride ? (uint64_t)((b->grid + b->tb_stride - 1) / b->tb_stride) * 64 + b->n_fill_waves : 0;   // + one helper lane per fill wave
        uint64_t need_q = std::max<uint64_t>(lanes, b->slots);
        uint32_t qs = 1;
        while (qs < need_q) qs <<= 1;
        b->tb_qsize = qs;
    }

    return 0;
}
static int batch_alloc_scratch(BaBatch* b) {
#define BA_ALLOC(buf, bytes) if (b->buf.alloc(bytes)) return 1
    BA_ALLOC(trace, b->pipe ? b->pipe_words * 4 : b->trace_stride * 4 * b->slots);
    BA_ALLOC(blocks, b->pipe ? b->pipe_recs * sizeof(BlockRec) : b->blocks_stride * sizeof(BlockRec) * b->slots);
    BA_ALLOC(ckpt, (size_t)(b->grid + b->cq_grid) * b->wpw * 8 * b->max_size * sizeof(short));
    BA_ALLOC(big, b->pclass == BA_PCLASS_BIG ? (size_t)b->grid * b->wpw * ba::big_wave_shorts(b->max_size) * sizeof(short)
                                              : (b->multi ? (size_t)b->grid * b->wpw * ba::MQ_WAVE_BYTES : (b->small ? (size_t)b->grid * b->wpw * ba::SM_WAVE_BYTES : 0)));
    BA_ALLOC(tb_queue, (size_t)b->tb_qsize * 4); BA_ALLOC(tb_ctrl, 256); BA_ALLOC(prof, 2048); BA_ALLOC(params_dev, sizeof(BatchParams));
    BA_ALLOC(slot_free, (size_t)b->slots * 4); BA_ALLOC(slot_info, (size_t)b->slots * sizeof(ba::SlotInfo)); BA_ALLOC(counter, 128);
    if (b->multi) BA_ALLOC(donate, ((size_t)b->grid * b->wpw + 64) * 4);   // (+ two counters in their own cache lines)
#undef BA_ALLOC
    return 0;
}

// Sequence images into b->pool; the per-pair offset / length arrays must already be on the device.
static int upload_images(BaBatch* b, const Packed& P, size_t n) {
    if (!P.on_device) { HIP_TRY(hipMemcpy(b->pool.p, P.image.data(), P.total, hipMemcpyHostToDevice)); return 0; }
    DevBuf raw, raw_q, raw_r, err;
    if (raw.alloc(P.raw_bytes) || raw_q.alloc(n * 8) || raw_r.alloc(n * 8) || err.alloc(8)) return 1;
    HIP_TRY(hipMemcpy(raw.p, P.raw, P.raw_bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(raw_q.p, P.raw_qo.data(), n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(raw_r.p, P.raw_ro.data(), n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(err.p, 0xff, 8));
    HIP_TRY(hipMemset((uint8_t*)b->pool.p + P.total - POOL_SLACK, null_byte(b->kind), POOL_SLACK));   // slack behind the last image
    HIP_TRY(ba_launch_pack_sequences(b->stream, seq_kind(b->kind), raw.as<uint8_t>(), raw_q.as<uint64_t>(), raw_r.as<uint64_t>(),
                                     b->q_off.as<uint64_t>(), b->q_len.as<uint32_t>(), b->r_off.as<uint64_t>(), b->r_len.as<uint32_t>(),
                                     b->pool.as<uint8_t>(), P.pad, (uint32_t)n, err.as<unsigned long long>()));
    HIP_TRY(hipStreamSynchronize(b->stream));
    unsigned long long e = 0;
    HIP_TRY(hipMemcpy(&e, err.p, 8, hipMemcpyDeviceToHost));
    if (e != ~0ull) return fail("pair %zu: byte 0x%02x is outside the matrix alphabet", P.who((size_t)(e >> 8)), (unsigned)(e & 0xff));
    return 0;
}

static int upload_dev_of(BaBatch* b);
static void plan_walks(BaBatch* b, const std::vector<uint32_t>& ql, const std::vector<uint32_t>& rl);
static void plan_exclusive(BaBatch* b, const std::vector<uint32_t>& ql, const std::vector<uint32_t>& rl);
// `get(p, which, &ptr, &len)` yields pair p's query (0) / reference (1); for kind PROFILE the reference comes from
// `getp(p)` instead.
template <class GetSeq, class GetProfile = NoProfiles>
static BaBatch* batch_build(int kind, const void* matrix, Gaps gaps, SizeRange size, int32_t x_drop, uint32_t mode, size_t n,
                            bool already_converted, GetSeq get, GetProfile getp = GetProfile()) {
    if (ensure_device()) return nullptr;
    const bool verbose = dev_env("BA_SETUP_TIMING") != nullptr;   // development: where the batch set-up time goes
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!verbose) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "block_aligner_hip setup: %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    if (kind < 0 || kind > 3) { fail("unknown matrix kind %d", kind); return nullptr; }
    const bool profile = kind == BA_KIND_PROFILE_;
    const size_t min_size = size.min < 16 ? 16 : size.min, max_size = size.max < 16 ? 16 : size.max;   // clamp to L (scan_block.rs:853-854)
    std::string why;
    if (check_align_params(profile, gaps, min_size, max_size, x_drop, mode, &why)) { fail("%s", why.c_str()); return nullptr; }
    if (min_size > max_size) { fail("min block size exceeds max block size"); return nullptr; }
    if (profile && (mode & BA_CIGAR_EQ)) { fail("=/X CIGARs need two sequences; a profile alignment has none to compare"); return nullptr; }
    if (kind == BA_KIND_BYTES && (mode & BA_X_DROP)) { /* allowed by the reference, documented as inaccurate (scores.rs:235-239) */ }
    int pc = pclass_of(max_size);
    if (pc < 0) { fail("max block size %zu not supported by the HIP backend (16..%zu)", max_size, (size_t)BA_MAX_BLOCK); return nullptr; }
    if (n == 0 || n > 0x7fffffffu) { fail("batch must hold between 1 and 2^31-1 pairs"); return nullptr; }
    // Round 6: block ranges that end above 2048 cells but start well below (percent_len 1 % .. 10 % of a 20 .. 40 kbp read: 256 / 512 .. 4096). The
    // row-tiled class keeps a pair's borders in global memory and has neither the eight-cells-per-lane rectangles nor the untraced end-game grows --
    // and such a pair's block rarely passes 2048 cells. The batch is launched in the 2048-cell class (LDS borders, every fast path) with its own maximum;
    // a pair whose block wants to grow past 2048 cells stops with ST_CLASS_OVERFLOW and batch_wait runs it again in the row-tiled class (batch_retry).
    // 2500 pairs of 32 kbp at 512..4096, none of which passes 2048 cells, same box: 192.9 -> 140.1 ms with traceback, 132.5 -> 81.8 score only
    // (tools/dev/big_probe.sh).
    bool opt_class = false;
    // (from 128 cells: the small-block pipelines are not in the bet. A batch that loses the bet -- more than an eighth of its pairs re-run -- is launched
    // in the row-tiled class from its next run on: batch_drop_class_bet.)
    if (pc == BA_PCLASS_BIG && min_size >= 128 && min_size <= 1024 && !dev_env("BA_NO_OPT_CLASS")) { pc = 4; opt_class = true; }

    std::unique_ptr<BaBatch> b(new BaBatch);
    b->device = g_device; b->kind = kind; b->mode = mode; b->n = (uint32_t)n;
    b->min_size = (uint32_t)min_size; b->max_size = (uint32_t)max_size; b->pclass = (uint32_t)pc; b->opt_class = opt_class;
    b->gap_open = gaps.open; b->gap_extend = gaps.extend; b->x_drop = x_drop;

    Packed P;
    if (pack_pairs(kind, gaps, min_size, max_size, mode, n, already_converted, get, getp, P, lap)) return nullptr;
    std::vector<uint64_t>& qo = P.qo; std::vector<uint64_t>& ro = P.ro; std::vector<uint32_t>& ql = P.ql; std::vector<uint32_t>& rl = P.rl;
    std::vector<uint64_t>& cig_off = P.cig_off;
    const uint64_t total = P.total, maxlen2 = P.maxlen2, cig_total = P.cig_total;
    b->h_q_off = qo; b->h_r_off = ro; b->h_order = P.order;
    plan_walks(b.get(), ql, rl);
    b->cap_n = n; b->cap_pool = total; b->cap_cig = cig_total; b->cap_maxlen2 = maxlen2;
    b->pool_bytes = total;

    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) { fail("hipStreamCreate failed"); return nullptr; }
    if (hipEventCreate(&b->ev0) != hipSuccess || hipEventCreate(&b->ev1) != hipSuccess) { fail("hipEventCreate failed"); return nullptr; }

    const bool trace = mode & BA_TRACE;
    const size_t mat_bytes = kind == BA_KIND_AA ? 27 * 32 : (kind == BA_KIND_NUC ? 8 * 16 : (profile ? 0 : 2));
    // ---- launch geometry: one wave per workgroup, as many resident waves as LDS / registers allow
    uint64_t sum_len2 = 0;
    for (size_t p = 0; p < n; p++) sum_len2 += (uint64_t)ql[p] + rl[p];
    // Small blocks: a 32-cell block keeps 16 of a wave's 64 lanes busy, so batches that start at 32 cells begin in k_quad, four
    // pairs to a wave, and only the pairs that need more than plain shift steps go on to the per-pair kernel (see batch_launch).
    // Small batches are bound by their longest pair's chain of steps, which the hand-over only lengthens: the pipeline is taken
    // from the sizes where it wins (measured, GCUPS with / without: protein pairs 32k 371 / 405, 64k 745 / 527, 400k 1375 / 597;
    // PSSM 8k 108 / 105, 20k 211 / 162, 80k 458 / 172; 1 kbp DNA 200k 1347 / 606). BA_FORCE_QUAD / BA_NO_QUAD override.
    // (1 kbp DNA, round 3 -- the per-pair kernel walks with whole waves now: 2.5 k pairs 0.78 / 1.89 ms without / with traceback against
    // 1.34 / 2.96 through the pipeline, 4 k 0.82 / 1.96 against 1.40 / 3.07, 8 k 1.37 / 3.18 against 1.44 / 3.31, 10 k 3.59 against 3.33)
    const size_t quad_from = kind == BA_KIND_AA ? 65536u : 8192u;
    const bool trace_mode = mode & BA_TRACE;
    b->quad = !special_of(mode) && pc != BA_PCLASS_BIG && min_size == 32 && !dev_env("BA_NO_QUAD") && (dev_env("BA_FORCE_QUAD") || n >= quad_from);
    // Round 4: sequence kinds take k_small instead (ba_small.hpp) -- sixteen pairs per wave at 32 cells, eight cells per lane, and the grow /
    // shrink / X-drop end game of a pair by the same wave in solo mode: no queue, no launch beside. Block classes up to 1024 cells.
    // From the batch sizes at which sixteen pairs per wave still fill the machine (same-box sweeps, tools/dev/small_sweep.sh; GCUPS k_small /
    // round-3 pipeline / per-pair kernel): 1 kbp DNA X-drop 40 k pairs 941 / 963 / 580, 80 k 1356 / 1154 / 622; with traceback 40 k 512 / 580 /
    // 348, 80 k 791 / 742 / 393; protein pairs 30 k 388 / 303 / 369, 70 k 792 / 651 / 519; with traceback 70 k 258 / 283 / 210, 150 k 487 / 441 /
    // 216. Below them the round-3 rules apply.
    size_t small_from = kind == BA_KIND_AA ? (trace_mode ? 98304u : 32768u) : (trace_mode ? 57344u : 49152u);
    if (profile) small_from = 10000u;   // (round 5: 11 k PSSM pairs 131 against 128 GCUPS, 20 k 227 against 208, 80 k 641 against 459; below: the round-2 pipeline)
    // (round 5: LOCAL_START / FREE_QUERY_START_GAPS batches of the sequence kinds too -- k_small's special instantiations; FREE_QUERY_END_GAPS stays per pair)
    // Same-box sweep (tools/dev/local_sweep.py: 1 kbp DNA pairs behind 100..300 unrelated bases, X-drop 50, block 32..256; GCUPS k_small / per-pair
    // kernel), since the per-pair kernel's register path takes these modes' steps: LOCAL_START with traceback 50 k pairs 367 / 470, 150 k 629 / 614,
    // 300 k 714 / 643; without 150 k 1077 / 1183, 300 k 1222 / 1187. (Before: with traceback 150 k 529 / 478, without 250 k 1009 / 857; with the zero
    // mask at four words per trace word 150 k 440 / 327.) FREE_QUERY_START_GAPS on the same pairs stays behind the per-pair kernel (150 k pairs with
    // traceback 662 / 782, without 1034 / 1294): those batches take k_small only when forced.
    const bool small_mode = !special_of(mode) || (!profile && !(mode & BA_FREE_QUERY_END_GAPS));
    if (mode & BA_LOCAL_START) small_from = trace_mode ? 196608u : 262144u;
    if ((mode & BA_FREE_QUERY_START_GAPS) && !dev_env("BA_FORCE_SMALL")) small_from = ~(size_t)0;
    b->small = small_mode && pc <= 3 && min_size == ba::SM_B_HOST && !dev_env("BA_NO_SMALL") && (dev_env("BA_FORCE_SMALL") || (n >= small_from && !dev_env("BA_FORCE_QUAD")));
    if (b->small) b->quad = false;
    // Pair-slot batches: every pair's trace stack stays in its own region of the arenas until the fill is over, then k_walk
    // walks all paths with one pair per lane. The small-block pipeline needs this form with TRACE; profile batches without small
    // blocks take it in place of the hand-off ring (50..500 positions, 20k pairs: 143 -> 158 GCUPS, 80k: 172 -> 225). Short
    // sequence pairs do not: their walks are few hundred steps, cheapest done at once by the fill wave's lane 0 (protein pairs,
    // 8k..65k pairs: 3 .. 20 % slower with k_walk, whose latest wave ends one longest-path walk after the fill).
    std::vector<uint64_t> toff, boff;
    if (trace && (!special_of(mode) || b->small) && pc != BA_PCLASS_BIG && (b->quad || b->small || (profile && n >= 4096) || dev_env("BA_FORCE_PIPE"))) {
        const uint64_t fixed = total + cig_total * 4 + (uint64_t)n * (64 + sizeof(ba::PairCont) + 40);
        if (!dev_env("BA_NO_PIPE") && !pipe_cut(b.get(), ql.data(), rl.data(), n, fixed, toff, boff)) {
            b->pipe = true; b->pipe_words = toff[n] + toff[n] / 8; b->pipe_recs = boff[n] + boff[n] / 8;   // (headroom for ba_batch_reload)
        }
    }
    if (trace && !b->pipe) b->quad = b->small = false;   // with TRACE the pipeline needs every pair's trace stack resident until the end
    // Batches that start at 128 cells (the reference's nanopore set-up, examples/nanopore_bench.rs: 1 % .. 10 % of 10 kbp): four pairs
    // per wave while a pair's block is 128 cells (ba_multi.hpp), from the sizes at which every wave still finds four pairs.
    // (round 5: LOCAL_START / FREE_QUERY_START_GAPS batches too -- k_multi's special instantiations; FREE_QUERY_END_GAPS stays per pair)
    // Round 5 (tools/dev/multi_from.py, tools/dev/tail_knobs.sh; GCUPS k_multi / per-pair kernel, block 128..512, same box): the pairs must be long enough
    // -- a pair's first block and its end game are the solo driver's, whose traced instantiation spills (section 4): with traceback, 30 k pairs of
    // 1000 bases 1040 / 1105, 1500 bases 1181 / 1085, 2000 bases 1257 / 1075; 300..1500 bases with a 10..120-base indel each 415 / 1014 at 20 k
    // pairs, 667 / 1112 at 200 k (before this rule those batches took k_multi); score-only 1361 / 1808 at 20 k, 2109 / 1925 at 60 k. Long pairs from
    // 12 288 pairs on (config 3's pairs: 10 k 1005 / 975, 12.5 k 1107 / 1054, 14 k 1198 / 1119).
    const uint64_t avg_len2 = n ? sum_len2 / n : 0;
    // (round 6, the cheaper lane walk and one traceback wave per 3 workgroups: config 3's pairs 10 k 1068 / 1050, 12 k 1290 / 1071, 12.5 k 1342 / 1105 -> from 10 000 pairs)
    bool multi_fits = avg_len2 >= 3000 ? n >= 10000 : (!trace_mode && n >= 49152);
    // Round 6: batches that start at 256 cells -- what percent_len gives reads above 12.8 kbp (lib.rs:109-111, examples/nanopore_bench_global.rs:144-171) --
    // take k_multi with two slots of 256 cells per wave: DNA, block classes 512 .. 2048, the plain modes. From the sizes at which every wave finds its two pairs.
    b->multi_b = 128;
    bool wide = false;
    if (min_size == 256 && kind == BA_KIND_NUC && !special_of(mode) && max_size >= 512 && pc >= 2 && pc <= 4) {
        wide = true; b->multi_b = 256;
        // (round 6, later: a wave whose second slot stays empty goes on stepping with the first -- from 256 pairs: 600 pairs of 22 kbp at 256..4096 34.9 ms against the
        // per-pair kernel's 45.4, 1500 of 13 kbp at 256..2048 21.5 / 27.2. The figures of the earlier rule, from 2048 pairs:)
        multi_fits = avg_len2 >= 6000 && n >= 256;   // (13 kbp reads, 256..2048, with traceback, GCUPS two-pair slots / per-pair kernel, same box: 1.5 k pairs 353 / 410, 2.5 k 576 / 401, 4 k 990 / 552, 8 k 1155 / 603, 20 k 1555 / 654, 70 k 1873 / 720)
    }
    // Round 6: the launch geometry of k_multi by the batch size. The kernel is bound by vector issue, so a round of the batch -- every slot of every resident
    // wave filled once -- takes as long as a SIMD has waves; a batch of about one round is fastest on the geometry whose slots it just fills (four-wave
    // workgroups, two / three of them per CU = 32 / 48 slots per CU; at three waves per SIMD the traced driver also spills 171 registers instead of 344), while
    // batches of many rounds want four waves per SIMD (config 3's 100 k pairs: 157.5 ms against 168.7 at three). DNA, block classes 512 and 1024, plain modes.
    // 10 kbp pairs at 128..1024, ms at four / three / two waves per SIMD / per-pair kernel, same box (tools/dev/geom_sweep.sh, geom_sweep2.sh) --
    // with traceback: 6 k pairs 38.4 / 38.6 / 22.2 / 22.5, 8 k 40.9 / 27.0 / 23.3 / 28.6, 10 k 32.6 / 27.3 / 27.4 / 34.1, 12.5 k 33.3 / 30.4 / 32.0 / -, 14 k 34.7 / 32.2 / 34.4,
    // 16 k 35.4 / 34.2 / 36.7, 20 k 41.6 / 42.2, 25 k 48.7 / 53.1; score only: 8 k 18.2 / 22.5 / 16.4 / 15.0, 10 k 22.6 / 22.7 / 18.3 / 19.5, 12 k 27.5 / 18.1 / 23.7 / 22.1,
    // 14 k 24.1 / 19.1, 16 k 22.9 / 26.0, 20 k 26.6 / 30.0.
    b->geom = 0; b->wpw = ba::WAVES_PER_WG;
    const bool geom_ok = !wide && kind == BA_KIND_NUC && !special_of(mode) && (pc == 2 || pc == 3) && min_size == ba::MQ_B_HOST;   // (the instantiations that exist: ba_kernels.hip)
    if (geom_ok && avg_len2 >= 3000) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device) != hipSuccess || cus <= 0) cus = 256;
        const uint64_t s2 = 32ull * (uint64_t)cus, s3 = 48ull * (uint64_t)cus;   // slots of the two geometries
        // (where the geometries change over, tools/dev/geom_edges.sh -- with traceback, two / three waves per SIMD: 8.4 k pairs 25.6 / 27.5 ms, 9 k 26.4 / 27.0, 9.5 k 27.1 / 27.1;
        // three / four: 16 k 34.0 / 35.4, 17.5 k 40.1 / 39.7, 19 k 40.9 / 40.2; two / per-pair kernel: 5.5 k 20.4 / 22.3, 5 k - / 19.2. Score only (with the
        // workgroups cut to the batch, batch_plan; geom_edges3.sh): two / per-pair 7 k 16.1 / 14.1, 7.6 k 12.0 / 15.0, 9 k 13.5 / 17.2; two / three 9.2 k 13.4 / 17.8,
        // 9.5 k 18.3 / 17.7, 10.5 k 19.1 / 17.9; three / four 15 k 20.5 / 22.8, 15.5 k 26.0 / 23.0 -- steps, not slopes: a wave whose slots cannot all be refilled
        // finishes its pairs one at a time)
        if (trace_mode) {
            b->geom = n <= s2 * 115 / 100 ? 2u : (n <= s3 * 136 / 100 ? 3u : 0u);
            multi_fits = n >= s2 * 65 / 100;
        } else {
            b->geom = n <= s2 * 113 / 100 ? 2u : (n <= s3 * 124 / 100 ? 3u : 0u);
            multi_fits = n >= s2 * 92 / 100;
        }
    }
    if (const char* env = dev_env("BA_MQ_GEOM")) { const int v = atoi(env); b->geom = (geom_ok && (v == 2 || v == 3)) ? (uint32_t)v : 0u; }   // (development / tests: 0 = the eight-wave workgroups)
    // ... and batches that start at 512 cells (percent_len 1 % of reads above 25.6 kbp): one slot of 512 cells per wave -- the same pair per wave as the per-pair
    // kernel, but its plain steps in the slots' loop (the state in registers, the driver's decisions in vector code) instead of the per-pair driver's shell.
    if (min_size == 512 && kind == BA_KIND_NUC && !special_of(mode) && max_size >= 1024 && pc >= 3 && pc <= 4) {
        wide = true; b->multi_b = 512;
        // (32 kbp pairs at 512..4096, one slot per wave / per-pair kernel, same box, ms with traceback (tools/dev/m512_ab.sh): 300 pairs 54.3 / 84.6, 1200 54.6 / 85.0,
        // 2500 84.2 / 136.9, 5000 117.1 / 212.0; score only 600 pairs 26.2 / 48.8, 2500 64.9 / 81.6 -- a pair's own chain of steps is a third shorter)
        multi_fits = avg_len2 >= 8000 && n >= 256;
    }
    b->multi = !profile && small_mode && pc != BA_PCLASS_BIG && (min_size == ba::MQ_B_HOST || wide) && !dev_env("BA_NO_MULTI") && (dev_env("BA_FORCE_MULTI") || multi_fits);
    if (!b->multi) { b->multi_b = 128; b->geom = 0; }
    if (b->geom) b->wpw = (uint32_t)ba::MQ_GEOM_WPW;
    if (b->multi && batch_plan(b.get(), n, total + cig_total * 4 + (uint64_t)n * 64, maxlen2, false, sum_len2 / n)) { b->multi = false; b->geom = 0; b->wpw = ba::WAVES_PER_WG; }
    if (b->small && batch_plan(b.get(), n, total + cig_total * 4 + (uint64_t)n * 64, maxlen2, false, sum_len2 / n)) b->small = false;   // (e.g. LDS: falls back to the per-pair kernel)
    if (!b->multi && !b->small)
    if (batch_plan(b.get(), n, total + cig_total * 4 + (uint64_t)n * 64, maxlen2, false, sum_len2 / n)) return nullptr;
    if (b->pipe) b->adaptive = true;
    b->plan_fixed = total + cig_total * 4 + (uint64_t)n * 64; b->plan_maxlen2 = maxlen2; b->plan_avg = sum_len2 / n;
    plan_exclusive(b.get(), ql, rl);
    b->cig_total = trace ? cig_total : 0;

    lap("launch geometry");
#define BA_ALLOC(buf, bytes) if (b->buf.alloc(bytes)) return nullptr
    BA_ALLOC(pool, total); BA_ALLOC(q_off, n * 8); BA_ALLOC(q_len, n * 4); BA_ALLOC(r_off, n * 8); BA_ALLOC(r_len, n * 4);
    BA_ALLOC(matrix, 1024);
    BA_ALLOC(score, n * 4); BA_ALLOC(qidx, n * 4); BA_ALLOC(ridx, n * 4); BA_ALLOC(cig_len, n * 4); BA_ALLOC(cells, n * 8);
    BA_ALLOC(status, n * 4); BA_ALLOC(nblocks, n * 4); BA_ALLOC(pair_slot, n * 4); BA_ALLOC(trace_words, n * 4);
    BA_ALLOC(cig_off, (n + 1) * 8);
    BA_ALLOC(cig_ops, b->cig_total * 4);
#undef BA_ALLOC
    if (batch_alloc_scratch(b.get())) return nullptr;
    if (b->quad && (b->contA.alloc(n * sizeof(ba::PairCont)) || b->cont_n.alloc(n * 4))) return nullptr;
    if (b->pipe && (b->trace_off.alloc((n + 1) * 8) || b->blocks_off.alloc((n + 1) * 8))) return nullptr;
    if (b->quad) {
        if (b->cq_queue.alloc(n * 4) || b->cq_ctrl.alloc(256)) return nullptr;
        if (g_quad_grid[kind](trace ? 1 : 0, (mode & BA_X_DROP) ? 1 : 0, &b->quad_grid) != hipSuccess || !b->quad_grid) { fail("occupancy query failed for the small-block kernel"); return nullptr; }
    }
    if ((b->quad || b->small) && (hipStreamCreateWithFlags(&b->stream2, hipStreamNonBlocking) != hipSuccess ||
                               hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming) != hipSuccess)) {
        fail("hipStreamCreate / hipEventCreate failed"); return nullptr;
    }
    lap("device allocation");
#define BA_H2D(buf, src, bytes) if (hipMemcpy(b->buf.p, src, bytes, hipMemcpyHostToDevice) != hipSuccess) { fail("hipMemcpy H2D failed"); return nullptr; }
    BA_H2D(q_off, qo.data(), n * 8); BA_H2D(q_len, ql.data(), n * 4);
    BA_H2D(r_off, ro.data(), n * 8); BA_H2D(r_len, rl.data(), n * 4); BA_H2D(cig_off, cig_off.data(), (n + 1) * 8);
    if (b->pipe) { BA_H2D(trace_off, toff.data(), (n + 1) * 8); BA_H2D(blocks_off, boff.data(), (n + 1) * 8); }
    if (upload_images(b.get(), P, n)) return nullptr;
    {
        int8_t tmp[1024] = {0};
        if (kind == BA_KIND_BYTES) { const ByteMatrix* bm = (const ByteMatrix*)matrix; tmp[0] = bm->match_score; tmp[1] = bm->mismatch_score; }
        else if (mat_bytes) memcpy(tmp, matrix, mat_bytes);
        BA_H2D(matrix, tmp, 1024);
    }
#undef BA_H2D
    lap("host-to-device copies");
    if (hipMemset(b->cig_len.p, 0, n * 4) != hipSuccess || hipMemset(b->status.p, 0, n * 4) != hipSuccess) { fail("hipMemset failed"); return nullptr; }
    if (upload_dev_of(b.get())) return nullptr;
    return b.release();
}


// device copy of the caller-order -> device-order map (for the gather of CIGAR runs in the caller's order, ba_batch_compact_cigars)
static int upload_dev_of(BaBatch* b) {
    if (!(b->mode & BA_TRACE) || b->h_order.empty()) return 0;
    std::vector<uint32_t> dev_of(b->n);
    for (uint32_t s = 0; s < b->n; s++) dev_of[b->h_order[s]] = s;
    if (!b->dev_of.p && b->dev_of.alloc((size_t)b->cap_n * 4)) return 1;
    HIP_TRY(hipMemcpy(b->dev_of.p, dev_of.data(), (size_t)b->n * 4, hipMemcpyHostToDevice));
    return 0;
}

// Replace the pairs of an existing batch (same matrix, gaps, block range and modes); the device buffers -- above all the
// trace arena, whose allocation dominates the set-up time -- are reused, so the new set must fit what they were sized for.
template <class GetSeq, class GetProfile = NoProfiles>
static int batch_reload(BaBatch* b, size_t n, bool already_converted, GetSeq get, GetProfile getp = GetProfile()) {
    if (!b) return fail("null batch");
    if (b->in_flight) return fail("reload: the batch has a launch in flight (ba_batch_wait first)");
    if (n == 0 || n > b->cap_n) return fail("reload: %zu pairs exceed the batch's capacity of %llu", n, (unsigned long long)b->cap_n);
    HIP_TRY(hipSetDevice(b->device));
    Packed P;
    if (pack_pairs(b->kind, Gaps{(int8_t)b->gap_open, (int8_t)b->gap_extend}, b->min_size, b->max_size, b->mode, n, already_converted, get, getp, P, [](const char*) {})) return 1;
    if (P.total > b->cap_pool) return fail("reload: %llu sequence bytes exceed the batch's capacity of %llu", (unsigned long long)P.total, (unsigned long long)b->cap_pool);
    if (P.maxlen2 > b->cap_maxlen2) return fail("reload: a pair is longer (%llu) than the longest pair the batch was created with (%llu)", (unsigned long long)P.maxlen2 - 2, (unsigned long long)b->cap_maxlen2 - 2);
    if ((b->mode & BA_TRACE) && P.cig_total > b->cap_cig) return fail("reload: CIGAR capacity exceeded");
    std::vector<uint64_t> toff, boff;
    if (b->pipe && pipe_cut(b, P.ql.data(), P.rl.data(), n, 0, toff, boff)) return fail("reload: the new pairs' trace regions exceed the batch's trace arena");
    // from here on the device arrays change: a failure leaves no pairs loaded (a later launch says so) instead of a mix
    b->n = 0; b->ran = false;
    HIP_TRY(hipMemcpy(b->q_off.p, P.qo.data(), n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->q_len.p, P.ql.data(), n * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->r_off.p, P.ro.data(), n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(b->r_len.p, P.rl.data(), n * 4, hipMemcpyHostToDevice));
    if (upload_images(b, P, n)) return 1;
    HIP_TRY(hipMemcpy(b->cig_off.p, P.cig_off.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    if (b->pipe) {
        HIP_TRY(hipMemcpy(b->trace_off.p, toff.data(), (n + 1) * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(b->blocks_off.p, boff.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemset(b->cig_len.p, 0, n * 4));
    HIP_TRY(hipMemset(b->status.p, 0, n * 4));
    b->n = (uint32_t)n; b->h_q_off = P.qo; b->h_r_off = P.ro; b->h_order = P.order; b->pool_bytes = P.total;
    plan_walks(b, P.ql, P.rl);
    plan_exclusive(b, P.ql, P.rl);
    b->cig_total = (b->mode & BA_TRACE) ? P.cig_total : 0;
    b->ran = false;
    return upload_dev_of(b);
}

// Enqueue one pass over the batch on its stream and return; batch_wait collects it. (Two batches on two streams
// overlap; with ba_batch_reload the host packs the next set while the device aligns the current one.)
// k_walk ends with its longest walk. A lane walks at ~0.55 us per cell, a whole wave on one path (walk_wave) several times faster but one
// path at a time and in scalar code (one scalar unit per CU: a few such walks per CU at most): the leading pairs of the batch order with at
// least 1 / BA_WALK_WAVE_FRAC of the longest pair's |q| + |r| -- at most BA_WALK_WAVE_MAX of them -- go one to a wave, the others one to a
// lane beside them, and the launch's makespan becomes the longest path at the wave's pace.
static void plan_walks(BaBatch* b, const std::vector<uint32_t>& ql, const std::vector<uint32_t>& rl) {
    uint32_t frac = 3, kmax = 1024;
    if (const char* e = dev_env("BA_WALK_WAVE_FRAC")) frac = (uint32_t)std::max(1, atoi(e));
    if (const char* e = dev_env("BA_WALK_WAVE_MAX")) kmax = (uint32_t)std::max(0, atoi(e));
    const size_t n = ql.size();
    uint32_t longest = 0;
    for (size_t p = 0; p < n; p++) longest = std::max(longest, ql[p] + rl[p]);
    const uint32_t from = std::max(longest / frac, 512u);
    // The walkers take the leading walk_wave_n pairs of the batch order. That order is longest first only when pack_pairs sorted it (it keeps
    // the caller's order when that is already sorted -- or when told to): then count the leading run; otherwise no pair is singled out rather
    // than an arbitrary prefix (round-3 advisor finding: an unsorted order stopped the prefix test at its first short pair).
    bool sorted = true;
    for (size_t p = 1; p < n && sorted; p++) sorted = (uint64_t)ql[p] + rl[p] <= (uint64_t)ql[p - 1] + rl[p - 1] + 64;   // (cost order: ties and near-ties in any order)
    size_t cnt = 0;
    if (sorted) while (cnt < n && cnt < kmax && ql[cnt] + rl[cnt] >= from) cnt++;
    // equal-length batches (every pair qualifies): whole-wave walks are for a FEW long paths beside many lane walks, not for a prefix of a
    // uniform batch
    if (cnt == std::min<size_t>(n, kmax) && n > kmax) cnt = 0;
    b->walk_wave_n = (uint32_t)cnt;
}
// k_small: sixteen pairs to a wave make every pair sixteen times as long in flight (and a neighbour's solo episode stops a slot), so a pair
// many times the average length would end the launch alone. The leading pairs of the batch order (longest first) whose own chain of steps
// in a slot would take more than about half of what the whole batch takes -- |q| + |r| above half the batch's residues per slot -- run one
// to a wave instead, on all lanes, before the wave starts its slots; at most one for every second wave.
static void plan_exclusive(BaBatch* b, const std::vector<uint32_t>& ql, const std::vector<uint32_t>& rl) {
    b->sm_excl_n = 0;
    if (!b->small || dev_env("BA_NO_EXCL")) return;
    const size_t n = ql.size();
    uint64_t total = 0;
    for (size_t p = 0; p < n; p++) total += (uint64_t)ql[p] + rl[p];
    const uint64_t slots = (uint64_t)b->grid * ba::WAVES_PER_WG * ba::SM_SLOTS;
    uint64_t thr = std::max<uint64_t>(total / std::max<uint64_t>(slots, 1) / 2, 1024);
    // ... and many times the batch's average length: a batch of equal pairs has none. (Score only: six times (~4000 for the protein set) -- same box, threshold
    // 2400 / 3600 / 4800 / 6000: 3.86 / 3.75 / 3.71 / 4.08 ms; with traceback the pairs' walks are part of the chain: 9.68 / 9.95 / - / 10.06 ms)
    thr = std::max<uint64_t>(thr, ((b->mode & BA_TRACE) ? 4 : 6) * (total / std::max<size_t>(n, 1)));
    // (no "the run did not end inside the cap" rule as in plan_walks: the protein set's 2048 longest of 400 k pairs fill the cap and are exactly the pairs
    // meant -- with that rule none ran alone and the launch took 8.05 instead of 3.8 ms)
    if (const char* e = dev_env("BA_EXCL_LEN2")) thr = (uint64_t)std::max(0, atoi(e));
    const size_t cap = (size_t)b->grid * ba::WAVES_PER_WG / 2;
    // (as in plan_walks: the leading pairs are the longest only when the device order is sorted -- it is not under BA_CALLER_ORDER --; an unsorted order
    // singles out no pair rather than an arbitrary prefix: round-5 advisor finding)
    bool sorted = true;
    for (size_t p = 1; p < n && sorted; p++) sorted = (uint64_t)ql[p] + rl[p] <= (uint64_t)ql[p - 1] + rl[p - 1] + 64;
    size_t cnt = 0;
    if (sorted) while (cnt < n && cnt < cap && (uint64_t)ql[cnt] + rl[cnt] > thr) cnt++;   // (device order: longest first)
    b->sm_excl_n = (uint32_t)cnt;
    // k_walk behind the fill: those pairs are walked by their own waves, so the paths it walks one to a wave (plan_walks: the leading pairs of the
    // batch order) are counted from the first pair that is NOT among them -- round 5: until then every pair k_walk saw went to a lane, and the launch
    // was as long as one lane's walk of the longest of them (400 k protein pairs: 2.08 ms for paths of up to ~2400 cells)
    if ((b->mode & BA_TRACE) && cnt && !special_of(b->mode) && !dev_env("BA_NO_WALK_AFTER_EXCL")) {
        const uint64_t first = (uint64_t)ql[cnt < n ? cnt : n - 1] + rl[cnt < n ? cnt : n - 1];
        const uint64_t from = std::max<uint64_t>(first / 3, 512);
        size_t w = cnt;
        while (w < n && w - cnt < 1024 && (uint64_t)ql[w] + rl[w] >= from) w++;
        b->walk_wave_n = (uint32_t)w;
    }
    // TRACE: the longest of them -- at least half the longest pair's length, at most one wave in thirty-two -- get a launch of their own
    // beside the main one (batch_launch): their fill + walk is a serial chain that outlasts the rest of the batch (400 k protein pairs with
    // traceback: the 8881-residue pair alone takes 8.6 ms, everything else 6 ms), and the walks of all other pairs need not wait for it
    size_t side = 0;
    // (development switch only: the two launches ran beside each other in one batch object and one after the other in the next -- 9.0 against
    // 15.3 ms for 400 k protein pairs with traceback, 10.8 without the second launch)
    if (dev_env("BA_EXCL_SIDE") && ((b->mode & BA_TRACE) || dev_env("BA_EXCL_SIDE_ALL")) && cnt) {
        const uint64_t longest = (uint64_t)ql[0] + rl[0];
        const size_t side_cap = std::max<size_t>(4, (size_t)b->grid / 8);   // (a workgroup of the side launch runs four pairs: at most one workgroup in 32)
        while (side < cnt && side < side_cap && ((uint64_t)ql[side] + rl[side]) * 2 >= longest) side++;
        if (b->grid < 16) side = 0;
    }
    b->sm_side_n = (uint32_t)side;
}
static uint32_t walk_grid(const BaBatch* b) {   // workgroups of k_walk (4 waves x 64 walking lanes each)
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, b->device) != hipSuccess) return 1;
    uint32_t per_cu = 8;
    if (const char* env = dev_env("BA_WALK_WGS_PER_CU")) { int v = atoi(env); if (v > 0) per_cu = (uint32_t)v; }
    return std::min((b->n + 255u) / 256u + (b->walk_wave_n + 3u) / 4u, (uint32_t)prop.multiProcessorCount * per_cu);
}
// A batch launched in the 2048-cell class on the bet that few of its pairs grow past it (opt_class) whose last run lost the bet: plan and allocate it
// again as what its block range says -- the row-tiled class, per-pair kernel.
static int batch_drop_class_bet(BaBatch* b) {
    b->opt_class = false; b->bet_lost = false;
    b->pclass = (uint32_t)BA_PCLASS_BIG; b->multi = false; b->multi_b = 128; b->geom = 0; b->wpw = ba::WAVES_PER_WG;
    if (batch_plan(b, b->n, b->plan_fixed, b->plan_maxlen2, false, b->plan_avg)) return 1;
    return batch_alloc_scratch(b);
}
static int batch_launch(BaBatch* b) {
    if (b->in_flight) return fail("the batch already has a launch in flight (ba_batch_wait first)");
    b->compacted = false;
    if (b->n == 0) return fail("the batch holds no pairs (its last reload failed)");
    HIP_TRY(hipSetDevice(b->device));
    if (b->bet_lost && batch_drop_class_bet(b)) return 1;
    if (!b->handle_mode) {   // (a handle's work counter arrives zeroed with its upload; it has no hand-off structures)
        HIP_TRY(hipMemsetAsync(b->counter.p, 0, 128, b->stream));
        HIP_TRY(hipMemsetAsync(b->tb_ctrl.p, 0, 256, b->stream));
        HIP_TRY(hipMemsetAsync(b->prof.p, 0, 2048, b->stream));
        HIP_TRY(hipMemsetAsync(b->tb_queue.p, 0, (size_t)b->tb_qsize * 4, b->stream));
        HIP_TRY(hipMemsetAsync(b->slot_free.p, 1, (size_t)b->slots * 4, b->stream));   // any non-zero value = free
        if (b->multi && b->donate.p) HIP_TRY(hipMemsetAsync(b->donate.p, 0, ((size_t)b->grid * b->wpw + 64) * 4, b->stream));
    }
    const BatchParams bp = b->params();
    HIP_TRY(hipEventRecord(b->ev0, b->stream));
    if (b->ev_l0) HIP_TRY(hipEventRecord(b->ev_l0, b->stream));   // (a re-run sub-batch: batch_retry re-uses ev0 for the merge)
    const LaunchFn launch = b->pclass == BA_PCLASS_BIG ? g_launch_big[special_of(b->mode)][b->kind] : (b->multi ? (b->geom ? g_launch_mg[b->geom - 2][b->pclass] : b->multi_b == 512 ? g_launch_m512[b->pclass] : b->multi_b == 256 ? g_launch_m256[b->pclass] : special_of(b->mode) ? g_launch_ms[b->kind][b->pclass] : g_launch_m[b->kind][b->pclass]) : g_launch[special_of(b->mode)][b->kind][b->pclass]);
    if (b->quad && b->n <= b->cap_n) {
        // k_quad starts every pair -- its first block and plain shift steps, four pairs per wave -- and finishes the global
        // alignments that never need more. A pair that does (a grow, X-drop termination, fewer than 32 residues) goes through a
        // queue to the per-pair kernel: one small launch of it runs beside k_quad on a second stream (the longest pairs come
        // first and leave early: their remaining steps are a long serial chain best started at once), a full-size one after it.
        // TRACE (a pair-slot batch): k_quad only stacks trace words; the paths of the pairs it finished are walked by k_walk, one
        // pair per lane, the per-pair kernel walks its own pairs' paths at once.
        const int tr = (b->mode & BA_TRACE) ? 1 : 0, xd = (b->mode & BA_X_DROP) ? 1 : 0;
        uint32_t* flagA = b->cont_n.as<uint32_t>();
        HIP_TRY(hipMemsetAsync(flagA, 0, (size_t)b->n * 4, b->stream));
        HIP_TRY(hipMemsetAsync(b->cq_queue.p, 0, (size_t)b->n * 4, b->stream));
        HIP_TRY(hipMemsetAsync(b->cq_ctrl.p, 0, 256, b->stream));
        HIP_TRY(hipEventRecord(b->ev_fork, b->stream));
        BatchParams pq = bp; pq.cont_out = b->contA.as<ba::PairCont>(); pq.cont_out_flag = flagA;
        pq.cq_queue = b->cq_queue.as<uint32_t>(); pq.cq_ctrl = b->cq_ctrl.as<uint32_t>(); pq.cq_producers = b->quad_grid * ba::WAVES_PER_WG;
        pq.work_chunk = 4;   // (one position per slot: pairs come longest first, and a long chunk would queue the longest pairs on one wave)
        HIP_TRY(g_launch_quad[b->kind](tr, xd, b->quad_grid, b->stream, &pq));
        // (the pairs that reach the per-pair kernel -- with X-drop all of them, since termination is not a plain shift step: their
        // paths go to a second k_walk; without, only the pairs that grow: lane 0 walks each at once, nothing is left for the end)
        const bool walk2 = tr && bp.cig_ops && xd;
        pq.inline_len2 = walk2 ? ~0u : 0u;
        BatchParams pc = pq; pc.cont_mode = 2; pc.cont_in = b->contA.as<ba::PairCont>(); pc.cont_in_flag = flagA; pc.cont_out = nullptr; pc.cont_out_flag = nullptr;
        // Beside k_quad -- global alignments only: there the pairs that leave are the ones that grow, few and each a long serial
        // chain (protein pairs, 400k: 905 -> 1350 GCUPS; PSSM 80k: 349 -> 432); with X-drop every pair leaves and the per-pair
        // kernel has a full machine's worth of work after k_quad anyway (1 kbp DNA: 1320 -> 1226 with the side launch).
        // (launched after k_quad: a kernel that only waits must never be the one that holds the device)
        const bool beside = (!xd || dev_env("BA_CQ_BESIDE")) && !dev_env("BA_CQ_AFTER");
        if (beside) {
            BatchParams pa = pc; pa.ckpt_wave0 = b->grid * ba::WAVES_PER_WG; pa.cq_side = 1;
            HIP_TRY(hipStreamWaitEvent(b->stream2, b->ev_fork, 0));
            HIP_TRY(launch(tr, xd, b->cq_grid, b->lds, b->stream2, &pa));
            HIP_TRY(hipEventRecord(b->ev_join, b->stream2));
        }
        if (tr && bp.cig_ops) {
            BatchParams w1 = bp; w1.cont_mode = 1; w1.cont_in_flag = flagA; w1.work_counter = b->counter.as<uint32_t>() + 8;   // (its own counter, zeroed above)
            HIP_TRY(ba_launch_walk(b->stream, &w1, walk_grid(b)));
        }
        HIP_TRY(launch(tr, xd, b->grid, b->lds, b->stream, &pc));
        if (beside) HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_join, 0));
        if (walk2) {
            BatchParams w2 = bp; w2.cont_mode = 2; w2.cont_in_flag = flagA; w2.work_counter = b->counter.as<uint32_t>() + 12;
            HIP_TRY(ba_launch_walk(b->stream, &w2, walk_grid(b)));
        }
    } else if (b->small) {
        // k_small takes every pair from start to end (slots at 32 cells, everything else by the same wave in solo mode); TRACE: a pair-slot
        // batch -- the fill only stacks, k_walk (its form for slot rectangles) walks all paths afterwards
        // (the pairs run one to a wave at the start of the launch -- the batch's longest -- walk their paths at once, with the whole wave:
        // the longest walk of the batch overlaps with the fill instead of ending the launch)
        BatchParams p1 = bp; p1.inline_len2 = ~0u;
        const LaunchFn launch_sm = special_of(b->mode) ? g_launch_sms[b->kind][b->pclass] : g_launch_sm[b->kind][b->pclass];
        const uint32_t side_n = (b->stream2 && !special_of(b->mode) && (bp.cig_ops || !(b->mode & BA_TRACE)) && b->sm_side_n && b->sm_side_n <= b->sm_excl_n) ? b->sm_side_n : 0u;
        uint32_t grid_main = b->grid;
        if (side_n) {
            // The longest pairs' launch: one wave per pair, workgroups taken off the main launch (together they fill the device as one
            // launch would); its waves use the tail of the wave-indexed scratch (arena, checkpoints). k_walk_l2 skips these pairs -- their
            // waves walk them at once --, also while they are still being filled: their hand-over records say "nothing to walk" beforehand.
            const uint32_t grid_side = (side_n + 3) / 4;   // (four of a workgroup's eight waves take pairs: one per SIMD, see k_small)
            grid_main = b->grid - grid_side;
            HIP_TRY(hipMemsetAsync(b->slot_info.p, 0xff, (size_t)side_n * sizeof(ba::SlotInfo), b->stream));
            HIP_TRY(hipEventRecord(b->ev_fork, b->stream));
            BatchParams ps = p1;
            ps.n = side_n; ps.sm_excl_n = side_n; ps.sm_excl_first = 0;
            ps.work_counter = b->counter.as<uint32_t>() + 8;
            ps.big = (short*)((char*)b->big.p + (size_t)grid_main * ba::WAVES_PER_WG * ba::SM_WAVE_BYTES);
            ps.ckpt_wave0 = grid_main * ba::WAVES_PER_WG; ps.cq_side = 1;
            HIP_TRY(hipStreamWaitEvent(b->stream2, b->ev_fork, 0));
            HIP_TRY(launch_sm((b->mode & BA_TRACE) != 0, (b->mode & BA_X_DROP) != 0, grid_side, b->lds, b->stream2, &ps));
            HIP_TRY(hipEventRecord(b->ev_join, b->stream2));
            p1.sm_excl_first = side_n;
        }
        HIP_TRY(launch_sm((b->mode & BA_TRACE) != 0, (b->mode & BA_X_DROP) != 0, grid_main, b->lds, b->stream, &p1));
        if ((b->mode & BA_TRACE) && bp.cig_ops) {
            BatchParams pw = bp; pw.work_counter = b->counter.as<uint32_t>() + 16;   // (its own counters, zeroed with the others: the side launch may still be counting)
            if (special_of(b->mode)) HIP_TRY(ba_launch_walk_loc(b->stream, &pw, walk_grid(b)));   // (records with the zero-mask bits, the early stops)
            else HIP_TRY(ba_launch_walk_l2(b->stream, &pw, walk_grid(b)));
        }
        if (side_n) HIP_TRY(hipStreamWaitEvent(b->stream, b->ev_join, 0));
    } else if (b->pipe) {   // pair-slot batch: the fill stacks, k_walk walks
        BatchParams p1 = bp; p1.cig_ops = nullptr; p1.inline_len2 = ~0u;
        HIP_TRY(launch(1, (b->mode & BA_X_DROP) != 0, b->grid, b->lds, b->stream, &p1));
        if (bp.cig_ops) {
            HIP_TRY(hipMemsetAsync(b->counter.p, 0, 64, b->stream));
            HIP_TRY(ba_launch_walk(b->stream, &bp, walk_grid(b)));
        }
    } else
    HIP_TRY(launch((b->mode & BA_TRACE) != 0, (b->mode & BA_X_DROP) != 0, b->grid, b->lds, b->stream, &bp));
    HIP_TRY(hipEventRecord(b->ev1, b->stream));
    b->in_flight = true;
    return 0;
}
template <class T>
static int d2h(const DevBuf& buf, T* dst, size_t count) {
    if (!dst) return 0;
    HIP_TRY(hipMemcpy(dst, buf.p, count * sizeof(T), hipMemcpyDeviceToHost));
    return 0;
}
// Re-run the pairs `idx` (device order) of a TRACE batch with trace slots of the reference's full bound and put their results
// where the first pass left BA_ST_TRACE_OVERFLOW. The sub-batch reads the parent's resident images and matrix.
static int batch_retry(BaBatch* b, const std::vector<uint32_t>& idx, float* retry_ms) {
    const size_t k = idx.size(), n = b->n;
    std::vector<uint32_t> ql(n), rl(n);
    if (d2h(b->q_len, ql.data(), n) || d2h(b->r_len, rl.data(), n)) return 1;
    BaBatch sub;
    sub.device = b->device; sub.kind = b->kind; sub.mode = b->mode; sub.n = (uint32_t)k; sub.min_size = b->min_size; sub.max_size = b->max_size;
    sub.pclass = b->opt_class ? (uint32_t)BA_PCLASS_BIG : b->pclass; sub.gap_open = b->gap_open; sub.gap_extend = b->gap_extend; sub.x_drop = b->x_drop;
    std::vector<uint64_t> qo(k), ro(k), co(k + 1);
    std::vector<uint32_t> sql(k), srl(k);
    uint64_t maxlen2 = 0, cig_total = 0;
    for (size_t t = 0; t < k; t++) {
        const uint32_t p = idx[t];
        qo[t] = b->h_q_off[p]; ro[t] = b->h_r_off[p]; sql[t] = ql[p]; srl[t] = rl[p];
        maxlen2 = std::max<uint64_t>(maxlen2, (uint64_t)ql[p] + rl[p] + 2);
        co[t] = cig_total; cig_total += (uint64_t)ql[p] + rl[p] + 1;
    }
    co[k] = cig_total;
    if (hipStreamCreateWithFlags(&sub.stream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
    if (hipEventCreate(&sub.ev0) != hipSuccess || hipEventCreate(&sub.ev1) != hipSuccess || hipEventCreate(&sub.ev_l0) != hipSuccess ||
        hipEventCreate(&sub.ev_m1) != hipSuccess) return fail("hipEventCreate failed");
    if (batch_plan(&sub, k, cig_total * 4 + (uint64_t)k * 64, maxlen2, true)) return 1;
    sub.cig_total = cig_total;
    sub.pool.view(b->pool.p, b->pool.bytes); sub.matrix.view(b->matrix.p, b->matrix.bytes);
    DevBuf d_idx;
    if (sub.q_off.alloc(k * 8) || sub.q_len.alloc(k * 4) || sub.r_off.alloc(k * 8) || sub.r_len.alloc(k * 4) || sub.cig_off.alloc((k + 1) * 8) ||
        sub.score.alloc(k * 4) || sub.qidx.alloc(k * 4) || sub.ridx.alloc(k * 4) || sub.cig_len.alloc(k * 4) || sub.cells.alloc(k * 8) ||
        sub.status.alloc(k * 4) || sub.nblocks.alloc(k * 4) || sub.pair_slot.alloc(k * 4) || sub.trace_words.alloc(k * 4) ||
        sub.cig_ops.alloc(cig_total * 4) || d_idx.alloc(k * 4) || batch_alloc_scratch(&sub)) return 1;
    HIP_TRY(hipMemcpy(sub.q_off.p, qo.data(), k * 8, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(sub.q_len.p, sql.data(), k * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(sub.r_off.p, ro.data(), k * 8, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(sub.r_len.p, srl.data(), k * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(sub.cig_off.p, co.data(), (k + 1) * 8, hipMemcpyHostToDevice)); HIP_TRY(hipMemcpy(d_idx.p, idx.data(), k * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(sub.cig_len.p, 0, k * 4)); HIP_TRY(hipMemset(sub.status.p, 0, k * 4));
    if (batch_launch(&sub)) return 1;
    HIP_TRY(hipStreamSynchronize(sub.stream));
    sub.in_flight = false;
    const BatchParams sp = sub.params(), dp = b->params();
    HIP_TRY(hipEventRecord(sub.ev0, sub.stream));
    HIP_TRY(ba_launch_merge_retry(sub.stream, d_idx.as<uint32_t>(), (uint32_t)k, &sp, &dp, sub.trace_words.as<uint32_t>(), b->trace_words.as<uint32_t>()));
    HIP_TRY(hipEventRecord(sub.ev_m1, sub.stream));
    HIP_TRY(hipStreamSynchronize(sub.stream));
    // the re-run belongs to the batch's device time: its kernels (ev0 .. ev1 of the sub-batch's launch) and the merge
    if (retry_ms) {
        float a = 0, c = 0;
        HIP_TRY(hipEventElapsedTime(&a, sub.ev_l0, sub.ev1));
        HIP_TRY(hipEventElapsedTime(&c, sub.ev0, sub.ev_m1));
        *retry_ms = a + c;
    }
    return 0;
}
// The kernels wait for each other without a give-up (a fill wave for a trace slot, a traceback lane for a ring entry: round 4), so a hand-off that
// were ever lost would hang the launch: the host side bounds the wait instead (round-4 advisor finding) -- the launch is polled, and after
// g_wait_limit_ms (ba_set_wait_limit_ms; default 10 minutes, 0 = wait for ever) the call fails with a message instead of never returning.
static std::atomic<uint64_t> g_wait_limit_ms{600000};
void ba_set_wait_limit_ms(uint64_t ms) { g_wait_limit_ms.store(ms); }
// Returns 0 = done, 1 = the stream reported an error, 2 = the limit passed and the launch is STILL RUNNING (round-5 advisor finding: a timeout is not a
// failed launch -- the batch stays "in flight": results and a relaunch are refused until a later wait succeeds; ba_batch_destroy blocks in hipFree
// until the device lets go).
static int stream_wait_bounded(hipStream_t s, const char* what) {
    const uint64_t limit = g_wait_limit_ms.load();
    if (!limit) { HIP_TRY(hipStreamSynchronize(s)); return 0; }
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 0;; spins++) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return 0;
        if (e != hipErrorNotReady) return fail("%s: %s", what, hipGetErrorString(e));
        if (spins < 256) continue;                                   // (short launches: the first polls back to back)
        const auto dt = std::chrono::steady_clock::now() - t0;
        if (std::chrono::duration_cast<std::chrono::milliseconds>(dt).count() > (long long)limit) {
            fail("%s: the launch did not finish within %llu ms (ba_set_wait_limit_ms) and is still running: wait again, or the device may be hung", what, (unsigned long long)limit);
            return 2;
        }
        if (dt > std::chrono::microseconds(200)) std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}
static int batch_wait(BaBatch* b, float* kernel_ms) {
    if (!b->in_flight) return fail("nothing was launched on this batch");
    HIP_TRY(hipSetDevice(b->device));
    if (const int w = stream_wait_bounded(b->stream, "ba_batch_wait")) {
        if (w != 2) b->in_flight = false;   // (a stream error ends the launch; a timeout does not: the batch stays in flight)
        return 1;
    }
    if (kernel_ms) HIP_TRY(hipEventElapsedTime(kernel_ms, b->ev0, b->ev1));
    b->in_flight = false;
    b->retried = 0;
    if (b->quad) {
        uint32_t ctl[64] = {0};
        HIP_TRY(hipMemcpy(ctl, b->cq_ctrl.p, sizeof ctl, hipMemcpyDeviceToHost));
        if (ctl[48]) return fail("a per-pair kernel gave up waiting for the small-block kernel's queue");
        // every queued pair was taken by a per-pair wave: positions handed out (head) cover the entries appended (tail)
        if (ctl[16] < ctl[0]) return fail("small-block pipeline: %u queued pairs were never taken by the per-pair kernel", ctl[0] - ctl[16]);
    }
    if (b->adaptive || b->opt_class) {   // pairs whose trace stack outgrew the expected size: once more, with the reference's full bound
        // (... and, round 6, pairs whose block outgrew the launch's class: once more in the row-tiled class, opt_class)
        std::vector<uint32_t> st(b->n), again;
        if (d2h(b->status, st.data(), b->n)) return 1;
        for (uint32_t p = 0; p < b->n; p++) if (st[p] & (BA_ST_TRACE_OVERFLOW | ba::ST_CLASS_OVERFLOW)) again.push_back(p);
        if (b->opt_class) {
            size_t grew = 0;
            for (uint32_t p : again) if (st[p] & ba::ST_CLASS_OVERFLOW) grew++;
            if (grew * 8 > b->n && !dev_env("BA_KEEP_CLASS_BET")) b->bet_lost = true;
        }
        if (!again.empty()) {
            float retry_ms = 0;
            if (batch_retry(b, again, &retry_ms)) return 1;
            if (kernel_ms) *kernel_ms += retry_ms;   // the merged results include the re-run pairs' cells: so does the time
            b->retried = (uint32_t)again.size();
        }
    }
    b->ran = true;
    return 0;
}
static int batch_run(BaBatch* b, float* kernel_ms) {
    if (batch_launch(b)) return 1;
    return batch_wait(b, kernel_ms);
}


// dst holds one value per pair in device order: rearrange to the caller's order (dst[order[s]] = value of device entry s)
template <class T>
static void to_caller_order(const std::vector<uint32_t>& order, T* dst) {
    if (!dst || order.empty()) return;
    std::vector<T> tmp(dst, dst + order.size());
    for (size_t s = 0; s < order.size(); s++) dst[order[s]] = tmp[s];
}

// ------------------------------------------------------------------ C ABI, Part 2
extern "C" {

const char* ba_last_error(void) { return g_err.c_str(); }
// Page-locked host memory for a caller's result buffers (CIGAR runs above all: 600 MB per 100 k 10 kbp pairs come back at 5 GB/s into
// pageable memory, at PCIe speed into this).
void* ba_host_alloc(uint64_t bytes) {
    if (ensure_device()) return nullptr;
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 4, hipHostMallocDefault) != hipSuccess) { fail("hipHostMalloc(%llu bytes) failed", (unsigned long long)bytes); return nullptr; }
    return p;
}
void ba_host_free(void* p) { if (p) (void)hipHostFree(p); }
int ba_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int ba_set_device(int device) {
    int n = ba_device_count();
    if (device < 0 || device >= n) return fail("device %d out of range (%d devices)", device, n);
    g_device = device;
    HIP_TRY(hipSetDevice(device));
    return 0;
}
int ba_device_memory(uint64_t* free_bytes, uint64_t* total_bytes) {
    if (ensure_device()) return 1;
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return 0;
}
uintptr_t block_percent_len(uintptr_t len, float p) {   // lib.rs:109-111
    size_t v = (size_t)std::round(p * (float)len);
    if (v < 32) v = 32;
    size_t pw = 1;
    while (pw < v) pw <<= 1;
    return pw < ((size_t)1 << 14) ? pw : ((size_t)1 << 14);
}

BaBatch* ba_batch_create(int kind, const void* matrix, Gaps gaps, SizeRange size, int32_t x_drop, uint32_t mode, const uint8_t* pool,
                         const uint64_t* q_off, const uint32_t* q_len, const uint64_t* r_off, const uint32_t* r_len, uintptr_t n) {
    if (!matrix || !pool || !q_off || !q_len || !r_off || !r_len) { fail("null argument"); return nullptr; }
    if (kind == BA_KIND_PROFILE_) { fail("profile batches are created with ba_batch_create_profile"); return nullptr; }
    return batch_build(kind, matrix, gaps, size, x_drop, mode, n, false, [&](size_t p, int w, const uint8_t** ptr, size_t* len) {
        if (w == 0) { *ptr = pool + q_off[p]; *len = q_len[p]; } else { *ptr = pool + r_off[p]; *len = r_len[p]; }
    });
}
BaBatch* ba_batch_create_profile(const AAProfile* const* profiles, SizeRange size, int32_t x_drop, uint32_t mode, const uint8_t* pool,
                                 const uint64_t* q_off, const uint32_t* q_len, uintptr_t n) {
    if (!profiles || !pool || !q_off || !q_len || n == 0 || !profiles[0]) { fail("null argument"); return nullptr; }
    const Gaps g{0, profiles[0]->gap_extend};   // the open costs are per position, inside the profiles
    return batch_build(BA_KIND_PROFILE_, nullptr, g, size, x_drop, mode, n, false,
                       [&](size_t p, int, const uint8_t** ptr, size_t* len) { *ptr = pool + q_off[p]; *len = q_len[p]; },
                       [&](size_t p) { return profiles[p]; });
}
int ba_batch_reload(BaBatch* b, const uint8_t* pool, const uint64_t* q_off, const uint32_t* q_len, const uint64_t* r_off, const uint32_t* r_len, uintptr_t n) {
    if (!b || !pool || !q_off || !q_len || !r_off || !r_len) return fail("null argument");
    if (b->kind == BA_KIND_PROFILE_) return fail("profile batches are reloaded with ba_batch_reload_profile");
    return batch_reload(b, n, false, [&](size_t p, int w, const uint8_t** ptr, size_t* len) {
        if (w == 0) { *ptr = pool + q_off[p]; *len = q_len[p]; } else { *ptr = pool + r_off[p]; *len = r_len[p]; }
    });
}
int ba_batch_reload_profile(BaBatch* b, const AAProfile* const* profiles, const uint8_t* pool, const uint64_t* q_off, const uint32_t* q_len, uintptr_t n) {
    if (!b || !profiles || !pool || !q_off || !q_len) return fail("null argument");
    if (b->kind != BA_KIND_PROFILE_) return fail("not a profile batch");
    return batch_reload(b, n, false, [&](size_t p, int, const uint8_t** ptr, size_t* len) { *ptr = pool + q_off[p]; *len = q_len[p]; },
                        [&](size_t p) { return profiles[p]; });
}
int ba_batch_run(BaBatch* b, float* kernel_ms) { return b ? batch_run(b, kernel_ms) : fail("null batch"); }
int ba_batch_launch(BaBatch* b) { return b ? batch_launch(b) : fail("null batch"); }
int ba_batch_wait(BaBatch* b, float* kernel_ms) { return b ? batch_wait(b, kernel_ms) : fail("null batch"); }
int ba_batch_results(BaBatch* b, int32_t* score, uint32_t* qi, uint32_t* ri, uint64_t* cells, uint32_t* cigar_len, uint32_t* status) {
    if (!b) return fail("null batch");
    if (!b->ran) return fail("ba_batch_run has not been called");
    HIP_TRY(hipSetDevice(b->device));
    if (d2h(b->score, score, b->n) || d2h(b->qidx, qi, b->n) || d2h(b->ridx, ri, b->n) || d2h(b->cells, cells, b->n) ||
        d2h(b->cig_len, cigar_len, b->n) || d2h(b->status, status, b->n)) return 1;
    to_caller_order(b->h_order, score); to_caller_order(b->h_order, qi); to_caller_order(b->h_order, ri);
    to_caller_order(b->h_order, cells); to_caller_order(b->h_order, cigar_len); to_caller_order(b->h_order, status);
    return 0;
}
// Sum of width x height over the rectangles left on each pair's trace stack (Trace::blocks(), scan_block.rs:1676-1691): what
// the reference's "DP fraction" counts (examples/uc_accuracy.rs:88-89). Rectangles popped by a checkpoint restore are not in it.
int ba_batch_surviving_cells(BaBatch* b, uint64_t* cells) {
    if (!b || !cells) return fail("null argument");
    if (!(b->mode & BA_TRACE)) return fail("batch was created without BA_TRACE");
    if (!b->ran) return fail("ba_batch_run has not been called");
    HIP_TRY(hipSetDevice(b->device));
    std::vector<uint32_t> w(b->n);
    if (d2h(b->trace_words, w.data(), b->n)) return 1;
    const uint64_t per_word = (b->mode & BA_LOCAL_START) ? 2 : 1;   // LOCAL_START: a zero-mask word follows every trace word
    for (uint32_t s = 0; s < b->n; s++) cells[b->h_order.empty() ? s : b->h_order[s]] = (uint64_t)w[s] / per_word * 8;
    return 0;
}
// Enqueue, behind the launch in flight (call between ba_batch_launch and ba_batch_wait), the gather of every pair's CIGAR runs into one
// dense device buffer in the caller's pair order. ba_batch_cigars then only copies. Optional: without it ba_batch_cigars gathers on demand.
int ba_batch_compact_cigars(BaBatch* b, uint32_t* pinned_out, uint64_t pinned_capacity) {
    if (!b) return fail("null batch");
    if (!(b->mode & BA_TRACE)) return fail("batch was created without BA_TRACE");
    if (!b->in_flight) return fail("ba_batch_compact_cigars goes between ba_batch_launch and ba_batch_wait");
    if (b->adaptive && b->pipe) return 0;   // (pair-slot batches may re-run pairs after the wait: gather on demand)
    HIP_TRY(hipSetDevice(b->device));
    if (!b->compact_off.p && b->compact_off.alloc((size_t)b->cap_n * 8)) return 1;
    if (!b->h_total) { HIP_TRY(hipHostMalloc((void**)&b->h_total, 64, hipHostMallocDefault)); }
    // pinned_out (memory from ba_host_alloc, which the device can write): the gather writes the runs straight into the caller's host
    // buffer -- over PCIe, inside the stream, right behind the kernels; ba_batch_cigars on the same pointer then has nothing left to copy
    // (a device-to-host copy issued while another batch's persistent launch fills the machine waits for that launch: copies of this size
    // are done by copy kernels). Otherwise: into a device buffer.
    uint32_t* out = pinned_out; uint64_t cap = pinned_capacity;
    if (out) {
        // a kernel writes through this pointer: it must be host memory the device can reach (ba_host_alloc / hipHostMalloc / hipHostRegister)
        // and hold pinned_capacity runs -- an ordinary malloc or numpy buffer here would be a GPU page fault that takes the process down
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, out) != hipSuccess) { (void)hipGetLastError(); return fail("ba_batch_compact_cigars: pinned_out is not page-locked host memory (use ba_host_alloc)"); }
        if (at.type != hipMemoryTypeHost) return fail("ba_batch_compact_cigars: pinned_out must be page-locked HOST memory (use ba_host_alloc)");
        void* base = nullptr; size_t range = 0;
        if (hipMemGetAddressRange((hipDeviceptr_t*)&base, &range, (hipDeviceptr_t)at.devicePointer) == hipSuccess && base) {
            const size_t used = (size_t)((const char*)at.devicePointer - (const char*)base);
            if (range < used || (range - used) / 4 < cap) return fail("ba_batch_compact_cigars: pinned_out holds %zu runs, pinned_capacity says %llu", (range - used) / 4, (unsigned long long)cap);
        } else (void)hipGetLastError();
        out = (uint32_t*)at.devicePointer;   // (the address the device uses for this allocation; the same value for ba_host_alloc memory)
    }
    if (!out) {
        if (!b->compact.p) {
            // capacity: a third of the worst case (every cell of a pair's path its own run) covers any real alignment; if it ever does not,
            // the gather is skipped on the device and ba_batch_cigars falls back
            b->compact_cap = std::max<uint64_t>(b->cap_cig / 3, 1u << 20);
            if (b->compact.alloc(b->compact_cap * 4)) return 1;
        }
        out = b->compact.as<uint32_t>(); cap = b->compact_cap;
    }
    HIP_TRY(ba_launch_cigar_offsets_and_compact(b->stream, b->cig_ops.as<uint32_t>(), b->cig_off.as<uint64_t>(), b->cig_len.as<uint32_t>(),
                                                b->h_order.empty() ? nullptr : b->dev_of.as<uint32_t>(), b->compact_off.as<uint64_t>(), out,
                                                b->h_total, cap, b->n));
    b->compacted = true; b->compact_host = pinned_out; b->compact_used_cap = cap;
    return 0;
}
int ba_batch_cigars(BaBatch* b, uint32_t* runs, uint64_t capacity) {
    if (!b) return fail("null batch");
    if (!(b->mode & BA_TRACE)) return fail("batch was created without BA_TRACE");
    if (!b->ran) return fail("ba_batch_run has not been called");
    HIP_TRY(hipSetDevice(b->device));
    if (b->compacted && b->retried == 0) {   // gathered behind the launch: one copy
        const unsigned long long total = *(volatile unsigned long long*)b->h_total;   // (written by the gather; the stream was synchronised by ba_batch_wait)
        if (total <= b->compact_used_cap) {
            if (total > capacity) return fail("cigar buffer too small: need %llu entries", total);
            if (b->compact_host) { if (runs != b->compact_host && total) memcpy(runs, b->compact_host, total * 4); }   // already in host memory
            else if (total) HIP_TRY(hipMemcpy(runs, b->compact.p, total * 4, hipMemcpyDeviceToHost));
            return 0;
        }
    }
    std::vector<uint32_t> len(b->n);
    if (d2h(b->cig_len, len.data(), b->n)) return 1;
    // runs go out pair after pair in the caller's order; the device arrays are in device order
    std::vector<uint64_t> out_off(b->n);
    uint64_t total = 0;
    if (b->h_order.empty()) for (uint32_t p = 0; p < b->n; p++) { out_off[p] = total; total += len[p]; }
    else {
        std::vector<uint32_t> dev_of(b->n);
        for (uint32_t s = 0; s < b->n; s++) dev_of[b->h_order[s]] = s;
        for (uint32_t p = 0; p < b->n; p++) { out_off[dev_of[p]] = total; total += len[dev_of[p]]; }
    }
    if (total > capacity) return fail("cigar buffer too small: need %llu entries", (unsigned long long)total);
    if (total == 0) return 0;
    DevBuf d_off, d_out;
    if (d_off.alloc(b->n * 8) || d_out.alloc(total * 4)) return 1;
    HIP_TRY(hipMemcpy(d_off.p, out_off.data(), b->n * 8, hipMemcpyHostToDevice));
    HIP_TRY(ba_launch_compact_cigars(b->stream, b->cig_ops.as<uint32_t>(), b->cig_off.as<uint64_t>(), b->cig_len.as<uint32_t>(),
                                     d_off.as<uint64_t>(), d_out.as<uint32_t>(), b->n));
    HIP_TRY(hipStreamSynchronize(b->stream));
    HIP_TRY(hipMemcpy(runs, d_out.p, total * 4, hipMemcpyDeviceToHost));
    return 0;
}
#ifdef BA_DEV
// Development library only: the device-side known-answer test of the lane primitives (k_lane_kat, ba_kernels.hip). x: n_cells int16 values of
// D11_open, whole columns one after the other (form 0: 128 cells per column, four columns per wave; 1: 32 cells, sixteen per wave; 2 / 3 / 4: one
// column of 128 / 64 / 32 cells per wave); out: the columns' R11.
int ba_dev_lane_scan(int form, const int16_t* x, uint32_t n_cells, int gap_extend, int16_t* out) {
    if (form < 0 || form > 4 || !x || !out) return fail("ba_dev_lane_scan: bad arguments");
    const uint32_t per_wave = form <= 1 ? 512u : (form == 2 ? 128u : (form == 3 ? 64u : 32u));
    if (n_cells == 0 || n_cells % per_wave) return fail("ba_dev_lane_scan: %u cells are not whole waves of %u", n_cells, per_wave);
    if (gap_extend > 0 || gap_extend < -128) return fail("ba_dev_lane_scan: gap_extend %d", gap_extend);
    DevBuf dx, dout;
    if (dx.alloc((size_t)n_cells * 2) || dout.alloc((size_t)n_cells * 2)) return 1;
    HIP_TRY(hipMemcpy(dx.p, x, (size_t)n_cells * 2, hipMemcpyHostToDevice));
    HIP_TRY(ba_launch_lane_kat(nullptr, form, dx.as<short>(), dout.as<short>(), gap_extend, n_cells / per_wave));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout.p, (size_t)n_cells * 2, hipMemcpyDeviceToHost));
    return 0;
}
#endif
int ba_batch_prof(BaBatch* b, uint64_t out[128]) {   // development: phase timers of a -DBA_TIMING build (all per-pair launches of a batch add up)
    if (!b) return fail("null batch");
    HIP_TRY(hipMemcpy(out, b->prof.p, 1024, hipMemcpyDeviceToHost));
    return 0;
}
int ba_batch_info(BaBatch* b, uint64_t out[4]) {
    if (!b) return fail("null batch");
    out[0] = (uint64_t)b->grid * b->wpw; out[1] = b->lds / b->wpw; out[2] = b->trace.bytes; out[3] = b->pool_bytes;
    return 0;
}
// Cells of the speculative, untraced rectangles of the last run (X-drop + TRACE batches: the chain of grows that closes an alignment,
// ba_driver.hpp run()): part of the computed cells, filled without trace flags and location bookkeeping (14 instead of 20 int16
// operations per cell). Pairs re-run for a trace overflow are not in it.
int ba_batch_spec_cells(BaBatch* b, uint64_t* cells) {
    if (!b || !cells) return fail("null argument");
    if (!b->ran) return fail("ba_batch_run has not been called");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipMemcpy(cells, (const char*)b->prof.p + 60 * 8, 8, hipMemcpyDeviceToHost));
    return 0;
}
int ba_batch_kernel(BaBatch* b) { return !b ? -1 : (b->small ? 3 : (b->quad ? 2 : (b->multi ? 1 : 0))); }
int ba_batch_geometry(BaBatch* b) { return !b ? -1 : (int)b->geom; }
int ba_batch_retried(BaBatch* b) { return b ? (int)b->retried : -1; }
void ba_batch_destroy(BaBatch* b) { delete b; }

int block_batch_align(int kind, const void* matrix, Gaps gaps, SizeRange size, int32_t x_drop, uint32_t mode, const uint8_t* pool,
                      const uint64_t* q_off, const uint32_t* q_len, const uint64_t* r_off, const uint32_t* r_len, uintptr_t n,
                      AlignResult* results, uint32_t* cigar_runs, uint64_t cigar_capacity, uint32_t* cigar_len) {
    std::unique_ptr<BaBatch> b(ba_batch_create(kind, matrix, gaps, size, x_drop, mode, pool, q_off, q_len, r_off, r_len, n));
    if (!b) return 1;
    if (ba_batch_run(b.get(), nullptr)) return 1;
    std::vector<int32_t> sc(n); std::vector<uint32_t> qi(n), ri(n), st(n);
    if (ba_batch_results(b.get(), sc.data(), qi.data(), ri.data(), nullptr, cigar_len, st.data())) return 1;
    for (size_t p = 0; p < n; p++) {
        if (st[p]) return fail("pair %zu failed on the device (status 0x%x)", p, st[p]);
        if (results) results[p] = AlignResult{sc[p], qi[p], ri[p]};
    }
    if ((mode & BA_TRACE) && cigar_runs) return ba_batch_cigars(b.get(), cigar_runs, cigar_capacity);
    return 0;
}

// ------------------------------------------------------------------ nucleotide / byte objects
NucMatrix* block_new_simple_nucmatrix(int8_t match_score, int8_t mismatch_score) { return new NucMatrix(nuc_simple(match_score, mismatch_score)); }
void block_set_nucmatrix(NucMatrix* m, uint8_t a, uint8_t b, int8_t score) {   // scores.rs:174-183
    a = upper(a); b = upper(b);
    if (!(a >= 'A' && a <= 'Z' && b >= 'A' && b <= 'Z')) die("NucMatrix::set: bytes must be in 'A'..='Z'");
    m->scores[(a & 7) * 16 + (b & 15)] = score;
    m->scores[(b & 7) * 16 + (a & 15)] = score;
}
void block_free_nucmatrix(NucMatrix* m) { delete m; }

static PaddedBytes* padded_new(int kind, size_t len, size_t max_size) {   // scan_block.rs:1798-1803
    PaddedBytes* p = new PaddedBytes;
    p->s.assign(1 + len + max_size, null_byte(kind));
    p->len = len; p->kind = kind;
    return p;
}
static void padded_set(PaddedBytes* p, const uint8_t* s, size_t len, size_t max_size, bool rev) {   // scan_block.rs:1806-1822
    if (1 + len + max_size > p->s.size()) die("PaddedBytes::set_bytes: %zu bytes + %zu padding exceed the allocated %zu", len, max_size, p->s.size());
    p->s[0] = null_byte(p->kind);
    for (size_t k = 0; k < len; k++) {
        uint8_t c;
        if (!convert_char(p->kind, s[rev ? len - 1 - k : k], &c)) die("byte 0x%02x is outside the matrix alphabet", s[rev ? len - 1 - k : k]);
        p->s[1 + k] = c;
    }
    std::fill(p->s.begin() + 1 + len, p->s.begin() + 1 + len + max_size, null_byte(p->kind));
    p->len = len;
}
PaddedBytes* block_new_padded_nuc(uintptr_t len, uintptr_t max_size) { return padded_new(BA_KIND_NUC, len, max_size); }
void block_set_bytes_padded_nuc(PaddedBytes* p, const uint8_t* s, uintptr_t len, uintptr_t max_size) { padded_set(p, s, len, max_size, false); }
void block_set_bytes_rev_padded_nuc(PaddedBytes* p, const uint8_t* s, uintptr_t len, uintptr_t max_size) { padded_set(p, s, len, max_size, true); }
void block_free_padded_nuc(PaddedBytes* p) { delete p; }
PaddedBytes* block_new_padded_bytes(uintptr_t len, uintptr_t max_size) { return padded_new(BA_KIND_BYTES, len, max_size); }
void block_set_bytes_padded_bytes(PaddedBytes* p, const uint8_t* s, uintptr_t len, uintptr_t max_size) { padded_set(p, s, len, max_size, false); }
void block_free_padded_bytes(PaddedBytes* p) { delete p; }

// ------------------------------------------------------------------ Part 1: AAMatrix, PaddedBytes (aa), Cigar
AAMatrix* block_new_simple_aamatrix(int8_t match_score, int8_t mismatch_score) {   // scores.rs:48-61
    AAMatrix* m = new AAMatrix;
    memset(m->scores, 0x80, sizeof m->scores);
    for (int i = 0; i < 26; i++)
        for (int j = 0; j < 26; j++) m->scores[i * 32 + j] = i == j ? match_score : mismatch_score;
    return m;
}
void block_set_aamatrix(AAMatrix* m, uint8_t a, uint8_t b, int8_t score) {   // scores.rs:89-98
    a = upper(a); b = upper(b);
    if (!(a >= 'A' && a <= 'Z' + 1 && b >= 'A' && b <= 'Z' + 1)) die("AAMatrix::set: bytes must be in 'A'..='['");
    m->scores[(a - 'A') * 32 + (b - 'A')] = score;
    m->scores[(b - 'A') * 32 + (a - 'A')] = score;
}
void block_free_aamatrix(AAMatrix* m) { delete m; }

PaddedBytes* block_new_padded_aa(uintptr_t len, uintptr_t max_size) { return padded_new(BA_KIND_AA, len, max_size); }
void block_set_bytes_padded_aa(PaddedBytes* p, const uint8_t* s, uintptr_t len, uintptr_t max_size) { padded_set(p, s, len, max_size, false); }
void block_set_bytes_rev_padded_aa(PaddedBytes* p, const uint8_t* s, uintptr_t len, uintptr_t max_size) { padded_set(p, s, len, max_size, true); }
void block_free_padded_aa(PaddedBytes* p) { delete p; }

Cigar* block_new_cigar(uintptr_t query_len, uintptr_t reference_len) {   // cigar.rs:49-54
    Cigar* c = new Cigar;
    c->capacity = query_len + reference_len + 5;
    return c;
}
OpLen block_get_cigar(const Cigar* c, uintptr_t i) {   // cigar.rs:92-94 (i-th run from the start of the alignment)
    if (i >= c->ops.size()) die("Cigar::get: index %zu out of range (%zu runs)", (size_t)i, c->ops.size());
    return c->ops[i];
}
uintptr_t block_len_cigar(const Cigar* c) { return c->ops.size(); }
void block_free_cigar(Cigar* c) { delete c; }

// ------------------------------------------------------------------ Part 1: AAProfile (host object; scores.rs:470-715)
AAProfile* block_new_aaprofile(uintptr_t str_len, uintptr_t block_size, int8_t gap_extend) {
    AAProfile* p = new AAProfile;
    p->max_len = p->curr_len = str_len + block_size + 1;
    p->str_len = str_len; p->gap_extend = gap_extend;
    p->pos_aa.assign(p->max_len * 32, (int8_t)-128);
    p->gap_open_C.assign(p->max_len, -128); p->gap_close_C.assign(p->max_len, -128); p->gap_open_R.assign(p->max_len, -128);
    return p;
}
uintptr_t block_len_aaprofile(const AAProfile* p) { return p->str_len; }
void block_clear_aaprofile(AAProfile* p, uintptr_t str_len, uintptr_t block_size) {
    const size_t cl = str_len + block_size + 1;
    if (cl > p->max_len) die("AAProfile::clear: length exceeds the allocation");
    std::fill(p->pos_aa.begin(), p->pos_aa.begin() + cl * 32, (int8_t)-128);
    std::fill(p->gap_open_C.begin(), p->gap_open_C.begin() + cl, (int16_t)-128);
    std::fill(p->gap_close_C.begin(), p->gap_close_C.begin() + cl, (int16_t)-128);
    std::fill(p->gap_open_R.begin(), p->gap_open_R.begin() + cl, (int16_t)-128);
    p->str_len = str_len; p->curr_len = cl;
}
void block_set_aaprofile(AAProfile* p, uintptr_t i, uint8_t b, int8_t score) {
    b = upper(b);
    if (!(b >= 'A' && b <= 'Z' + 1)) die("AAProfile::set: byte must be in 'A'..='['");
    if (i >= p->curr_len) die("AAProfile::set: position out of range");
    p->pos_aa[i * 32 + (b - 'A')] = score;
}
static void profile_set_all(AAProfile* p, const uint8_t* order, size_t order_len, const int8_t* scores, size_t scores_len,
                            size_t left_shift, size_t right_shift, bool rev) {   // scores.rs:677-714
    if (order_len == 0 || order_len > 32) die("AAProfile::set_all: order length must be in 1..=32");
    uint8_t o[32];
    for (size_t k = 0; k < order_len; k++) {
        const uint8_t b = upper(order[k]);
        if (!(b >= 'A' && b <= 'Z' + 1)) die("AAProfile::set_all: order byte out of range");
        o[k] = (uint8_t)(b - 'A');
    }
    if (scores_len / order_len != p->str_len) die("AAProfile::set_all: scores length does not match the profile length");
    size_t idx = 0;
    for (size_t n = 0; n < p->str_len; n++) {
        const size_t i = rev ? p->str_len - n : 1 + n;
        for (size_t j = 0; j < order_len; j++) {
            const int8_t sc = (int8_t)((int8_t)((uint8_t)scores[idx] << left_shift) >> right_shift);
            p->pos_aa[i * 32 + o[j]] = sc;
            idx++;
        }
    }
}
void block_set_all_aaprofile(AAProfile* p, const uint8_t* order, uintptr_t order_len, const int8_t* scores, uintptr_t scores_len,
                             uintptr_t left_shift, uintptr_t right_shift) { profile_set_all(p, order, order_len, scores, scores_len, left_shift, right_shift, false); }
void block_set_all_rev_aaprofile(AAProfile* p, const uint8_t* order, uintptr_t order_len, const int8_t* scores, uintptr_t scores_len,
                                 uintptr_t left_shift, uintptr_t right_shift) { profile_set_all(p, order, order_len, scores, scores_len, left_shift, right_shift, true); }
void block_set_gap_open_C_aaprofile(AAProfile* p, uintptr_t i, int8_t gap) { if (gap >= 0) die("Gap open cost must be negative!"); p->gap_open_C.at(i) = gap; }
void block_set_gap_close_C_aaprofile(AAProfile* p, uintptr_t i, int8_t gap) { p->gap_close_C.at(i) = gap; }
void block_set_gap_open_R_aaprofile(AAProfile* p, uintptr_t i, int8_t gap) { if (gap >= 0) die("Gap open cost must be negative!"); p->gap_open_R.at(i) = gap; }
void block_set_all_gap_open_C_aaprofile(AAProfile* p, int8_t gap) { if (gap >= 0) die("Gap open cost must be negative!"); std::fill(p->gap_open_C.begin(), p->gap_open_C.begin() + p->str_len + 1, (int16_t)gap); }
void block_set_all_gap_close_C_aaprofile(AAProfile* p, int8_t gap) { std::fill(p->gap_close_C.begin(), p->gap_close_C.begin() + p->str_len + 1, (int16_t)gap); }
void block_set_all_gap_open_R_aaprofile(AAProfile* p, int8_t gap) { if (gap >= 0) die("Gap open cost must be negative!"); std::fill(p->gap_open_R.begin(), p->gap_open_R.begin() + p->str_len + 1, (int16_t)gap); }
int8_t block_get_aaprofile(const AAProfile* p, uintptr_t i, uint8_t b) {
    b = upper(b);
    if (!(b >= 'A' && b <= 'Z' + 1)) die("AAProfile::get: byte must be in 'A'..='['");
    return p->pos_aa.at(i * 32 + (b - 'A'));
}
int8_t block_get_gap_extend_aaprofile(const AAProfile* p) { return p->gap_extend; }
// Part 2: bulk form of set / set_gap_* for bindings that already hold the profile as arrays. Copies the first
// `positions` rows of pos_aa ([position][32]) and entries of the three gap arrays; no range checks beyond the length.
int ba_aaprofile_set_raw(AAProfile* p, const int8_t* pos_aa, const int8_t* gap_open_C, const int8_t* gap_close_C,
                         const int8_t* gap_open_R, uintptr_t positions) {
    if (!p || !pos_aa || !gap_open_C || !gap_close_C || !gap_open_R) return fail("null argument");
    if (positions > p->curr_len) return fail("%zu positions exceed the profile's %zu", (size_t)positions, p->curr_len);
    memcpy(p->pos_aa.data(), pos_aa, (size_t)positions * 32);
    for (size_t i = 0; i < positions; i++) { p->gap_open_C[i] = gap_open_C[i]; p->gap_close_C[i] = gap_close_C[i]; p->gap_open_R[i] = gap_open_R[i]; }
    return 0;
}
void block_free_aaprofile(AAProfile* p) { delete p; }

}  // extern "C"


// Block::align_exp / align_profile_exp over a batch (scan_block.rs:884-902, 974-992): a second (third, ...) kernel
// pass over the subset of pairs that stayed below the target score, with the min block size doubled each time.
template <class MakeBatch>
static int batch_exp(size_t n, SizeRange size, int32_t target, MakeBatch make, AlignResult* results, uintptr_t* reached_min) {
    if (!results || !reached_min) return fail("null argument");
    size_t min_size = size.min < 16 ? 16 : size.min;
    const size_t max_size = size.max < 16 ? 16 : size.max;
    std::vector<size_t> todo(n);
    for (size_t p = 0; p < n; p++) { todo[p] = p; reached_min[p] = 0; }
    while (min_size <= max_size && !todo.empty()) {
        std::unique_ptr<BaBatch> b(make(todo, SizeRange{min_size, max_size}));
        if (!b) return 1;
        if (batch_run(b.get(), nullptr)) return 1;
        const size_t m = todo.size();
        std::vector<int32_t> sc(m); std::vector<uint32_t> qi(m), ri(m), st(m);
        if (ba_batch_results(b.get(), sc.data(), qi.data(), ri.data(), nullptr, nullptr, st.data())) return 1;
        std::vector<size_t> next;
        for (size_t k = 0; k < m; k++) {
            const size_t p = todo[k];
            if (st[k]) return fail("pair %zu failed on the device (status 0x%x)", p, st[k]);
            results[p] = AlignResult{sc[k], qi[k], ri[k]};
            if (sc[k] >= target) reached_min[p] = min_size; else next.push_back(p);
        }
        todo.swap(next);
        min_size *= 2;
    }
    return 0;
}

extern "C" {
int block_batch_align_exp(int kind, const void* matrix, Gaps gaps, SizeRange size, int32_t x_drop, int32_t target_score, uint32_t mode,
                          const uint8_t* pool, const uint64_t* q_off, const uint32_t* q_len, const uint64_t* r_off, const uint32_t* r_len,
                          uintptr_t n, AlignResult* results, uintptr_t* reached_min) {
    if (!matrix || !pool || !q_off || !q_len || !r_off || !r_len) return fail("null argument");
    if (kind == BA_KIND_PROFILE_) return fail("use block_batch_align_profile_exp for profiles");
    mode &= ~(uint32_t)(BA_TRACE | BA_CIGAR_EQ);   // scores only: trace the finished pairs with a plain batch at reached_min
    return batch_exp(n, size, target_score, [&](const std::vector<size_t>& idx, SizeRange s) {
        return batch_build(kind, matrix, gaps, s, x_drop, mode, idx.size(), false, [&](size_t k, int w, const uint8_t** ptr, size_t* len) {
            const size_t p = idx[k];
            if (w == 0) { *ptr = pool + q_off[p]; *len = q_len[p]; } else { *ptr = pool + r_off[p]; *len = r_len[p]; }
        });
    }, results, reached_min);
}
int block_batch_align_profile_exp(const AAProfile* const* profiles, SizeRange size, int32_t x_drop, int32_t target_score, uint32_t mode,
                                  const uint8_t* pool, const uint64_t* q_off, const uint32_t* q_len, uintptr_t n,
                                  AlignResult* results, uintptr_t* reached_min) {
    if (!profiles || !pool || !q_off || !q_len || n == 0 || !profiles[0]) return fail("null argument");
    mode &= ~(uint32_t)(BA_TRACE | BA_CIGAR_EQ);
    const Gaps g{0, profiles[0]->gap_extend};
    return batch_exp(n, size, target_score, [&](const std::vector<size_t>& idx, SizeRange s) {
        return batch_build(BA_KIND_PROFILE_, nullptr, g, s, x_drop, mode, idx.size(), false,
                           [&](size_t k, int, const uint8_t** ptr, size_t* len) { *ptr = pool + q_off[idx[k]]; *len = q_len[idx[k]]; },
                           [&](size_t k) { return profiles[idx[k]]; });
    }, results, reached_min);
}
}  // extern "C"

// ------------------------------------------------------------------ one batch over several GPUs
// Pairs are independent, so a batch shards without any exchange step: contiguous slices of the caller's pair
// list, balanced by cost (|q| + |r|, what the number of driver steps follows), one BaBatch per device, each created by its
// own host thread (packing and upload run in parallel), launched on its own stream; results come back in the caller's order.
extern "C" int ba_shard_slices(const uint32_t* q_len, const uint32_t* r_len, uintptr_t n, int parts, uint64_t* bounds) {
    if (!q_len || !r_len || !bounds || parts < 1) return fail("null argument");
    std::vector<uint64_t> pre(n + 1, 0);
    for (size_t p = 0; p < n; p++) pre[p + 1] = pre[p] + (uint64_t)q_len[p] + r_len[p] + 16;   // (+16: even empty pairs cost a launch slot)
    bounds[0] = 0;
    for (int k = 1; k < parts; k++) {
        const uint64_t target = pre[n] / (uint64_t)parts * (uint64_t)k + pre[n] % (uint64_t)parts * (uint64_t)k / (uint64_t)parts;
        size_t lo = std::lower_bound(pre.begin(), pre.end(), target) - pre.begin();
        if (lo > n) lo = n;
        if (lo < bounds[k - 1]) lo = bounds[k - 1];
        bounds[k] = lo;
    }
    bounds[parts] = n;
    return 0;
}

struct BaMultiBatch {
    std::vector<std::unique_ptr<BaBatch>> part;
    std::vector<uint64_t> bounds;     // part k holds the caller's pairs [bounds[k], bounds[k + 1])
    std::vector<float> last_ms;       // kernel time of every part in the last ba_multibatch_run (HIP events on the part's own stream)
    uint32_t mode = 0;
};

extern "C" {
BaMultiBatch* ba_multibatch_create(int kind, const void* matrix, Gaps gaps, SizeRange size, int32_t x_drop, uint32_t mode, const uint8_t* pool,
                                   const uint64_t* q_off, const uint32_t* q_len, const uint64_t* r_off, const uint32_t* r_len, uintptr_t n,
                                   const int* devices, int n_devices) {
    if (!matrix || !pool || !q_off || !q_len || !r_off || !r_len || !devices || n_devices < 1) { fail("null argument"); return nullptr; }
    if (n == 0) { fail("batch must hold at least one pair"); return nullptr; }
    const int have = ba_device_count();
    for (int k = 0; k < n_devices; k++)
        if (devices[k] < 0 || devices[k] >= have) { fail("device %d out of range (%d devices)", devices[k], have); return nullptr; }
    std::unique_ptr<BaMultiBatch> m(new BaMultiBatch);
    m->mode = mode;
    m->bounds.resize(n_devices + 1);
    if (ba_shard_slices(q_len, r_len, n, n_devices, m->bounds.data())) return nullptr;
    m->part.resize(n_devices);
    std::vector<std::string> errs(n_devices);
    auto build = [&](int k) {
        const size_t lo = m->bounds[k], cnt = m->bounds[k + 1] - lo;
        if (cnt == 0) return;                                  // more devices than pairs: this one stays idle
        g_device = devices[k];                                 // (thread-local)
        BaBatch* b = ba_batch_create(kind, matrix, gaps, size, x_drop, mode, pool, q_off + lo, q_len + lo, r_off + lo, r_len + lo, cnt);
        if (!b) errs[k] = g_err; else m->part[k].reset(b);
    };
    std::vector<std::thread> th;
    for (int k = 1; k < n_devices; k++) th.emplace_back(build, k);
    const int keep = g_device;
    build(0);
    g_device = keep;
    for (auto& t : th) t.join();
    for (int k = 0; k < n_devices; k++)
        if (!errs[k].empty()) { fail("device %d: %s", devices[k], errs[k].c_str()); return nullptr; }
    return m.release();
}
int ba_multibatch_run(BaMultiBatch* m, float* kernel_ms) {
    if (!m) return fail("null batch");
    // all devices first, then collect. A failure does not leave the other parts in flight: every part that was launched is
    // waited for before the first error is reported.
    std::string first_err;
    for (auto& b : m->part) {
        if (!b || !first_err.empty()) continue;
        if (batch_launch(b.get())) first_err = g_err;
    }
    float worst = 0;
    m->last_ms.assign(m->part.size(), 0.f);
    for (size_t k = 0; k < m->part.size(); k++) {
        auto& b = m->part[k];
        if (!b || !b->in_flight) continue;
        float ms = 0;
        if (batch_wait(b.get(), &ms)) { if (first_err.empty()) first_err = g_err; continue; }   // (batch_wait itself says whether the part is still in flight: a timeout leaves it so)
        m->last_ms[k] = ms;
        worst = std::max(worst, ms);
    }
    if (!first_err.empty()) return fail("%s", first_err.c_str());
    if (kernel_ms) *kernel_ms = worst;
    return 0;
}
int ba_multibatch_results(BaMultiBatch* m, int32_t* score, uint32_t* qi, uint32_t* ri, uint64_t* cells, uint32_t* cigar_len, uint32_t* status) {
    if (!m) return fail("null batch");
    for (size_t k = 0; k < m->part.size(); k++) {
        if (!m->part[k]) continue;
        const size_t lo = m->bounds[k];
        if (ba_batch_results(m->part[k].get(), score ? score + lo : nullptr, qi ? qi + lo : nullptr, ri ? ri + lo : nullptr, cells ? cells + lo : nullptr,
                             cigar_len ? cigar_len + lo : nullptr, status ? status + lo : nullptr)) return 1;
    }
    return 0;
}
int ba_multibatch_cigars(BaMultiBatch* m, uint32_t* runs, uint64_t capacity) {
    if (!m) return fail("null batch");
    uint64_t at = 0;
    for (auto& b : m->part) {
        if (!b) continue;
        std::vector<uint32_t> len(b->n);
        HIP_TRY(hipSetDevice(b->device));
        if (d2h(b->cig_len, len.data(), b->n)) return 1;
        uint64_t total = 0;
        for (uint32_t x : len) total += x;
        if (at + total > capacity) return fail("cigar buffer too small: need more than %llu entries", (unsigned long long)capacity);
        if (total && ba_batch_cigars(b.get(), runs + at, capacity - at)) return 1;
        at += total;
    }
    return 0;
}
// ---- every pair with its own block range (lib.rs:109-111 percent_len per pair, as examples/nanopore_bench_global.rs:163,171 calls it):
// pairs are binned by (min, max), every bin is an ordinary batch of this library -- the multi-pair kernels where the bin's minimum size has one --
// launched one after the other on the device; results and CIGAR runs come back in the caller's order.
struct BaSizedBatch {
    std::vector<std::unique_ptr<BaBatch>> part;
    std::vector<SizeRange> range;                 // part k's block range
    std::vector<std::vector<uint32_t>> idx;       // part k's pairs (caller indices, ascending)
    std::vector<float> last_ms;
    std::vector<double> work;                     // part k's share of the work (residues x maximum block size): launch order, memory share
    size_t n = 0;
    uint32_t mode = 0;
};
BaSizedBatch* ba_sized_batch_create(int kind, const void* matrix, Gaps gaps, const SizeRange* size_per_pair, int32_t x_drop, uint32_t mode, const uint8_t* pool,
                                    const uint64_t* q_off, const uint32_t* q_len, const uint64_t* r_off, const uint32_t* r_len, uintptr_t n) {
    if (!matrix || !size_per_pair || !pool || !q_off || !q_len || !r_off || !r_len) { fail("null argument"); return nullptr; }
    if (n == 0) { fail("batch must hold at least one pair"); return nullptr; }
    if (n >= (1ull << 32)) { fail("too many pairs"); return nullptr; }
    std::unique_ptr<BaSizedBatch> m(new BaSizedBatch);
    m->n = n; m->mode = mode;
    std::map<std::pair<uintptr_t, uintptr_t>, size_t> cls;
    for (size_t p = 0; p < n; p++) {
        const auto key = std::make_pair(size_per_pair[p].min, size_per_pair[p].max);
        auto it = cls.find(key);
        if (it == cls.end()) { it = cls.emplace(key, m->idx.size()).first; m->idx.emplace_back(); m->range.push_back(size_per_pair[p]); }
        m->idx[it->second].push_back((uint32_t)p);
    }
    m->part.resize(m->idx.size());
    // every range's batch gets a share of the free device memory in proportion to what its trace stacks ask for (residues x maximum block size)
    std::vector<double> weight(m->idx.size(), 0.0);
    double weight_left = 0;
    for (size_t k = 0; k < m->idx.size(); k++) {
        for (uint32_t p : m->idx[k]) weight[k] += ((double)q_len[p] + r_len[p] + 64.0) * (double)m->range[k].max;
        weight_left += weight[k];
    }
    m->work = weight;
    struct CapReset { ~CapReset() { g_mem_cap = ~0ull; } } cap_reset;
    for (size_t k = 0; k < m->idx.size(); k++) {
        {
            size_t free_b = 0, total_b = 0;
            (void)hipMemGetInfo(&free_b, &total_b);
            g_mem_cap = m->idx.size() > 1 ? (uint64_t)((double)free_b * 0.95 * (weight[k] / std::max(weight_left, 1.0))) : ~0ull;
            g_mem_cap = std::max<uint64_t>(g_mem_cap, 3ull << 30);   // (batch_plan keeps 1 GB aside; a range of a few pairs still needs its scratch)
            weight_left -= weight[k];
        }
        const auto& ix = m->idx[k];
        std::vector<uint64_t> qo(ix.size()), ro(ix.size());
        std::vector<uint32_t> ql(ix.size()), rl(ix.size());
        for (size_t i = 0; i < ix.size(); i++) { qo[i] = q_off[ix[i]]; ro[i] = r_off[ix[i]]; ql[i] = q_len[ix[i]]; rl[i] = r_len[ix[i]]; }
        BaBatch* b = ba_batch_create(kind, matrix, gaps, m->range[k], x_drop, mode, pool, qo.data(), ql.data(), ro.data(), rl.data(), ix.size());
        if (!b) { const std::string e = g_err; fail("block range %llu..%llu (%llu pairs): %s", (unsigned long long)m->range[k].min, (unsigned long long)m->range[k].max, (unsigned long long)ix.size(), e.c_str()); return nullptr; }
        m->part[k].reset(b);
    }
    return m.release();
}
BaSizedBatch* ba_sized_batch_create_percent(int kind, const void* matrix, Gaps gaps, float min_percent, float max_percent, int32_t x_drop, uint32_t mode, const uint8_t* pool,
                                            const uint64_t* q_off, const uint32_t* q_len, const uint64_t* r_off, const uint32_t* r_len, uintptr_t n) {
    if (!q_len || !r_len) { fail("null argument"); return nullptr; }
    std::vector<SizeRange> sz(n);
    for (size_t p = 0; p < n; p++) {   // examples/nanopore_bench_global.rs:144-171: both percentages of the longer sequence's length
        const uintptr_t len = std::max(q_len[p], r_len[p]);
        sz[p].min = block_percent_len(len, min_percent); sz[p].max = block_percent_len(len, max_percent);
    }
    return ba_sized_batch_create(kind, matrix, gaps, sz.data(), x_drop, mode, pool, q_off, q_len, r_off, r_len, n);
}
int ba_sized_batch_run(BaSizedBatch* m, float* kernel_ms) {
    if (!m) return fail("null batch");
    // Round 5: all ranges' launches at once, each on its batch's own stream, the ones with the most work first: a range of a few thousand pairs
    // does not fill the device (its grid is what its pairs need), and the long ranges' launches end with a few long pairs. kernel_ms: the host's
    // clock from the first launch to the last completion (the device is idle before: the previous run was waited for). BA_SIZED_SERIAL (development):
    // one after the other, kernel_ms the sum of the launches' event times.
    m->last_ms.assign(m->part.size(), 0.f);
    if (dev_env("BA_SIZED_SERIAL")) {
        float total = 0;
        for (size_t k = 0; k < m->part.size(); k++) {
            float ms = 0;
            if (ba_batch_run(m->part[k].get(), &ms)) return 1;
            m->last_ms[k] = ms; total += ms;
        }
        if (kernel_ms) *kernel_ms = total;
        return 0;
    }
    std::vector<size_t> order(m->part.size());
    for (size_t k = 0; k < order.size(); k++) order[k] = k;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return m->work[a] > m->work[b]; });
    const auto t0 = std::chrono::steady_clock::now();
    int rc = 0;
    std::string first_err;
    // Round 6: a launch whose waves wait for each other -- the hand-off ring of a traced batch (its traceback waves and helpers wait for the fill waves'
    // hand-offs), k_multi's end-of-batch slot donation (its idle waves wait until every fill wave has been counted) -- needs all of its workgroups resident
    // to be sure to end. Alone it is (the grid is what the device holds); beside launches that simply run out it still is, later; but two or more such
    // launches that each hold a part of the device wait for workgroups the other's waiting waves keep out (three ranges of k_multi in one per-pair-ranges
    // batch: 9 .. 25 s instead of 0.2, every wait ended by a time-out). So: the ranges without such waits all at once, the ones with them one after the
    // other beside those.
    auto waits_inside = [&](const BaBatch* b) { return b->tb_stride != 0 || (b->multi && (b->mode & BA_TRACE) && b->donate.p != nullptr); };
    std::vector<size_t> free_run, chained;
    for (size_t k : order) (waits_inside(m->part[k].get()) ? chained : free_run).push_back(k);
    std::vector<char> launched(m->part.size(), 0);
    auto wait_one = [&](size_t k) { float ms = 0; if (ba_batch_wait(m->part[k].get(), &ms) && !rc) { rc = 1; first_err = g_err; } m->last_ms[k] = ms; launched[k] = 0; };
    auto launch_one = [&](size_t k) { if (rc) return; if (ba_batch_launch(m->part[k].get())) { rc = 1; first_err = g_err; } else launched[k] = 1; };
    if (!chained.empty()) launch_one(chained[0]);   // (the longest chain first: the other ranges fill the device around it)
    for (size_t k : free_run) launch_one(k);
    for (size_t i = 0; i < chained.size(); i++) {
        if (launched[chained[i]]) wait_one(chained[i]);
        if (i + 1 < chained.size()) launch_one(chained[i + 1]);
    }
    for (size_t k : free_run) if (launched[k]) wait_one(k);   // (also after a failed launch: nothing is left in flight behind the caller's back)
    for (size_t k = 0; k < launched.size(); k++) if (launched[k]) wait_one(k);
    if (rc) return fail("%s", first_err.c_str());
    if (kernel_ms) *kernel_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}
int ba_sized_batch_results(BaSizedBatch* m, int32_t* score, uint32_t* qi, uint32_t* ri, uint64_t* cells, uint32_t* cigar_len, uint32_t* status) {
    if (!m) return fail("null batch");
    for (size_t k = 0; k < m->part.size(); k++) {
        const auto& ix = m->idx[k];
        const size_t c = ix.size();
        std::vector<int32_t> s(score ? c : 0); std::vector<uint32_t> a(qi ? c : 0), b(ri ? c : 0), l(cigar_len ? c : 0), st(status ? c : 0); std::vector<uint64_t> ce(cells ? c : 0);
        if (ba_batch_results(m->part[k].get(), score ? s.data() : nullptr, qi ? a.data() : nullptr, ri ? b.data() : nullptr, cells ? ce.data() : nullptr,
                             cigar_len ? l.data() : nullptr, status ? st.data() : nullptr)) return 1;
        for (size_t i = 0; i < c; i++) {
            const uint32_t p = ix[i];
            if (score) score[p] = s[i]; if (qi) qi[p] = a[i]; if (ri) ri[p] = b[i]; if (cells) cells[p] = ce[i]; if (cigar_len) cigar_len[p] = l[i]; if (status) status[p] = st[i];
        }
    }
    return 0;
}
int ba_sized_batch_cigars(BaSizedBatch* m, uint32_t* runs, uint64_t capacity) {   // concatenated in the caller's pair order, as ba_batch_cigars
    if (!m) return fail("null batch");
    std::vector<uint32_t> len(m->n, 0);
    if (ba_sized_batch_results(m, nullptr, nullptr, nullptr, nullptr, len.data(), nullptr)) return 1;
    std::vector<uint64_t> off(m->n + 1, 0);
    for (size_t p = 0; p < m->n; p++) off[p + 1] = off[p] + len[p];
    if (off[m->n] > capacity) return fail("cigar buffer too small: need %llu entries", (unsigned long long)off[m->n]);
    for (size_t k = 0; k < m->part.size(); k++) {
        const auto& ix = m->idx[k];
        uint64_t total = 0;
        for (uint32_t p : ix) total += len[p];
        if (!total) continue;
        std::vector<uint32_t> tmp(total);
        if (ba_batch_cigars(m->part[k].get(), tmp.data(), total)) return 1;
        uint64_t at = 0;
        for (uint32_t p : ix) { if (len[p]) std::memcpy(runs + off[p], tmp.data() + at, (size_t)len[p] * 4); at += len[p]; }
    }
    return 0;
}
int ba_sized_batch_classes(BaSizedBatch* m, SizeRange* ranges, uint64_t* counts, int32_t* kernels, float* kernel_ms, int capacity) {   // the bins; returns their number
    if (!m) return -1;
    for (int k = 0; k < capacity && k < (int)m->part.size(); k++) {
        if (ranges) ranges[k] = m->range[k];
        if (counts) counts[k] = m->idx[k].size();
        if (kernels) kernels[k] = ba_batch_kernel(m->part[k].get());
        if (kernel_ms) kernel_ms[k] = k < (int)m->last_ms.size() ? m->last_ms[k] : 0.f;
    }
    return (int)m->part.size();
}
void ba_sized_batch_destroy(BaSizedBatch* m) { delete m; }

int ba_multibatch_kernel_ms(BaMultiBatch* m, float* ms, int capacity) {   // per slice, of the last run; returns the number of slices
    if (!m) return -1;
    for (int k = 0; k < capacity && k < (int)m->last_ms.size(); k++) ms[k] = m->last_ms[k];
    return (int)m->part.size();
}
int ba_multibatch_parts(BaMultiBatch* m, uint64_t* bounds, int capacity) {   // slice boundaries (n_devices + 1 entries); returns n_devices
    if (!m) return -1;
    for (int k = 0; k < capacity && k < (int)m->bounds.size(); k++) bounds[k] = m->bounds[k];
    return (int)m->part.size();
}
void ba_multibatch_destroy(BaMultiBatch* m) { delete m; }
}  // extern "C"

// ------------------------------------------------------------------ Block handles (Part 1 + generic)
// Like the reference's Block (scan_block.rs:798-805, 1280-1340: Block::new allocates, align never does), a handle owns its
// device state from creation: stream, events, the two image slots, one trace slot, the rectangle list, the CIGAR buffer
// and the checkpoint scratch, sized for any pair with |q| + |r| <= query_len + reference_len and any block size up to
// max_size. An align uploads one buffer (images + the per-pair words), launches one workgroup and reads one 64-byte
// record back; nothing is allocated or freed.
struct HandleUpload {   // head of BaBatch::hblk; the images follow at `images`
    uint64_t q_off, r_off, cig_off[2];
    uint32_t q_len, r_len, work_counter, pad_;
    static constexpr size_t images = 64;
};
struct HandleResult {   // BaBatch::rblk
    int32_t score; uint32_t query_idx, reference_idx, cig_len, status, nblocks, slot, trace_words;
    unsigned long long cells; uint32_t pad_[6];
};
static_assert(sizeof(HandleUpload) <= HandleUpload::images && sizeof(HandleResult) == 64, "handle records");

struct BlockImpl {
    uint32_t mode;                 // BA_TRACE | BA_X_DROP | ...
    size_t query_len, reference_len, max_size;   // upper bounds from Block::new (scan_block.rs:798-805)
    AlignResult res{0, 0, 0};
    std::unique_ptr<BaBatch> dev;                // persistent device state (created with the handle when a device is usable)
    std::vector<uint8_t> staging;                // host image of hblk
    int8_t matrix_image[1024] = {0};             // host image of the matrix last uploaded
    bool matrix_valid = false;
    uint32_t last_ql = 0, last_rl = 0, last_nblocks = 0, last_slot = 0;
    bool aligned = false;
};

static uint64_t handle_trace_stride(size_t max_size, uint64_t maxlen2, uint32_t mode) {   // Trace::new's bound, scan_block.rs:1363-1366 (+ zero mask, + slack)
    return (uint64_t)(max_size / 16) * (maxlen2 + 2 * max_size) * 2 * ((mode & BA_LOCAL_START) ? 2 : 1) + 64;
}

// Allocate a handle's device state. Returns 0, or non-zero with the message in g_err (the caller decides whether that is fatal).
static int handle_prepare(BlockImpl* h) {
    if (ensure_device()) return 1;
    std::unique_ptr<BaBatch> b(new BaBatch);
    b->device = g_device; b->handle_mode = true; b->n = 1;
    const size_t max_size = h->max_size < 16 ? 16 : h->max_size, sum = h->query_len + h->reference_len;
    if (sum > 0x3fffffffu) return fail("sequences too long");
    const bool trace = h->mode & BA_TRACE;
    const size_t pad = max_size + 16;
    // either sequence may be as long as the sum; the reference side may be an AAProfile image instead
    const uint64_t seq_img = (1 + sum + pad + 3 + 8) & ~(size_t)7;
    const uint64_t ref_img = std::max<uint64_t>(seq_img, (ba::profile_image_bytes((uint32_t)sum, (uint32_t)max_size) + 7) & ~7ull);
    b->cap_pool = HandleUpload::images + seq_img + ref_img + POOL_SLACK;
    b->cap_maxlen2 = sum + 2; b->cap_n = 1; b->cap_cig = sum + 1;
    b->trace_stride = trace ? handle_trace_stride(max_size, b->cap_maxlen2, h->mode) : 0;
    b->blocks_stride = trace ? b->cap_maxlen2 : 0;
    if (b->trace_stride >= (1ull << 30)) return fail("trace stack of %llu words exceeds the 2^30 limit", (unsigned long long)b->trace_stride);
    if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
    if (hipEventCreate(&b->ev0) != hipSuccess || hipEventCreate(&b->ev1) != hipSuccess) return fail("hipEventCreate failed");
    if (b->hblk.alloc(b->cap_pool) || b->rblk.alloc(sizeof(HandleResult)) || b->matrix.alloc(1024) || b->cig_ops.alloc((trace ? b->cap_cig : 1) * 4) ||
        b->trace.alloc(b->trace_stride * 4) || b->blocks.alloc(b->blocks_stride * sizeof(BlockRec)) ||
        b->ckpt.alloc((size_t)ba::WAVES_PER_WG * 8 * max_size * sizeof(short)) ||
        b->big.alloc(max_size > 2048 ? (size_t)ba::WAVES_PER_WG * ba::big_wave_shorts((uint32_t)max_size) * sizeof(short) : 0) || b->prof.alloc(2048) || b->tb_ctrl.alloc(256) ||
        b->tb_queue.alloc(4) || b->slot_free.alloc(4) || b->slot_info.alloc(sizeof(ba::SlotInfo))) return 1;
    uint8_t* hb = b->hblk.as<uint8_t>(); uint8_t* rb = b->rblk.as<uint8_t>();
    b->pool.view(hb, b->cap_pool);   // the images' offsets are relative to the start of hblk
    b->q_off.view(hb + offsetof(HandleUpload, q_off), 8); b->r_off.view(hb + offsetof(HandleUpload, r_off), 8);
    b->cig_off.view(hb + offsetof(HandleUpload, cig_off), 16);
    b->q_len.view(hb + offsetof(HandleUpload, q_len), 4); b->r_len.view(hb + offsetof(HandleUpload, r_len), 4);
    b->counter.view(hb + offsetof(HandleUpload, work_counter), 4);
    b->score.view(rb + offsetof(HandleResult, score), 4); b->qidx.view(rb + offsetof(HandleResult, query_idx), 4);
    b->ridx.view(rb + offsetof(HandleResult, reference_idx), 4); b->cig_len.view(rb + offsetof(HandleResult, cig_len), 4);
    b->status.view(rb + offsetof(HandleResult, status), 4); b->nblocks.view(rb + offsetof(HandleResult, nblocks), 4);
    b->pair_slot.view(rb + offsetof(HandleResult, slot), 4); b->trace_words.view(rb + offsetof(HandleResult, trace_words), 4);
    b->cells.view(rb + offsetof(HandleResult, cells), 8);
    b->grid = 1; b->slots = 1; b->slots_per_wave = 0; b->tb_stride = 0;   // (slots_per_wave 0: whichever wave takes the pair uses slot 0)
    b->n_fill_waves = ba::WAVES_PER_WG; b->tb_qsize = 1;
    h->staging.assign(b->cap_pool, 0);
    h->dev = std::move(b);
    return 0;
}

// One alignment through a handle's persistent state. `profile` non-null: sequence-to-profile (kind PROFILE).
// q_s / r_s: a padded image's bytes as the reference keeps them (scan_block.rs:1790-1812: index 0 is the NULL pad, then the converted
// sequence); r_s unused with a profile.
static void handle_align(BlockImpl* h, int kind, const uint8_t* q_s, size_t q_len, const uint8_t* r_s, size_t r_len, const AAProfile* profile, const void* matrix,
                         Gaps g, size_t min_size, size_t max_size, int32_t x) {
    if (!h->dev && handle_prepare(h)) die("%s", g_err.c_str());
    BaBatch* b = h->dev.get();
    if (hipSetDevice(b->device) != hipSuccess) die("hipSetDevice failed");
    const int pc = pclass_of(max_size);
    if (pc < 0) die("max block size %zu not supported by the HIP backend (16..%zu)", max_size, (size_t)BA_MAX_BLOCK);
    const uint32_t mode = h->mode & ~(uint32_t)BA_CIGAR_EQ;
    const bool trace = mode & BA_TRACE;
    if ((mode & BA_FREE_QUERY_END_GAPS) && !(min_size > q_len)) die("Min block size must be larger than the query length for FREE_QUERY_END_GAPS!");   // scan_block.rs:860-862
    b->kind = kind; b->mode = mode; b->min_size = (uint32_t)min_size; b->max_size = (uint32_t)max_size; b->pclass = (uint32_t)pc;
    b->gap_open = g.open; b->gap_extend = g.extend; b->x_drop = x;
    b->lds = ba::lds_wg_bytes_h(kind, lds_class_cells(pc)) + (trace ? ba::TB_LDS_BYTES : 0u);
    // ---- the upload: per-pair words + the two padded images (scan_block.rs:1798-1812), built in the handle's staging buffer
    const size_t pad = max_size + 16;
    const uint32_t ql = (uint32_t)q_len, rl = (uint32_t)(profile ? profile->str_len : r_len);
    HandleUpload hu{};
    hu.q_off = HandleUpload::images; hu.q_len = ql; hu.r_len = rl; hu.work_counter = 0;
    const size_t q_img = (1 + (size_t)ql + pad + 3 + 8) & ~(size_t)7;
    hu.r_off = hu.q_off + q_img;
    const size_t r_img = profile ? (size_t)ba::profile_image_bytes(rl, (uint32_t)max_size) : ((1 + (size_t)rl + pad + 3) & ~(size_t)3);
    const size_t used = (size_t)hu.r_off + r_img + POOL_SLACK;
    if (used > b->cap_pool) die("sequence lengths exceed the bounds this Block was created with");
    hu.cig_off[0] = 0; hu.cig_off[1] = (uint64_t)ql + rl + 1;
    uint8_t* st = h->staging.data();
    memcpy(st, &hu, sizeof hu);
    const uint8_t nb = null_byte(kind);
    uint8_t* qi = st + hu.q_off;
    qi[0] = nb; memcpy(qi + 1, q_s + 1, ql); memset(qi + 1 + ql, nb, q_img - 1 - ql);
    uint8_t* ri = st + hu.r_off;
    if (profile) profile_image(profile, ba::profile_positions(rl, (uint32_t)max_size), ri);
    else { ri[0] = nb; memcpy(ri + 1, r_s + 1, rl); memset(ri + 1 + rl, nb, r_img - 1 - rl); }
    memset(st + hu.r_off + r_img, nb, POOL_SLACK);
    b->pool_bytes = used; b->cig_total = trace ? hu.cig_off[1] : 0;
    b->h_q_off.assign(1, hu.q_off); b->h_r_off.assign(1, hu.r_off);
    hipError_t e = hipMemcpyAsync(b->hblk.p, st, used, hipMemcpyHostToDevice, b->stream);
    if (e == hipSuccess && b->matrix.p) {
        const size_t mat_bytes = kind == BA_KIND_AA ? 27 * 32 : (kind == BA_KIND_NUC ? 8 * 16 : (kind == BA_KIND_BYTES ? 2 : 0));
        // (the image lives in the handle: the source of an asynchronous copy has to outlive the stream synchronisation;
        // unchanged matrices are not uploaded again)
        int8_t tmp[1024] = {0};
        if (kind == BA_KIND_BYTES) { const ByteMatrix* bm = (const ByteMatrix*)matrix; tmp[0] = bm->match_score; tmp[1] = bm->mismatch_score; }
        else if (mat_bytes) memcpy(tmp, matrix, mat_bytes);
        if (mat_bytes && (!h->matrix_valid || memcmp(tmp, h->matrix_image, 1024) != 0)) {
            memcpy(h->matrix_image, tmp, 1024);
            e = hipMemcpyAsync(b->matrix.p, h->matrix_image, 1024, hipMemcpyHostToDevice, b->stream);
            h->matrix_valid = e == hipSuccess;
        }
    }
    if (e != hipSuccess) die("hipMemcpy H2D failed: %s", hipGetErrorString(e));
    b->in_flight = false; b->ran = false;
    if (batch_run(b, nullptr)) die("%s", g_err.c_str());
    HandleResult hr;
    if (hipMemcpy(&hr, b->rblk.p, sizeof hr, hipMemcpyDeviceToHost) != hipSuccess) die("hipMemcpy D2H failed");
    if (hr.status) die("device alignment failed (status 0x%x)", hr.status);
    h->res = AlignResult{hr.score, hr.query_idx, hr.reference_idx};
    h->last_ql = ql; h->last_rl = rl; h->last_nblocks = hr.nblocks; h->last_slot = hr.slot; h->aligned = true;
}

static BlockImpl* block_new_impl(uint32_t mode, size_t query_len, size_t reference_len, size_t max_size) {
    if (max_size == 0 || (max_size & (max_size - 1))) die("Block size must be a power of two!");
    BlockImpl* b = new BlockImpl;
    b->mode = mode; b->query_len = query_len; b->reference_len = reference_len; b->max_size = max_size;
    // Block::new allocates (scan_block.rs:798-805). Without a usable device the handle still exists -- precondition checks
    // work anywhere -- and the first align reports the missing device.
    if (pclass_of(max_size < 16 ? 16 : max_size) >= 0) (void)handle_prepare(b);
    return b;
}

static void block_align_impl(BlockImpl* b, int kind, const PaddedBytes* q, const PaddedBytes* r, const void* matrix, Gaps g, SizeRange s, int32_t x) {
    if (q->kind != kind || r->kind != kind) die("PaddedBytes were built for a different matrix kind");
    const size_t min_size = s.min < 16 ? 16 : s.min, max_size = s.max < 16 ? 16 : s.max;
    std::string why;
    if (check_align_params(false, g, min_size, max_size, x, b->mode, &why)) die("%s", why.c_str());
    if (min_size > max_size) die("min block size exceeds max block size");
    // Allocated::clear (scan_block.rs:1324-1326)
    if (q->len + r->len > b->query_len + b->reference_len) die("sequence lengths exceed the bounds this Block was created with");
    if (max_size > b->max_size) die("max block size exceeds the bound this Block was created with");
    handle_align(b, kind, q->s.data(), q->len, r->s.data(), r->len, nullptr, matrix, g, min_size, max_size, x);
}

static void block_align_profile_impl(BlockImpl* b, const PaddedBytes* q, const AAProfile* pr, SizeRange s, int32_t x) {   // scan_block.rs:942-968
    if (q->kind != BA_KIND_AA) die("PaddedBytes were built for a different matrix kind");
    const size_t min_size = s.min < 16 ? 16 : s.min, max_size = s.max < 16 ? 16 : s.max;
    const Gaps g{0, pr->gap_extend};
    std::string why;
    if (check_align_params(true, g, min_size, max_size, x, b->mode, &why)) die("%s", why.c_str());
    if (min_size > max_size) die("min block size exceeds max block size");
    if (q->len + pr->str_len > b->query_len + b->reference_len) die("sequence lengths exceed the bounds this Block was created with");
    if (max_size > b->max_size) die("max block size exceeds the bound this Block was created with");
    handle_align(b, BA_KIND_PROFILE_, q->s.data(), q->len, nullptr, 0, pr, nullptr, g, min_size, max_size, x);
}

static void block_cigar_impl(BlockImpl* b, bool eq, const PaddedBytes* q, const PaddedBytes* r, size_t i, size_t j, Cigar* cigar) {
    if (!(b->mode & BA_TRACE)) die("trace() requires a Block created with TRACE");   // scan_block.rs:1241-1243
    BaBatch* d = b->dev.get();
    if (!d || !b->aligned) die("cigar requested before any alignment");
    if (eq && d->kind == BA_KIND_PROFILE_) die("cigar_eq needs two sequences; the last alignment was against a profile");
    if (!(i <= b->last_ql && j <= b->last_rl)) die("Traceback cigar end position must be in bounds!");   // scan_block.rs:1483
    if (i + j + 5 > cigar->capacity) die("Cigar was created for shorter sequences than this traceback needs");   // cigar.rs:58-60 slice bound
    (void)q; (void)r;   // the padded images of the aligned pair are already resident on the device
    BatchParams bp = d->params();
    bp.cig_ops = d->cig_ops.as<uint32_t>();
    bp.flags = eq ? (bp.flags | ba::F_CIGAR_EQ) : (bp.flags & ~ba::F_CIGAR_EQ);
    bp.tb_i = (uint32_t)i; bp.tb_j = (uint32_t)j; bp.tb_nblocks = b->last_nblocks; bp.tb_slot = b->last_slot;
    if (hipSetDevice(d->device) != hipSuccess || ba_launch_traceback(d->stream, &bp) != hipSuccess ||
        hipStreamSynchronize(d->stream) != hipSuccess) die("traceback kernel failed: %s", hipGetErrorString(hipGetLastError()));
    HandleResult hr;
    if (hipMemcpy(&hr, d->rblk.p, sizeof hr, hipMemcpyDeviceToHost) != hipSuccess) die("hipMemcpy D2H failed");
    if (hr.status) die("device traceback failed (status 0x%x)", hr.status);
    const uint32_t n = hr.cig_len;
    std::vector<uint32_t> runs(n);
    if (n && hipMemcpy(runs.data(), d->cig_ops.as<uint32_t>() + (d->cig_total - n), (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess)
        die("hipMemcpy of cigar runs failed");
    cigar->ops.resize(n);
    for (uint32_t k = 0; k < n; k++) cigar->ops[k] = OpLen{(Operation)(runs[k] & 15), (uintptr_t)(runs[k] >> 4)};
}

// Trace::blocks() (scan_block.rs:1676-1691): the rectangles on the trace stack of the last alignment, in fill order
static size_t block_trace_blocks_impl(BlockImpl* b, Rectangle* out, size_t capacity) {
    if (!(b->mode & BA_TRACE)) die("trace() requires a Block created with TRACE");
    BaBatch* d = b->dev.get();
    if (!d || !b->aligned) die("blocks requested before any alignment");
    const uint32_t nb = b->last_nblocks, slot = b->last_slot;
    if (!out) return nb;
    std::vector<BlockRec> recs(nb);
    if (nb && hipMemcpy(recs.data(), d->blocks.as<BlockRec>() + (size_t)slot * d->blocks_stride, (size_t)nb * sizeof(BlockRec), hipMemcpyDeviceToHost) != hipSuccess)
        die("hipMemcpy of the rectangle list failed");
    for (size_t k = 0; k < nb && k < capacity; k++) out[k] = Rectangle{recs[k].i & 0x7fffffffu, recs[k].j, recs[k].w, recs[k].h};   // (bit 31 of i: trace word layout)
    return nb;
}

extern "C" {

uintptr_t block_trace_blocks_generic(BlockHandle b, Rectangle* out, uintptr_t capacity) { return block_trace_blocks_impl((BlockImpl*)b, out, capacity); }
BlockHandle block_new_generic(uint32_t mode, uintptr_t ql, uintptr_t rl, uintptr_t max_size) { return block_new_impl(mode, ql, rl, max_size); }
void block_align_generic(BlockHandle b, int kind, const PaddedBytes* q, const PaddedBytes* r, const void* matrix, Gaps g, SizeRange s, int32_t x) {
    block_align_impl((BlockImpl*)b, kind, q, r, matrix, g, s, x);
}
void block_align_profile_generic(BlockHandle b, const PaddedBytes* q, const AAProfile* p, SizeRange s, int32_t x) {
    block_align_profile_impl((BlockImpl*)b, q, p, s, x);
}
// The same two calls for a caller that keeps its own PaddedBytes (the Rust crate behind the `simd_hip` feature: rust/src/): the
// padded image as the reference stores it -- [NULL] + converted bytes + NULL padding (scan_block.rs:1790-1812) -- by pointer.
void block_align_padded_generic(BlockHandle bh, int kind, const uint8_t* q_s, uintptr_t q_len, const uint8_t* r_s, uintptr_t r_len, const void* matrix,
                                Gaps g, SizeRange s, int32_t x) {
    BlockImpl* b = (BlockImpl*)bh;
    if (kind < 0 || kind > 2) die("unknown matrix kind %d", kind);
    const size_t min_size = s.min < 16 ? 16 : s.min, max_size = s.max < 16 ? 16 : s.max;
    std::string why;
    if (check_align_params(false, g, min_size, max_size, x, b->mode, &why)) die("%s", why.c_str());
    if (min_size > max_size) die("min block size exceeds max block size");
    if (q_len + r_len > b->query_len + b->reference_len) die("sequence lengths exceed the bounds this Block was created with");
    if (max_size > b->max_size) die("max block size exceeds the bound this Block was created with");
    handle_align(b, kind, q_s, q_len, r_s, r_len, nullptr, matrix, g, min_size, max_size, x);
}
void block_align_profile_padded_generic(BlockHandle bh, const uint8_t* q_s, uintptr_t q_len, const AAProfile* pr, SizeRange s, int32_t x) {
    BlockImpl* b = (BlockImpl*)bh;
    const size_t min_size = s.min < 16 ? 16 : s.min, max_size = s.max < 16 ? 16 : s.max;
    const Gaps g{0, pr->gap_extend};
    std::string why;
    if (check_align_params(true, g, min_size, max_size, x, b->mode, &why)) die("%s", why.c_str());
    if (min_size > max_size) die("min block size exceeds max block size");
    if (q_len + pr->str_len > b->query_len + b->reference_len) die("sequence lengths exceed the bounds this Block was created with");
    if (max_size > b->max_size) die("max block size exceeds the bound this Block was created with");
    handle_align(b, BA_KIND_PROFILE_, q_s, q_len, nullptr, 0, pr, nullptr, g, min_size, max_size, x);
}
AlignResult block_res_generic(BlockHandle b) { return ((BlockImpl*)b)->res; }
void block_cigar_generic(BlockHandle b, uintptr_t i, uintptr_t j, Cigar* c) { block_cigar_impl((BlockImpl*)b, false, nullptr, nullptr, i, j, c); }
void block_cigar_eq_generic(BlockHandle b, const PaddedBytes* q, const PaddedBytes* r, uintptr_t i, uintptr_t j, Cigar* c) {
    block_cigar_impl((BlockImpl*)b, true, q, r, i, j, c);
}
void block_free_generic(BlockHandle b) { delete (BlockImpl*)b; }

#define BA_DEFINE_BLOCK_FNS(S, CIG, CIGEQ, MODE)                                                                                    \
    BlockHandle block_new_##S(uintptr_t ql, uintptr_t rl, uintptr_t max_size) { return block_new_impl(MODE, ql, rl, max_size); }   \
    void block_align_##S(BlockHandle b, const PaddedBytes* q, const PaddedBytes* r, const AAMatrix* m, Gaps g, SizeRange s, int32_t x) { \
        block_align_impl((BlockImpl*)b, BA_KIND_AA, q, r, m, g, s, x);                                                            \
    }                                                                                                                             \
    void block_align_profile_##S(BlockHandle b, const PaddedBytes* q, const AAProfile* p, SizeRange s, int32_t x) {                \
        block_align_profile_impl((BlockImpl*)b, q, p, s, x);                                                                      \
    }                                                                                                                             \
    AlignResult block_res_##S(BlockHandle b) { return ((BlockImpl*)b)->res; }                                                      \
    void CIG(BlockHandle b, uintptr_t i, uintptr_t j, Cigar* c) { block_cigar_impl((BlockImpl*)b, false, nullptr, nullptr, i, j, c); } \
    void CIGEQ(BlockHandle b, const PaddedBytes* q, const PaddedBytes* r, uintptr_t i, uintptr_t j, Cigar* c) {                    \
        block_cigar_impl((BlockImpl*)b, true, q, r, i, j, c);                                                                     \
    }                                                                                                                             \
    void block_free_##S(BlockHandle b) { delete (BlockImpl*)b; }

BA_DEFINE_BLOCK_FNS(aa, _block_cigar_aa, _block_cigar_eq_aa, 0)
BA_DEFINE_BLOCK_FNS(aa_xdrop, _block_cigar_aa_xdrop, _block_cigar_eq_aa_xdrop, BA_X_DROP)
BA_DEFINE_BLOCK_FNS(aa_trace, block_cigar_aa_trace, block_cigar_eq_aa_trace, BA_TRACE)
BA_DEFINE_BLOCK_FNS(aa_trace_xdrop, block_cigar_aa_trace_xdrop, block_cigar_eq_aa_trace_xdrop, BA_TRACE | BA_X_DROP)

}  // extern "C"
