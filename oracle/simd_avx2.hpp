// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// 16-lane saturating-i16 vector layer, restated with the same x86 AVX2 intrinsics the
// reference backend uses, so per-128-bit-half behaviours (byte shifts that shift in zero,
// alignr, shufflehi, permute4x64) are reproduced by the hardware itself rather than modelled.
//
// Follows /root/reference/src/avx2.rs (file:line cited per function).
// Parity status: see DESIGN.md section 2 ("pinned by the reference's own known-answer tests").
#pragma once
#include <immintrin.h>
#include <cstdint>
#include <cstring>

namespace ba_oracle {

// avx2.rs:9-16
constexpr int L = 16;
constexpr int16_t ZERO = 1 << 14;
constexpr int16_t MIN = 0;

struct V16 {
    __m256i v;
};
struct H16 {  // 16 bytes of sequence / one LUT row half
    __m128i v;
};

// avx2.rs:24-46
static inline V16 v_adds(V16 a, V16 b) { return {_mm256_adds_epi16(a.v, b.v)}; }
static inline V16 v_subs(V16 a, V16 b) { return {_mm256_subs_epi16(a.v, b.v)}; }
static inline V16 v_max(V16 a, V16 b) { return {_mm256_max_epi16(a.v, b.v)}; }
static inline V16 v_cmpeq(V16 a, V16 b) { return {_mm256_cmpeq_epi16(a.v, b.v)}; }
static inline V16 v_blend8(V16 a, V16 b, V16 mask) { return {_mm256_blendv_epi8(a.v, b.v, mask.v)}; }
// avx2.rs:48-62
static inline V16 v_load(const int16_t* p) { return {_mm256_load_si256((const __m256i*)p)}; }
static inline V16 v_loadu(const int16_t* p) { return {_mm256_loadu_si256((const __m256i*)p)}; }
static inline void v_store(int16_t* p, V16 a) { _mm256_store_si256((__m256i*)p, a.v); }
static inline V16 v_set1(int16_t x) { return {_mm256_set1_epi16(x)}; }

// avx2.rs:173-182 (runtime-index extract through memory)
static inline int16_t v_extract(V16 a, int i) {
    alignas(32) int16_t t[L];
    v_store(t, a);
    return t[i];
}
// avx2.rs:81-92 with num = 0 (the only use, scan_block.rs:1131)
static inline V16 v_insert0(V16 a, int16_t x) { return {_mm256_insert_epi16(a.v, x, 0)}; }
// avx2.rs:96
static inline uint32_t v_movemask8(V16 a) { return (uint32_t)_mm256_movemask_epi8(a.v); }

// avx2.rs:100-115 with num = 1: out[0] = b[15], out[k] = a[k-1]
static inline V16 v_sl1(V16 a, V16 b) {
    __m256i t = _mm256_permute2x128_si256(a.v, b.v, 0x03);
    return {_mm256_alignr_epi8(a.v, t, 14)};
}
// avx2.rs:139-141: [b[8..16], a[0..8]]
static inline V16 v_step(V16 a, V16 b) { return {_mm256_permute2x128_si256(a.v, b.v, 0x03)}; }
// avx2.rs:166-169
static inline V16 v_broadcasthi(V16 a) {
    __m256i t = _mm256_shufflehi_epi16(a.v, 0xFF);
    return {_mm256_permute4x64_epi64(t, 0xFF)};
}
// avx2.rs:186-192
static inline int16_t v_hmax(V16 a) {
    __m256i v2 = _mm256_max_epi16(a.v, _mm256_srli_si256(a.v, 2));
    v2 = _mm256_max_epi16(v2, _mm256_srli_si256(v2, 4));
    v2 = _mm256_max_epi16(v2, _mm256_srli_si256(v2, 8));
    v2 = _mm256_max_epi16(v2, _mm256_permute2x128_si256(v2, v2, 0x03));
    return (int16_t)_mm256_extract_epi16(v2, 0);
}
// avx2.rs:221-242 with num = STEP = 8
static inline int16_t v_prefix_hmax8(V16 a) {
    __m256i v = a.v;
    v = _mm256_max_epi16(v, _mm256_srli_si256(v, 8));
    v = _mm256_max_epi16(v, _mm256_srli_si256(v, 4));
    v = _mm256_max_epi16(v, _mm256_srli_si256(v, 2));
    return (int16_t)_mm256_extract_epi16(v, 0);
}
// avx2.rs:246-267 with num = SHRINK_SUFFIX_LEN = 2
static inline int16_t v_suffix_hmax2(V16 a) {
    __m256i v = a.v;
    v = _mm256_max_epi16(v, _mm256_slli_si256(v, 2));
    return (int16_t)_mm256_extract_epi16(v, 15);
}
// avx2.rs:271-274
static inline int v_hargmax(V16 a, int16_t mx) {
    __m256i e = _mm256_cmpeq_epi16(a.v, _mm256_set1_epi16(mx));
    uint32_t m = (uint32_t)_mm256_movemask_epi8(e);
    return (m == 0 ? 32 : __builtin_ctz(m)) / 2;
}

struct ScanConsts {
    V16 gap_all;   // (k+1)*g
    V16 lane;      // per-128-bit-half ramp g..8g
};
// avx2.rs:297-310
static inline ScanConsts v_scan_consts(V16 gap) {
    __m256i s1 = _mm256_slli_si256(gap.v, 2);
    s1 = _mm256_adds_epi16(s1, gap.v);
    __m256i s2 = _mm256_slli_si256(s1, 4);
    s2 = _mm256_adds_epi16(s2, s1);
    __m256i s4 = _mm256_slli_si256(s2, 8);
    s4 = _mm256_adds_epi16(s4, s2);
    __m256i c = _mm256_srli_si256(_mm256_shufflehi_epi16(s4, 0xFF), 8);
    c = _mm256_permute4x64_epi64(c, 0x05);
    c = _mm256_adds_epi16(c, s4);
    return {{c}, {s4}};
}
// avx2.rs:315-338
static inline V16 v_prefix_scan(V16 r, V16 gap, V16 lane_consts) {
    __m256i s1 = _mm256_slli_si256(r.v, 2);
    s1 = _mm256_adds_epi16(s1, gap.v);
    s1 = _mm256_max_epi16(r.v, s1);
    __m256i s2 = _mm256_slli_si256(s1, 4);
    s2 = _mm256_adds_epi16(s2, _mm256_slli_epi16(gap.v, 1));
    s2 = _mm256_max_epi16(s1, s2);
    __m256i s4 = _mm256_slli_si256(s2, 8);
    s4 = _mm256_adds_epi16(s4, _mm256_slli_epi16(gap.v, 2));
    s4 = _mm256_max_epi16(s2, s4);
    __m256i c = _mm256_shufflehi_epi16(s4, 0xFF);
    c = _mm256_permute4x64_epi64(c, 0x50);
    c = _mm256_adds_epi16(c, lane_consts.v);
    return {_mm256_max_epi16(s4, c)};
}

// avx2.rs:372
static inline H16 h_loadu(const uint8_t* p) { return {_mm_loadu_si128((const __m128i*)p)}; }
static inline H16 h_set1(int8_t x) { return {_mm_set1_epi8(x)}; }
// avx2.rs:343-350 — two 16-byte LUT halves; bit 4 of each index byte picks the half
static inline V16 h_lookup2(const int8_t* row32, H16 idx) {
    __m128i l1 = _mm_loadu_si128((const __m128i*)row32);
    __m128i l2 = _mm_loadu_si128((const __m128i*)(row32 + 16));
    __m128i a = _mm_shuffle_epi8(l1, idx.v);
    __m128i b = _mm_shuffle_epi8(l2, idx.v);
    __m128i m = _mm_slli_epi16(idx.v, 3);
    return {_mm256_cvtepi8_epi16(_mm_blendv_epi8(a, b, m))};
}
// avx2.rs:354-356
static inline V16 h_lookup1(const int8_t* row16, H16 idx) {
    __m128i l = _mm_loadu_si128((const __m128i*)row16);
    return {_mm256_cvtepi8_epi16(_mm_shuffle_epi8(l, idx.v))};
}
// avx2.rs:360-364
static inline V16 h_lookup_bytes(int8_t match, int8_t mismatch, uint8_t c, H16 v) {
    __m128i m = _mm_cmpeq_epi8(_mm_set1_epi8((char)c), v.v);
    __m128i r = _mm_blendv_epi8(_mm_set1_epi8(mismatch), _mm_set1_epi8(match), m);
    return {_mm256_cvtepi8_epi16(r)};
}

static inline const char* simd_backend_name() { return "avx2-intrinsics"; }

}  // namespace ba_oracle
