// ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the shipped product path.
//
// Scalar lane model of the reference's 16-lane AVX2 layer: same interface as simd_avx2.hpp,
// but every op is written as its per-lane meaning (closed forms), with no intrinsics. This is the
// *specification* the HIP kernel implements; tests/test_oracle_lane_model.py proves it equal to the
// intrinsic version on random and adversarial (negative / saturating) inputs.
//
// Follows /root/reference/src/avx2.rs (semantics derived in SURVEY.md Appendix A.3-A.4).
#pragma once
#include <cstdint>
#include <cstring>
#include <algorithm>

namespace ba_oracle {

constexpr int L = 16;
constexpr int16_t ZERO = 1 << 14;
constexpr int16_t MIN = 0;

struct V16 {
    int16_t v[L];
};
struct H16 {
    uint8_t v[L];
};

static inline int16_t sat16(int32_t x) { return (int16_t)std::min(32767, std::max(-32768, x)); }

#define BA_LANEWISE(expr)              \
    V16 o;                             \
    for (int k = 0; k < L; k++) o.v[k] = (expr); \
    return o;

static inline V16 v_adds(V16 a, V16 b) { BA_LANEWISE(sat16((int32_t)a.v[k] + b.v[k])) }
static inline V16 v_subs(V16 a, V16 b) { BA_LANEWISE(sat16((int32_t)a.v[k] - b.v[k])) }
static inline V16 v_max(V16 a, V16 b) { BA_LANEWISE(std::max(a.v[k], b.v[k])) }
static inline V16 v_cmpeq(V16 a, V16 b) { BA_LANEWISE((int16_t)(a.v[k] == b.v[k] ? -1 : 0)) }
// byte-granular blend on the mask's byte MSBs
static inline V16 v_blend8(V16 a, V16 b, V16 mask) {
    V16 o;
    for (int k = 0; k < L; k++) {
        uint16_t ua = (uint16_t)a.v[k], ub = (uint16_t)b.v[k], um = (uint16_t)mask.v[k];
        uint16_t lo = (um & 0x0080) ? (ub & 0x00FF) : (ua & 0x00FF);
        uint16_t hi = (um & 0x8000) ? (ub & 0xFF00) : (ua & 0xFF00);
        o.v[k] = (int16_t)(lo | hi);
    }
    return o;
}
static inline V16 v_load(const int16_t* p) { V16 o; std::memcpy(o.v, p, sizeof o.v); return o; }
static inline V16 v_loadu(const int16_t* p) { return v_load(p); }
static inline void v_store(int16_t* p, V16 a) { std::memcpy(p, a.v, sizeof a.v); }
static inline V16 v_set1(int16_t x) { BA_LANEWISE(x) }
static inline int16_t v_extract(V16 a, int i) { return a.v[i]; }
static inline V16 v_insert0(V16 a, int16_t x) { a.v[0] = x; return a; }
static inline uint32_t v_movemask8(V16 a) {
    uint32_t m = 0;
    for (int k = 0; k < L; k++) {
        uint16_t u = (uint16_t)a.v[k];
        if (u & 0x0080) m |= 1u << (2 * k);
        if (u & 0x8000) m |= 1u << (2 * k + 1);
    }
    return m;
}
static inline V16 v_sl1(V16 a, V16 b) { BA_LANEWISE(k == 0 ? b.v[L - 1] : a.v[k - 1]) }
static inline V16 v_step(V16 a, V16 b) { BA_LANEWISE(k < 8 ? b.v[k + 8] : a.v[k - 8]) }
static inline V16 v_broadcasthi(V16 a) { BA_LANEWISE(a.v[L - 1]) }
static inline int16_t v_hmax(V16 a) { return *std::max_element(a.v, a.v + L); }
static inline int16_t v_prefix_hmax8(V16 a) { return *std::max_element(a.v, a.v + 8); }
static inline int16_t v_suffix_hmax2(V16 a) { return std::max(a.v[L - 2], a.v[L - 1]); }
static inline int v_hargmax(V16 a, int16_t mx) {
    for (int k = 0; k < L; k++) if (a.v[k] == mx) return k;
    return L;
}

struct ScanConsts {
    V16 gap_all;  // (k+1)*g
    V16 lane;     // unused by the scalar model
};
static inline ScanConsts v_scan_consts(V16 gap) {
    ScanConsts c;
    for (int k = 0; k < L; k++) { c.gap_all.v[k] = sat16((int32_t)(k + 1) * gap.v[0]); c.lane.v[k] = 0; }
    return c;
}
// Closed form of avx2.rs:315-338: true in-vector max-plus scan T, joined with the artefacts V of the
// zero shift-in: V = [g,2g,3g,4g,5g,6g,7g,12g, g,2g,3g,4g,5g,6g,7g, none].
static inline V16 v_prefix_scan(V16 r, V16 gap, V16 /*lane_consts*/) {
    const int32_t g = gap.v[0];
    static const int mult[L] = {1, 2, 3, 4, 5, 6, 7, 12, 1, 2, 3, 4, 5, 6, 7, 0};
    V16 o;
    int32_t run = -32768;
    for (int k = 0; k < L; k++) {
        run = std::max((int32_t)r.v[k], (int32_t)sat16(run + g));
        int32_t x = run;
        if (mult[k]) x = std::max(x, (int32_t)sat16(mult[k] * g));
        o.v[k] = (int16_t)x;
    }
    return o;
}

static inline H16 h_loadu(const uint8_t* p) { H16 o; std::memcpy(o.v, p, L); return o; }
static inline V16 h_lookup2(const int8_t* row32, H16 idx) {
    // pshufb zeroes lanes whose index has bit 7 set; blend picks the upper half on bit 4
    BA_LANEWISE((idx.v[k] & 0x80) ? 0 : (int16_t)row32[(idx.v[k] & 15) + ((idx.v[k] & 16) ? 16 : 0)])
}
static inline V16 h_lookup1(const int8_t* row16, H16 idx) {
    BA_LANEWISE((idx.v[k] & 0x80) ? 0 : (int16_t)row16[idx.v[k] & 15])
}
static inline V16 h_lookup_bytes(int8_t match, int8_t mismatch, uint8_t c, H16 v) {
    BA_LANEWISE((int16_t)(v.v[k] == c ? match : mismatch))
}
#undef BA_LANEWISE

static inline const char* simd_backend_name() { return "scalar-lane-model"; }

}  // namespace ba_oracle
