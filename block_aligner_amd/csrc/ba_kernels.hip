// One translation unit per (matrix kind, max-block class): compiled with -DBA_KIND=<0|1|2|3> -DBA_PMAX=<1|2|4|8|16>.
// BA_PMAX = max block size / 128 (packed VGPRs per border array per lane); a smaller class keeps the kernel's VGPR
// budget (and so its occupancy) matched to the batch's max block size.
#include "ba_driver.hpp"

#ifndef BA_KIND
#error "define BA_KIND"
#endif
#ifndef BA_PMAX
#error "define BA_PMAX"
#endif

#ifndef BA_SPECIAL
#define BA_SPECIAL 0   // 1: the kernels of this TU handle LOCAL_START / FREE_QUERY_START_GAPS / FREE_QUERY_END_GAPS batches
#endif

#define BA_CAT_(a, b, c, d) a##b##c##d
#define BA_CAT(a, b, c, d) BA_CAT_(a, b, c, d)
#if BA_BIG && BA_SPECIAL   // blocks of 4096 .. 32768 cells with LOCAL_START / FREE_QUERY_*_GAPS
#define BA_LAUNCH BA_CAT(ba_launch_bigs_k, BA_KIND, _p, BA_PMAX)
#define BA_OCC BA_CAT(ba_occupancy_bigs_k, BA_KIND, _p, BA_PMAX)
#elif BA_BIG   // blocks of 4096 .. 32768 cells (one class per kind)
#define BA_LAUNCH BA_CAT(ba_launch_big_k, BA_KIND, _p, BA_PMAX)
#define BA_OCC BA_CAT(ba_occupancy_big_k, BA_KIND, _p, BA_PMAX)
#elif BA_SPECIAL
#define BA_LAUNCH BA_CAT(ba_launch_s_k, BA_KIND, _p, BA_PMAX)
#define BA_OCC BA_CAT(ba_occupancy_s_k, BA_KIND, _p, BA_PMAX)
#else
#define BA_LAUNCH BA_CAT(ba_launch_k, BA_KIND, _p, BA_PMAX)
#define BA_OCC BA_CAT(ba_occupancy_k, BA_KIND, _p, BA_PMAX)
#endif

template <bool TRACE, bool XDROP>
static hipError_t launch1(unsigned grid, unsigned lds, hipStream_t s, const ba::BatchParams& bp) {
    if (lds > 64 * 1024) {   // more than the default dynamic-LDS limit: opt in (160 KB per CU on gfx950)
        hipError_t e = hipFuncSetAttribute((const void*)ba::k_align<BA_PMAX, BA_KIND, TRACE, XDROP, (BA_SPECIAL != 0)>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    ba::k_align<BA_PMAX, BA_KIND, TRACE, XDROP, (BA_SPECIAL != 0)><<<dim3(grid), dim3(ba::WAVES_PER_WG * 64), lds, s>>>(bp);
    return hipGetLastError();
}
template <bool TRACE, bool XDROP>
static hipError_t occ1(int* blocks_per_cu, unsigned lds) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)ba::k_align<BA_PMAX, BA_KIND, TRACE, XDROP, (BA_SPECIAL != 0)>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, ba::k_align<BA_PMAX, BA_KIND, TRACE, XDROP, (BA_SPECIAL != 0)>, ba::WAVES_PER_WG * 64, lds);
}

extern "C" hipError_t BA_LAUNCH(int trace, int xdrop, unsigned grid, unsigned lds, hipStream_t s, const ba::BatchParams* bp) {
    if (trace) return xdrop ? launch1<true, true>(grid, lds, s, *bp) : launch1<true, false>(grid, lds, s, *bp);
    return xdrop ? launch1<false, true>(grid, lds, s, *bp) : launch1<false, false>(grid, lds, s, *bp);
}
extern "C" hipError_t BA_OCC(int trace, int xdrop, unsigned lds, int* blocks_per_cu) {
    if (trace) return xdrop ? occ1<true, true>(blocks_per_cu, lds) : occ1<true, false>(blocks_per_cu, lds);
    return xdrop ? occ1<false, true>(blocks_per_cu, lds) : occ1<false, false>(blocks_per_cu, lds);
}

#if BA_KIND != 3 && !BA_SPECIAL && !BA_BIG
// four pairs per wave while the block is 128 cells, everything else by the same wave on all its lanes (ba_multi.hpp): one kernel per
// kind and block class
#include "ba_multi.hpp"
template <bool TRACE, bool XDROP>
static hipError_t launch_multi(unsigned grid, unsigned lds, hipStream_t s, const ba::BatchParams& bp) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP><<<dim3(grid), dim3(ba::WAVES_PER_WG * 64), lds, s>>>(bp);
    return hipGetLastError();
}
template <bool TRACE, bool XDROP>
static hipError_t occ_multi(int* blocks_per_cu, unsigned lds) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP>, ba::WAVES_PER_WG * 64, lds);
}
extern "C" hipError_t BA_CAT(ba_launch_m_k, BA_KIND, _p, BA_PMAX)(int trace, int xdrop, unsigned grid, unsigned lds, hipStream_t s, const ba::BatchParams* bp) {
    if (trace) return xdrop ? launch_multi<true, true>(grid, lds, s, *bp) : launch_multi<true, false>(grid, lds, s, *bp);
    return xdrop ? launch_multi<false, true>(grid, lds, s, *bp) : launch_multi<false, false>(grid, lds, s, *bp);
}
extern "C" hipError_t BA_CAT(ba_occupancy_m_k, BA_KIND, _p, BA_PMAX)(int trace, int xdrop, unsigned lds, int* blocks_per_cu) {
    if (trace) return xdrop ? occ_multi<true, true>(blocks_per_cu, lds) : occ_multi<true, false>(blocks_per_cu, lds);
    return xdrop ? occ_multi<false, true>(blocks_per_cu, lds) : occ_multi<false, false>(blocks_per_cu, lds);
}
#if BA_KIND == 1 && BA_PMAX >= 4
// ... with two slots of 256 cells per wave (round 6: DNA batches that start at 256 cells, block classes 512 .. 2048)
template <bool TRACE, bool XDROP>
static hipError_t launch_multi256(unsigned grid, unsigned lds, hipStream_t s, const ba::BatchParams& bp) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP, 0, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP, 0, 256><<<dim3(grid), dim3(ba::WAVES_PER_WG * 64), lds, s>>>(bp);
    return hipGetLastError();
}
template <bool TRACE, bool XDROP>
static hipError_t occ_multi256(int* blocks_per_cu, unsigned lds) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP, 0, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP, 0, 256>, ba::WAVES_PER_WG * 64, lds);
}
extern "C" hipError_t BA_CAT(ba_launch_m256_k, BA_KIND, _p, BA_PMAX)(int trace, int xdrop, unsigned grid, unsigned lds, hipStream_t s, const ba::BatchParams* bp) {
    if (trace) return xdrop ? launch_multi256<true, true>(grid, lds, s, *bp) : launch_multi256<true, false>(grid, lds, s, *bp);
    return xdrop ? launch_multi256<false, true>(grid, lds, s, *bp) : launch_multi256<false, false>(grid, lds, s, *bp);
}
extern "C" hipError_t BA_CAT(ba_occupancy_m256_k, BA_KIND, _p, BA_PMAX)(int trace, int xdrop, unsigned lds, int* blocks_per_cu) {
    if (trace) return xdrop ? occ_multi256<true, true>(blocks_per_cu, lds) : occ_multi256<true, false>(blocks_per_cu, lds);
    return xdrop ? occ_multi256<false, true>(blocks_per_cu, lds) : occ_multi256<false, false>(blocks_per_cu, lds);
}
#endif
#if BA_KIND == 1 && BA_PMAX >= 8
// ... with one slot of 512 cells per wave (round 6: DNA batches that start at 512 cells -- percent_len 1 % of reads above 25.6 kbp --, block classes 1024 and 2048)
template <bool TRACE, bool XDROP>
static hipError_t launch_multi512(unsigned grid, unsigned lds, hipStream_t s, const ba::BatchParams& bp) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP, 0, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP, 0, 512><<<dim3(grid), dim3(ba::WAVES_PER_WG * 64), lds, s>>>(bp);
    return hipGetLastError();
}
template <bool TRACE, bool XDROP>
static hipError_t occ_multi512(int* blocks_per_cu, unsigned lds) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP, 0, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP, 0, 512>, ba::WAVES_PER_WG * 64, lds);
}
extern "C" hipError_t BA_CAT(ba_launch_m512_k, BA_KIND, _p, BA_PMAX)(int trace, int xdrop, unsigned grid, unsigned lds, hipStream_t s, const ba::BatchParams* bp) {
    if (trace) return xdrop ? launch_multi512<true, true>(grid, lds, s, *bp) : launch_multi512<true, false>(grid, lds, s, *bp);
    return xdrop ? launch_multi512<false, true>(grid, lds, s, *bp) : launch_multi512<false, false>(grid, lds, s, *bp);
}
extern "C" hipError_t BA_CAT(ba_occupancy_m512_k, BA_KIND, _p, BA_PMAX)(int trace, int xdrop, unsigned lds, int* blocks_per_cu) {
    if (trace) return xdrop ? occ_multi512<true, true>(blocks_per_cu, lds) : occ_multi512<true, false>(blocks_per_cu, lds);
    return xdrop ? occ_multi512<false, true>(blocks_per_cu, lds) : occ_multi512<false, false>(blocks_per_cu, lds);
}
#endif
#if BA_KIND == 1 && (BA_PMAX == 4 || BA_PMAX == 8)
// ... in workgroups of four waves at three / two waves per SIMD (round 6: DNA batches whose pairs fill that many waves' slots about once -- ba_host.cpp batch_build)
template <bool TRACE, bool XDROP, int EU>
static hipError_t launch_multi_g(unsigned grid, unsigned lds, hipStream_t s, const ba::BatchParams& bp) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP, 0, 128, ba::MQ_GEOM_WPW, EU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP, 0, 128, ba::MQ_GEOM_WPW, EU><<<dim3(grid), dim3(ba::MQ_GEOM_WPW * 64), lds, s>>>(bp);
    return hipGetLastError();
}
template <bool TRACE, bool XDROP, int EU>
static hipError_t occ_multi_g(int* blocks_per_cu, unsigned lds) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP, 0, 128, ba::MQ_GEOM_WPW, EU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP, 0, 128, ba::MQ_GEOM_WPW, EU>, ba::MQ_GEOM_WPW * 64, lds);
}
#define BA_MULTI_GEOM(EU)                                                                                                                                              \
    extern "C" hipError_t BA_CAT(ba_launch_mg##EU##_k, BA_KIND, _p, BA_PMAX)(int trace, int xdrop, unsigned grid, unsigned lds, hipStream_t s, const ba::BatchParams* bp) { \
        if (trace) return xdrop ? launch_multi_g<true, true, EU>(grid, lds, s, *bp) : launch_multi_g<true, false, EU>(grid, lds, s, *bp);                              \
        return xdrop ? launch_multi_g<false, true, EU>(grid, lds, s, *bp) : launch_multi_g<false, false, EU>(grid, lds, s, *bp);                                       \
    }                                                                                                                                                                  \
    extern "C" hipError_t BA_CAT(ba_occupancy_mg##EU##_k, BA_KIND, _p, BA_PMAX)(int trace, int xdrop, unsigned lds, int* blocks_per_cu) {                               \
        if (trace) return xdrop ? occ_multi_g<true, true, EU>(blocks_per_cu, lds) : occ_multi_g<true, false, EU>(blocks_per_cu, lds);                                  \
        return xdrop ? occ_multi_g<false, true, EU>(blocks_per_cu, lds) : occ_multi_g<false, false, EU>(blocks_per_cu, lds);                                           \
    }
BA_MULTI_GEOM(3)
BA_MULTI_GEOM(2)
#undef BA_MULTI_GEOM
#endif
#endif

#if !BA_SPECIAL && !BA_BIG && BA_PMAX <= 8
#if BA_KIND == 3
#include "ba_multi.hpp"   // (k_small's column code builds on multi_rect's helpers; k_multi itself has no profile form)
#endif
// sixteen pairs per wave while the block is 32 cells, everything else by the same wave on all its lanes (ba_small.hpp): one kernel per
// kind and block class (up to 1024 cells: the solo driver's LDS borders fit in the slots' region)
#include "ba_small.hpp"
template <bool TRACE, bool XDROP>
static hipError_t launch_small(unsigned grid, unsigned lds, hipStream_t s, const ba::BatchParams& bp) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)ba::k_small<BA_PMAX, BA_KIND, TRACE, XDROP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    ba::k_small<BA_PMAX, BA_KIND, TRACE, XDROP><<<dim3(grid), dim3(ba::WAVES_PER_WG * 64), lds, s>>>(bp);
    return hipGetLastError();
}
template <bool TRACE, bool XDROP>
static hipError_t occ_small(int* blocks_per_cu, unsigned lds) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)ba::k_small<BA_PMAX, BA_KIND, TRACE, XDROP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, ba::k_small<BA_PMAX, BA_KIND, TRACE, XDROP>, ba::WAVES_PER_WG * 64, lds);
}
extern "C" hipError_t BA_CAT(ba_launch_sm_k, BA_KIND, _p, BA_PMAX)(int trace, int xdrop, unsigned grid, unsigned lds, hipStream_t s, const ba::BatchParams* bp) {
    if (trace) return xdrop ? launch_small<true, true>(grid, lds, s, *bp) : launch_small<true, false>(grid, lds, s, *bp);
    return xdrop ? launch_small<false, true>(grid, lds, s, *bp) : launch_small<false, false>(grid, lds, s, *bp);
}
extern "C" hipError_t BA_CAT(ba_occupancy_sm_k, BA_KIND, _p, BA_PMAX)(int trace, int xdrop, unsigned lds, int* blocks_per_cu) {
    if (trace) return xdrop ? occ_small<true, true>(blocks_per_cu, lds) : occ_small<true, false>(blocks_per_cu, lds);
    return xdrop ? occ_small<false, true>(blocks_per_cu, lds) : occ_small<false, false>(blocks_per_cu, lds);
}
#endif

#if BA_SPECIAL && !BA_BIG && BA_KIND != 3
// k_multi for LOCAL_START (spm 1) / FREE_QUERY_START_GAPS (spm 2) batches of the sequence kinds: the slots take these modes' plain steps too
// (FREE_QUERY_END_GAPS batches stay with the per-pair kernel: the mode's running column maxima are not a slot's)
#include "ba_multi.hpp"
template <bool TRACE, bool XDROP, int SPM>
static hipError_t launch_multi_s(unsigned grid, unsigned lds, hipStream_t s, const ba::BatchParams& bp) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP, SPM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP, SPM><<<dim3(grid), dim3(ba::WAVES_PER_WG * 64), lds, s>>>(bp);
    return hipGetLastError();
}
template <bool TRACE, bool XDROP, int SPM>
static hipError_t occ_multi_s(int* blocks_per_cu, unsigned lds) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP, SPM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, ba::k_multi<BA_PMAX, BA_KIND, TRACE, XDROP, SPM>, ba::WAVES_PER_WG * 64, lds);
}
template <int SPM>
static hipError_t launch_multi_m(int trace, int xdrop, unsigned grid, unsigned lds, hipStream_t s, const ba::BatchParams& bp) {
    if (trace) return xdrop ? launch_multi_s<true, true, SPM>(grid, lds, s, bp) : launch_multi_s<true, false, SPM>(grid, lds, s, bp);
    return xdrop ? launch_multi_s<false, true, SPM>(grid, lds, s, bp) : launch_multi_s<false, false, SPM>(grid, lds, s, bp);
}
template <int SPM>
static hipError_t occ_multi_m(int trace, int xdrop, unsigned lds, int* blocks_per_cu) {
    if (trace) return xdrop ? occ_multi_s<true, true, SPM>(blocks_per_cu, lds) : occ_multi_s<true, false, SPM>(blocks_per_cu, lds);
    return xdrop ? occ_multi_s<false, true, SPM>(blocks_per_cu, lds) : occ_multi_s<false, false, SPM>(blocks_per_cu, lds);
}
// (the batch's flags choose the instantiation)
extern "C" hipError_t BA_CAT(ba_launch_ms_k, BA_KIND, _p, BA_PMAX)(int trace, int xdrop, unsigned grid, unsigned lds, hipStream_t s, const ba::BatchParams* bp) {
    return (bp->flags & ba::F_LOCAL) ? launch_multi_m<1>(trace, xdrop, grid, lds, s, *bp) : launch_multi_m<2>(trace, xdrop, grid, lds, s, *bp);
}
extern "C" hipError_t BA_CAT(ba_occupancy_ms_k, BA_KIND, _p, BA_PMAX)(int trace, int xdrop, unsigned lds, int* blocks_per_cu) {
    int a = 0, b = 0;   // (one launch geometry for both)
    hipError_t e = occ_multi_m<1>(trace, xdrop, lds, &a);
    if (e != hipSuccess) return e;
    e = occ_multi_m<2>(trace, xdrop, lds, &b);
    *blocks_per_cu = a < b ? a : b;
    return e;
}
#endif

#if BA_SPECIAL && !BA_BIG && BA_PMAX <= 8 && BA_KIND != 3
// ... and k_small for the same two modes
#include "ba_small.hpp"
template <bool TRACE, bool XDROP, int SPM>
static hipError_t launch_small_s(unsigned grid, unsigned lds, hipStream_t s, const ba::BatchParams& bp) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)ba::k_small<BA_PMAX, BA_KIND, TRACE, XDROP, SPM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    ba::k_small<BA_PMAX, BA_KIND, TRACE, XDROP, SPM><<<dim3(grid), dim3(ba::WAVES_PER_WG * 64), lds, s>>>(bp);
    return hipGetLastError();
}
template <bool TRACE, bool XDROP, int SPM>
static hipError_t occ_small_s(int* blocks_per_cu, unsigned lds) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)ba::k_small<BA_PMAX, BA_KIND, TRACE, XDROP, SPM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, ba::k_small<BA_PMAX, BA_KIND, TRACE, XDROP, SPM>, ba::WAVES_PER_WG * 64, lds);
}
template <int SPM>
static hipError_t launch_small_m(int trace, int xdrop, unsigned grid, unsigned lds, hipStream_t s, const ba::BatchParams& bp) {
    if (trace) return xdrop ? launch_small_s<true, true, SPM>(grid, lds, s, bp) : launch_small_s<true, false, SPM>(grid, lds, s, bp);
    return xdrop ? launch_small_s<false, true, SPM>(grid, lds, s, bp) : launch_small_s<false, false, SPM>(grid, lds, s, bp);
}
template <int SPM>
static hipError_t occ_small_m(int trace, int xdrop, unsigned lds, int* blocks_per_cu) {
    if (trace) return xdrop ? occ_small_s<true, true, SPM>(blocks_per_cu, lds) : occ_small_s<true, false, SPM>(blocks_per_cu, lds);
    return xdrop ? occ_small_s<false, true, SPM>(blocks_per_cu, lds) : occ_small_s<false, false, SPM>(blocks_per_cu, lds);
}
// (the batch's flags choose the instantiation)
extern "C" hipError_t BA_CAT(ba_launch_sms_k, BA_KIND, _p, BA_PMAX)(int trace, int xdrop, unsigned grid, unsigned lds, hipStream_t s, const ba::BatchParams* bp) {
    return (bp->flags & ba::F_LOCAL) ? launch_small_m<1>(trace, xdrop, grid, lds, s, *bp) : launch_small_m<2>(trace, xdrop, grid, lds, s, *bp);
}
extern "C" hipError_t BA_CAT(ba_occupancy_sms_k, BA_KIND, _p, BA_PMAX)(int trace, int xdrop, unsigned lds, int* blocks_per_cu) {
    int a = 0, b = 0;   // (one launch geometry for both)
    hipError_t e = occ_small_m<1>(trace, xdrop, lds, &a);
    if (e != hipSuccess) return e;
    e = occ_small_m<2>(trace, xdrop, lds, &b);
    *blocks_per_cu = a < b ? a : b;
    return e;
}
#endif

#if BA_PMAX == 1 && !BA_SPECIAL && !BA_BIG
// four pairs per wave while the block is 32 cells (ba_quad.hpp): one kernel per kind
#include "ba_quad.hpp"
template <bool TRACE, bool XDROP>
static hipError_t quad_grid(unsigned* grid) {
    const unsigned lds = ba::lds_table_bytes_h(BA_KIND) + ba::WAVES_PER_WG * 4 * ba::QUAD_SLOT_BYTES;
    int per_cu = 0, dev = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ba::k_quad<BA_KIND, TRACE, XDROP>, ba::WAVES_PER_WG * 64, lds);
    if (e != hipSuccess) return e;
    hipDeviceProp_t prop;
    if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    if (per_cu * ba::WAVES_PER_WG > 32) per_cu = 32 / ba::WAVES_PER_WG;
    *grid = (unsigned)(prop.multiProcessorCount * per_cu);
    return hipSuccess;
}
template <bool TRACE, bool XDROP>
static hipError_t launch_quad(unsigned grid, hipStream_t s, const ba::BatchParams& bp) {
    const unsigned lds = ba::lds_table_bytes_h(BA_KIND) + ba::WAVES_PER_WG * 4 * ba::QUAD_SLOT_BYTES;
    ba::k_quad<BA_KIND, TRACE, XDROP><<<dim3(grid), dim3(ba::WAVES_PER_WG * 64), lds, s>>>(bp);
    return hipGetLastError();
}
// grid: workgroups, from ba_quad_grid_k* (bp->cq_producers must be grid * WAVES_PER_WG)
extern "C" hipError_t BA_CAT(ba_launch_quad_k, BA_KIND, , )(int trace, int xdrop, unsigned grid, hipStream_t s, const ba::BatchParams* bp) {
    if (trace) return xdrop ? launch_quad<true, true>(grid, s, *bp) : launch_quad<true, false>(grid, s, *bp);
    return xdrop ? launch_quad<false, true>(grid, s, *bp) : launch_quad<false, false>(grid, s, *bp);
}
extern "C" hipError_t BA_CAT(ba_quad_grid_k, BA_KIND, , )(int trace, int xdrop, unsigned* grid) {
    if (trace) return xdrop ? quad_grid<true, true>(grid) : quad_grid<true, false>(grid);
    return xdrop ? quad_grid<false, true>(grid) : quad_grid<false, false>(grid);
}
#endif

#if BA_KIND == 0 && BA_PMAX == 1 && !BA_SPECIAL && !BA_BIG
// kernels that exist once
// Traceback from an arbitrary end cell over slot 0's trace (the per-handle API: block_cigar_* after block_align_*).
__global__ void __launch_bounds__(64) k_traceback(const ba::BatchParams bp) {
    using namespace ba;
    __shared__ uint32_t words[4096];
    uint32_t st = 0, n = 0;
    if (bp.flags & (F_LOCAL | F_FQS)) {
        if (lane_id() == 0)
            n = traceback(bp.blocks + (uint64_t)bp.tb_slot * bp.blocks_stride, bp.tb_nblocks, bp.trace_arena + (uint64_t)bp.tb_slot * bp.trace_stride, bp.tb_i,
                          bp.tb_j, bp.pool + bp.q_off[0], bp.pool + bp.r_off[0], bp.flags, bp.cig_ops, bp.cig_off[0], bp.cig_off[1], &st);
    } else
        n = walk_wave<false>(bp.blocks + (uint64_t)bp.tb_slot * bp.blocks_stride, bp.tb_nblocks, bp.trace_arena + (uint64_t)bp.tb_slot * bp.trace_stride, bp.tb_i,
                             bp.tb_j, bp.pool + bp.q_off[0], bp.pool + bp.r_off[0], (bp.flags & F_CIGAR_EQ) != 0, bp.cig_ops, bp.cig_off[0], bp.cig_off[1], &st,
                             words, 4096u);
    if (lane_id() == 0) { bp.cig_len[0] = n; bp.status[0] = st; }
}

// Device-side known-answer test of the lane primitives (round 6; the reference's own: avx2.rs:469-489, plus the carry from vector to vector of
// scan_block.rs:1144-1150): columns of x = D11_open in, R11 out, through the very functions the fill kernels call -- the scan constants
// (make_fill_consts / make_multi_consts / make_small_consts) included. One wave per workgroup; a column starts from MIN = 0 above its first cell.
//   form 0: k_multi's eight cells per lane, 16 lanes to a slot (four 128-cell columns per wave): scan8_* + multi_carry (wave_prefix_max16, G / w0)
//   form 1: k_small's eight cells per lane, 4 lanes to a slot (sixteen 32-cell columns per wave): scan8_* + small_carry (quad_prefix_max)
//   form 2 / 3 / 4: two cells per lane, 64 / 32 / 16 lanes (one 128- / 64- / 32-cell column per wave): fast_scan (k_align, k_quad, the solo drivers)
// Reached only through the development library (ba_dev_lane_scan, ba_host.cpp).
__global__ void __launch_bounds__(64) k_lane_kat(int form, const short* __restrict__ x, short* __restrict__ out, int gap_extend) {
    using namespace ba;
    const int lane = lane_id();
    if (form <= 1) {
        const short* xi = x + (size_t)blockIdx.x * 512 + lane * 8;
        int xr[4], r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) xr[k] = pk((int)xi[2 * k], (int)xi[2 * k + 1]);
        const int ge2 = splat(gap_extend);
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = scan8_inreg(xr[k], ge2);
        int G[4], cs;
        if (form == 0) {
            const MultiConsts mc = make_multi_consts(lane & 15, gap_extend);
#pragma unroll
            for (int k = 0; k < 4; k++) G[k] = mc.G[k];
            scan8_chain(r, G[0]);
            cs = multi_carry(r[3], mc);
        } else {
            const SmallConsts mc = make_small_consts(lane & 3, gap_extend);
#pragma unroll
            for (int k = 0; k < 4; k++) G[k] = mc.G[k];
            scan8_chain(r, G[0]);
            cs = small_carry(r[3], mc, lane & 3);
        }
        short* oi = out + (size_t)blockIdx.x * 512 + lane * 8;
#pragma unroll
        for (int k = 0; k < 4; k++) { const int v = scan8_apply(r[k], cs, G[k], k == 3); oi[2 * k] = as_s(v).x; oi[2 * k + 1] = as_s(v).y; }
    } else {
        const int nl = form == 2 ? 64 : (form == 3 ? 32 : 16);
        const size_t base = (size_t)blockIdx.x * (size_t)(2 * nl);
        const int xv = lane < nl ? pk((int)x[base + 2 * lane], (int)x[base + 2 * lane + 1]) : 0;
        const FillConsts fc = make_fill_consts(lane, 0, gap_extend);
        const int v = form == 2 ? fast_scan<64>(xv, fc) : (form == 3 ? fast_scan<32>(xv, fc) : fast_scan<16>(xv, fc));
        if (lane < nl) { out[base + 2 * lane] = as_s(v).x; out[base + 2 * lane + 1] = as_s(v).y; }
    }
}
extern "C" hipError_t ba_launch_lane_kat(hipStream_t s, int form, const short* x, short* out, int gap_extend, unsigned waves) {
    k_lane_kat<<<dim3(waves), dim3(64), 0, s>>>(form, x, out, gap_extend);
    return hipGetLastError();
}

// Pair-slot batches: all tracebacks of the batch, one pair per lane (ba_driver.hpp traceback_all).
constexpr int WALK_WAVES = 4;
__global__ void __launch_bounds__(WALK_WAVES * 64) k_walk(const ba::BatchParams bp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char walk_lds[];
    ba::traceback_all(bp, (uint32_t)ba::F_CIGAR_EQ, walk_lds + ((uint32_t)threadIdx.x >> 6) * ba::TB_LDS_BYTES);
}

// ... of a k_small batch (slot rectangles: words of 4 cells x 2 columns; the lanes' records are larger)
__global__ void __launch_bounds__(WALK_WAVES * 64) k_walk_l2(const ba::BatchParams bp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char walk_lds[];
    ba::traceback_all<true>(bp, (uint32_t)ba::F_CIGAR_EQ, walk_lds + ((uint32_t)threadIdx.x >> 6) * ba::TB_LDS_BYTES_L2);
}

// ... of a LOCAL_START / FREE_QUERY_START_GAPS batch of k_small (the lanes' records also hold the cells' zero-mask bits)
__global__ void __launch_bounds__(WALK_WAVES * 64) k_walk_loc(const ba::BatchParams bp) {
    extern __shared__ __attribute__((aligned(16))) unsigned char walk_lds[];
    ba::traceback_all<true, true>(bp, (uint32_t)ba::F_CIGAR_EQ, walk_lds + ((uint32_t)threadIdx.x >> 6) * ba::TB_LDS_BYTES_LOC);
}

__global__ void __launch_bounds__(256) k_compact_cigars(const uint32_t* __restrict__ ops, const uint64_t* __restrict__ cig_off,
                                                        const uint32_t* __restrict__ cig_len, const uint64_t* __restrict__ out_off,
                                                        uint32_t* __restrict__ out, uint32_t n) {
    for (uint32_t p = blockIdx.x; p < n; p += gridDim.x) {
        const uint32_t len = cig_len[p];
        const uint64_t src = cig_off[p + 1] - len, dst = out_off[p];
        for (uint32_t k = threadIdx.x; k < len; k += blockDim.x) out[dst + k] = ops[src + k];
    }
}
// Offsets of the compacted CIGAR runs, pair after pair in the CALLER's order (dev_of[p] = device position of the caller's pair p):
// out_off[dev_of[p]] = sum of cig_len[dev_of[p']] over p' < p; *total = the sum over all pairs. One workgroup: per-thread chunk sums,
// a scan over the 1024 partial sums, a second pass.
__global__ void __launch_bounds__(1024) k_cigar_offsets(const uint32_t* __restrict__ cig_len, const uint32_t* __restrict__ dev_of, uint64_t* __restrict__ out_off,
                                                        unsigned long long* __restrict__ total, uint32_t n) {
    __shared__ unsigned long long part[1024];
    const uint32_t t = threadIdx.x, per = (n + 1023u) / 1024u;
    const uint32_t lo = min(t * per, n), hi = min(lo + per, n);
    unsigned long long sum = 0;
    for (uint32_t p = lo; p < hi; p++) sum += cig_len[dev_of ? dev_of[p] : p];
    part[t] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {   // inclusive scan
        const unsigned long long v = t >= d ? part[t - d] : 0ull;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned long long at = part[t] - sum;
    for (uint32_t p = lo; p < hi; p++) { const uint32_t s = dev_of ? dev_of[p] : p; out_off[s] = at; at += cig_len[s]; }
    if (t == 1023) *total = part[1023];
}
// the gather itself, only if everything fits (else *total says how much is needed and nothing is written)
__global__ void __launch_bounds__(256) k_compact_cigars_checked(const uint32_t* __restrict__ ops, const uint64_t* __restrict__ cig_off,
                                                                const uint32_t* __restrict__ cig_len, const uint64_t* __restrict__ out_off,
                                                                uint32_t* __restrict__ out, uint32_t n, const unsigned long long* __restrict__ total, unsigned long long capacity) {
    if (*total > capacity) return;
    for (uint32_t p = blockIdx.x; p < n; p += gridDim.x) {
        const uint32_t len = cig_len[p];
        const uint64_t src = cig_off[p + 1] - len, dst = out_off[p];
        for (uint32_t k = threadIdx.x; k < len; k += blockDim.x) out[dst + k] = ops[src + k];
    }
}
extern "C" hipError_t ba_launch_cigar_offsets_and_compact(hipStream_t s, const uint32_t* ops, const uint64_t* cig_off, const uint32_t* cig_len, const uint32_t* dev_of,
                                                          uint64_t* out_off, uint32_t* out, unsigned long long* total, unsigned long long capacity, uint32_t n) {
    k_cigar_offsets<<<dim3(1), dim3(1024), 0, s>>>(cig_len, dev_of, out_off, total, n);
    const unsigned grid = n < 4096 ? (n ? n : 1) : 4096;
    k_compact_cigars_checked<<<dim3(grid), dim3(256), 0, s>>>(ops, cig_off, cig_len, out_off, out, n, total, capacity);
    return hipGetLastError();
}
extern "C" hipError_t ba_launch_compact_cigars(hipStream_t s, const uint32_t* ops, const uint64_t* cig_off, const uint32_t* cig_len,
                                               const uint64_t* out_off, uint32_t* out, uint32_t n) {
    const unsigned grid = n < 4096 ? (n ? n : 1) : 4096;
    k_compact_cigars<<<dim3(grid), dim3(256), 0, s>>>(ops, cig_off, cig_len, out_off, out, n);
    return hipGetLastError();
}
// PaddedBytes images on the device (scan_block.rs:1798-1850: [NULL] + convert(bytes) + NULL padding; convert_char of
// scores.rs:130-134 / 212-216 / 270-272): one workgroup per sequence, four image bytes per thread and store. A byte the
// reference would assert on is reported through *err (lowest pair wins): pair << 8 | byte.
__global__ void __launch_bounds__(256) k_pack_sequences(int kind, const uint8_t* __restrict__ raw, const uint64_t* __restrict__ raw_q,
                                                        const uint64_t* __restrict__ raw_r, const uint64_t* __restrict__ q_off,
                                                        const uint32_t* __restrict__ q_len, const uint64_t* __restrict__ r_off,
                                                        const uint32_t* __restrict__ r_len, uint8_t* __restrict__ image, uint32_t pad,
                                                        unsigned long long* err) {
    const uint32_t p = blockIdx.x >> 1;
    const bool ref = blockIdx.x & 1;
    const uint8_t* src = raw + (ref ? raw_r[p] : raw_q[p]);
    const uint32_t len = ref ? r_len[p] : q_len[p];
    uint32_t* dst = (uint32_t*)(image + (ref ? r_off[p] : q_off[p]));   // images start 4-byte aligned
    const uint32_t null_b = kind == ba::KIND_AA ? 26u : (kind == ba::KIND_NUC ? (uint32_t)'Z' : 0u);
    const uint32_t words = ((1u + len + pad + 3u) & ~3u) / 4u;
    for (uint32_t k = threadIdx.x; k < words; k += blockDim.x) {
        uint32_t wd = 0;
#pragma unroll
        for (uint32_t b = 0; b < 4; b++) {
            const uint32_t pos = 4 * k + b;
            uint32_t c = null_b;
            if (pos >= 1 && pos <= len) {
                c = src[pos - 1];
                if (kind != ba::KIND_BYTES) {
                    if (c >= 'a' && c <= 'z') c -= 32;
                    const bool ok = kind == ba::KIND_AA ? (c >= 'A' && c <= 'A' + 26) : (c >= 'A' && c <= 'Z');
                    if (!ok) { atomicMin(err, ((unsigned long long)p << 8) | src[pos - 1]); c = null_b; }
                    else if (kind == ba::KIND_AA) c -= 'A';
                }
            }
            wd |= c << (8 * b);
        }
        dst[k] = wd;
    }
}
extern "C" hipError_t ba_launch_pack_sequences(hipStream_t s, int kind, const uint8_t* raw, const uint64_t* raw_q, const uint64_t* raw_r,
                                               const uint64_t* q_off, const uint32_t* q_len, const uint64_t* r_off, const uint32_t* r_len,
                                               uint8_t* image, uint32_t pad, uint32_t n, unsigned long long* err) {
    k_pack_sequences<<<dim3(2 * n), dim3(256), 0, s>>>(kind, raw, raw_q, raw_r, q_off, q_len, r_off, r_len, image, pad, err);
    return hipGetLastError();
}
// Results of a re-run (pairs whose trace stack outgrew an adaptively sized slot, see batch_plan) back into the batch's own
// arrays: one workgroup per pair; sub-batch entry k belongs to the batch's (device-order) pair idx[k].
__global__ void __launch_bounds__(256) k_merge_retry(const uint32_t* __restrict__ idx, const ba::BatchParams sub, const ba::BatchParams dst,
                                                     const uint32_t* __restrict__ sub_trace_words, uint32_t* __restrict__ dst_trace_words) {
    const uint32_t k = blockIdx.x, p = idx[k];
    const uint32_t len = sub.cig_len[k];
    if (threadIdx.x == 0) {
        dst.score[p] = sub.score[k]; dst.query_idx[p] = sub.query_idx[k]; dst.reference_idx[p] = sub.reference_idx[k];
        dst.cig_len[p] = len; dst.status[p] = sub.status[k]; dst.cells[p] = sub.cells[k];
        dst_trace_words[p] = sub_trace_words[k];
    }
    if (!sub.cig_ops || !dst.cig_ops) return;
    const uint64_t src = sub.cig_off[k + 1] - len, to = dst.cig_off[p + 1] - len;   // runs are right-aligned in a pair's range
    for (uint32_t t = threadIdx.x; t < len; t += blockDim.x) dst.cig_ops[to + t] = sub.cig_ops[src + t];
}
extern "C" hipError_t ba_launch_merge_retry(hipStream_t s, const uint32_t* idx, uint32_t k, const ba::BatchParams* sub, const ba::BatchParams* dst,
                                            const uint32_t* sub_tw, uint32_t* dst_tw) {
    k_merge_retry<<<dim3(k), dim3(256), 0, s>>>(idx, *sub, *dst, sub_tw, dst_tw);
    return hipGetLastError();
}
extern "C" hipError_t ba_launch_walk(hipStream_t s, const ba::BatchParams* bp, uint32_t grid) {
    k_walk<<<dim3(grid), dim3(WALK_WAVES * 64), WALK_WAVES * ba::TB_LDS_BYTES, s>>>(*bp);
    return hipGetLastError();
}
extern "C" hipError_t ba_launch_walk_l2(hipStream_t s, const ba::BatchParams* bp, uint32_t grid) {
    k_walk_l2<<<dim3(grid), dim3(WALK_WAVES * 64), WALK_WAVES * ba::TB_LDS_BYTES_L2, s>>>(*bp);
    return hipGetLastError();
}
extern "C" hipError_t ba_launch_walk_loc(hipStream_t s, const ba::BatchParams* bp, uint32_t grid) {
    k_walk_loc<<<dim3(grid), dim3(WALK_WAVES * 64), WALK_WAVES * ba::TB_LDS_BYTES_LOC, s>>>(*bp);
    return hipGetLastError();
}
extern "C" hipError_t ba_launch_traceback(hipStream_t s, const ba::BatchParams* bp) {
    k_traceback<<<dim3(1), dim3(64), 0, s>>>(*bp);
    return hipGetLastError();
}
#endif
