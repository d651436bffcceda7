// Microbenchmark (gfx950): issue cost of the VALU / SALU instructions the block fill uses or could use.
// Every kernel runs `iters` x 64 instructions on 8 independent accumulators per wave; W waves per SIMD (W x 256 CUs x 4
// waves). Two clocks are reported: cycles per wave-instruction per SIMD from the HIP-event wall time at the nominal
// 2.4 GHz, and from the wave's own s_memtime delta (shader cycles; independent of DVFS and of launch overheads).
//   hipcc --offload-arch=gfx950 -O2 tools/dev/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ALL8(INS) INS("%0") INS("%1") INS("%2") INS("%3") INS("%4") INS("%5") INS("%6") INS("%7")
#define REP8(x) x x x x x x x x
#define ACC "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)

#define I_PKADD(r) "v_pk_add_i16 " r ", " r ", %8 clamp\n"
#define I_PKMAX(r) "v_pk_max_i16 " r ", " r ", %8\n"
#define I_PKMAXSEL(r) "v_pk_max_i16 " r ", " r ", %9 op_sel_hi:[1,0]\n"
#define I_PKSUBU(r) "v_pk_sub_u16 " r ", " r ", %9\n"
#define I_PKMINU(r) "v_pk_min_u16 " r ", " r ", 1 op_sel_hi:[1,0]\n"
#define I_PKMAD(r) "v_pk_mad_u16 " r ", " r ", %8, %9\n"
#define I_PKASHR(r) "v_pk_ashrrev_i16 " r ", 15, " r "\n"
#define I_ADDU32(r) "v_add_u32 " r ", " r ", %8\n"
#define I_MAXI32(r) "v_max_i32 " r ", " r ", %9\n"
#define I_MAX3(r) "v_max3_i32 " r ", " r ", %8, %9\n"
#define I_MAXDPP(r) "v_max_i32_dpp " r ", " r ", " r " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_MAXDPPB(r) "v_max_i32_dpp " r ", " r ", " r " row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define I_MOVDPP(r) "v_mov_b32_dpp " r ", " r " wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADDDPP(r) "v_add_u32_dpp " r ", " r ", %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define I_ALIGNBIT(r) "v_alignbit_b32 " r ", " r ", %9, 16\n"
#define I_PERM(r) "v_perm_b32 " r ", " r ", %9, %8\n"
#define I_BFI(r) "v_bfi_b32 " r ", %8, " r ", %9\n"
#define I_AND(r) "v_and_b32 " r ", %8, " r "\n"
#define I_LSHLOR(r) "v_lshl_or_b32 " r ", " r ", 4, %9\n"
#define I_ANDOR(r) "v_and_or_b32 " r ", " r ", %8, %9\n"
#define I_SUBSDWA(r) "v_sub_u32_sdwa " r ", sext(" r "), %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define I_MAXI16SDWA(r) "v_max_i16_sdwa " r ", " r ", %9 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0\n"
#define I_CMPSDWA(r) "v_cmp_eq_u16_sdwa s[20:21], " r ", %9 src0_sel:WORD_1 src1_sel:WORD_1\n"
#define I_CMP32(r) "v_cmp_lt_i32 vcc, " r ", %9\n"
#define I_CNDMASK(r) "v_cndmask_b32 " r ", " r ", %9, vcc\n"
#define I_FMA(r) "v_fma_f32 " r ", " r ", %9, %9\n"
#define I_ADDF32(r) "v_add_f32 " r ", " r ", %9\n"
#define I_MAXI16(r) "v_max_i16 " r ", " r ", %9\n"
#define I_ADDI16(r) "v_add_i16 " r ", " r ", %9 clamp\n"
#define I_MADI32I16(r) "v_mad_i32_i16 " r ", " r ", %9, %9\n"
#define I_DOT2(r) "v_dot2_i32_i16 " r ", " r ", %9, " r "\n"
#define I_READLANE(r) "v_readlane_b32 s20, " r ", 5\n"
#define I_WRITELANE(r) "v_writelane_b32 " r ", %8, 5\n"
#define I_ASHR(r) "v_ashrrev_i32 " r ", 16, " r "\n"
#define I_SALU(r) "s_add_u32 s20, s20, %8\n"
#define I_MIX_VS(r) "v_pk_add_i16 " r ", " r ", %8 clamp\n s_add_u32 s20, s20, %8\n"
#define I_MIX_VSS(r) "v_pk_add_i16 " r ", " r ", %8 clamp\n s_add_u32 s20, s20, %8\n s_max_i32 s21, s21, %8\n"
#define I_NOP(r) "v_pk_max_i16 " r ", " r ", %8\n s_nop 1\n"
#define I_DEP(r) "v_pk_add_i16 %0, %0, %8 clamp\n"


// ---- survey of candidate fast-path instructions (VGPR sources unless the name says otherwise)
#define J_ADDU32V(r) "v_add_u32 " r ", " r ", %9\n"
#define J_SUBU32V(r) "v_sub_u32 " r ", " r ", %9\n"
#define J_MINI32(r) "v_min_i32 " r ", " r ", %9\n"
#define J_MAXU32(r) "v_max_u32 " r ", " r ", %9\n"
#define J_LSHL(r) "v_lshlrev_b32 " r ", 3, " r "\n"
#define J_LSHR(r) "v_lshrrev_b32 " r ", 3, " r "\n"
#define J_ANDV(r) "v_and_b32 " r ", " r ", %9\n"
#define J_ORV(r) "v_or_b32 " r ", " r ", %9\n"
#define J_XORV(r) "v_xor_b32 " r ", " r ", %9\n"
#define J_MAXF32(r) "v_max_f32 " r ", " r ", %9\n"
#define J_MINF32(r) "v_min_f32 " r ", " r ", %9\n"
#define J_MULF32(r) "v_mul_f32 " r ", " r ", %9\n"
#define J_SUBF32(r) "v_sub_f32 " r ", " r ", %9\n"
#define J_FMAC(r) "v_fmac_f32 " r ", %9, %9\n"
#define J_ADDU16(r) "v_add_u16 " r ", " r ", %9\n"
#define J_SUBU16(r) "v_sub_u16 " r ", " r ", %9\n"
#define J_MAXU16(r) "v_max_u16 " r ", " r ", %9\n"
#define J_MINI16(r) "v_min_i16 " r ", " r ", %9\n"
#define J_LSHL16(r) "v_lshlrev_b16 " r ", 3, " r "\n"
#define J_ASHR16(r) "v_ashrrev_i16 " r ", 3, " r "\n"
#define J_MULLO16(r) "v_mul_lo_u16 " r ", " r ", %9\n"
#define J_MULI24(r) "v_mul_i32_i24 " r ", " r ", %9\n"
#define J_MOV(r) "v_mov_b32 " r ", %9\n"
#define J_CVTF(r) "v_cvt_f32_i32 " r ", " r "\n"
#define J_CVTI(r) "v_cvt_i32_f32 " r ", " r "\n"
#define J_NOT(r) "v_not_b32 " r ", " r "\n"
#define J_MAXF32DPP(r) "v_max_f32_dpp " r ", " r ", " r " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define J_MAXF32DPPB(r) "v_max_f32_dpp " r ", " r ", " r " row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define J_ADDF32DPP(r) "v_add_f32_dpp " r ", " r ", %9 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
#define J_MAXI16DPP(r) "v_max_i16_dpp " r ", " r ", " r " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define J_ASHRDPP(r) "v_ashrrev_i32_dpp " r ", %9, " r " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define J_MED3I(r) "v_med3_i32 " r ", " r ", %8, %9\n"
#define J_MED3F(r) "v_med3_f32 " r ", " r ", %9, %9\n"
#define J_MAX3F(r) "v_max3_f32 " r ", " r ", %9, %9\n"
#define J_ADD3(r) "v_add3_u32 " r ", " r ", %8, %9\n"
#define J_MAXF32E64(r) "v_max_f32_e64 " r ", " r ", %9\n"
#define J_PKMAXF16(r) "v_pk_max_f16 " r ", " r ", %9\n"
#define J_PKADDF16(r) "v_pk_add_f16 " r ", " r ", %9\n"
#define J_ADDF32S(r) "v_add_f32 " r ", %8, " r "\n"
#define J_ADDF32K(r) "v_add_f32 " r ", 0x4b400000, " r "\n"
#define J_CNDMASK2(r) "v_cndmask_b32 " r ", " r ", %9, vcc\n"
#define J_MAXI16S(r) "v_max_i16 " r ", %8, " r "\n"
#define J_SUBREV(r) "v_subrev_u32 " r ", %9, " r "\n"
#define J_LSHLADD(r) "v_lshl_add_u32 " r ", " r ", 3, %9\n"
#define J_ADDLSHL(r) "v_add_lshl_u32 " r ", " r ", %9, 3\n"
#define J_SADU16(r) "v_sad_u16 " r ", " r ", %9, " r "\n"
#define J_MIX2(r) "v_pk_add_i16 " r ", " r ", %8 clamp\n v_max_i16 " r ", " r ", %9\n"
#define J_MIX3(r) "v_add_f32 " r ", " r ", %9\n v_max_i16 " r ", " r ", %9\n"
#define K_CNDE64(r) "v_cndmask_b32_e64 " r ", " r ", %9, s[20:21]\n"
#define K_CNDMIX(r) "v_cndmask_b32 " r ", " r ", %9, vcc\n v_pk_add_i16 " r ", " r ", %9 clamp\n v_pk_add_i16 " r ", " r ", %9 clamp\n v_pk_add_i16 " r ", " r ", %9 clamp\n"
#define K_CND0(r) "v_cndmask_b32_e64 " r ", " r ", %9, s[22:23]\n"
#define K_CNDEXEC(r) "v_cndmask_b32_e64 " r ", " r ", %9, exec\n"

// ---- round 4: groups shaped like multi_rect's mix (recurrence: packed VOP3P; trace packing / shifts / moves: the candidates for 2.3-cycle VOP2 forms)
#define L_3PK_AND(r) "v_pk_add_i16 " r ", " r ", %9 clamp\n v_pk_max_i16 " r ", " r ", %9\n v_pk_add_i16 " r ", " r ", %9 clamp\n v_and_b32 " r ", " r ", %9\n"
#define L_3PK_BFI(r) "v_pk_add_i16 " r ", " r ", %9 clamp\n v_pk_max_i16 " r ", " r ", %9\n v_pk_add_i16 " r ", " r ", %9 clamp\n v_bfi_b32 " r ", %9, " r ", %9\n"
#define L_4PK(r) "v_pk_add_i16 " r ", " r ", %9 clamp\n v_pk_max_i16 " r ", " r ", %9\n v_pk_add_i16 " r ", " r ", %9 clamp\n v_pk_max_i16 " r ", " r ", %9\n"
#define L_SHR_BFI(r) "v_lshrrev_b32 " r ", 1, " r "\n v_bfi_b32 " r ", %9, " r ", %9\n"
#define L_SHR_AND_AND_OR(r) "v_lshrrev_b32 " r ", 1, " r "\n v_and_b32 " r ", " r ", %9\n v_and_b32 " r ", " r ", %9\n v_or_b32 " r ", " r ", %9\n"
#define L_PK_FAST_PK_FAST(r) "v_pk_add_i16 " r ", " r ", %9 clamp\n v_and_b32 " r ", " r ", %9\n v_pk_max_i16 " r ", " r ", %9\n v_lshrrev_b32 " r ", 1, " r "\n"
#define L_2PK_2FAST(r) "v_pk_add_i16 " r ", " r ", %9 clamp\n v_pk_max_i16 " r ", " r ", %9\n v_and_b32 " r ", " r ", %9\n v_lshrrev_b32 " r ", 1, " r "\n"
#define L_MAXI16_PAIR(r) "v_max_i16 " r ", " r ", %9\n v_max_i16_sdwa " r ", " r ", %9 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1\n"
#define L_PERM(r) "v_perm_b32 " r ", " r ", %9, %9\n"
#define L_SHL16_OR(r) "v_lshlrev_b16 " r ", 8, " r "\n v_or_b32 " r ", " r ", %9\n"
struct Case { const char* name; int id; int per_group; };   // per_group: VALU instructions counted per accumulator visit

template <int KIND>
__global__ void __launch_bounds__(256) k(int* out, unsigned long long* ticks, int iters, int s) {
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const int vb = threadIdx.x * 3 + 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[20:21], 0x3333\n s_mov_b64 s[22:23], 0" ::: "vcc", "s20", "s21", "s22", "s23");
    for (int i = 0; i < iters; i++) {
#define CASE(N, INS) if (KIND == N) { REP8(asm volatile(ALL8(INS) : ACC : "s"(s), "v"(vb) : "s20", "s21", "vcc", "scc");) }
        CASE(0, I_PKADD) CASE(1, I_PKMAX) CASE(2, I_PKMAXSEL) CASE(3, I_PKSUBU) CASE(4, I_PKMINU) CASE(5, I_PKMAD) CASE(6, I_PKASHR)
        CASE(7, I_ADDU32) CASE(8, I_MAXI32) CASE(9, I_MAX3) CASE(10, I_MAXDPP) CASE(11, I_MAXDPPB) CASE(12, I_MOVDPP) CASE(13, I_ADDDPP)
        CASE(14, I_ALIGNBIT) CASE(15, I_PERM) CASE(16, I_BFI) CASE(17, I_AND) CASE(18, I_LSHLOR) CASE(19, I_ANDOR) CASE(20, I_SUBSDWA)
        CASE(21, I_MAXI16SDWA) CASE(22, I_CMPSDWA) CASE(23, I_CMP32) CASE(24, I_CNDMASK) CASE(25, I_FMA) CASE(26, I_ADDF32) CASE(27, I_MAXI16)
        CASE(28, I_ADDI16) CASE(29, I_MADI32I16) CASE(30, I_DOT2) CASE(31, I_READLANE) CASE(32, I_WRITELANE) CASE(33, I_ASHR)
        CASE(34, I_SALU) CASE(35, I_MIX_VS) CASE(36, I_MIX_VSS) CASE(37, I_NOP) CASE(38, I_DEP)
        CASE(40, J_ADDU32V) CASE(41, J_SUBU32V) CASE(42, J_MINI32) CASE(43, J_MAXU32) CASE(44, J_LSHL) CASE(45, J_LSHR) CASE(46, J_ANDV)
        CASE(47, J_ORV) CASE(48, J_XORV) CASE(49, J_MAXF32) CASE(50, J_MINF32) CASE(51, J_MULF32) CASE(52, J_SUBF32) CASE(53, J_FMAC)
        CASE(54, J_ADDU16) CASE(55, J_SUBU16) CASE(56, J_MAXU16) CASE(57, J_MINI16) CASE(58, J_LSHL16) CASE(59, J_ASHR16) CASE(60, J_MULLO16)
        CASE(61, J_MULI24) CASE(62, J_MOV) CASE(63, J_CVTF) CASE(64, J_CVTI) CASE(65, J_NOT) CASE(66, J_MAXF32DPP) CASE(67, J_MAXF32DPPB)
        CASE(68, J_ADDF32DPP) CASE(69, J_MAXI16DPP) CASE(70, J_ASHRDPP) CASE(71, J_MED3I) CASE(72, J_MED3F) CASE(73, J_MAX3F) CASE(74, J_ADD3)
        CASE(75, J_MAXF32E64) CASE(76, J_PKMAXF16) CASE(77, J_PKADDF16) CASE(78, J_ADDF32S) CASE(79, J_ADDF32K) CASE(80, J_CNDMASK2)
        CASE(88, K_CNDE64) CASE(89, K_CNDMIX) CASE(90, K_CND0) CASE(91, K_CNDEXEC)
        CASE(92, L_3PK_AND) CASE(93, L_3PK_BFI) CASE(94, L_4PK) CASE(95, L_SHR_BFI) CASE(96, L_SHR_AND_AND_OR) CASE(97, L_PK_FAST_PK_FAST) CASE(98, L_2PK_2FAST)
        CASE(99, L_MAXI16_PAIR) CASE(100, L_PERM) CASE(101, L_SHL16_OR)
        CASE(81, J_MAXI16S) CASE(82, J_SUBREV) CASE(83, J_LSHLADD) CASE(84, J_ADDLSHL) CASE(85, J_SADU16) CASE(86, J_MIX2) CASE(87, J_MIX3)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

static int* g_out; static unsigned long long* g_ticks;
template <int KIND>
static void run(const char* name, FILE* md) {
    const int iters = 4000;
    fprintf(md, "| `%s` |", name);
    printf("%-34s", name);
    for (int w : {1, 2, 4, 8}) {
        const int grid = 256 * w;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<KIND><<<grid, 256>>>(g_out, g_ticks, 10, 3);
        hipEventRecord(e0); k<KIND><<<grid, 256>>>(g_out, g_ticks, iters, 3); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> t(grid);
        hipMemcpy(t.data(), g_ticks, grid * 8, hipMemcpyDeviceToHost);
        double mean = 0; for (auto x : t) mean += (double)x; mean /= grid;
        const double groups = (double)iters * 64;            // instruction groups per wave
        // per SIMD: w waves each issue `groups`; cycles per group per SIMD = wave ticks / (groups x w)
        const double cyc_wall = ms * 1e-3 * 2.4e9 / (groups * w);
        const double cyc_tick = mean / (groups * w);
        printf("  w=%d %5.2f/%5.2f", w, cyc_wall, cyc_tick);
        fprintf(md, " %.2f / %.2f |", cyc_wall, cyc_tick);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    printf("\n"); fprintf(md, "\n"); fflush(stdout); fflush(md);
}

int main(int argc, char** argv) {
    hipMalloc(&g_out, 256 * 8 * 256 * 4); hipMalloc(&g_ticks, 256 * 8 * 8);
    FILE* md = fopen(argc > 1 ? argv[1] : "/dev/null", "w");
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    fprintf(md, "# gfx950 instruction issue cost (tools/dev/valu_rate.hip on %s, %d CUs, clockRate %d kHz)\n\n"
                "SIMD cycles per instruction group per SIMD with W waves resident per SIMD (256 CUs x 4 SIMDs x W waves, 8 independent\n"
                "accumulators per wave, 64 groups per loop iteration). First number: from the HIP-event wall time at the nominal 2.4 GHz;\n"
                "second: from each wave's own s_memtime delta (shader clock). A group is ONE instruction unless the name says otherwise.\n\n"
                "| instruction | W=1 | W=2 | W=4 | W=8 |\n|---|---|---|---|---|\n", p.name, p.multiProcessorCount, p.clockRate);
#define RUN(N, NAME) if (N >= first) run<N>(NAME, md);
    const int first = argc > 2 ? atoi(argv[2]) : 0;
    if (first == 2) {   // round 4 (profiles/r04_valu_rate.md)
        RUN(94, "group: 4 packed (add, max, add, max)") RUN(93, "group: 3 packed + v_bfi_b32") RUN(92, "group: 3 packed + v_and_b32 (VOP2, vgpr)")
        RUN(98, "group: 2 packed, then v_and_b32 + v_lshrrev_b32") RUN(97, "group: packed, v_and_b32, packed, v_lshrrev_b32 (interleaved)")
        RUN(95, "group: v_lshrrev_b32 + v_bfi_b32 (one insert of the trace packing)") RUN(96, "group: v_lshrrev_b32 + 2 v_and_b32 + v_or_b32 (the same insert in VOP2 forms)")
        RUN(1, "v_pk_max_i16") RUN(99, "group: v_max_i16 + v_max_i16_sdwa WORD_1 (the same maximum in two halves)")
        RUN(100, "v_perm_b32") RUN(101, "group: v_lshlrev_b16 + v_or_b32")
        fclose(md); return 0;
    }
    if (first) { RUN(24, "v_cndmask_b32 vcc") RUN(88, "v_cndmask_b32_e64 sgpr pair") RUN(89, "group: v_cndmask vcc + 3 v_pk_add_i16") RUN(90, "v_cndmask_b32_e64 sgpr pair = 0") RUN(91, "v_cndmask_b32_e64 exec") RUN(16, "v_bfi_b32") fclose(md); return 0; }
    RUN(0, "v_pk_add_i16 clamp") RUN(1, "v_pk_max_i16") RUN(2, "v_pk_max_i16 op_sel_hi") RUN(3, "v_pk_sub_u16") RUN(4, "v_pk_min_u16 inline const")
    RUN(5, "v_pk_mad_u16") RUN(6, "v_pk_ashrrev_i16") RUN(7, "v_add_u32") RUN(8, "v_max_i32") RUN(9, "v_max3_i32") RUN(10, "v_max_i32_dpp row_shr:1")
    RUN(11, "v_max_i32_dpp row_bcast:15") RUN(12, "v_mov_b32_dpp wave_shr:1") RUN(13, "v_add_u32_dpp wave_shr:1") RUN(14, "v_alignbit_b32")
    RUN(15, "v_perm_b32") RUN(16, "v_bfi_b32") RUN(17, "v_and_b32") RUN(18, "v_lshl_or_b32") RUN(19, "v_and_or_b32") RUN(20, "v_sub_u32_sdwa sext WORD_1")
    RUN(21, "v_max_i16_sdwa WORD_1 preserve") RUN(22, "v_cmp_eq_u16_sdwa -> sgpr pair") RUN(23, "v_cmp_lt_i32 -> vcc") RUN(24, "v_cndmask_b32")
    RUN(25, "v_fma_f32") RUN(26, "v_add_f32") RUN(27, "v_max_i16 (VOP2)") RUN(28, "v_add_i16 clamp (VOP3)") RUN(29, "v_mad_i32_i16") RUN(30, "v_dot2_i32_i16")
    RUN(31, "v_readlane_b32") RUN(32, "v_writelane_b32") RUN(33, "v_ashrrev_i32") RUN(34, "s_add_u32 (SALU only)")
    RUN(35, "group: v_pk_add_i16 + 1 SALU") RUN(36, "group: v_pk_add_i16 + 2 SALU") RUN(37, "group: v_pk_max_i16 + s_nop 1")
    RUN(38, "v_pk_add_i16 dependent chain")
    RUN(40, "v_add_u32 (vgpr)") RUN(41, "v_sub_u32") RUN(42, "v_min_i32") RUN(43, "v_max_u32") RUN(44, "v_lshlrev_b32") RUN(45, "v_lshrrev_b32")
    RUN(46, "v_and_b32 (vgpr)") RUN(47, "v_or_b32") RUN(48, "v_xor_b32") RUN(49, "v_max_f32") RUN(50, "v_min_f32") RUN(51, "v_mul_f32") RUN(52, "v_sub_f32")
    RUN(53, "v_fmac_f32") RUN(54, "v_add_u16") RUN(55, "v_sub_u16") RUN(56, "v_max_u16") RUN(57, "v_min_i16") RUN(58, "v_lshlrev_b16") RUN(59, "v_ashrrev_i16")
    RUN(60, "v_mul_lo_u16") RUN(61, "v_mul_i32_i24") RUN(62, "v_mov_b32") RUN(63, "v_cvt_f32_i32") RUN(64, "v_cvt_i32_f32") RUN(65, "v_not_b32")
    RUN(66, "v_max_f32_dpp row_shr:1") RUN(67, "v_max_f32_dpp row_bcast:15") RUN(68, "v_add_f32_dpp wave_shr:1") RUN(69, "v_max_i16_dpp row_shr:1")
    RUN(70, "v_ashrrev_i32_dpp row_shr:1") RUN(71, "v_med3_i32") RUN(72, "v_med3_f32") RUN(73, "v_max3_f32") RUN(74, "v_add3_u32") RUN(75, "v_max_f32_e64")
    RUN(76, "v_pk_max_f16") RUN(77, "v_pk_add_f16") RUN(78, "v_add_f32 sgpr src") RUN(79, "v_add_f32 literal") RUN(80, "v_cndmask_b32 (vcc set)")
    RUN(81, "v_max_i16 sgpr src") RUN(82, "v_subrev_u32") RUN(83, "v_lshl_add_u32") RUN(84, "v_add_lshl_u32") RUN(85, "v_sad_u16")
    RUN(86, "group: v_pk_add_i16 + v_max_i16") RUN(87, "group: v_add_f32 + v_max_i16")
    fclose(md);
    return 0;
}
