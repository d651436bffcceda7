// Microbenchmark: issue rate of the VALU instructions the fill uses (gfx950). Many waves, independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void __launch_bounds__(256) k(int* out, int iters, int s) {
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { REP8(asm volatile("v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %1, %1, %8 clamp\n v_pk_add_i16 %2, %2, %8 clamp\n v_pk_add_i16 %3, %3, %8 clamp\n v_pk_add_i16 %4, %4, %8 clamp\n v_pk_add_i16 %5, %5, %8 clamp\n v_pk_add_i16 %6, %6, %8 clamp\n v_pk_add_i16 %7, %7, %8 clamp" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));) }
        if (KIND == 1) { REP8(asm volatile("v_pk_max_i16 %0, %0, %8\n v_pk_max_i16 %1, %1, %8\n v_pk_max_i16 %2, %2, %8\n v_pk_max_i16 %3, %3, %8\n v_pk_max_i16 %4, %4, %8\n v_pk_max_i16 %5, %5, %8\n v_pk_max_i16 %6, %6, %8\n v_pk_max_i16 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));) }
        if (KIND == 2) { REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));) }
        if (KIND == 3) { REP8(asm volatile("v_max_i32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %1, %2, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %3, %4, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %5, %6, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %6, %7, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %7, %0, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));) }
        if (KIND == 4) { REP8(asm volatile("v_alignbit_b32 %0, %0, %1, 16\n v_alignbit_b32 %1, %1, %2, 16\n v_alignbit_b32 %2, %2, %3, 16\n v_alignbit_b32 %3, %3, %4, 16\n v_alignbit_b32 %4, %4, %5, 16\n v_alignbit_b32 %5, %5, %6, 16\n v_alignbit_b32 %6, %6, %7, 16\n v_alignbit_b32 %7, %7, %0, 16" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));) }
        if (KIND == 5) { REP8(asm volatile("v_pk_mad_u16 %0, %0, %8, %1\n v_pk_mad_u16 %1, %1, %8, %2\n v_pk_mad_u16 %2, %2, %8, %3\n v_pk_mad_u16 %3, %3, %8, %4\n v_pk_mad_u16 %4, %4, %8, %5\n v_pk_mad_u16 %5, %5, %8, %6\n v_pk_mad_u16 %6, %6, %8, %7\n v_pk_mad_u16 %7, %7, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));) }
        if (KIND == 6) { REP8(asm volatile("v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %0, %0, %8 clamp\n v_pk_add_i16 %0, %0, %8 clamp" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int KIND> void run(const char* name, int blocks_per_cu) {
    int* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 256 * blocks_per_cu;
    k<KIND><<<grid, 256>>>(out, 10, 3);
    hipEventRecord(e0); k<KIND><<<grid, 256>>>(out, iters, 3); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr = (double)iters * 64 * grid * 4;    // wave-instructions
    double per_simd_per_cyc = instr / (ms * 1e-3) / (1024.0 * 2.4e9);
    printf("%-28s waves/SIMD=%d  %.3f ms  %.3f wave-instr/cycle/SIMD (at 2.4 GHz nominal) => %.2f cycles/instr\n", name, blocks_per_cu, ms, per_simd_per_cyc, 1.0 / per_simd_per_cyc);
}
int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_pk_add_i16 clamp", w); run<1>("v_pk_max_i16", w); run<2>("v_add_u32", w); run<3>("v_max_i32_dpp", w);
        run<4>("v_alignbit_b32", w); run<5>("v_pk_mad_u16", w); run<6>("v_pk_add_i16 dependent", w);
    }
    return 0;
}
