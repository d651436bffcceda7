// Standalone check of the cross-lane primitives used by ba_device.hpp on real gfx950 hardware.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../block_aligner_amd/csrc/ba_device.hpp"
__global__ void k(const int* in, int* out) {
    int v = in[threadIdx.x];
    out[threadIdx.x] = ba::wave_prefix_max(v);
    out[64 + threadIdx.x] = ba::wave_shr1(v, -7);
    out[128 + threadIdx.x] = ba::wave_max(v);
    out[192 + threadIdx.x] = __builtin_amdgcn_readlane(v, 5);
}
int main() {
    int h[64], o[256], *di, *dout;
    srand(1);
    for (int i = 0; i < 64; i++) h[i] = rand() % 1000 - 500;
    hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof o);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    k<<<1, 64>>>(di, dout);
    hipError_t e = hipDeviceSynchronize();
    printf("sync: %s\n", hipGetErrorString(e));
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0, run = -100000, mx = -100000;
    for (int i = 0; i < 64; i++) mx = h[i] > mx ? h[i] : mx;
    for (int i = 0; i < 64; i++) {
        run = h[i] > run ? h[i] : run;
        if (o[i] != run) { bad++; if (bad < 5) printf("prefix_max lane %d got %d want %d\n", i, o[i], run); }
        int want = i ? h[i - 1] : -7;
        if (o[64 + i] != want) { bad++; if (bad < 10) printf("shr1 lane %d got %d want %d\n", i, o[64 + i], want); }
        if (o[128 + i] != mx) { bad++; if (bad < 15) printf("wave_max lane %d got %d want %d\n", i, o[128 + i], mx); }
        if (o[192 + i] != h[5]) bad++;
    }
    printf("dpp_test bad=%d\n", bad);
    return bad != 0;
}
