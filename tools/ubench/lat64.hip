// Micro-benchmark: dependent-chain latency of fp64 add/mul/fma and a 64-bit DPP quad broadcast, one wave. Development aid.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_chain(double* out, int iters, double a, double b, int mode) {
    double x = a + threadIdx.x * 1e-9, y = b;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (mode == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) x = x + y;                     // 16 dependent adds
        } else if (mode == 1) {
#pragma unroll
            for (int k = 0; k < 16; ++k) x = x * y;
        } else if (mode == 2) {
#pragma unroll
            for (int k = 0; k < 16; ++k) x = fma(x, y, y);
        } else if (mode == 3) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                int lo = __double2loint(x), hi = __double2hiint(x);
                lo = __builtin_amdgcn_update_dpp(lo, lo, 0x00, 0xf, 0xf, false);
                hi = __builtin_amdgcn_update_dpp(hi, hi, 0x00, 0xf, 0xf, false);
                x = __hiloint2double(hi, lo) + y;                       // dpp bcast + add
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) { x = x + y; y = y * a; }      // two independent chains
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x + y;
    if (threadIdx.x == 0) out[64] = (double)(t1 - t0) / (16.0 * iters);
}
int main() {
    double* d; hipMalloc(&d, 128 * sizeof(double));
    const char* names[] = {"add chain", "mul chain", "fma chain", "dpp+add chain", "add+indep mul"};
    for (int m = 0; m < 5; ++m) {
        k_chain<<<1, 64>>>(d, 1000, 1.0, 1.0000001, m);
        hipDeviceSynchronize();
        double h; hipMemcpy(&h, d + 64, 8, hipMemcpyDeviceToHost);
        printf("%s: %.1f memtime ticks per link\n", names[m], h);
    }
    return 0;
}
