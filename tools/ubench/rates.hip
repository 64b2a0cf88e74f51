// Micro-benchmark: sustained issue cost (cycles per wave-instruction per SIMD) of a few VALU instructions with
// 1, 2 and 4 waves per SIMD, all CUs busy.  Development aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    double d0 = threadIdx.x, d1 = 1, d2 = 2, d3 = 3, d4 = 4, d5 = 5, d6 = 6, d7 = 7;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0, p5 = p1, p6 = p2, p7 = p3, pc = {1.0001f, 0.9999f};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %2, %2, %1, %1\n v_fma_f32 %3, %3, %1, %1\n v_fma_f32 %4, %4, %1, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4));) }
        if (MODE == 1) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));) }
        if (MODE == 2) { REP16(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d4));) }
        if (MODE == 3) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
        if (MODE == 4) { REP16(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4));) }
        if (MODE == 5) { REP16(asm volatile("v_cvt_f32_f64 %0, %4\n v_cvt_f32_f64 %1, %5\n v_cvt_f32_f64 %2, %6\n v_cvt_f32_f64 %3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(d0), "v"(d1), "v"(d2), "v"(d3));) }
        if (MODE == 6) { REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pc));) }
        if (MODE == 7) { REP16(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d4));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + (float)(d0 + d1 + d2 + d3) + p0.x + p1.y + p2.x + p3.y + a5+a6+a7+(float)(d5+d6+d7)+p4.x+p5.x+p6.x+p7.x;
}
template <int MODE> void run(const char* name) {
    float* d; hipMalloc(&d, 256 * 1024 * sizeof(float));
    for (int threads : {256, 512, 1024}) {
        const int iters = 500;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        k<MODE><<<256, threads>>>(d, 10); hipDeviceSynchronize();
        hipEventRecord(a); k<MODE><<<256, threads>>>(d, iters); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double insts_per_simd = (double)iters * 64 * (threads / 256.0);
        printf("%-14s %d waves/SIMD: %.2f ns per wave-instr per SIMD\n", name, threads / 256, ms * 1e6 / insts_per_simd);
    }
    hipFree(d);
}
int main() {
    run<0>("v_fma_f32"); run<1>("v_pk_fma_f32"); run<2>("v_fma_f64"); run<7>("v_mul/add_f64"); run<3>("v_add_f32"); run<6>("v_pk_add_f32");
    run<4>("v_mov_b32"); run<5>("v_cvt_f32_f64");
    return 0;
}
