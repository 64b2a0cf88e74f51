// Micro-benchmark: issue rate of v_fma_f64 / v_fma_f32 / v_pk_fma_f32 in the register-tiled pattern of the
// sliding dot (9 accumulators, rotating window), no memory traffic.  Development aid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <typename T, int R>
__global__ __launch_bounds__(512) void k_fma(T* out, int iters, T seed) {
    T acc[R], w[R], c[R];
    for (int r = 0; r < R; ++r) { acc[r] = 0; w[r] = seed + threadIdx.x * (T)1e-3 + r; c[r] = seed * (T)0.5 + r; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < R; ++t) {
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = fma(w[(r + t) % R], c[t], acc[r]);
        }
        asm volatile("" ::: "memory");
    }
    T s = 0; for (int r = 0; r < R; ++r) s += acc[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename T, int R> void run(const char* name, int threads, int blocks = 256) {
    T* d; hipMalloc(&d, 256 * 1024 * sizeof(T));
    const int iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_fma<T, R><<<blocks, threads>>>(d, 10, (T)1.0);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k_fma<T, R><<<blocks, threads>>>(d, iters, (T)1.0);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double fmas_per_wave = (double)iters * R * R;
    double waves_per_simd = threads / 64.0 / 4.0;
    double ns_per_fma_simd = ms * 1e6 / (fmas_per_wave * waves_per_simd);
    printf("%s blocks=%d threads=%d: %.3f ms, %.2f ns per wave-FMA per SIMD (= %.2f cycles @2.4GHz), %.1f TFLOP/s\n", name, blocks, threads, ms,
           ns_per_fma_simd, ns_per_fma_simd * 2.4, (double)blocks * threads / 64.0 * 64.0 * fmas_per_wave * 2 / (ms * 1e-3) / 1e12);
    hipFree(d);
}
int main() {
    run<double, 9>("f64 R=9", 512, 256); run<double, 9>("f64 R=9", 256, 256); run<double, 9>("f64 R=9", 64, 256);
    run<double, 9>("f64 R=9", 512, 16); run<double, 9>("f64 R=9", 256, 16); run<double, 9>("f64 R=9", 64, 16);
    run<double, 9>("f64 R=9", 64, 1);
    run<float, 9>("f32 R=9", 512, 256); run<float, 9>("f32 R=9", 512, 16); run<float, 9>("f32 R=9", 64, 1);
    return 0;
}
