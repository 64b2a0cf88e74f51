// Test-only library: the wavefront-level primitives of the kernels (DPP folds, lane reads, ballots, combined LDS
// atomics), each run alone on caller-supplied lane values so that a GPU test can compare them with their definition
// evaluated on the host.  The host emulation of the phases (tests/host_emul) has no wavefronts and cannot cover them.
// Built by tests/gpu_unit/Makefile, loaded by tests/test_gpu_wave_units.py only; nothing of the product links it.
#include <hip/hip_runtime.h>
#include "../../dsp_amd/csrc/long_phases.h"

namespace {

__global__ void k_fold(const double* in, double* out, int op) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    out[i] = echo::wave_fold(op, in[i]);
}

__global__ void k_max_int(const int* in, int* out) {
#if defined(__HIP_DEVICE_COMPILE__)                 // the primitive exists in the device pass only
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    out[i] = wave_max_int(in[i]);
#endif
}

// the argmax butterfly of the band kernel's detector partials: larger value wins, equal values keep the lower index
__global__ void k_argmax(const double* val, const int* idx, double* best_out, int* idx_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double best = val[i];
    int bidx = idx[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double b2 = __shfl_xor(best, o);
        const int i2 = __shfl_xor(bidx, o);
        echo::combine(best, bidx, b2, i2);
    }
    best_out[i] = best;
    idx_out[i] = bidx;
}

// rounds of 256 elements: hist[digit] += 1 for the active ones, *count += number of flagged ones
__global__ void k_hist(const int* digit, const int* active, const int* flag, int rounds, int* hist_out, int* count_out) {
    __shared__ int hist[256];
    __shared__ int count;
    const int tid = threadIdx.x;
    hist[tid] = 0;
    if (tid == 0) count = 0;
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        const int e = (blockIdx.x * rounds + r) * 256 + tid;
        ECHO_HIST_ADD(hist, digit[e], active[e] != 0);
        ECHO_WAVE_COUNT_ADD(&count, flag[e] != 0);
    }
    __syncthreads();
    hist_out[blockIdx.x * 256 + tid] = hist[tid];
    if (tid == 0) count_out[blockIdx.x] = count;
}

// inclusive scan over the lanes of a wavefront as the detector's bin search does it (ph_sel_pick, ph_fs_scan_b)
__global__ void k_scan(const int* in, int* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63;
    int inc = in[i];
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
    out[i] = inc;
}

int done(hipStream_t s) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    return (int)e;
}

}  // namespace

// n must be a multiple of 256 in every call; pointers are device pointers; returns a hipError_t
extern "C" int wave_unit_fold(const double* in, double* out, int n, int op, void* stream) {
    if (n <= 0 || n % 256 || op < 0 || op > 2) return -1;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_fold, dim3(n / 256), dim3(256), 0, s, in, out, op);
    return done(s);
}
extern "C" int wave_unit_max_int(const int* in, int* out, int n, void* stream) {
    if (n <= 0 || n % 256) return -1;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_max_int, dim3(n / 256), dim3(256), 0, s, in, out);
    return done(s);
}
extern "C" int wave_unit_argmax(const double* val, const int* idx, double* best_out, int* idx_out, int n, void* stream) {
    if (n <= 0 || n % 256) return -1;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_argmax, dim3(n / 256), dim3(256), 0, s, val, idx, best_out, idx_out);
    return done(s);
}
// blocks workgroups of 256 threads, each over rounds * 256 consecutive elements; digits must lie in 0..255;
// hist_out [blocks][256], count_out [blocks]
extern "C" int wave_unit_hist(const int* digit, const int* active, const int* flag, int blocks, int rounds,
                              int* hist_out, int* count_out, void* stream) {
    if (blocks <= 0 || rounds <= 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_hist, dim3(blocks), dim3(256), 0, s, digit, active, flag, rounds, hist_out, count_out);
    return done(s);
}
extern "C" int wave_unit_scan(const int* in, int* out, int n, void* stream) {
    if (n <= 0 || n % 256) return -1;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_scan, dim3(n / 256), dim3(256), 0, s, in, out);
    return done(s);
}
