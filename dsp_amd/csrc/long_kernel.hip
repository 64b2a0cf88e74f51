// Kernels of the long-frame path (long_phases.h): tiled FIR, tiled matched filter, median/MAD detector.
// Compiled with -ffp-contract=off: the detector threshold med + 6*mad is a rounded product and a rounded sum, as
// NumPy evaluates it (airSonar.py:111-113); the sliding dot products use explicit fma().
#include <hip/hip_runtime.h>

#include "../../include/echo_hip.h"
#include "echo_internal.h"
#include "long_phases.h"

namespace echo {

__global__ __launch_bounds__(kLFirNT) void long_fir_kernel(LongParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, tile = blockIdx.x, blk = blockIdx.y;
    ph_lfir_load(tid, kLFirNT, tile, blk, p, smem);
    __syncthreads();
    double acc[kR];
    ph_lfir_run(tid, kLFirNT, tile, blk, p, smem, acc);
    __syncthreads();
    ph_lfir_stage(tid, kLFirNT, smem, acc);
    __syncthreads();
    ph_lfir_store(tid, kLFirNT, tile, blk, p, smem);
}

__global__ __launch_bounds__(kLCorrNT) void long_corr_kernel(LongParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, tile = blockIdx.x, blk = blockIdx.y;
    ph_lcorr_load(tid, kLCorrNT, blk, p, smem);
    __syncthreads();
    ph_lcorr_run(tid, kLCorrNT, tile, blk, p, smem);
}

__global__ __launch_bounds__(kLDetNT) void long_detect_kernel(LongParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
#define ECHO_PHASE(...) { __VA_ARGS__; }
#define ECHO_SYNC __syncthreads();
    ECHO_LDET_SEQUENCE(ECHO_PHASE, ECHO_SYNC, p, smem, blk, kLDetNT);
#undef ECHO_PHASE
#undef ECHO_SYNC
}

__global__ __launch_bounds__(kLDetNT) void long_twopeak_kernel(LongParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
    ph_tp_load(tid, kLDetNT, blk, p, smem);
    __syncthreads();
    ph_tp_ref_fold(tid, kLDetNT, p, smem);
    __syncthreads();
    ph_tp_ref(tid, kLDetNT, p, smem);
    __syncthreads();
    ph_tp_second(tid, kLDetNT, p, smem);
    __syncthreads();
    ph_tp_second_fold(tid, kLDetNT, p, smem);
    __syncthreads();
    ph_tp_finish(tid, kLDetNT, blk, p, smem);
}

__global__ __launch_bounds__(kLDetNT) void long_median_kernel(LongParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
#define ECHO_PHASE(...) { __VA_ARGS__; }
#define ECHO_SYNC __syncthreads();
    ECHO_LMED_SEQUENCE(ECHO_PHASE, ECHO_SYNC, p, smem, blk, kLDetNT);
#undef ECHO_PHASE
#undef ECHO_SYNC
}

__global__ __launch_bounds__(kLSpecNT) void long_spec_kernel(LongParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
    ph_lspec_load(tid, kLSpecNT, blk, p, smem);
    __syncthreads();
    int cur = 0;
    for (int ps = 0; ps < p.splan.npass; ++ps) {
        ph_lspec_pass(tid, kLSpecNT, p, smem, ps, cur);
        __syncthreads();
        cur ^= 1;
    }
    ph_lspec_store(tid, kLSpecNT, blk, p, smem, cur);
}

__global__ __launch_bounds__(kLSpecNT) void long_spec_odd_kernel(LongParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
#define ECHO_PHASE(...) { __VA_ARGS__; }
#define ECHO_SYNC __syncthreads();
    ECHO_LSPEC_ODD_SEQUENCE(ECHO_PHASE, ECHO_SYNC, p, smem, blk, kLSpecNT);
#undef ECHO_PHASE
#undef ECHO_SYNC
}

static inline int ok() { return hipGetLastError() == hipSuccess ? 0 : ECHO_ERR_HIP; }

int long_kernels_prepare(const LongParams& p) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(long_fir_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            p.fir_smem) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(long_corr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            p.corr_smem) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(long_detect_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            p.det_smem) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(long_twopeak_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            p.tp_smem) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(long_median_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            p.lm_smem) != hipSuccess ||
        (p.has_spec && hipFuncSetAttribute(reinterpret_cast<const void*>(p.spec_odd ? long_spec_odd_kernel : long_spec_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, p.spec_smem) != hipSuccess))
        return ECHO_ERR_HIP;
    return ECHO_OK;
}

// grid.y carries the (frame, band) index and is limited to 65535: callers pass at most that many blocks per launch
int launch_long_fir(const LongParams& p, int B, hipStream_t s) {
    if (B <= 0) return 0;
    const int nblk = B * p.nb;
    if (nblk > 65535) return ECHO_ERR_INVALID;
    const int tiles = (p.N + kLFirNT * kR - 1) / (kLFirNT * kR);
    hipLaunchKernelGGL(long_fir_kernel, dim3(tiles, nblk), dim3(kLFirNT), p.fir_smem, s, p);
    return ok();
}

int launch_long_corr(const LongParams& p, int B, hipStream_t s) {
    if (B <= 0) return 0;
    const int nblk = B * p.nb;
    if (nblk > 65535 || p.k_begin < 0 || p.k_begin >= p.L) return ECHO_ERR_INVALID;
    if (p.cf_use && p.cf_g) return launch_long_corr_fft(p, B, s);      // long chirp: overlap-save FFT
    const int tiles = (p.L - p.k_begin + kLCorrNT * kR - 1) / (kLCorrNT * kR);
    hipLaunchKernelGGL(long_corr_kernel, dim3(tiles, nblk), dim3(kLCorrNT), p.corr_smem, s, p);
    return ok();
}

int launch_long_spec(const LongParams& p, int B, hipStream_t s) {
    if (B <= 0) return 0;
    if (!p.has_spec || !p.spec_in || !p.spec_out) return ECHO_ERR_UNSUPPORTED;
    if (p.spec_odd) {
        if (!p.sbw || !p.sbhat || p.sb_m < p.N + p.nbins - 1) return ECHO_ERR_INVALID;
        hipLaunchKernelGGL(long_spec_odd_kernel, dim3(B * p.nb), dim3(kLSpecNT), p.spec_smem, s, p);
        return ok();
    }
    hipLaunchKernelGGL(long_spec_kernel, dim3(B * p.nb), dim3(kLSpecNT), p.spec_smem, s, p);
    return ok();
}

int launch_long_detect(const LongParams& p, int B, hipStream_t s) {
    if (B <= 0) return 0;
    if (p.detector == ECHO_DET_TWOPEAK) {
        hipLaunchKernelGGL(long_twopeak_kernel, dim3(B * p.nb), dim3(kLDetNT), p.tp_smem, s, p);
        return ok();
    }
    if (p.detector == ECHO_DET_MEDIAN || p.detector == ECHO_DET_MEANPOW) {
        hipLaunchKernelGGL(long_median_kernel, dim3(B * p.nb), dim3(kLDetNT), p.lm_smem, s, p);
        return ok();
    }
    hipLaunchKernelGGL(long_detect_kernel, dim3(B * p.nb), dim3(kLDetNT), p.det_smem, s, p);
    return ok();
}

}  // namespace echo
