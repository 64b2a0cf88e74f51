// Matched filter of long frames by overlap-save FFT in LDS (cfft_phases.h), float64.  Compiled with the default
// floating-point contraction: the transform is compared at 1e-13 of the correlation's maximum, not bit for bit.
#include <hip/hip_runtime.h>

#include "../../include/echo_hip.h"
#include "echo_internal.h"
#include "cfft_phases.h"
#include "rcf_phases.h"
#include "ospec_phases.h"

namespace echo {

__global__ __launch_bounds__(kCfNT) void long_corr_fft_kernel(LongParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, pair = blockIdx.x, blk = blockIdx.y;
#define ECHO_PHASE(...) { __VA_ARGS__; }
#define ECHO_SYNC __syncthreads();
    ECHO_CF_SEQUENCE(ECHO_PHASE, ECHO_SYNC, p, smem, pair, blk, kCfNT);
#undef ECHO_PHASE
#undef ECHO_SYNC
}

// The same transform followed by first_strong_peak of the original parameter set (airSonar.py:104-117) on the lags
// it has just produced: the correlation never leaves LDS.
__global__ __launch_bounds__(kCfNT) void long_corr_det_kernel(LongParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
#define ECHO_PHASE(...) { __VA_ARGS__; }
#define ECHO_SYNC __syncthreads();
    ECHO_CF_DET_SEQUENCE(ECHO_PHASE, ECHO_SYNC, p, smem, blk, kCfNT);
#undef ECHO_PHASE
#undef ECHO_SYNC
}

__global__ __launch_bounds__(kCfNT) void long_corr_tp_kernel(LongParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
#define ECHO_PHASE(...) { __VA_ARGS__; }
#define ECHO_SYNC __syncthreads();
    ECHO_CF_TP_SEQUENCE(ECHO_PHASE, ECHO_SYNC, p, smem, blk, kCfNT);
#undef ECHO_PHASE
#undef ECHO_SYNC
}

#ifdef ECHO_RC_VGPR
#define ECHO_RC_ATTR __attribute__((amdgpu_num_vgpr(ECHO_RC_VGPR)))
#else
#define ECHO_RC_ATTR
#endif

// The single real transform of rcf_phases.h with the detector behind it: the form the chain takes when its plan exists.
__global__ __launch_bounds__(kRcNT) ECHO_RC_ATTR void long_rcorr_det_kernel(LongParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
#define ECHO_PHASE(...) { __VA_ARGS__; }
#define ECHO_SYNC __syncthreads();
    ECHO_RC_DET_SEQUENCE(ECHO_PHASE, ECHO_SYNC, p, smem, blk, kRcNT);
#undef ECHO_PHASE
#undef ECHO_SYNC
}

__global__ __launch_bounds__(kRcNT) ECHO_RC_ATTR void long_rcorr_tp_kernel(LongParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
#define ECHO_PHASE(...) { __VA_ARGS__; }
#define ECHO_SYNC __syncthreads();
    ECHO_RC_TP_SEQUENCE(ECHO_PHASE, ECHO_SYNC, p, smem, blk, kRcNT);
#undef ECHO_PHASE
#undef ECHO_SYNC
}

// The transform alone, one workgroup per block of lags: small enough (LDS, registers) for two workgroups to share a
// compute unit, so that one's barriers and memory waits are covered by the other's arithmetic.
__global__ __launch_bounds__(kRcSplitNT, 4) void long_rcorr_kernel(LongParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, sub = blockIdx.x, blk = blockIdx.y;
#define ECHO_PHASE(...) { __VA_ARGS__; }
#define ECHO_SYNC __syncthreads();
    ECHO_RC_SEQUENCE(ECHO_PHASE, ECHO_SYNC, p, smem, sub, blk, kRcSplitNT, kRcSplitMaxR);
#undef ECHO_PHASE
#undef ECHO_SYNC
}

static int rcf_smem(const LongParams& p) { return p.rc_smem + (kDetExtra > kCfTpExtra ? kDetExtra : kCfTpExtra); }

int rcf_kernel_prepare(const LongParams& p) {
    static_assert(kLDetNT <= kRcNT, "two-peak detector behind the transform");
    if (!p.rc_use) return ECHO_OK;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(long_rcorr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            p.rc_smem) != hipSuccess) return ECHO_ERR_HIP;
    if (p.rc_nblk != 1) return ECHO_OK;
    if (rcf_smem(p) > 160 * 1024) return ECHO_ERR_INVALID;
    const void* ks[] = {reinterpret_cast<const void*>(long_rcorr_det_kernel), reinterpret_cast<const void*>(long_rcorr_tp_kernel)};
    for (const void* k : ks)
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, rcf_smem(p)) != hipSuccess) return ECHO_ERR_HIP;
    return ECHO_OK;
}

static bool rcf_ready(const LongParams& p) {
    return p.rc_use && p.rc_wh && p.rc_wk && p.rc_roots && p.rc_g && p.rc_pos && p.rc_k0 == p.k_begin && p.rc_nblk >= 1 &&
           2 * p.rc_H >= p.rc_V + p.M - 1 && (long)p.rc_nblk * p.rc_V >= p.L - p.k_begin;
}

// true when launch_long_rcorr can replace launch_long_corr for these parameters
// (the stand-alone kernel holds the butterflies up to radix kRcSplitMaxR only)
bool long_rcorr_fits(const LongParams& p) {
    for (int i = 0; i < p.rc_npass; ++i) if (p.rc_radix[i] > kRcSplitMaxR) return false;
    return rcf_ready(p);
}

// true when launch_long_rcorr_det can replace launch_long_corr + launch_long_detect for these parameters
bool long_rcorr_det_fits(const LongParams& p) {
    if (!rcf_ready(p) || p.rc_nblk != 1) return false;
    if (p.detector == ECHO_DET_MAD) return p.L - p.half > p.min_delay && p.k_begin == p.half + p.min_delay;
    return p.detector == ECHO_DET_TWOPEAK && p.k_begin == 0 && p.L - p.half > 0;
}

int launch_long_rcorr_det(const LongParams& q, int B, hipStream_t s) {
    if (B <= 0) return 0;
    LongParams p = q;
    if (!long_rcorr_det_fits(p) || !p.res_out) return ECHO_ERR_INVALID;
    p.det_view = 2; p.det_q0 = p.k_begin - p.half; p.det_lds_off = p.rc_smem;
    if (p.detector == ECHO_DET_TWOPEAK) hipLaunchKernelGGL(long_rcorr_tp_kernel, dim3(B * p.nb), dim3(kRcNT), rcf_smem(p), s, p);
    else hipLaunchKernelGGL(long_rcorr_det_kernel, dim3(B * p.nb), dim3(kRcNT), rcf_smem(p), s, p);
    return hipGetLastError() == hipSuccess ? 0 : ECHO_ERR_HIP;
}

// lags [k_begin, L) to corr_out (the layout of launch_long_corr_fft)
int launch_long_rcorr(const LongParams& p, int B, hipStream_t s) {
    if (B <= 0) return 0;
    if (!long_rcorr_fits(p) || !p.corr_out || (long)B * p.nb > 65535) return ECHO_ERR_INVALID;
    hipLaunchKernelGGL(long_rcorr_kernel, dim3(p.rc_nblk, B * p.nb), dim3(kRcSplitNT), p.rc_smem, s, p);
    return hipGetLastError() == hipSuccess ? 0 : ECHO_ERR_HIP;
}

// Spectrum of an odd-length long frame, chirp-z convolution with the transform in place (ospec_phases.h)
__global__ __launch_bounds__(kOsNT) void long_spec_odd_ip_kernel(LongParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
#define ECHO_PHASE(...) { __VA_ARGS__; }
#define ECHO_SYNC __syncthreads();
    ECHO_OSPEC_SEQUENCE(ECHO_PHASE, ECHO_SYNC, p, smem, blk, kOsNT);
#undef ECHO_PHASE
#undef ECHO_SYNC
}

int ospec_kernel_prepare(const LongParams& p) {
    if (!p.os_use) return ECHO_OK;
    if (p.os_smem > 160 * 1024) return ECHO_ERR_INVALID;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(long_spec_odd_ip_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, p.os_smem) == hipSuccess ? ECHO_OK : ECHO_ERR_HIP;
}

int launch_long_spec_odd_ip(const LongParams& p, int B, hipStream_t s) {
    if (B <= 0) return 0;
    if (!p.os_use || !p.os_tw || !p.os_roots || !p.os_bhat || !p.sbw || !p.spec_in || !p.spec_out ||
        p.os_m < p.N + p.nbins - 1) return ECHO_ERR_INVALID;
    hipLaunchKernelGGL(long_spec_odd_ip_kernel, dim3(B * p.nb), dim3(kOsNT), p.os_smem, s, p);
    return hipGetLastError() == hipSuccess ? 0 : ECHO_ERR_HIP;
}

int cfft_kernel_prepare() {
    static_assert(sizeof(TpMisc) <= 64 && kLDetNT <= kCfNT, "two-peak detector behind the transform");
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(long_corr_tp_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kCfDetSmem) != hipSuccess) return ECHO_ERR_HIP;
    static_assert(kCfDetSmem <= 160 * 1024, "transform and detector scratch exceed the LDS of a compute unit");
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(long_corr_det_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, kCfDetSmem) != hipSuccess) return ECHO_ERR_HIP;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(long_corr_fft_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kCfSmem) == hipSuccess ? ECHO_OK : ECHO_ERR_HIP;
}

// true when launch_long_corr_det can replace launch_long_corr + launch_long_detect for these parameters (MAD and
// two-peak detectors, all wanted lags in one pair of sub-blocks)
bool long_corr_det_fits(const LongParams& p) {
    if (!p.cf_use || !p.cf_g || !p.cf_w2 || !p.cf_tw || cf_pairs(p) != 1) return false;
    if (p.detector == ECHO_DET_MAD) return p.L - p.half > p.min_delay && p.k_begin == p.half + p.min_delay;
    return p.detector == ECHO_DET_TWOPEAK && p.k_begin == 0 && p.L - p.half > 0;
}

int launch_long_corr_det(const LongParams& q, int B, hipStream_t s) {
    if (B <= 0) return 0;
    LongParams p = q;
    if (!long_corr_det_fits(p) || !p.res_out) return ECHO_ERR_INVALID;
    p.det_view = 1; p.det_q0 = p.k_begin - p.half; p.det_lds_off = kCfSmem;
    if (p.detector == ECHO_DET_TWOPEAK) hipLaunchKernelGGL(long_corr_tp_kernel, dim3(B * p.nb), dim3(kCfNT), kCfDetSmem, s, p);
    else hipLaunchKernelGGL(long_corr_det_kernel, dim3(B * p.nb), dim3(kCfNT), kCfDetSmem, s, p);
    return hipGetLastError() == hipSuccess ? 0 : ECHO_ERR_HIP;
}

int launch_long_corr_fft(const LongParams& p, int B, hipStream_t s) {
    if (B <= 0) return 0;
    const int nblk = B * p.nb;
    if (nblk > 65535 || p.k_begin < 0 || p.k_begin >= p.L || !p.cf_use || !p.cf_g || !p.cf_w2 || !p.cf_tw || p.cf_V < 1)
        return ECHO_ERR_INVALID;
    hipLaunchKernelGGL(long_corr_fft_kernel, dim3(cf_pairs(p), nblk), dim3(kCfNT), kCfSmem, s, p);
    return hipGetLastError() == hipSuccess ? 0 : ECHO_ERR_HIP;
}

}  // namespace echo
