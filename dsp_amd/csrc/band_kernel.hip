// One workgroup per (frame, band): FIR -> correlation + peak statistics -> Bluestein rfft -> dB spectrum,
// with every intermediate resident in LDS.  The algorithm is in band_phases.h (shared with the host
// emulation used by the unit tests); this file only sequences the phases between workgroup barriers.
#include <hip/hip_runtime.h>

#include "echo_internal.h"

namespace echo {

__global__ __launch_bounds__(kNT) void band_kernel(BandParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
    ph_load(tid, kNT, blk, p, smem);
    __syncthreads();
    ph_fir(tid, kNT, blk, p, smem);
    __syncthreads();
    if (p.do_corr) {
        ph_corr(tid, kNT, blk, p, smem);
        __syncthreads();
        ph_reduce_a(tid, kNT, p, smem);
        __syncthreads();
        ph_reduce_b(tid, kNT, p, smem);
        ph_rank(tid, kNT, p, smem);           // touches the noise window and two scalars only
        __syncthreads();
        ph_finish(tid, kNT, blk, p, smem);
    }
    if (p.do_spec) {
        __syncthreads();
        ph_spec_load(tid, kNT, p, smem);
        __syncthreads();
        ph_fft_r8<false>(tid, kNT, 1024, p, smem);
        __syncthreads();
        ph_fft_r8<false>(tid, kNT, 128, p, smem);
        __syncthreads();
        ph_fft_r8<false>(tid, kNT, 16, p, smem);
        __syncthreads();
        ph_fft_mid(tid, kNT, p, smem);
        __syncthreads();
        ph_fft_r8<true>(tid, kNT, 16, p, smem);
        __syncthreads();
        ph_fft_r8<true>(tid, kNT, 128, p, smem);
        __syncthreads();
        ph_fft_r8<true>(tid, kNT, 1024, p, smem);
        __syncthreads();
        ph_spec_store(tid, kNT, blk, p, smem);
    }
}

int band_kernel_prepare(int smem_bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(band_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    return e == hipSuccess ? 0 : -4;
}

int launch_band(const BandParams& p, int B, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(band_kernel, dim3((unsigned)B * p.nb), dim3(kNT), p.smem_bytes, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace echo
