// One workgroup per (frame, band): FIR -> correlation + peak statistics -> Bluestein rfft -> dB spectrum,
// with every intermediate resident in LDS.  The algorithm is in band_phases.h (shared with the host
// emulation used by the unit tests); this file only sequences the phases between workgroup barriers.
#include <hip/hip_runtime.h>

#include "echo_internal.h"

namespace echo {

// Diagnostic build only (-DECHO_DIAG_STAMPS, `make diag`): thread 0 of the first blocks records the shader clock
// after every phase into a buffer of its own.  The production library contains none of this.
#ifdef ECHO_DIAG_STAMPS
constexpr int kStampBlocks = 4096, kStampSlots = 24;
__device__ unsigned long long g_stamps[kStampBlocks * kStampSlots];
#define STAMP(k) do { if (tid == 0 && blk < kStampBlocks) g_stamps[blk * kStampSlots + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

__global__ __launch_bounds__(kNT) void band_kernel(BandParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
    STAMP(0);
    ph_load(tid, kNT, blk, p, smem);
    __syncthreads(); STAMP(1);
    double fir_acc[kR];
    ph_fir(tid, kNT, blk, p, smem, fir_acc);
    __syncthreads();                          // every lane has read its FIR inputs: sig may overwrite ybuf
    ph_fir_store(tid, kNT, blk, p, smem, fir_acc);
    __syncthreads(); STAMP(2);
    if (p.do_corr) {
        ph_corr(tid, kNT, blk, p, smem);
        __syncthreads(); STAMP(3);
        ph_reduce(tid, kNT, p, smem);
        ph_rank_count(tid, kNT, p, smem);     // reads the noise window, writes counts beyond the reduction scratch
        __syncthreads(); STAMP(4);
        ph_rank_pick(tid, kNT, p, smem);
        __syncthreads(); STAMP(5);
        ph_finish(tid, kNT, blk, p, smem);
        STAMP(6);
    }
}

// Spectrum: one workgroup per (frame, band); Bluestein rfft of the band signal in a 64-KB LDS buffer, two
// workgroups per CU.  Data-independent table loads are issued one phase ahead of their use, in front of a barrier.
__global__ __launch_bounds__(kNT, 4) void spec_kernel(BandParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
    STAMP(7);
    Tw3 tw = fft_tw_load(tid, 1024, p);
    ph_spec_load(tid, kNT, blk, p, smem);
    __syncthreads(); STAMP(8);
    ph_fft_r8<false>(tid, 1024, p, smem, tw);
    tw = fft_tw_load(tid, 128, p);
    __syncthreads(); STAMP(9);
    ph_fft_r8<false>(tid, 128, p, smem, tw);
    tw = fft_tw_load(tid, 16, p);
    __syncthreads(); STAMP(10);
    ph_fft_r8<false>(tid, 16, p, smem, tw);
    __syncthreads(); STAMP(11);
    {
        const Bh8 bh = fft_bh_load(tid, p);
        ph_fft_mid(tid, p, smem, bh);
    }
    tw = fft_tw_load(tid, 16, p);
    __syncthreads(); STAMP(12);
    ph_fft_r8<true>(tid, 16, p, smem, tw);
    tw = fft_tw_load(tid, 128, p);
    __syncthreads(); STAMP(13);
    ph_fft_r8<true>(tid, 128, p, smem, tw);
    tw = fft_tw_load(tid, 1024, p);
    __syncthreads(); STAMP(14);
    ph_fft_r8<true>(tid, 1024, p, smem, tw);
    __syncthreads(); STAMP(15);
    ph_spec_store(tid, kNT, blk, p, smem);
    STAMP(16);
}

// The 8192-point cyclic convolution of the Bluestein transform, in place in LDS (entered after a barrier, leaves
// after one): forward passes, register-resident middle with the chirp-filter product, inverse passes.
__device__ __forceinline__ void bluestein_round(int tid, const BandParams& p, unsigned char* smem) {
    Tw3 tw = fft_tw_load(tid, 1024, p);
    ph_fft_r8<false>(tid, 1024, p, smem, tw);
    tw = fft_tw_load(tid, 128, p);
    __syncthreads();
    ph_fft_r8<false>(tid, 128, p, smem, tw);
    tw = fft_tw_load(tid, 16, p);
    __syncthreads();
    ph_fft_r8<false>(tid, 16, p, smem, tw);
    __syncthreads();
    {
        const Bh8 bh = fft_bh_load(tid, p);
        ph_fft_mid(tid, p, smem, bh);
    }
    tw = fft_tw_load(tid, 16, p);
    __syncthreads();
    ph_fft_r8<true>(tid, 16, p, smem, tw);
    tw = fft_tw_load(tid, 128, p);
    __syncthreads();
    ph_fft_r8<true>(tid, 128, p, smem, tw);
    tw = fft_tw_load(tid, 1024, p);
    __syncthreads();
    ph_fft_r8<true>(tid, 1024, p, smem, tw);
    __syncthreads();
}

// The same with the 4608-point cyclic convolution (fft9_phases.h), 576 threads.
__device__ __forceinline__ void bluestein9_round(int tid, const BandParams& p, unsigned char* smem) {
    ph9_r9<false>(tid, p, smem);
    __syncthreads();
    ph9_r8<false>(tid, 64, p, smem);
    __syncthreads();
    ph9_r8<false>(tid, 8, p, smem);
    __syncthreads();
    ph9_mid(tid, p, smem);
    __syncthreads();
    ph9_r8<true>(tid, 8, p, smem);
    __syncthreads();
    ph9_r8<true>(tid, 64, p, smem);
    __syncthreads();
    ph9_r9<true>(tid, p, smem);
    __syncthreads();
}

__global__ __launch_bounds__(kF9NT) void spec9_kernel(BandParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
    ph9_load(tid, kF9NT, blk, p, smem);
    __syncthreads();
    bluestein9_round(tid, p, smem);
    ph9_store(tid, kF9NT, blk, p, smem);
}

__global__ __launch_bounds__(kF9NT) void spec9_hilbert_kernel(BandParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
    ph9_load(tid, kF9NT, blk, p, smem);
    __syncthreads();
    bluestein9_round(tid, p, smem);
    ph_hilb_mid<true>(tid, kF9NT, blk, p, smem);
    __syncthreads();
    bluestein9_round(tid, p, smem);
    ph_hilb_env<true>(tid, kF9NT, blk, p, smem);
    __syncthreads();
    ph_hilb_smooth(tid, kF9NT, blk, p, smem);
    __syncthreads();
    ph_hilb_stats1(tid, kF9NT, p, smem);
    __syncthreads();
    ph_hilb_dev(tid, kF9NT, blk, p, smem);
    __syncthreads();
    ph_hilb_finish(tid, kF9NT, blk, p, smem);
}

// Spectrum plus the envelope branch of the band signal (hilb_phases.h): dB spectrum and signal_quality.
__global__ __launch_bounds__(kNT, 4) void spec_hilbert_kernel(BandParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
    ph_spec_load(tid, kNT, blk, p, smem);
    __syncthreads();
    bluestein_round(tid, p, smem);
    ph_hilb_mid<false>(tid, kNT, blk, p, smem);
    __syncthreads();
    bluestein_round(tid, p, smem);
    ph_hilb_env<false>(tid, kNT, blk, p, smem);
    __syncthreads();
    ph_hilb_smooth(tid, kNT, blk, p, smem);
    __syncthreads();
    ph_hilb_stats1(tid, kNT, p, smem);
    __syncthreads();
    ph_hilb_dev(tid, kNT, blk, p, smem);
    __syncthreads();
    ph_hilb_finish(tid, kNT, blk, p, smem);
}

#ifdef ECHO_DIAG_STAMPS
extern "C" int echo_diag_read_stamps(unsigned long long* host, int n) {
    if (n > kStampBlocks * kStampSlots) n = kStampBlocks * kStampSlots;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -4;
}
extern "C" int echo_diag_stamp_shape(int* blocks, int* slots) { *blocks = kStampBlocks; *slots = kStampSlots; return 0; }
#endif

int band_kernel_prepare(int smem_bytes, int spec_smem_bytes, int spec_hilb_smem_bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(band_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(spec_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, spec_smem_bytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(spec9_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, spec_smem_bytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(spec_hilbert_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, spec_hilb_smem_bytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(spec9_hilbert_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, spec_hilb_smem_bytes);
    return e == hipSuccess ? 0 : -4;
}

int launch_spec(const BandParams& p, int B, hipStream_t s) {
    if (B <= 0) return 0;
    if (p.fft9) hipLaunchKernelGGL(spec9_kernel, dim3((unsigned)B * p.nb), dim3(kF9NT), p.spec_smem_bytes, s, p);
    else hipLaunchKernelGGL(spec_kernel, dim3((unsigned)B * p.nb), dim3(kNT), p.spec_smem_bytes, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_spec_hilbert(const BandParams& p, int B, hipStream_t s) {
    if (B <= 0) return 0;
    if (!p.specf_in || !p.sq_out || !p.spec_out || !p.uk || !hilb_fits(p.N, p.n2)) return -1;
    if (p.fft9) hipLaunchKernelGGL(spec9_hilbert_kernel, dim3((unsigned)B * p.nb), dim3(kF9NT), p.spec_hilb_smem_bytes, s, p);
    else hipLaunchKernelGGL(spec_hilbert_kernel, dim3((unsigned)B * p.nb), dim3(kNT), p.spec_hilb_smem_bytes, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_band(const BandParams& p, int B, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(band_kernel, dim3((unsigned)B * p.nb), dim3(kNT), p.smem_bytes, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace echo
