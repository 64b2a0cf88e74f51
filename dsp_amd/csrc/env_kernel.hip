// Kernel of the correlation envelope branch (env_phases.h): one 512-thread workgroup per (frame, band), float64.
// Compiled with -ffp-contract=off so that thresholds (mean + 2*std) round as NumPy's; the FFT inner loops use
// explicit fma().
#include <hip/hip_runtime.h>

#include "../../include/echo_hip.h"
#include "echo_internal.h"
#include "env_phases.h"

namespace echo {

__global__ __launch_bounds__(kEnvNT, 4) void corr_env_kernel(EnvParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
#define ECHO_PHASE(...) { __VA_ARGS__; }
#define ECHO_SYNC __syncthreads();
    ECHO_ENV_SEQUENCE(ECHO_PHASE, ECHO_SYNC, EnvPart1 part1_, part1_, p, smem, blk, kEnvNT);
#undef ECHO_PHASE
#undef ECHO_SYNC
}

// FIR, correlation as a real cyclic convolution, and the envelope branch on it (fce_phases.h): the full chain's band
// stage and envelope kernel in one workgroup per (frame, band).
__global__ __launch_bounds__(kEnvNT, 4) void fce_kernel(EnvParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
#define ECHO_PHASE(...) { __VA_ARGS__; }
#define ECHO_SYNC __syncthreads();
    ECHO_FCE_SEQUENCE(ECHO_PHASE, ECHO_SYNC, EnvPart1 part1_, part1_, FceKeep keep_, keep_, p, smem, blk, kEnvNT);
#undef ECHO_PHASE
#undef ECHO_SYNC
}

// The limit is a property of the kernel, not of a plan: it only ever grows (per device), so that plans with a smaller
// LDS footprint prepared later do not lower it under the larger ones.
int env_kernel_prepare(const EnvParams& p) {
    static int prepared[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ECHO_ERR_HIP;
    if (p.smem_bytes <= prepared[dev]) return ECHO_OK;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(corr_env_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            p.smem_bytes) != hipSuccess) return ECHO_ERR_HIP;
    prepared[dev] = p.smem_bytes;
    return ECHO_OK;
}

int fce_kernel_prepare(const EnvParams& p) {
    static int prepared[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ECHO_ERR_HIP;
    if (p.smem_bytes <= prepared[dev]) return ECHO_OK;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(fce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            p.smem_bytes) != hipSuccess) return ECHO_ERR_HIP;
    prepared[dev] = p.smem_bytes;
    return ECHO_OK;
}

int launch_fce(const EnvParams& p, int B, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(fce_kernel, dim3(B * p.nb), dim3(kEnvNT), p.smem_bytes, s, p);
    return hipGetLastError() == hipSuccess ? ECHO_OK : ECHO_ERR_HIP;
}

int launch_corr_env(const EnvParams& p, int B, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(corr_env_kernel, dim3(B * p.nb), dim3(kEnvNT), p.smem_bytes, s, p);
    return hipGetLastError() == hipSuccess ? ECHO_OK : ECHO_ERR_HIP;
}

}  // namespace echo
