// On-device synthesis of rx frames: a delayed, scaled copy of the transmitted chirp mix (optionally a second, weaker
// one) in Gaussian noise, rounded to the int16 grid the way AudioIO.record delivers samples
// (np.frombuffer(raw, np.int16).astype(np.float32) / 2**15, airSonar_optimized.py:573).  Frame g is a pure function
// of (seed, g): Philox4x32-10 with key (seed, seed >> 32) and counter (sample block, g, stream id, 0), so any rank can
// generate any shard.  Same recipe as the host generator (range U[1.5, 14] m, amplitude U[0.05, 0.4], noise sigma
// U[0.002, 0.02], a second echo at half amplitude in 10 % of the frames) with its own random stream: Box-Muller
// normals in float32.  dsp_amd/synth.py holds the NumPy mirror of the parameter draw.
#include <hip/hip_runtime.h>

#include "../../include/echo_hip.h"
#include "echo_internal.h"

namespace echo {

struct Philox { uint32_t v[4]; };

__device__ __forceinline__ Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    Philox o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}
// uniform in (0, 1]: never 0, so its logarithm is finite
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f); }

struct SynthParams {
    float* rx;             // [B][N]
    const float* tx_mix;   // [M] transmitted mix / 32768
    int B, N, M, fs;
    long long first;       // global index of frame 0 of this call
    unsigned long long seed;
    float c_sound;
};

__global__ __launch_bounds__(256) void synth_kernel(SynthParams p) {
    const int f = blockIdx.x;
    const uint32_t g_lo = (uint32_t)(p.first + f), g_hi = (uint32_t)((unsigned long long)(p.first + f) >> 32);
    const uint32_t k0 = (uint32_t)p.seed, k1 = (uint32_t)(p.seed >> 32);
    // per-frame parameters: stream id 1, block 0
    const Philox q = philox4x32_10(0u, g_lo, 1u, g_hi, k0, k1);
    const Philox q2 = philox4x32_10(1u, g_lo, 1u, g_hi, k0, k1);
    const float r1 = 1.5f + 12.5f * u01(q.v[0]);
    const float amp = 0.05f + 0.35f * u01(q.v[1]);
    const float sigma = 0.002f + 0.018f * u01(q.v[2]);
    const bool second = u01(q.v[3]) < 0.10f;
    const float r2 = 1.5f + 12.5f * u01(q2.v[0]);
    const int d1 = (int)rintf(2.0f * r1 / p.c_sound * (float)p.fs);
    const int d2 = (int)rintf(2.0f * r2 / p.c_sound * (float)p.fs);
    float* out = p.rx + (size_t)f * p.N;
    // four samples per Philox block: two Box-Muller pairs
    for (int b = threadIdx.x; b * 4 < p.N; b += blockDim.x) {
        const Philox r = philox4x32_10((uint32_t)b, g_lo, 0u, g_hi, k0, k1);
        float z[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float rad = sqrtf(-2.0f * logf(u01(r.v[2 * h])));
            float sn, cs;
            sincosf(6.283185307179586f * u01(r.v[2 * h + 1]), &sn, &cs);
            z[2 * h] = rad * cs; z[2 * h + 1] = rad * sn;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = 4 * b + j;
            if (n >= p.N) break;
            float x = sigma * z[j];
            const int e1 = n - d1, e2 = n - d2;
            if (e1 >= 0 && e1 < p.M) x += amp * p.tx_mix[e1];
            if (second && e2 >= 0 && e2 < p.M) x += 0.5f * amp * p.tx_mix[e2];
            float qv = rintf(x * 32768.0f);
            qv = fminf(fmaxf(qv, -32768.0f), 32767.0f);
            out[n] = qv * (1.0f / 32768.0f);
        }
    }
}

int launch_synth(float* rx, const float* tx_mix, int B, int N, int M, int fs, long long first, unsigned long long seed,
                 float c_sound, hipStream_t s) {
    if (B <= 0) return 0;
    SynthParams p;
    p.rx = rx; p.tx_mix = tx_mix; p.B = B; p.N = N; p.M = M; p.fs = fs; p.first = first; p.seed = seed; p.c_sound = c_sound;
    hipLaunchKernelGGL(synth_kernel, dim3(B), dim3(256), 0, s, p);
    return hipGetLastError() == hipSuccess ? ECHO_OK : ECHO_ERR_HIP;
}

}  // namespace echo
