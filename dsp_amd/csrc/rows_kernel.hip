// Per-row kernels behind the drop-in functions that take signals of any length (no handle, no plan):
//   first_strong_peak(corr, fs, min_delay_samples) for any correlation length and blind zone   airSonar_optimized.py:200-219
//   |np.fft.rfft(x)| / mag2db of it for lengths none of the FFT kernels holds                  airSonar_optimized.py:332-343, :710
// Compiled with -ffp-contract=off: one rounding per multiply and per add, as NumPy evaluates the detector.
#include <hip/hip_runtime.h>
#include <math.h>

#include "echo_internal.h"

namespace echo {

constexpr int kFspNT = 256;

// np.argmax order on (value, index): the first NaN wins over everything, otherwise the larger value, the lower index on ties
__device__ __forceinline__ bool fsp_better(double v, int i, double b, int bi) {
    const bool vn = v != v, bn = b != b;
    if (vn || bn) return vn && (!bn || i < bi);
    return v > b || (v == b && i < bi);
}

// One workgroup per row of corr[rows][len]; pos = corr[half:].  Mirrors the reference line by line:
//   pos.size <= min_delay -> (None, 0.0), corr untouched                                   :206-207
//   pos[:min_delay] = 0 in place                                                            :208
//   noise_floor = median(|pos[min_delay : min_delay + noise_n]|) ** 2 (slice truncated)     :209-210
//   peak_idx = argmax(pos); snr_db = 10 log10(pos[peak]^2 / noise_floor) if noise_floor > 0 else 0; gate   :211-219
// The median is the exact order statistic (mean of the two middle ones for an even count), found by counting.
__global__ __launch_bounds__(kFspNT) void fsp_rows_kernel(double* corr, int len, int min_delay, int noise_n, double gate_db,
                                                          int32_t* peak_idx, double* snr_db) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsp_smem[];
    double* nw = reinterpret_cast<double*>(fsp_smem);
    __shared__ double s_best[kFspNT / 64];
    __shared__ int s_idx[kFspNT / 64];
    __shared__ int s_nan;
    __shared__ double s_med[2];
    const int tid = threadIdx.x, row = blockIdx.x;
    const int half = len / 2, npos = len - half;
    double* pos = corr + (size_t)row * len + half;
    if (npos <= min_delay) {
        if (tid == 0) { peak_idx[row] = -1; snr_db[row] = 0.0; }
        return;
    }
    if (tid == 0) { s_nan = 0; s_med[0] = 0.0; s_med[1] = 0.0; }
    __syncthreads();
    for (int i = tid; i < min_delay; i += kFspNT) pos[i] = 0.0;
    int nn = npos - min_delay < noise_n ? npos - min_delay : noise_n;
    if (nn < 0) nn = 0;
    bool any_nan = false;
    for (int i = tid; i < nn; i += kFspNT) { const double v = fabs(pos[min_delay + i]); nw[i] = v; any_nan |= (v != v); }
    if (any_nan) s_nan = 1;
    // signed maximum; the zeroed blind zone is the value 0 at index 0
    double best = min_delay > 0 ? 0.0 : -INFINITY;
    int bi = min_delay > 0 ? 0 : 0x7fffffff;
    for (int i = min_delay + tid; i < npos; i += kFspNT) { const double v = pos[i]; if (fsp_better(v, i, best, bi)) { best = v; bi = i; } }
    for (int o = 32; o > 0; o >>= 1) {
        const double b2 = __shfl_xor(best, o); const int i2 = __shfl_xor(bi, o);
        if (fsp_better(b2, i2, best, bi)) { best = b2; bi = i2; }
    }
    if ((tid & 63) == 0) { s_best[tid >> 6] = best; s_idx[tid >> 6] = bi; }
    __syncthreads();
    if (!s_nan && nn > 0) {
        const int k0 = (nn - 1) / 2, k1 = nn / 2;
        for (int i = tid; i < nn; i += kFspNT) {
            const double v = nw[i];
            int less = 0, eq = 0;
            for (int j = 0; j < nn; ++j) { const double u = nw[j]; less += (u < v) ? 1 : 0; eq += (u == v) ? 1 : 0; }
            if (less <= k0 && k0 < less + eq) s_med[0] = v;
            if (less <= k1 && k1 < less + eq) s_med[1] = v;
        }
    }
    __syncthreads();
    if (tid != 0) return;
    for (int w = 1; w < kFspNT / 64; ++w)
        if (fsp_better(s_best[w], s_idx[w], best, bi)) { best = s_best[w]; bi = s_idx[w]; }
    double noise_floor;
    if (s_nan || nn == 0) noise_floor = NAN;                   // np.median of an empty slice, or of one holding a NaN
    else { const double med = (s_med[0] + s_med[1]) / 2.0; noise_floor = med * med; }
    const double peak_power = best * best;
    double snr = 0.0;
    if (noise_floor > 0.0) snr = 10.0 * log10(peak_power / noise_floor);
    if (snr < gate_db) { peak_idx[row] = -1; snr_db[row] = 0.0; }
    else               { peak_idx[row] = bi; snr_db[row] = snr; }
}

int launch_fsp_rows(double* corr, int rows, int len, int min_delay, int noise_n, double gate_db, int32_t* peak_idx,
                    double* snr_db, hipStream_t s) {
    if (rows <= 0) return 0;
    const int npos = len - len / 2;
    int nn = npos - min_delay < noise_n ? npos - min_delay : noise_n;
    if (nn < 1) nn = 1;
    const size_t smem = (size_t)nn * sizeof(double);
    if (smem > 64 * 1024) return -2;                           // noise window beyond the default dynamic LDS limit
    for (int r0 = 0; r0 < rows; r0 += 65535) {
        const int nr = rows - r0 < 65535 ? rows - r0 : 65535;
        hipLaunchKernelGGL(fsp_rows_kernel, dim3(nr), dim3(kFspNT), smem, s, corr + (size_t)r0 * len, len, min_delay,
                           noise_n, gate_db, peak_idx + r0, snr_db + r0);
        if (hipGetLastError() != hipSuccess) return -4;
    }
    return 0;
}

// |X[k]| (or its dB), X = np.fft.rfft(x), as direct sums in float64: one thread per bin, x staged through LDS in tiles,
// exp(-2 pi i (k j mod n) / n) from a table of n roots with the index advanced by k per sample.  O(n^2) per row: the
// path of lengths no FFT kernel takes (huge prime factors, odd lengths beyond one workgroup's LDS).
constexpr int kDftNT = 256, kDftTile = 2048;
__global__ __launch_bounds__(kDftNT) void dft_rows_kernel(const double* x, const double2* w, float* spec, int n, int nbins,
                                                          int db) {
    __shared__ double xs[kDftTile];
    const int tid = threadIdx.x, row = blockIdx.y;
    const int k = blockIdx.x * kDftNT + tid;
    const int kk = k < nbins ? k : 0;
    const double* xr = x + (size_t)row * n;
    double re = 0.0, im = 0.0;
    int idx = 0;                                               // (kk * j) mod n
    for (int j0 = 0; j0 < n; j0 += kDftTile) {
        const int m = n - j0 < kDftTile ? n - j0 : kDftTile;
        __syncthreads();
        for (int i = tid; i < m; i += kDftNT) xs[i] = xr[j0 + i];
        __syncthreads();
        for (int j = 0; j < m; ++j) {
            const double2 t = w[idx];
            const double v = xs[j];
            re = fma(v, t.x, re); im = fma(v, t.y, im);
            idx += kk; if (idx >= n) idx -= n;
        }
    }
    if (k >= nbins) return;
    double mag = sqrt(re * re + im * im);
    if (db) mag = 20.0 * log10(fmax(mag, 1e-12));
    spec[(size_t)row * nbins + k] = (float)mag;
}

int launch_dft_rows(const double* x, const void* roots, float* spec, int db, int rows, int n, hipStream_t s) {
    if (rows <= 0) return 0;
    const int nbins = n / 2 + 1;
    for (int r0 = 0; r0 < rows; r0 += 65535) {
        const int nr = rows - r0 < 65535 ? rows - r0 : 65535;
        hipLaunchKernelGGL(dft_rows_kernel, dim3((nbins + kDftNT - 1) / kDftNT, nr), dim3(kDftNT), 0, s,
                           x + (size_t)r0 * n, static_cast<const double2*>(roots), spec + (size_t)r0 * nbins, n, nbins, db);
        if (hipGetLastError() != hipSuccess) return -4;
    }
    return 0;
}

}  // namespace echo
