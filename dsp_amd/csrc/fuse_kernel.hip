// Per-frame range conversion, band confidence, confidence-weighted fusion and the scalar Kalman filter,
// batched over frames / streams, plus the small utility kernels (argmax, mag2db, record packing).
// Arithmetic is float64 in the reference's operation order; this file is compiled with -ffp-contract=off.
//
// Reference: calculate_band_confidence airSonar_optimized.py:222-243; range :662-663; normalize_confidences
// :245-262; fusion :733-738; ScalarKalman :265-283.  Variants: realtime_sonar.py:336-339, :393;
// airSonar.py:203-205, :225-229.
#include <hip/hip_runtime.h>
#include <math.h>

#include "../../include/echo_hip.h"
#include "echo_internal.h"

namespace echo {

__global__ void fuse_kernel(FuseParams p) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.B) return;
    EchoResult r;
    double d[ECHO_MAX_BANDS], c[ECHO_MAX_BANDS], sn[ECHO_MAX_BANDS];
    int nvalid = 0;
    r.valid_mask = 0;
    for (int j = 0; j < ECHO_MAX_BANDS; ++j) { r.dist[j] = 0.f; r.conf[j] = 0.f; r.snr_db[j] = 0.f; r.peak_idx[j] = -1; }
    for (int j = 0; j < p.nb; ++j) {
        BandRes br = p.res[(size_t)i * p.nb + j];
        double qfactor = 0.0;
        if (p.env) {
            // _process_band_gpu (airSonar_optimized.py:647-671): the envelope peak when one lies at or beyond
            // min_delay (index into corr, SNR from the envelope, amplitude of the unmodified corr), otherwise the
            // first_strong_peak result; confidence scaled by the mean of the two quality scores
            const EnvRes e = p.env[(size_t)i * p.nb + j];
            const double sq = p.sq[(size_t)i * p.nb + j];
            if (e.peak_idx >= 0) { br.peak_idx = e.peak_idx; br.snr_db = e.snr_db; br.amplitude = e.amplitude; }
            qfactor = (sq + e.quality) / 2.0;
            if (p.quality_out) {
                p.quality_out[((size_t)i * p.nb + j) * 2] = sq;
                p.quality_out[((size_t)i * p.nb + j) * 2 + 1] = e.quality;
            }
        }
        // the two-peak detector reports a delay that may be zero or negative; its validity travels in `pad`
        const bool band_valid = p.detector == ECHO_DET_TWOPEAK ? br.pad != 0 : br.peak_idx >= 0;
        r.peak_idx[j] = band_valid ? br.peak_idx : -1;
        if (!band_valid) continue;
        const double delay = (double)(br.peak_idx - p.range_offset);
        const double dist = delay / (double)p.fs * p.sound_speed / 2.0;
        double conf = 0.0;
        if (p.confidence == ECHO_CONF_CLAMPED) {
            const double sw = fmin(br.snr_db / 10.0, 1.0);
            const double aw = fmin(br.amplitude / 0.1, 1.0);
            conf = fmin(sw * 0.5 + aw * 0.3 + p.fw[j] * 0.2, 1.0);
        } else if (p.confidence == ECHO_CONF_RAW) {
            conf = fmin((br.snr_db / 10.0) * 0.5 + (br.amplitude / 0.1) * 0.3 + p.fw[j] * 0.2, 1.0);
        }
        if (p.env) conf = conf * (0.5 + 0.5 * qfactor);
        r.dist[j] = (float)dist; r.conf[j] = (float)conf; r.snr_db[j] = (float)br.snr_db;
        r.valid_mask |= 1u << j;
        d[nvalid] = dist; c[nvalid] = conf; sn[nvalid] = br.snr_db; ++nvalid;
    }
    double fused = 0.0, avg = 0.0;
    if (nvalid > 0) {
        if (p.fusion == ECHO_FUSE_MEAN) {
            double s = 0.0;
            for (int k = 0; k < nvalid; ++k) s += d[k];
            fused = s / (double)nvalid;
        } else if (p.fusion == ECHO_FUSE_SNRNORM) {
            // 666.py:372-385: SNRs scaled so that the largest is 100 (zeros when none is positive), + 1e-9 as weights
            double mx = sn[0];
            for (int k = 1; k < nvalid; ++k) mx = sn[k] > mx ? sn[k] : mx;
            double sw = 0.0, swd = 0.0, sn_sum = 0.0;
            for (int k = 0; k < nvalid; ++k) {
                const double nk = mx > 0.0 ? sn[k] / mx * 100.0 : 0.0;
                const double w = nk + 1e-9;
                swd += d[k] * w; sw += w; sn_sum += nk;
            }
            fused = swd / sw;
            avg = sn_sum / (double)nvalid;
        } else {
            double total = 0.0;
            for (int k = 0; k < nvalid; ++k) total += c[k];
            double w[ECHO_MAX_BANDS], cn[ECHO_MAX_BANDS];
            double sw = 0.0, swd = 0.0, scn = 0.0;
            for (int k = 0; k < nvalid; ++k) {
                cn[k] = total > 0.0 ? (c[k] / total) * 100.0 : 100.0 / (double)nvalid;
                w[k] = cn[k] / 100.0;
                if (p.fusion == ECHO_FUSE_WEIGHTED_EPS) w[k] += 1e-9;
            }
            for (int k = 0; k < nvalid; ++k) { swd += d[k] * w[k]; sw += w[k]; scn += cn[k]; }
            fused = swd / sw;
            avg = scn / (double)nvalid;
        }
    }
    r.fused_dist = (float)fused; r.avg_conf = (float)avg; r.kalman = (float)fused;
    reinterpret_cast<EchoResult*>(p.out)[i] = r;
    if (p.fused) p.fused[i] = fused;
    if (p.valid_any) p.valid_any[i] = nvalid > 0 ? 1 : 0;
}

// One lane per stream, sequential over the stream's T frames (time-major layout: index t*S + s).  Inputs are
// fetched eight steps ahead so the loop is not a chain of dependent global loads.
__global__ void kalman_kernel(KalmanParams p) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= p.S) return;
    double x = p.x[s], pp = p.p[s];
    bool init = p.init[s] != 0;
    EchoResult* rec = reinterpret_cast<EchoResult*>(p.rec);
    constexpr int U = 8;
    for (int t0 = 0; t0 < p.T; t0 += U) {
        double z[U]; uint8_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u < p.T ? t0 + u : p.T - 1;
            const size_t i = (size_t)t * p.S + s;
            z[u] = p.z[i]; v[u] = p.valid[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (t0 + u >= p.T) break;
            const size_t i = (size_t)(t0 + u) * p.S + s;
            if (v[u]) {
                if (!init) { x = z[u]; init = true; }
                else {
                    pp += p.q;
                    const double k = pp / (pp + p.r);
                    x += k * (z[u] - x);
                    pp *= (1 - k);
                }
            }
            const double o = init ? x : 0.0;
            if (p.out) p.out[i] = o;
            if (rec) rec[i].kalman = (float)o;
        }
    }
    p.x[s] = x; p.p[s] = pp; p.init[s] = init ? 1 : 0;
}

__global__ void pack_res_kernel(const int32_t* peak, const double* snr, const double* amp, BandRes* res, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BandRes r; r.peak_idx = peak[i]; r.snr_db = snr[i]; r.amplitude = amp ? amp[i] : 0.0; r.pad = 0;
    res[i] = r;
}

__global__ void unpack_res_kernel(const BandRes* res, int32_t* peak, double* snr, double* amp, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BandRes r = res[i];
    if (peak) peak[i] = r.peak_idx;
    if (snr) snr[i] = r.snr_db;
    if (amp) amp[i] = r.amplitude;
}

__global__ void unpack_env_kernel(const EnvRes* res, int32_t* peak, double* snr, double* quality, int32_t* npeaks, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const EnvRes r = res[i];
    if (peak) peak[i] = r.peak_idx;
    if (snr) snr[i] = r.snr_db;
    if (quality) quality[i] = r.quality;
    if (npeaks) npeaks[i] = r.n_peaks;
}

// gpu_argmax: first index of the row maximum (np.argmax), one wavefront per row
__global__ __launch_bounds__(256) void argmax_kernel(const double* x, int32_t* out, int rows, int n) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= rows) return;
    const double* r = x + (size_t)wave * n;
    double best = -INFINITY; int bi = 0x7fffffff;
    for (int i = lane; i < n; i += 64) { const double v = r[i]; if (v > best) { best = v; bi = i; } }
    for (int o = 32; o > 0; o >>= 1) {
        const double b2 = __shfl_down(best, o); const int i2 = __shfl_down(bi, o);
        if (b2 > best || (b2 == best && i2 < bi)) { best = b2; bi = i2; }
    }
    if (lane == 0) out[wave] = bi == 0x7fffffff ? 0 : bi;
}

__global__ void mag2db_kernel(const float* x, float* out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = 20.0f * log10f(fmaxf(fabsf(x[i]), 1e-12f));
}

// calculate_band_confidence (airSonar_optimized.py:222-243; realtime_sonar.py:393 for the unclamped variant),
// elementwise with an explicit band index per element
__global__ void band_confidence_kernel(const double* snr, const double* amp, const int32_t* band, double* out, int n,
                                       int kind, double fw0, double fw1, double fw2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = band[i];
    const double fw = j == 0 ? fw0 : (j == 1 ? fw1 : fw2);
    double c;
    if (kind == ECHO_CONF_CLAMPED) c = fmin(fmin(snr[i] / 10.0, 1.0) * 0.5 + fmin(amp[i] / 0.1, 1.0) * 0.3 + fw * 0.2, 1.0);
    else if (kind == ECHO_CONF_RAW) c = fmin((snr[i] / 10.0) * 0.5 + (amp[i] / 0.1) * 0.3 + fw * 0.2, 1.0);
    else c = 0.0;
    out[i] = c;
}

// normalize_confidences (airSonar_optimized.py:245-262): each row of n values scaled to sum to 100 (np.sum adds a
// short row left to right), or 100 / n each when the sum is not positive; one lane per row
__global__ void normalize_conf_kernel(const double* conf, double* out, int rows, int n) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const double* c = conf + (size_t)r * n;
    double total = 0.0;
    for (int k = 0; k < n; ++k) total += c[k];
    for (int k = 0; k < n; ++k) out[(size_t)r * n + k] = total > 0.0 ? (c[k] / total) * 100.0 : 100.0 / (double)n;
}

// filtfilt with the identity filter b = [b0], a = [1] that design_filters falls back to (airSonar_optimized.py:178-182):
// lfilter of order zero forward and backward, y = (x * b0) * b0
__global__ void iir_order0_kernel(const float* rx, double* y, const double* b, int B, int N, int nb) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * nb * N;
    if (i >= total) return;
    const int n = (int)(i % N); const size_t fb = i / N;
    const int band = (int)(fb % nb); const size_t f = fb / nb;
    const double b0 = b[band];
    y[i] = ((double)rx[f * N + n] * b0) * b0;
}

// gpu_correlate for arbitrary operands (airSonar_optimized.py:290-303): out[row][k] = sum_j apad[k + j] * b[j],
// apad = a[row] with m - 1 zeros on either side, k < n + m - 1.  256 lanes x kR consecutive lags per workgroup, the
// window of a and the whole of b staged in LDS, the same sliding dot (and summation order) as the band kernel.
constexpr int kXcNT = 256;
__global__ __launch_bounds__(kXcNT) void xcorr_kernel(const double* a, const double* b, double* out, int n, int m,
                                                      int mgroups) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int L = n + m - 1, T = kXcNT * kR;
    const int wlen = T + mgroups * kR, clen = (mgroups + 1) * kR;
    double* win = reinterpret_cast<double*>(smem);
    double* cb = win + wlen;
    const int k0 = blockIdx.x * T;
    const double* arow = a + (size_t)blockIdx.y * n;
    for (int i = threadIdx.x; i < wlen; i += kXcNT) {
        const int src = k0 + i - (m - 1);
        win[i] = (src >= 0 && src < n) ? arow[src] : 0.0;
    }
    for (int i = threadIdx.x; i < clen; i += kXcNT) cb[i] = i < m ? b[i] : 0.0;
    __syncthreads();
    const int kk = k0 + threadIdx.x * kR;
    if (kk >= L) return;
    double acc[kR];
    sliding_dot(win + threadIdx.x * kR, cb, mgroups, acc);
#pragma unroll
    for (int r = 0; r < kR; ++r)
        if (kk + r < L) out[(size_t)blockIdx.y * L + kk + r] = acc[r];
}

static inline int ok() { return hipGetLastError() == hipSuccess ? 0 : -4; }

int launch_band_confidence(const double* snr, const double* amp, const int32_t* band, double* out, int n, int kind,
                           const double* fw, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(band_confidence_kernel, dim3((n + 255) / 256), dim3(256), 0, s, snr, amp, band, out, n, kind,
                       fw[0], fw[1], fw[2]);
    return ok();
}
int launch_normalize_conf(const double* conf, double* out, int rows, int n, hipStream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(normalize_conf_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, conf, out, rows, n);
    return ok();
}
int launch_iir_order0(const float* rx, double* y, const double* b, int B, int N, int nb, hipStream_t s) {
    const size_t total = (size_t)B * nb * N;
    if (total == 0) return 0;
    hipLaunchKernelGGL(iir_order0_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, rx, y, b, B, N, nb);
    return ok();
}
int launch_xcorr(const double* a, const double* b, double* out, int rows, int n, int m, hipStream_t s) {
    if (rows <= 0) return 0;
    const int mgroups = (m + kR - 1) / kR, T = kXcNT * kR, L = n + m - 1;
    const size_t smem = (size_t)(T + mgroups * kR + (mgroups + 1) * kR) * sizeof(double);
    if (smem > 160 * 1024 || rows > 65535) return -2;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(xcorr_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem) != hipSuccess) return -4;
    hipLaunchKernelGGL(xcorr_kernel, dim3((L + T - 1) / T, rows), dim3(kXcNT), smem, s, a, b, out, n, m, mgroups);
    return ok();
}

int launch_fuse(const FuseParams& p, hipStream_t s) {
    if (p.B <= 0) return 0;
    hipLaunchKernelGGL(fuse_kernel, dim3((p.B + 255) / 256), dim3(256), 0, s, p);
    return ok();
}
int launch_kalman(const KalmanParams& p, hipStream_t s) {
    if (p.S <= 0 || p.T <= 0) return 0;
    hipLaunchKernelGGL(kalman_kernel, dim3((p.S + 63) / 64), dim3(64), 0, s, p);
    return ok();
}
int launch_pack_res(const int32_t* peak, const double* snr, const double* amp, BandRes* res, int n, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pack_res_kernel, dim3((n + 255) / 256), dim3(256), 0, s, peak, snr, amp, res, n);
    return ok();
}
int launch_unpack_res(const BandRes* res, int32_t* peak, double* snr, double* amp, int n, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(unpack_res_kernel, dim3((n + 255) / 256), dim3(256), 0, s, res, peak, snr, amp, n);
    return ok();
}
int launch_unpack_env(const EnvRes* res, int32_t* peak, double* snr, double* quality, int32_t* npeaks, int n,
                      hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(unpack_env_kernel, dim3((n + 255) / 256), dim3(256), 0, s, res, peak, snr, quality, npeaks, n);
    return ok();
}
int launch_argmax(const double* x, int32_t* out, int rows, int n, hipStream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(argmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, out, rows, n);
    return ok();
}
int launch_mag2db(const float* x, float* out, size_t n, hipStream_t s) {
    if (n == 0) return 0;
    size_t blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(mag2db_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, out, n);
    return ok();
}

}  // namespace echo
