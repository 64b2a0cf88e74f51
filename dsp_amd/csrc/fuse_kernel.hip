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

static inline int ok() { return hipGetLastError() == hipSuccess ? 0 : -4; }

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
