// Envelope branch of _process_band_gpu for the correlation (airSonar_optimized.py:642-654):
//   corr_envelope, corr_peaks, corr_quality = gpu_envelope_detection(corr, 'hilbert', smooth_window=5)   :431-475
//   valid_peaks = corr_peaks[corr_peaks >= min_delay]; peak = valid_peaks[argmax(corr_envelope[valid_peaks])]
//   snr = 10*log10(corr_envelope[peak]^2 / (mean(corr_envelope[:min_delay]^2) + 1e-10))
// One workgroup per (frame, band), float64 throughout:
//   analytic signal: only its imaginary part H(x) is transformed (the real part of scipy.signal.hilbert's result is x
//   up to rounding): X = rfft(x) by a packed half-length complex FFT (gfft.h), Y = -i X on the positive bins,
//   H(x) = irfft(Y) by one more half-length FFT; envelope = hypot(x, H(x))
//   uniform_filter1d(size, mode='reflect'); mean, std; find_peaks(height = mean + 2 std, distance = L // 50,
//   prominence = std): strict local maxima (plateau midpoints), height filter, distance filter by decreasing height,
//   prominence filter; calculate_signal_quality (:477-490).
// Phases of (tid, nt) as in band_phases.h; plain C++.
#pragma once
#include "gfft.h"
#include "envlin_phases.h"

namespace echo {

ECHO_HD int env_lds_add(int* p, int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(p, v);
#else
    const int o = *p; *p += v; return o;
#endif
}

constexpr int kEnvNT = 512;
constexpr int kEnvIterN = 5;           // n <= kEnvIterN * nt: per-thread trip count of the loops over n, unrolled so
                                       // that the global loads of all trips are in flight together
constexpr int kEnvMaxPeaks = 64;       // find_peaks with distance L // 50 returns at most 51 peaks
constexpr int kEnvChunk = 64;          // prominence scan: per-chunk max/min
constexpr int kEnvSuper = 8;           // ... and per group of 8 chunks
constexpr int kEnvMaxKept = 64;        // peaks the distance rule can keep (L // distance + 1 <= 51), rounded up

struct EnvRes {                        // per (frame, band)
    double snr_db;                     // envelope-path SNR (valid when peak_idx >= 0)
    double amplitude;                  // max|corr| over every lag (corr is not mutated on the envelope path)
    double quality;                    // corr_quality
    int32_t peak_idx;                  // index into corr (full coordinates), -1: no valid envelope peak -> fall back
    int32_t n_peaks;                   // len(corr_peaks), diagnostic
};

struct EnvParams {
    int L, n, nb, min_delay, smooth, distance;
    GfftPlan plan;
    const cpx<double>* roots;          // global, plan.nroots
    const cpx<double>* wn;             // global, n
    const cpx<double>* wl;             // global, n + 1: exp(-2*pi*i*k/L)
    const double* corr;                // [B][nb][L] raw correlation
    double* env_out;                   // optional [B][nb][L] smoothed envelope
    EnvRes* res_out;                   // [B][nb]
    int32_t* peaks_out;                // optional [B][nb][kEnvMaxPeaks]: corr_peaks in ascending order, -1 padded
    int off_buf0, off_buf1, off_roots, off_misc, smem_bytes;
    int aux_cmax, aux_cmin, aux_cpos, cand_cap;  // byte offsets inside the auxiliary buffer
    int aux_smax, aux_smin, aux_kept, aux_side;  // extrema of groups of 8 chunks; kept-peak list; per-side minima
    // Lengths whose half n has a prime factor beyond the mixed-radix passes (e.g. the 4282-sample band signal,
    // n = 2141 prime) take the half-length transforms as Bluestein convolutions of cyclic length bs_m >= 2n - 1 (a
    // length the mixed-radix FFT handles with small radices): `plan` is then the plan of bs_m, not of n.
    int bs_m;                          // 0: direct transform of length n
    const cpx<double>* bs_w;           // global, n: exp(-i*pi*k^2/n)
    const cpx<double>* bs_bhat;        // global, bs_m: FFT_m of the wrapped conj(w) sequence, divided by bs_m
    int stop_after;                    // development aid: leave the sequence after stage 1..5 (0 = run everything)
    // Long signals (kEnvLinMinL <= L <= 4608) take the Hilbert transform as one real cyclic convolution of length 9216
    // through an in-place 4608-point transform (envlin_phases.h): `plan`, roots, wn, wl and the Bluestein fields are unused.
    int lin;
    const cpx<double>* lin_tw;         // global, kLinTwN pass twiddles
    const cpx<double>* lin_wk;         // global, kLinH + 1
    const double* lin_gam;             // global, kLinH + 1
    const cpx<double>* lin_wmid;       // global, 8 * 288: lin_wk in the middle phase's lane order
    const cpx<double>* lin_gmid;       // global, 8 * 288: (gam[k], gam[H - k]) in that order
    // Fused kernel of the full chain (fce_phases.h): FIR of the IIR output and the correlation as a real cyclic
    // convolution in the same workgroup, ahead of the Hilbert transform.  `corr` is then where the kernel writes the
    // correlation (and reads it back for the envelope).
    int N, K, kgroups, ypad, ybuf_len; // frame length, FIR taps, groups of kR taps, zero padding of the FIR input
    int off_ybuf, off_ctap;            // LDS: FIR input and this band's reversed taps, behind the slots of the signal
    const double* y_in;                // [B][nb][N] zero-phase IIR output
    const double* taps_rev;            // [nb][(kgroups + 1) * kR] reversed taps, zero padded
    float* bandf_out;                  // optional [B][nb][N] float32 band signal (for the spectrum kernel)
    double* corr_w;                    // [B][nb][L] == corr
    const cpx<double>* fce_tw;         // global, kFceTwN pass twiddles of the 2304-point transform
    const cpx<double>* fce_wk;         // global, kFceH + 1: exp(-2 pi i k / 4608)
    const cpx<double>* fce_g;          // global, [nb][kFceH + 1]: DFT_4608(reversed chirp) / kFceH
    const cpx<double>* fce_wmid;       // global, 8 * 144: fce_wk in the middle phase's lane order
    const cpx<double>* fce_gmid;       // global, [nb][8 * 144][2]: (g[k], g[H - k]) in that order
};
constexpr int kEnvLinMinL = 3072;      // shorter signals keep the half-length transforms (fewer points than 4608)

struct EnvMisc {
    double sum_sm, max_sm, min_sm, sum_x, sum_x2, amp, sum_noise;     // first statistics pass
    double ss_sm, ss_x, ss_d;                                          // second statistics pass
    double mean_sm, mean_x, mean_d, std_sm, thr;
    double best_val;
    int undecided, best_idx, n_final, n_cand, n_kept, pad_;
};

constexpr int kEnvBsMaxRadix = 16;     // Bluestein: largest radix of the cyclic length's passes

inline bool plan_env(EnvParams& p, int L, int nb, int min_delay, int smooth, bool allow_lin = true) {
    p.L = L; p.n = L / 2; p.nb = nb; p.min_delay = min_delay; p.smooth = smooth; p.distance = L / 50;
    p.bs_m = 0; p.bs_w = nullptr; p.bs_bhat = nullptr;
    p.lin = 0; p.lin_tw = nullptr; p.lin_wk = nullptr; p.lin_gam = nullptr; p.lin_wmid = nullptr; p.lin_gmid = nullptr;
    if (L % 2 != 0 || L < 8 || smooth < 1 || smooth % 2 == 0 || smooth > 9) return false;
    if (p.distance < 1) p.distance = 1;
    p.lin = (allow_lin && L >= kEnvLinMinL && L <= kLinH) ? 1 : 0;
    if (p.lin) {
        p.plan.n = p.n; p.plan.npass = 0; p.plan.nroots = 0;
    } else if (!gfft_plan(p.plan, p.n)) {
        bool found = false;
        for (int m = 2 * p.n - 1; m <= 2 * p.n + 512 && !found; ++m)
            if (gfft_plan(p.plan, m) && p.plan.radix[0] <= kEnvBsMaxRadix) { p.bs_m = m; found = true; }
        if (!found) return false;
    }
    // After the smoothing the buffer that held the raw envelope is reused: first as reduction scratch (7 quantities
    // x nt partials of the first statistics pass, 2 x nt of the second behind them), then for the candidate states, the
    // per-chunk extrema and the list of kept peaks (which stay below the second pass's partials).
    const int nchunk = (L + kEnvChunk - 1) / kEnvChunk;
    int aux = 0;
    p.aux_cmax = aux; aux += 5 * nchunk * 8;         // extrema of quarter chunks [4*nchunk], then of chunks [nchunk]
    p.aux_cmin = aux; aux += 5 * nchunk * 8;
    p.cand_cap = L / 2 + 2;                          // strict local maxima are at least two samples apart
    p.aux_cpos = aux; aux += p.cand_cap * 7;         // int pos[cap]; byte state[cap], fk[cap], fu[cap]
    aux = (aux + 15) & ~15;
    const int nsuper = (nchunk + kEnvSuper - 1) / kEnvSuper;
    p.aux_smax = aux; aux += nsuper * 8;
    p.aux_smin = aux; aux += nsuper * 8;
    p.aux_side = aux; aux += 2 * kEnvMaxKept * 8;     // double side_ok[kept][2]: 1 when that side allows the prominence
    p.aux_kept = aux; aux += kEnvMaxKept * 4;         // int candidate slot of each kept peak
    // each of the two buffers holds the transform (n, or bs_m complex doubles) and, for short signals, at least the
    // reduction scratch and the auxiliary arrays
    int bufbytes = (p.bs_m ? p.bs_m : p.n) * 16;
    if (bufbytes < 9 * kEnvNT * 8) bufbytes = 9 * kEnvNT * 8;        // 7 + 2 partials per thread
    if (bufbytes < aux) bufbytes = aux;
    bufbytes = (bufbytes + 15) & ~15;
    int off = 0;
    if (p.lin) {
        // the work buffers of the later phases lie inside the transform buffer: buffer 0 over the slots that hold
        // the result, buffer 1 behind them
        if (bufbytes > kLinBuf1 || kLinBuf1 + bufbytes > kLinBytes || zl(p.n - 1) * 16 + 16 > kLinBuf1) return false;
        p.off_buf0 = 0; p.off_buf1 = kLinBuf1; off = kLinBytes;
    } else {
    p.off_buf0 = off; off += bufbytes;
    p.off_buf1 = off; off += bufbytes;
    }
    p.off_roots = off; off += p.plan.nroots * 16;
    p.off_misc = off; off += ((int)sizeof(EnvMisc) + 15) & ~15;
    p.smem_bytes = off;
    if (p.n > kEnvIterN * kEnvNT) return false;
    return off <= 160 * 1024;
}

// Bluestein tables: w[k] = exp(-i*pi*k^2/n) with k^2 reduced mod 2n in integers; bhat = DFT_m of
// c[j] = conj(w[|j|]) for |j| < n wrapped to length m, divided by m (direct O(m * n) sums in long double: once per plan).
inline void env_bs_tables(const EnvParams& p, std::vector<cpx<double>>& w, std::vector<cpx<double>>& bhat) {
    const long double pi = 3.141592653589793238462643383279502884L;
    const int n = p.n, m = p.bs_m;
    w.resize(n); bhat.resize(m);
    std::vector<long double> cr(n), ci(n);
    for (int k = 0; k < n; ++k) {
        const long long k2 = ((long long)k * k) % (2LL * n);
        const long double a = pi * (long double)k2 / (long double)n;
        cr[k] = cosl(a); ci[k] = sinl(a);              // conj(w[k]) = exp(+i a)
        w[k] = cpx<double>{(double)cr[k], (double)(-ci[k])};
    }
    // exp(-2*pi*i*t/m) table, indices reduced mod m in integers
    std::vector<long double> er(m), ei(m);
    for (int t = 0; t < m; ++t) { const long double a = -2.0L * pi * (long double)t / (long double)m; er[t] = cosl(a); ei[t] = sinl(a); }
    for (int q = 0; q < m; ++q) {
        // sum_j c[j] e(-q j) + sum_{j>=1} c[j] e(+q j)  =  c[0] + sum_{j>=1} c[j] * 2 cos(2 pi q j / m)
        long double sr = cr[0], si = ci[0];
        for (int j = 1; j < n; ++j) {
            const long double c2 = 2.0L * er[(int)(((long long)q * j) % m)];
            sr += cr[j] * c2; si += ci[j] * c2;
        }
        bhat[q] = cpx<double>{(double)(sr / (long double)m), (double)(si / (long double)m)};
    }
}

inline void env_tables(const EnvParams& p, std::vector<cpx<double>>& roots, std::vector<cpx<double>>& wn,
                       std::vector<cpx<double>>& wl) {
    gfft_tables(p.plan, roots, wn);
    const double pi = 3.14159265358979323846;
    wl.resize(p.n + 1);
    for (int k = 0; k <= p.n; ++k) {
        const double a = -2.0 * pi * (double)k / (double)p.L;
        wl[k] = cpx<double>{cos(a), sin(a)};
    }
}

ECHO_HD cpx<double>* env_buf(unsigned char* s, const EnvParams& p, int which) {
    return reinterpret_cast<cpx<double>*>(s + (which ? p.off_buf1 : p.off_buf0));
}
ECHO_HD EnvMisc* env_misc(unsigned char* s, const EnvParams& p) { return reinterpret_cast<EnvMisc*>(s + p.off_misc); }

// -------------------------------------------------------------------------------- Hilbert transform
struct EnvPart1 { double v[7]; };       // per-thread partials: sum sm, max sm, min sm, sum x, sum x^2, max|x|, noise sum
ECHO_HD EnvPart1 ph_env_load(int tid, int nt, int blk, const EnvParams& p, unsigned char* smem) {
    cpx<double>* z = env_buf(smem, p, 0);
    const double* x = p.corr + (size_t)blk * p.L;
    EnvPart1 a;
    a.v[0] = 0.0; a.v[1] = -INFINITY; a.v[2] = INFINITY; a.v[3] = 0.0; a.v[4] = 0.0; a.v[5] = 0.0; a.v[6] = 0.0;
    cpx<double> xv[kEnvIterN];
#pragma unroll
    for (int t = 0; t < kEnvIterN; ++t) {
        const int m = tid + t * nt, mc = m < p.n ? m : p.n - 1;
        xv[t] = cpx<double>{x[2 * mc], x[2 * mc + 1]};
    }
#pragma unroll
    for (int t = 0; t < kEnvIterN; ++t) {
        const int m = tid + t * nt;
        if (m >= p.n) continue;
        z[m] = xv[t];
        const double a0 = fabs(xv[t].x), a1 = fabs(xv[t].y);
        a.v[3] += xv[t].x + xv[t].y; a.v[4] += xv[t].x * xv[t].x + xv[t].y * xv[t].y;
        a.v[5] = a0 > a.v[5] ? a0 : a.v[5]; a.v[5] = a1 > a.v[5] ? a1 : a.v[5];
    }
    cpx<double>* r = reinterpret_cast<cpx<double>*>(smem + p.off_roots);
    for (int i = tid; i < p.plan.nroots; i += nt) r[i] = p.roots[i];
    if (tid == 0) {
        EnvMisc* m = env_misc(smem, p);
        m->undecided = 0; m->best_idx = -1; m->best_val = 0.0; m->n_final = 0; m->n_cand = 0; m->n_kept = 0;
    }
    return a;
}

// the same for the 4608-point transform: padded slots, zeros behind the signal
ECHO_HD EnvPart1 ph_envlin_load(int tid, int nt, int blk, const EnvParams& p, unsigned char* smem) {
    cpx<double>* z = reinterpret_cast<cpx<double>*>(smem);
    const double* x = p.corr + (size_t)blk * p.L;
    EnvPart1 a;
    a.v[0] = 0.0; a.v[1] = -INFINITY; a.v[2] = INFINITY; a.v[3] = 0.0; a.v[4] = 0.0; a.v[5] = 0.0; a.v[6] = 0.0;
    cpx<double> xv[kEnvIterN];
#pragma unroll
    for (int t = 0; t < kEnvIterN; ++t) {
        const int m = tid + t * nt, mc = m < p.n ? m : p.n - 1;
        xv[t] = cpx<double>{x[2 * mc], x[2 * mc + 1]};
    }
#pragma unroll
    for (int t = 0; t < kEnvIterN; ++t) {
        const int m = tid + t * nt;
        if (m >= p.n) continue;
        z[zl(m)] = xv[t];
        const double a0 = fabs(xv[t].x), a1 = fabs(xv[t].y);
        a.v[3] += xv[t].x + xv[t].y; a.v[4] += xv[t].x * xv[t].x + xv[t].y * xv[t].y;
        a.v[5] = a0 > a.v[5] ? a0 : a.v[5]; a.v[5] = a1 > a.v[5] ? a1 : a.v[5];
    }
    for (int m = p.n + tid; m < kLinH; m += nt) z[zl(m)] = cpx<double>{0.0, 0.0};
    if (tid == 0) {
        EnvMisc* m = env_misc(smem, p);
        m->undecided = 0; m->best_idx = -1; m->best_val = 0.0; m->n_final = 0; m->n_cand = 0; m->n_kept = 0;
    }
    return a;
}
// envelope[i] = |x[i] + i H(x)[i]|, H(x)[2m] + i H(x)[2m+1] in slot m of the transform buffer; into buffer 1
ECHO_HD void ph_envlin_abs(int tid, int nt, int blk, const EnvParams& p, unsigned char* smem) {
    const cpx<double>* v = reinterpret_cast<const cpx<double>*>(smem);
    double* env = reinterpret_cast<double*>(env_buf(smem, p, 1));
    const double* x = p.corr + (size_t)blk * p.L;
    cpx<double> xv[kEnvIterN];
#pragma unroll
    for (int t = 0; t < kEnvIterN; ++t) {
        const int m = tid + t * nt, mc = m < p.n ? m : p.n - 1;
        xv[t] = cpx<double>{x[2 * mc], x[2 * mc + 1]};
    }
#pragma unroll
    for (int t = 0; t < kEnvIterN; ++t) {
        const int m = tid + t * nt;
        if (m >= p.n) continue;
        const cpx<double> h = v[zl(m)];
        env[2 * m] = hypot(xv[t].x, h.x);
        env[2 * m + 1] = hypot(xv[t].y, h.y);
    }
}

ECHO_HD void ph_env_fft_pass(int tid, int nt, const EnvParams& p, unsigned char* smem, int ps, int src, bool inv) {
    const cpx<double>* r = reinterpret_cast<const cpx<double>*>(smem + p.off_roots) + p.plan.root_off[ps];
    gfft_pass<double>(tid, nt, p.plan, ps, env_buf(smem, p, src), env_buf(smem, p, src ^ 1), r, p.wn, inv);
}

// Bluestein steps around the passes of length bs_m (in place on buffer `which`).
// pre:  a[i] = z[i] * w[i] (z conjugated first for the inverse transform), zero beyond n
ECHO_HD void ph_env_bs_pre(int tid, int nt, const EnvParams& p, unsigned char* smem, int which, bool conj_in) {
    cpx<double>* z = env_buf(smem, p, which);
    for (int i = tid; i < p.bs_m; i += nt) {
        if (i >= p.n) { z[i] = cpx<double>{0.0, 0.0}; continue; }
        const cpx<double> w = p.bs_w[i];
        const double zx = z[i].x, zy = conj_in ? -z[i].y : z[i].y;
        z[i] = cpx<double>{zx * w.x - zy * w.y, zx * w.y + zy * w.x};
    }
}
ECHO_HD void ph_env_bs_mul(int tid, int nt, const EnvParams& p, unsigned char* smem, int which) {
    cpx<double>* z = env_buf(smem, p, which);
    for (int i = tid; i < p.bs_m; i += nt) {
        const cpx<double> b = p.bs_bhat[i], v = z[i];
        z[i] = cpx<double>{v.x * b.x - v.y * b.y, v.x * b.y + v.y * b.x};
    }
}
// post: X[k] = conv[k] * w[k] (conjugated for the inverse transform), k < n
ECHO_HD void ph_env_bs_post(int tid, int nt, const EnvParams& p, unsigned char* smem, int which, bool conj_out) {
    cpx<double>* z = env_buf(smem, p, which);
    for (int k = tid; k < p.n; k += nt) {
        const cpx<double> w = p.bs_w[k], v = z[k];
        const double re = v.x * w.x - v.y * w.y, im = v.x * w.y + v.y * w.x;
        z[k] = cpx<double>{re, conj_out ? -im : im};
    }
}

// Z = FFT_n(x[2m] + i x[2m+1]) -> V, the packed spectrum whose inverse half-length transform is
// H(x)[2m] + i H(x)[2m+1] (times n).  Pair (k, n-k) is owned by one thread.
ECHO_HD void ph_env_mid(int tid, int nt, const EnvParams& p, unsigned char* smem, int which) {
    cpx<double>* Z = env_buf(smem, p, which);
    const int n = p.n;
    constexpr int kIter = (kEnvIterN + 1) / 2;
    cpx<double> wv[kIter];
#pragma unroll
    for (int t = 0; t < kIter; ++t) { const int k = tid + t * nt; wv[t] = p.wl[k <= n / 2 ? k : 0]; }
#pragma unroll
    for (int t = 0; t < kIter; ++t) {
        const int k = tid + t * nt;
        if (k > n / 2) continue;
        if (k == 0) { Z[0] = cpx<double>{0.0, 0.0}; continue; }
        const int kc = n - k;
        const cpx<double> zk = Z[k], zc = cpx<double>{Z[kc].x, -Z[kc].y};
        const cpx<double> w = wv[t];
        // E, O: transforms of the even and odd samples;  X[k] = E + w O,  X[n-k] = conj(E - w O)
        const cpx<double> E{0.5 * (zk.x + zc.x), 0.5 * (zk.y + zc.y)};
        const cpx<double> Bv{0.5 * (zk.x - zc.x), 0.5 * (zk.y - zc.y)};
        const cpx<double> O{Bv.y, -Bv.x};                                  // -i Bv
        const cpx<double> wO{w.x * O.x - w.y * O.y, w.x * O.y + w.y * O.x};
        const cpx<double> Xk{E.x + wO.x, E.y + wO.y};
        const cpx<double> Xc{E.x - wO.x, -(E.y - wO.y)};                   // X[n-k]
        // Y = -i X on the positive bins
        const cpx<double> Yk{Xk.y, -Xk.x}, Yc{Xc.y, -Xc.x};
        const cpx<double> Ycc{Yc.x, -Yc.y};                                // conj(Y[n-k])
        const cpx<double> E2{0.5 * (Yk.x + Ycc.x), 0.5 * (Yk.y + Ycc.y)};
        const cpx<double> D2{0.5 * (Yk.x - Ycc.x), 0.5 * (Yk.y - Ycc.y)};
        const cpx<double> O2{w.x * D2.x + w.y * D2.y, w.x * D2.y - w.y * D2.x};   // conj(w) D2
        // V[k] = E2 + i O2,  V[n-k] = conj(E2) + i conj(O2)
        Z[k] = cpx<double>{E2.x - O2.y, E2.y + O2.x};
        if (kc != k) Z[kc] = cpx<double>{E2.x + O2.y, -E2.y + O2.x};
    }
}

// envelope[i] = |x[i] + i H(x)[i]|, H(x)[2m] = Re v[m] / n, H(x)[2m+1] = Im v[m] / n
ECHO_HD void ph_env_abs(int tid, int nt, int blk, const EnvParams& p, unsigned char* smem, int which) {
    const cpx<double>* v = env_buf(smem, p, which);
    double* env = reinterpret_cast<double*>(env_buf(smem, p, which ^ 1));
    const double* x = p.corr + (size_t)blk * p.L;
    const double inv_n = 1.0 / (double)p.n;
    cpx<double> xv[kEnvIterN];
#pragma unroll
    for (int t = 0; t < kEnvIterN; ++t) {
        const int m = tid + t * nt, mc = m < p.n ? m : p.n - 1;
        xv[t] = cpx<double>{x[2 * mc], x[2 * mc + 1]};
    }
#pragma unroll
    for (int t = 0; t < kEnvIterN; ++t) {
        const int m = tid + t * nt;
        if (m >= p.n) continue;
        const cpx<double> h = v[m];
        env[2 * m] = hypot(xv[t].x, h.x * inv_n);
        env[2 * m + 1] = hypot(xv[t].y, h.y * inv_n);
    }
}

// -------------------------------------------------------------------------------- smoothing and statistics
ECHO_HD int reflect_idx(int i, int n) {
    while (i < 0 || i >= n) i = i < 0 ? -i - 1 : 2 * n - i - 1;
    return i;
}

// deterministic block reduction: K partials per thread at scratch[q*nt + tid]; op 0 sum, 1 max, 2 min
ECHO_HD double red_op(int op, double a, double b) { return op == 0 ? a + b : (op == 1 ? (a > b ? a : b) : (a < b ? a : b)); }
// On the device the per-thread partials are first folded inside each wavefront (red_wave, fixed-order butterfly), so
// only lane 0 of a wavefront stores, and the fold reads one value per wavefront.
ECHO_HD double red_identity(int op) { return op == 0 ? 0.0 : (op == 1 ? -INFINITY : INFINITY); }
ECHO_HD double red_wave(int op, double a) { return wave_fold(op, a); }      // band_phases.h
// Total of quantity q over the workgroup from the stored partials (device: one per wavefront, at the slot of its first
// lane; host: one per thread).  Every thread that needs a total folds it by itself, in the same order: no fold phase of
// one wavefront between two barriers.
ECHO_HD double red_total(const double* scratch, int nt, int q, int op) {
    double a = scratch[q * nt];
#if defined(__HIP_DEVICE_COMPILE__)
    for (int w = 1; w < nt / 64; ++w) a = red_op(op, a, scratch[q * nt + w * 64]);
#else
    for (int t = 1; t < nt; ++t) a = red_op(op, a, scratch[q * nt + t]);
#endif
    return a;
}

// sm = uniform_filter1d(env, size) into the other buffer; the partials stay with the thread (the raw envelope is
// still being read by neighbours) and are stored in the next phase
ECHO_HD void ph_env_smooth(int tid, int nt, int blk, const EnvParams& p, unsigned char* smem, int env_in, EnvPart1& a) {
    const double* env = reinterpret_cast<const double*>(env_buf(smem, p, env_in));
    double* sm = reinterpret_cast<double*>(env_buf(smem, p, env_in ^ 1));
    const int h = p.smooth / 2;
    const double inv_size = 1.0 / (double)p.smooth;
    for (int i = tid; i < p.L; i += nt) {
        double s = 0.0;
        if (i >= h && i + h < p.L) { for (int d = -h; d <= h; ++d) s += env[i + d]; }
        else { for (int d = -h; d <= h; ++d) s += env[reflect_idx(i + d, p.L)]; }
        s = s * inv_size;
        sm[i] = s;
        if (p.env_out) p.env_out[(size_t)blk * p.L + i] = s;
        a.v[0] += s; a.v[1] = s > a.v[1] ? s : a.v[1]; a.v[2] = s < a.v[2] ? s : a.v[2];
        if (i < p.min_delay) a.v[6] += s * s;
    }
}
ECHO_HD void ph_env_part1_store(int tid, int nt, const EnvParams& p, unsigned char* smem, int scratch_buf, const EnvPart1& a) {
    double* scratch = reinterpret_cast<double*>(env_buf(smem, p, scratch_buf));
    const int ops[7] = {0, 1, 2, 0, 0, 1, 0};
    for (int q = 0; q < 7; ++q) scratch[q * nt + tid] = red_wave(ops[q], a.v[q]);
}
// second pass: squared deviations (np.std is the two-pass population deviation).  The totals of the first pass are
// folded here by every thread (the means) and by one thread for the record.
ECHO_HD void ph_env_part2(int tid, int nt, int blk, const EnvParams& p, unsigned char* smem, int sm_buf) {
    const double* sm = reinterpret_cast<const double*>(env_buf(smem, p, sm_buf));
    double* scratch = reinterpret_cast<double*>(env_buf(smem, p, sm_buf ^ 1));
    EnvMisc* m = env_misc(smem, p);
    const double sum_sm = red_total(scratch, nt, 0, 0);
    const double mean_sm = sum_sm / (double)p.L;
    const double mean_d = p.L > 1 ? (sm[p.L - 1] - sm[0]) / (double)(p.L - 1) : 0.0;
    if (tid == nt - 1) {
        m->sum_sm = sum_sm; m->max_sm = red_total(scratch, nt, 1, 1);
        m->min_sm = red_total(scratch, nt, 2, 2); m->sum_x = red_total(scratch, nt, 3, 0);
        m->sum_x2 = red_total(scratch, nt, 4, 0); m->amp = red_total(scratch, nt, 5, 1);
        m->sum_noise = red_total(scratch, nt, 6, 0);
        m->mean_sm = mean_sm; m->mean_x = m->sum_x / (double)p.L; m->mean_d = mean_d;
    }
    double a = 0.0, c = 0.0;
    for (int i = tid; i < p.L; i += nt) {
        const double ds = sm[i] - mean_sm;
        a += ds * ds;
        if (i + 1 < p.L) { const double dd = (sm[i + 1] - sm[i]) - mean_d; c += dd * dd; }
    }
    scratch[7 * nt + tid] = red_wave(0, a); scratch[8 * nt + tid] = red_wave(0, c);
}
// totals of the second pass, by one thread of the phase that follows it
ECHO_HD void ph_env_part2_final(int tid, int nt, const EnvParams& p, unsigned char* smem, int scratch_buf) {
    if (tid != nt - 1) return;
    const double* scratch = reinterpret_cast<const double*>(env_buf(smem, p, scratch_buf)) + 7 * nt;
    EnvMisc* m = env_misc(smem, p);
    m->ss_sm = red_total(scratch, nt, 0, 0); m->ss_d = red_total(scratch, nt, 1, 0);
    // sum (x - mean)^2 = sum x^2 - (sum x)^2 / L: the correlation of a band-pass signal has a mean far below its
    // deviation, so nothing cancels
    m->ss_x = m->sum_x2 - m->sum_x * m->sum_x / (double)p.L;
    if (m->ss_x < 0.0) m->ss_x = 0.0;
    m->std_sm = sqrt(m->ss_sm / (double)p.L);
    const double two_std = 2.0 * m->std_sm;
    m->thr = m->mean_sm + two_std;
}

// -------------------------------------------------------------------------------- find_peaks
// state[i]: 0 not a candidate, 1 undecided, 2 kept, 3 removed by the distance rule
// Auxiliary buffer (the dead raw-envelope buffer): per-chunk extrema and the candidate slots.  A candidate is a local
// maximum that passed the height filter; slot arrays: position, state (1 undecided, 2 kept, 3 removed) and the two
// flags of a distance round (bytes: several threads may set a flag, none reads-modifies-writes it).
ECHO_HD unsigned char* env_aux(unsigned char* smem, const EnvParams& p, int sm_buf) {
    return reinterpret_cast<unsigned char*>(env_buf(smem, p, sm_buf ^ 1));
}
struct EnvCand { int* pos; unsigned char* state; unsigned char* fk; unsigned char* fu; };
ECHO_HD EnvCand env_cand(unsigned char* smem, const EnvParams& p, int sm_buf) {
    unsigned char* a = env_aux(smem, p, sm_buf);
    EnvCand c;
    c.pos = reinterpret_cast<int*>(a + p.aux_cpos);
    c.state = a + p.aux_cpos + 4 * p.cand_cap; c.fk = c.state + p.cand_cap; c.fu = c.fk + p.cand_cap;
    return c;
}
ECHO_HD void ph_env_candidates(int tid, int nt, const EnvParams& p, unsigned char* smem, int sm_buf) {
    ph_env_part2_final(tid, nt, p, smem, sm_buf ^ 1);       // its partials lie behind everything this phase writes
    const double* sm = reinterpret_cast<const double*>(env_buf(smem, p, sm_buf));
    double* cmax = reinterpret_cast<double*>(env_aux(smem, p, sm_buf) + p.aux_cmax);
    double* cmin = reinterpret_cast<double*>(env_aux(smem, p, sm_buf) + p.aux_cmin);
    const int nchunk = (p.L + kEnvChunk - 1) / kEnvChunk;
    // four threads per chunk, 16 samples each; the chunk's extrema are folded by the prominence scan's reader
    for (int t = tid; t < 4 * nchunk; t += nt) {
        const int c = t >> 2, part = t & 3;
        double mx = -INFINITY, mn = INFINITY;
        const int b0 = c * kEnvChunk + part * (kEnvChunk / 4);
        const int e = b0 + kEnvChunk / 4 < p.L ? b0 + kEnvChunk / 4 : p.L;
        for (int i = b0; i < e; ++i) { mx = sm[i] > mx ? sm[i] : mx; mn = sm[i] < mn ? sm[i] : mn; }
        cmax[t] = mx; cmin[t] = mn;
    }
}
// local maxima as scipy's _local_maxima_1d: x[i-1] < x[i], plateau x[i..j] equal, x[j+1] < x[i] -> (i + j) // 2
ECHO_HD void ph_env_maxima(int tid, int nt, const EnvParams& p, unsigned char* smem, int sm_buf) {
    const double* sm = reinterpret_cast<const double*>(env_buf(smem, p, sm_buf));
    const EnvCand c = env_cand(smem, p, sm_buf);
    EnvMisc* m = env_misc(smem, p);
    const double thr = m->thr;
    {
        double* cmax = reinterpret_cast<double*>(env_aux(smem, p, sm_buf) + p.aux_cmax);
        double* cmin = reinterpret_cast<double*>(env_aux(smem, p, sm_buf) + p.aux_cmin);
        const int nchunk = (p.L + kEnvChunk - 1) / kEnvChunk;
        for (int ch = tid; ch < nchunk; ch += nt) {
            double mx = cmax[4 * ch], mn = cmin[4 * ch];
            for (int q = 1; q < 4; ++q) {
                mx = cmax[4 * ch + q] > mx ? cmax[4 * ch + q] : mx;
                mn = cmin[4 * ch + q] < mn ? cmin[4 * ch + q] : mn;
            }
            cmax[4 * nchunk + ch] = mx; cmin[4 * nchunk + ch] = mn;
        }
    }
    for (int i = tid + 1; i < p.L - 1; i += nt) {
        const double v = sm[i];
        if (!(v >= thr) || !(sm[i - 1] < v)) continue;   // height filter first: it rejects almost every sample
        int j = i;
        while (j + 1 < p.L && sm[j + 1] == v) ++j;
        if (j + 1 >= p.L || !(sm[j + 1] < v)) continue;
        const int slot = env_lds_add(&m->n_cand, 1);
        c.pos[slot] = (i + j) / 2; c.state[slot] = 1; c.fk[slot] = 0; c.fu[slot] = 0;
    }
}
// Distance rule (scipy _select_by_peak_distance: highest first, a kept peak removes every peak within distance-1
// samples) as a fixpoint: a candidate is removed when a kept candidate of higher priority lies in that range, kept
// when every higher-priority candidate in range is already removed.  One round = all (candidate, other) pairs in
// parallel setting the two flags, then one update per candidate.
ECHO_HD bool env_higher(double vj, int pj, double vi, int pi) { return vj > vi || (vj == vi && pj > pi); }
ECHO_HD void ph_env_distance_pairs(int tid, int nt, const EnvParams& p, unsigned char* smem, int sm_buf) {
    const double* sm = reinterpret_cast<const double*>(env_buf(smem, p, sm_buf));
    const EnvCand c = env_cand(smem, p, sm_buf);
    const int nc = env_misc(smem, p)->n_cand;
    for (int e = tid; e < nc * nc; e += nt) {
        const int i = e / nc, u = e - i * nc;
        if (i == u || c.state[i] != 1) continue;
        const int su = c.state[u];
        if (su == 3) continue;
        const int d = c.pos[u] > c.pos[i] ? c.pos[u] - c.pos[i] : c.pos[i] - c.pos[u];
        if (d >= p.distance || !env_higher(sm[c.pos[u]], c.pos[u], sm[c.pos[i]], c.pos[i])) continue;
        if (su == 2) c.fk[i] = 1; else c.fu[i] = 1;
    }
}
ECHO_HD void ph_env_distance_update(int tid, int nt, const EnvParams& p, unsigned char* smem, int sm_buf) {
    const EnvCand c = env_cand(smem, p, sm_buf);
    EnvMisc* m = env_misc(smem, p);
    const int nc = m->n_cand;
    for (int t = tid; t < nc; t += nt) {
        if (c.state[t] != 1) continue;
        if (c.fk[t]) c.state[t] = 3;
        else if (!c.fu[t]) c.state[t] = 2;
        else m->undecided = 1;
        c.fk[t] = 0; c.fu[t] = 0;
    }
}
// prominence filter of the kept peaks (scipy _peak_prominences, wlen = None; prominence >= std of the envelope): the two
// sides of every kept peak are walked by two threads (env_side_ok), peaks below the required prominence are dropped.
struct EnvScan { const double *sm, *qmax, *qmin, *cmax, *cmin, *smax, *smin; int L; };
ECHO_HD EnvScan env_scan(unsigned char* smem, const EnvParams& p, int sm_buf) {
    unsigned char* a = env_aux(smem, p, sm_buf);
    const int nchunk = (p.L + kEnvChunk - 1) / kEnvChunk;
    EnvScan s;
    s.sm = reinterpret_cast<const double*>(env_buf(smem, p, sm_buf));
    s.qmax = reinterpret_cast<const double*>(a + p.aux_cmax); s.qmin = reinterpret_cast<const double*>(a + p.aux_cmin);
    s.cmax = s.qmax + 4 * nchunk; s.cmin = s.qmin + 4 * nchunk;
    s.smax = reinterpret_cast<const double*>(a + p.aux_smax); s.smin = reinterpret_cast<const double*>(a + p.aux_smin);
    s.L = p.L;
    return s;
}
// One side of a kept peak against the required prominence: scipy's _peak_prominences walks from the peak to the first
// sample above it (or the end of the signal) and takes the minimum on the way; the peak stays when
// x[peak] - max(left_min, right_min) >= prominence.  Rounding is monotone, so that holds exactly when on each side some
// sample v of the walk has fl(h - v) >= prom: the walk can stop at the first such sample (almost always within a few
// samples of a real peak) instead of running to the end of the signal for the minimum itself.  Whole groups of 8 chunks
// (512 samples), chunks (64) and quarter chunks (16) whose maximum does not exceed the peak are judged by their minimum.
ECHO_HD bool env_side_ok(const EnvScan& s, int peak, int dir, double prom) {
    constexpr int Q = kEnvChunk / 4, S = kEnvChunk * kEnvSuper;
    const int L = s.L;
    const double h = s.sm[peak];
    if (h - h >= prom) return true;                         // the walk starts on the peak itself
    int i = peak + dir;
    while (i >= 0 && i < L) {
        const int g = i / S, gb = g * S, ge = gb + S;
        if (ge <= L && (dir < 0 ? i == ge - 1 : i == gb) && s.smax[g] <= h) {
            if (h - s.smin[g] >= prom) return true;
            i = dir < 0 ? gb - 1 : ge;
            continue;
        }
        const int c = i / kEnvChunk, cb = c * kEnvChunk, ce = cb + kEnvChunk;
        if (ce <= L && (dir < 0 ? i == ce - 1 : i == cb) && s.cmax[c] <= h) {
            if (h - s.cmin[c] >= prom) return true;
            i = dir < 0 ? cb - 1 : ce;
            continue;
        }
        const int q = i / Q, qb = q * Q, qe = qb + Q;
        if (qe <= L && (dir < 0 ? i == qe - 1 : i == qb) && s.qmax[q] <= h) {
            if (h - s.qmin[q] >= prom) return true;
            i = dir < 0 ? qb - 1 : qe;
            continue;
        }
        const double v = s.sm[i];
        if (v > h) break;
        if (h - v >= prom) return true;
        i += dir;
    }
    return false;
}
// extrema of the groups of 8 chunks, and the list of the candidates the distance rule kept
ECHO_HD void ph_env_kept_list(int tid, int nt, const EnvParams& p, unsigned char* smem, int sm_buf) {
    unsigned char* a = env_aux(smem, p, sm_buf);
    const int nchunk = (p.L + kEnvChunk - 1) / kEnvChunk, nsuper = (nchunk + kEnvSuper - 1) / kEnvSuper;
    const EnvScan s = env_scan(smem, p, sm_buf);
    double* smax = reinterpret_cast<double*>(a + p.aux_smax);
    double* smin = reinterpret_cast<double*>(a + p.aux_smin);
    for (int g = tid; g < nsuper; g += nt) {
        double mx = -INFINITY, mn = INFINITY;
        const int c1 = (g + 1) * kEnvSuper < nchunk ? (g + 1) * kEnvSuper : nchunk;
        for (int c = g * kEnvSuper; c < c1; ++c) { mx = s.cmax[c] > mx ? s.cmax[c] : mx; mn = s.cmin[c] < mn ? s.cmin[c] : mn; }
        smax[g] = mx; smin[g] = mn;
    }
    const EnvCand c = env_cand(smem, p, sm_buf);
    EnvMisc* m = env_misc(smem, p);
    int* kept = reinterpret_cast<int*>(a + p.aux_kept);
    const int nc = m->n_cand;
    for (int t = tid; t < nc; t += nt) {
        if (c.state[t] != 2) continue;
        const int slot = env_lds_add(&m->n_kept, 1);
        if (slot < kEnvMaxKept) kept[slot] = t;
    }
}
ECHO_HD void ph_env_side_mins(int tid, int nt, const EnvParams& p, unsigned char* smem, int sm_buf) {
    unsigned char* a = env_aux(smem, p, sm_buf);
    const EnvCand c = env_cand(smem, p, sm_buf);
    const int nk = env_misc(smem, p)->n_kept;
    if (nk > kEnvMaxKept) return;                              // cannot happen under the distance rule; handled serially below
    const int* kept = reinterpret_cast<const int*>(a + p.aux_kept);
    double* side = reinterpret_cast<double*>(a + p.aux_side);
    const EnvScan s = env_scan(smem, p, sm_buf);
    const double prom = env_misc(smem, p)->std_sm;
    for (int e = tid; e < 2 * nk; e += nt)
        side[e] = env_side_ok(s, c.pos[kept[e >> 1]], (e & 1) ? +1 : -1, prom) ? 1.0 : 0.0;
}
ECHO_HD void ph_env_prominence(int tid, int nt, const EnvParams& p, unsigned char* smem, int sm_buf) {
    unsigned char* a = env_aux(smem, p, sm_buf);
    const EnvCand c = env_cand(smem, p, sm_buf);
    EnvMisc* m = env_misc(smem, p);
    const EnvScan s = env_scan(smem, p, sm_buf);
    const int nk = m->n_kept;
    if (nk > kEnvMaxKept) {
        const int nc = m->n_cand;
        for (int t = tid; t < nc; t += nt) {
            if (c.state[t] != 2) continue;
            const int i = c.pos[t];
            if (!(env_side_ok(s, i, -1, m->std_sm) && env_side_ok(s, i, +1, m->std_sm))) c.state[t] = 3;
            else env_lds_add(&m->n_final, 1);
        }
        return;
    }
    const int* kept = reinterpret_cast<const int*>(a + p.aux_kept);
    const double* side = reinterpret_cast<const double*>(a + p.aux_side);
    for (int k = tid; k < nk; k += nt) {
        const int t = kept[k];
        if (!(side[2 * k] != 0.0 && side[2 * k + 1] != 0.0)) c.state[t] = 3;
        else env_lds_add(&m->n_final, 1);
    }
}
// valid_peaks[np.argmax(corr_envelope[valid_peaks])]: highest surviving peak at index >= min_delay, first on ties
ECHO_HD void ph_env_select(int tid, int nt, const EnvParams& p, unsigned char* smem, int sm_buf) {
    if (tid != 0) return;
    const double* sm = reinterpret_cast<const double*>(env_buf(smem, p, sm_buf));
    const EnvCand c = env_cand(smem, p, sm_buf);
    EnvMisc* m = env_misc(smem, p);
    const int nc = m->n_cand;
    int best = -1; double bv = 0.0;
    for (int t = 0; t < nc; ++t) {
        if (c.state[t] != 2 || c.pos[t] < p.min_delay) continue;
        const double v = sm[c.pos[t]];
        if (best < 0 || v > bv || (v == bv && c.pos[t] < best)) { best = c.pos[t]; bv = v; }
    }
    m->best_idx = best;
    m->best_val = best >= 0 ? bv : 0.0;
}
// corr_peaks: the surviving candidates in ascending order (rank by counting; they are few)
ECHO_HD void ph_env_peaks_out(int tid, int nt, int blk, const EnvParams& p, unsigned char* smem, int sm_buf) {
    if (!p.peaks_out) return;
    const EnvCand c = env_cand(smem, p, sm_buf);
    const int nc = env_misc(smem, p)->n_cand;
    int32_t* dst = p.peaks_out + (size_t)blk * kEnvMaxPeaks;
    // every slot has exactly one writer: ranks below n_final belong to the kept peaks, the rest is padding
    const int n_final = env_misc(smem, p)->n_final;
    for (int t = tid; t < kEnvMaxPeaks; t += nt) if (t >= n_final) dst[t] = -1;
    for (int t = tid; t < nc; t += nt) {
        if (c.state[t] != 2) continue;
        int rank = 0;
        for (int u = 0; u < nc; ++u) rank += (c.state[u] == 2 && c.pos[u] < c.pos[t]) ? 1 : 0;
        if (rank < kEnvMaxPeaks) dst[rank] = c.pos[t];
    }
}
ECHO_HD double clip01(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }
ECHO_HD void ph_env_finish(int tid, int blk, const EnvParams& p, unsigned char* smem) {
    if (tid != 0) return;
    const EnvMisc* m = env_misc(smem, p);
    EnvRes r;
    r.peak_idx = m->best_idx; r.n_peaks = m->n_final; r.amplitude = m->amp; r.snr_db = 0.0;
    if (m->best_idx >= 0) {
        const double sp = m->best_val * m->best_val;
        const double npow = (p.min_delay > 0 ? m->sum_noise / (double)p.min_delay : NAN) + 1e-10;
        r.snr_db = 10.0 * log10(sp / npow);
    }
    // calculate_signal_quality(corr, corr_envelope)
    const double L = (double)p.L;
    const double signal_power = m->sum_x2 / L;
    const double noise_est = sqrt(m->ss_x / L);
    const double snr = 10.0 * log10(signal_power / (noise_est * noise_est + 1e-10));
    const double std_d = p.L > 1 ? sqrt(m->ss_d / (L - 1.0)) : 0.0;
    const double smooth = std_d / (m->std_sm + 1e-10);
    const double dyn = (m->max_sm - m->min_sm) / (m->mean_sm + 1e-10);
    const double q = clip01(snr / 20.0) * 0.5 + clip01(1.0 - smooth) * 0.3 + clip01(dyn / 2.0) * 0.2;
    r.quality = clip01(q);
    p.res_out[blk] = r;
}

// Barrier-separated sequence of one workgroup.  PHASE runs a statement for every thread (tid in scope); SYNC is a
// workgroup barrier.  PART1 declares per-thread storage for the smoothing partials: a local on the device, an array
// indexed by tid on the host.
// One half-length transform (forward, or unnormalised inverse) of buffer cur_: the plan's passes directly, or as a
// Bluestein convolution (pre-multiply, forward passes of length bs_m, product with bhat, inverse passes, post-multiply;
// the inverse transform of V is conj(DFT(conj V))).
#define ECHO_ENV_TRANSFORM(PHASE, SYNC, p, smem, nt, INV)                                                    \
    if ((p).bs_m == 0) {                                                                                     \
        for (int ps_ = 0; ps_ < (p).plan.npass; ++ps_) {                                                     \
            PHASE(ph_env_fft_pass(tid, nt, p, smem, ps_, cur_, INV)) SYNC                                    \
            cur_ ^= 1;                                                                                       \
        }                                                                                                    \
    } else {                                                                                                 \
        PHASE(ph_env_bs_pre(tid, nt, p, smem, cur_, INV)) SYNC                                               \
        for (int ps_ = 0; ps_ < (p).plan.npass; ++ps_) {                                                     \
            PHASE(ph_env_fft_pass(tid, nt, p, smem, ps_, cur_, false)) SYNC                                  \
            cur_ ^= 1;                                                                                       \
        }                                                                                                    \
        PHASE(ph_env_bs_mul(tid, nt, p, smem, cur_)) SYNC                                                    \
        for (int ps_ = 0; ps_ < (p).plan.npass; ++ps_) {                                                     \
            PHASE(ph_env_fft_pass(tid, nt, p, smem, ps_, cur_, true)) SYNC                                   \
            cur_ ^= 1;                                                                                       \
        }                                                                                                    \
        PHASE(ph_env_bs_post(tid, nt, p, smem, cur_, INV)) SYNC                                              \
    }

// Hilbert transform of the signal in the padded slots (ph_envlin_load has run); the raw envelope follows
#define ECHO_ENV_LIN_HILBERT(PHASE, SYNC, p, smem, blk, nt)                                                  \
            PHASE(ph_envlin_pass<kLinH, 9, 4608, false>(tid, nt, (p).lin_tw, smem)) SYNC                     \
            PHASE(ph_envlin_pass<kLinH, 8, 512, false>(tid, nt, (p).lin_tw + kLinTw1, smem)) SYNC            \
            PHASE(ph_envlin_pass<kLinH, 8, 64, false>(tid, nt, (p).lin_tw + kLinTw2, smem)) SYNC             \
            if ((p).stop_after == 1) break;                                                                  \
            PHASE(ph_envlin_mid(tid, nt, (p).lin_wk, (p).lin_gam, (p).lin_wmid, (p).lin_gmid, smem)) SYNC      \
            PHASE(ph_envlin_pass<kLinH, 8, 64, true>(tid, nt, (p).lin_tw + kLinTw2, smem)) SYNC              \
            PHASE(ph_envlin_pass<kLinH, 8, 512, true>(tid, nt, (p).lin_tw + kLinTw1, smem)) SYNC             \
            PHASE(ph_envlin_pass<kLinH, 9, 4608, true>(tid, nt, (p).lin_tw, smem)) SYNC                      \
            if ((p).stop_after == 2) break;

// from the raw envelope in buffer cur_ to the record
#define ECHO_ENV_TAIL(PHASE, SYNC, PART1_REF, p, smem, blk, nt)                                              \
        PHASE(ph_env_smooth(tid, nt, blk, p, smem, cur_, PART1_REF)) SYNC                                   \
        const int sm_ = cur_ ^ 1, scr_ = cur_;   /* smoothed envelope; scratch over the raw envelope */      \
        if ((p).stop_after == 10) break;                                                                     \
        PHASE(ph_env_part1_store(tid, nt, p, smem, scr_, PART1_REF)) SYNC                                    \
        if ((p).stop_after == 11) break;                                                                     \
        PHASE(ph_env_part2(tid, nt, blk, p, smem, sm_)) SYNC                                                 \
        if ((p).stop_after == 3) break;                                                                      \
        PHASE(ph_env_candidates(tid, nt, p, smem, sm_)) SYNC                                                 \
        if ((p).stop_after == 12) break;                                                                     \
        PHASE(ph_env_maxima(tid, nt, p, smem, sm_)) SYNC                                                     \
        if ((p).stop_after == 13) break;                                                                     \
        for (int round_ = 0; round_ < (p).L; ++round_) {                                                     \
            PHASE(ph_env_distance_pairs(tid, nt, p, smem, sm_)) SYNC                                         \
            PHASE(ph_env_distance_update(tid, nt, p, smem, sm_)) SYNC                                        \
            const int left_ = env_misc(smem, p)->undecided;                                                  \
            SYNC                                                                                             \
            if (left_ == 0) break;                                                                           \
            PHASE(if (tid == 0) env_misc(smem, p)->undecided = 0)                                            \
        }                                                                                                    \
        if ((p).stop_after == 4) break;                                                                      \
        PHASE(ph_env_kept_list(tid, nt, p, smem, sm_)) SYNC                                                  \
        PHASE(ph_env_side_mins(tid, nt, p, smem, sm_)) SYNC                                                  \
        PHASE(ph_env_prominence(tid, nt, p, smem, sm_)) SYNC                                                 \
        if ((p).stop_after == 5) break;                                                                      \
        PHASE(ph_env_select(tid, nt, p, smem, sm_); ph_env_peaks_out(tid, nt, blk, p, smem, sm_)) SYNC        \
        PHASE(ph_env_finish(tid, blk, p, smem))

#define ECHO_ENV_SEQUENCE(PHASE, SYNC, PART1_DECL, PART1_REF, p, smem, blk, nt)                              \
    do {                                                                                                     \
        PART1_DECL;                                                                                          \
        int cur_ = 0;                                                                                        \
        if ((p).lin) {                                                                                       \
            PHASE(PART1_REF = ph_envlin_load(tid, nt, blk, p, smem)) SYNC                                    \
            if ((p).stop_after == 6) break;                                                                  \
            ECHO_ENV_LIN_HILBERT(PHASE, SYNC, p, smem, blk, nt)                                              \
            PHASE(ph_envlin_abs(tid, nt, blk, p, smem)) SYNC                                                 \
        } else {                                                                                             \
        PHASE(PART1_REF = ph_env_load(tid, nt, blk, p, smem)) SYNC                                                       \
        if ((p).stop_after == 6) break;                                                                      \
        ECHO_ENV_TRANSFORM(PHASE, SYNC, p, smem, nt, false)                                                  \
        if ((p).stop_after == 1) break;                                                                      \
        PHASE(ph_env_mid(tid, nt, p, smem, cur_)) SYNC                                                       \
        ECHO_ENV_TRANSFORM(PHASE, SYNC, p, smem, nt, true)                                                   \
        if ((p).stop_after == 2) break;                                                                      \
        PHASE(ph_env_abs(tid, nt, blk, p, smem, cur_)) SYNC                                                  \
        }                                                                                                    \
        cur_ ^= 1;                               /* cur_: envelope */                                        \
        ECHO_ENV_TAIL(PHASE, SYNC, PART1_REF, p, smem, blk, nt);                                             \
    } while (0)

}  // namespace echo

#include "fce_phases.h"
