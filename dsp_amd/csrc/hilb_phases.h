// Envelope branch for the band signal: signal_quality of gpu_envelope_detection(band_sig, 'hilbert', smooth_window=3)
// (airSonar_optimized.py:636, :431-475, :477-490).  Only the quality score of this call is used by the reference
// (its peaks are discarded), and the score enters the confidence at 1e-4, so the work is float32 and rides on the
// spectrum kernel: the rfft it already computes gives X; V (the packed spectrum of the Hilbert transform, see
// env_phases.h) goes through the same Bluestein machinery once more (inverse transform as conj(DFT(conj V))).
// Phases of (tid, nt); plain C++ apart from the wavefront reduction.
#pragma once
#include "band_phases.h"
#include "fft9_phases.h"

namespace echo {

struct HilbMisc {
    double sum_sm, max_sm, min_sm, sum_x, sum_x2;
    double ss_sm, ss_x, ss_d;
    double mean_sm, mean_x, mean_d;
};

// byte offsets inside the FFT buffer.  The second transform's output occupies the first 2144 elements; the raw envelope
// goes behind it, the smoothed one over it (it is dead by then), the reduction scratch behind the envelope: 39.6 KB in
// all, inside the 41.5 KB of the 4608-point transform, so that three workgroups share a compute unit's LDS as in the
// plain spectrum kernel.
constexpr int kHilbOffEnv = 20480;                 // float env[N]
constexpr int kHilbOffSm = 0;                      // float sm[N]
constexpr int kHilbOffRed = 38912;                 // double red[8][slots] (slots = wavefronts on the device): five
                                                   // partials of the first round, three of the second; then HilbMisc
constexpr int kHilbRedRows = 8;
constexpr int kHilbMaxN = (kHilbOffRed - kHilbOffEnv) / 4;

// frame sizes the layout above holds: the second transform's output (n2 elements, whole 32-element swizzle blocks)
// must end before env[], and env[] / sm[] hold N floats each
ECHO_HD bool hilb_fits(int N, int n2) { return ((n2 + 31) & ~31) * 9 <= kHilbOffEnv && N <= kHilbMaxN && N * 4 <= kHilbOffEnv - kHilbOffSm; }

ECHO_HD int hilb_slots(int nt) {
#if defined(__HIP_DEVICE_COMPILE__)
    return nt / 64;
#else
    return nt;
#endif
}
ECHO_HD HilbMisc* hilb_misc(unsigned char* smem, int nt) {
    return reinterpret_cast<HilbMisc*>(smem + kHilbOffRed + kHilbRedRows * 8 * hilb_slots(nt));
}

// LDS position of element i and the cyclic length, for the 8192-point (F9 = false) and 4608-point transforms
template <bool F9> ECHO_HD int hpos(int i) { return F9 ? pos9(i) : fa(i); }
template <bool F9> ECHO_HD int hlen() { return F9 ? kF9M : kFftM; }

// X from the first transform (dB spectrum out), V in, as the next Bluestein input a2[k] = conj(V[k]) w[k]
template <bool F9> ECHO_HD void ph_hilb_mid(int tid, int nt, int blk, const BandParams& p, unsigned char* smem) {
    float2* f = lds_c(smem, p.off_fft);
    const int n2 = p.n2;
    for (int k = tid; k <= n2 / 2; k += nt) {
        const float2 wk = p.bw[k];
        const float2 zk = cmul(f[hpos<F9>(k)], wk);
        if (k == 0) {
            float m0 = fabsf(zk.x + zk.y), m1 = fabsf(zk.x - zk.y);
            if (p.spec_db) { m0 = echo_db_from_sq(m0 * m0); m1 = echo_db_from_sq(m1 * m1); }
            p.spec_out[(size_t)blk * p.nbins] = m0;
            p.spec_out[(size_t)blk * p.nbins + n2] = m1;
            f[hpos<F9>(0)] = mk2(0.f, 0.f);
            continue;
        }
        const int kc = n2 - k;
        const float2 wc = p.bw[kc];
        float2 zc = cmul(f[hpos<F9>(kc)], wc); zc.y = -zc.y;                      // conj(Z[n2-k])
        const float2 u = p.uk[k];                                            // exp(-2*pi*i*k/N)
        const float2 E = mk2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
        const float2 Bv = mk2(0.5f * (zk.x - zc.x), 0.5f * (zk.y - zc.y));
        const float2 O = mk2(Bv.y, -Bv.x);
        const float2 uO = cmul(u, O);
        const float2 Xk = mk2(E.x + uO.x, E.y + uO.y);
        const float2 Xc = mk2(E.x - uO.x, -(E.y - uO.y));
        const float sk_ = Xk.x * Xk.x + Xk.y * Xk.y, sc_ = Xc.x * Xc.x + Xc.y * Xc.y;
        const float mk_ = p.spec_db ? echo_db_from_sq(sk_) : sqrtf(sk_), mc_ = p.spec_db ? echo_db_from_sq(sc_) : sqrtf(sc_);
        p.spec_out[(size_t)blk * p.nbins + k] = mk_;
        if (kc != k) p.spec_out[(size_t)blk * p.nbins + kc] = mc_;
        const float2 Yk = mk2(Xk.y, -Xk.x), Ycc = mk2(Xc.y, Xc.x);           // Y = -i X; conj(Y[n2-k])
        const float2 E2 = mk2(0.5f * (Yk.x + Ycc.x), 0.5f * (Yk.y + Ycc.y));
        const float2 D2 = mk2(0.5f * (Yk.x - Ycc.x), 0.5f * (Yk.y - Ycc.y));
        const float2 O2 = cmulc(D2, u);                                      // conj(u) D2
        const float2 Vk = mk2(E2.x - O2.y, E2.y + O2.x), Vc = mk2(E2.x + O2.y, -E2.y + O2.x);
        f[hpos<F9>(k)] = cmul(mk2(Vk.x, -Vk.y), wk);
        if (kc != k) f[hpos<F9>(kc)] = cmul(mk2(Vc.x, -Vc.y), wc);
    }
    for (int i = n2 + tid; i < hlen<F9>(); i += nt) f[hpos<F9>(i)] = mk2(0.f, 0.f);
}

// envelope = |x + i H(x)|: D = DFT(conj V) = c2 w; H(x)[2m] = Re D / n2, H(x)[2m+1] = -Im D / n2
template <bool F9> ECHO_HD void ph_hilb_env(int tid, int nt, int blk, const BandParams& p, unsigned char* smem) {
    const float2* f = lds_c(smem, p.off_fft);
    float* env = reinterpret_cast<float*>(smem + p.off_fft + kHilbOffEnv);
    const float2* x = reinterpret_cast<const float2*>(p.specf_in + (size_t)blk * p.N);
    const float inv = 1.0f / (float)p.n2;
    for (int m = tid; m < p.n2; m += nt) {
        const float2 d = cmul(f[hpos<F9>(m)], p.bw[m]);
        const float2 xv = x[m];
        env[2 * m] = echo_abs2f(xv.x, d.x * inv);
        env[2 * m + 1] = echo_abs2f(xv.y, -d.y * inv);
    }
}

// per-wavefront partials on the device (wave_fold, band_phases.h), per-thread partials on the host; op 0 sum, 1 max,
// 2 min; rows row0 .. row0 + k - 1 of the scratch
ECHO_HD void hilb_red_store(int tid, int nt, unsigned char* smem, double (&v)[5], int k, const int* ops, int row0) {
    double* red = reinterpret_cast<double*>(smem + kHilbOffRed);
    const int slots = hilb_slots(nt);
#if defined(__HIP_DEVICE_COMPILE__)
    for (int q = 0; q < k; ++q) v[q] = wave_fold(ops[q], v[q]);
    if ((tid & 63) == 0)
        for (int q = 0; q < k; ++q) red[(row0 + q) * slots + tid / 64] = v[q];
#else
    for (int q = 0; q < k; ++q) red[(row0 + q) * slots + tid] = v[q];
#endif
}
ECHO_HD double hilb_red_total(int nt, const unsigned char* smem, int q, int op) {
    const double* red = reinterpret_cast<const double*>(smem + kHilbOffRed);
    const int slots = hilb_slots(nt);
    double a = red[q * slots];
    for (int t = 1; t < slots; ++t) {
        const double b = red[q * slots + t];
        a = op == 0 ? a + b : (op == 1 ? (b > a ? b : a) : (b < a ? b : a));
    }
    return a;
}

ECHO_HD int hilb_reflect(int i, int n) { return i < 0 ? -i - 1 : (i >= n ? 2 * n - i - 1 : i); }

// uniform_filter1d(env, 3, mode='reflect') and the first-round sums
ECHO_HD void ph_hilb_smooth(int tid, int nt, int blk, const BandParams& p, unsigned char* smem) {
    const float* env = reinterpret_cast<const float*>(smem + p.off_fft + kHilbOffEnv);
    float* sm = reinterpret_cast<float*>(smem + p.off_fft + kHilbOffSm);
    const float* x = p.specf_in + (size_t)blk * p.N;
    double v[5] = {0.0, -INFINITY, INFINITY, 0.0, 0.0};
    for (int i = tid; i < p.N; i += nt) {
        const float s = (env[hilb_reflect(i - 1, p.N)] + env[i] + env[hilb_reflect(i + 1, p.N)]) / 3.0f;
        sm[i] = s;
        const double sd = (double)s, xd = (double)x[i];
        v[0] += sd; v[1] = sd > v[1] ? sd : v[1]; v[2] = sd < v[2] ? sd : v[2];
        v[3] += xd; v[4] += xd * xd;
    }
    const int ops[5] = {0, 1, 2, 0, 0};
    hilb_red_store(tid, nt, smem, v, 5, ops, 0);
}
// second round: squared deviations.  Every thread folds the first round's totals it needs (the means) by itself, one
// thread keeps the rest for the score: no phase of one thread between two barriers.
ECHO_HD void ph_hilb_dev(int tid, int nt, int blk, const BandParams& p, unsigned char* smem) {
    const float* sm = reinterpret_cast<const float*>(smem + p.off_fft + kHilbOffSm);
    const float* x = p.specf_in + (size_t)blk * p.N;
    const double sum_sm = hilb_red_total(nt, smem, 0, 0), sum_x = hilb_red_total(nt, smem, 3, 0);
    const double mean_sm = sum_sm / (double)p.N, mean_x = sum_x / (double)p.N;
    const double mean_d = p.N > 1 ? ((double)sm[p.N - 1] - (double)sm[0]) / (double)(p.N - 1) : 0.0;
    if (tid == 0) {
        HilbMisc* m = hilb_misc(smem, nt);
        m->sum_sm = sum_sm; m->max_sm = hilb_red_total(nt, smem, 1, 1);
        m->min_sm = hilb_red_total(nt, smem, 2, 2); m->sum_x = sum_x;
        m->sum_x2 = hilb_red_total(nt, smem, 4, 0);
        m->mean_sm = mean_sm; m->mean_x = mean_x; m->mean_d = mean_d;
    }
    double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = tid; i < p.N; i += nt) {
        const double a = (double)sm[i] - mean_sm, b = (double)x[i] - mean_x;
        v[0] += a * a; v[1] += b * b;
        if (i + 1 < p.N) { const double d = ((double)sm[i + 1] - (double)sm[i]) - mean_d; v[2] += d * d; }
    }
    const int ops[5] = {0, 0, 0, 0, 0};
    hilb_red_store(tid, nt, smem, v, 3, ops, 5);
}
ECHO_HD double hilb_clip01(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }
// calculate_signal_quality(band_sig, envelope)
ECHO_HD void ph_hilb_finish(int tid, int nt, int blk, const BandParams& p, unsigned char* smem) {
    if (tid != 0) return;
    const HilbMisc* m = hilb_misc(smem, nt);
    const double N = (double)p.N;
    const double ss_sm = hilb_red_total(nt, smem, 5, 0), ss_x = hilb_red_total(nt, smem, 6, 0);
    const double ss_d = hilb_red_total(nt, smem, 7, 0);
    const double std_sm = sqrt(ss_sm / N), noise = sqrt(ss_x / N);
    const double snr = 10.0 * log10((m->sum_x2 / N) / (noise * noise + 1e-10));
    const double std_d = p.N > 1 ? sqrt(ss_d / (N - 1.0)) : 0.0;
    const double smooth = std_d / (std_sm + 1e-10);
    const double dyn = (m->max_sm - m->min_sm) / (m->mean_sm + 1e-10);
    const double q = hilb_clip01(snr / 20.0) * 0.5 + hilb_clip01(1.0 - smooth) * 0.3 + hilb_clip01(dyn / 2.0) * 0.2;
    p.sq_out[blk] = hilb_clip01(q);
}

}  // namespace echo
