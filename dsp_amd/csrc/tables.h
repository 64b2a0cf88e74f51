// Host-side construction of the constant tables and the LDS layout of the band kernel.
// Plain C++ (no HIP): shared by the library's echo_create and by the host emulation used in unit tests.
#pragma once
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <vector>

#include "band_phases.h"
#include "hilb_phases.h"
#include "cfft_phases.h"
#include "rcf_phases.h"
#include "ospec_phases.h"
#include "fft9_phases.h"

namespace echo {

struct HostTables {
    std::vector<double> taps_rev;   // [nb][kgroups*kR]
    std::vector<double> chirp;      // [nb][mgroups*kR]
    std::vector<float2> bw, bhat, tw, post, uk, f9tw9, f9twa, f9twb, f9bhat, f9ghat, taps32x2;
};

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int align_up(int a, int b) { return ceil_div(a, b) * b; }

// In-place iterative radix-2 FFT, double precision, forward sign.
inline void fft_pow2(std::vector<std::complex<double>>& a) {
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    const double pi = 3.14159265358979323846;
    for (size_t len = 2; len <= n; len <<= 1) {
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const double ang = -2.0 * pi * (double)k / (double)len;
                const std::complex<double> w(std::cos(ang), std::sin(ang));
                const std::complex<double> u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v; a[i + k + len / 2] = u - v;
            }
    }
}

// Recursive mixed-radix FFT (factors 2 and 3), double precision, forward sign; for the one-off table of the
// 4608-point variant.
inline void fft_23(std::vector<std::complex<double>>& a) {
    const size_t n = a.size();
    if (n <= 1) return;
    const size_t r = (n % 2 == 0) ? 2 : 3;
    const size_t m = n / r;
    std::vector<std::vector<std::complex<double>>> sub(r, std::vector<std::complex<double>>(m));
    for (size_t i = 0; i < n; ++i) sub[i % r][i / r] = a[i];
    for (size_t j = 0; j < r; ++j) fft_23(sub[j]);
    const double pi = 3.14159265358979323846;
    for (size_t k = 0; k < n; ++k) {
        std::complex<double> acc(0, 0);
        for (size_t j = 0; j < r; ++j) {
            const double ang = -2.0 * pi * (double)((j * k) % n) / (double)n;
            acc += sub[j][k % m] * std::complex<double>(std::cos(ang), std::sin(ang));
        }
        a[k] = acc;
    }
}

// Tables of the FFT matched filter (cfft_phases.h): g[band][e] = conj(DFT_F(chirp_band))[cf_pass_freq(e)] / F,
// w2[n] = exp(-2 pi i n / F), and per radix-8 pass of stride q the runs W^i, W^2i, W^4i with W = exp(-2 pi i / (8q)).
inline void cf_tables(int nb, int M, const double* chirps, std::vector<zd>& g, std::vector<zd>& w2, std::vector<zd>& tw) {
    const double pi = 3.14159265358979323846;
    g.assign((size_t)nb * kCfF, zmk(0, 0));
    for (int b = 0; b < nb; ++b) {
        std::vector<std::complex<double>> h(kCfF, std::complex<double>(0, 0));
        for (int j = 0; j < M && j < kCfF; ++j) h[j] = chirps[(size_t)b * M + j];
        fft_pow2(h);
        for (int e = 0; e < kCfF; ++e) {
            const std::complex<double> c = std::conj(h[cf_pass_freq(e)]) / (double)kCfF;
            g[(size_t)b * kCfF + e] = zmk(c.real(), c.imag());
        }
    }
    w2.resize(kCfF / 2);
    for (int n = 0; n < kCfF / 2; ++n) {
        const double ang = -2.0 * pi * (double)n / (double)kCfF;
        w2[n] = zmk(std::cos(ang), std::sin(ang));
    }
    tw.assign(kCfTwLen, zmk(0, 0));
    const int qs[3] = {512, 64, 8};
    for (int qi = 0; qi < 3; ++qi) {
        const int q = qs[qi], off = cf_tw_offset(q);
        for (int i = 0; i < q; ++i)
            for (int r = 0; r < 3; ++r) {
                const double ang = -2.0 * pi * (double)(i << r) / (double)(8 * q);
                tw[off + r * q + i] = zmk(std::cos(ang), std::sin(ang));
            }
    }
}

// Frequency index held at LDS position p after the forward passes (radices 8,8,8,4,4; sub-sizes 1024,128,16,4,1).
inline int pass_order_freq(int p) {
    const int m1 = p / 1024, m2 = (p / 128) % 8, m3 = (p / 16) % 8, m4 = (p / 4) % 4, m5 = p % 4;
    return m1 + 8 * m2 + 64 * m3 + 512 * m4 + 2048 * m5;
}

inline float2 mk(double re, double im) { float2 r; r.x = (float)re; r.y = (float)im; return r; }

// Fills the size/layout fields of BandParams that depend only on the configuration.  Returns false when the
// configuration does not fit the kernel (frame too long for one workgroup or for the Bluestein length).
inline bool plan_band(BandParams& p, int N, int M, int K, int nb, int min_delay, int noise_n, int detector,
                      double snr_gate) {
    p.N = N; p.M = M; p.L = N + M - 1; p.half = p.L / 2; p.nb = nb; p.K = K;
    p.min_delay = min_delay; p.detector = detector; p.snr_gate = snr_gate;
    const int npos = p.L - p.half;
    int nn = noise_n;
    if (detector == 0 && min_delay + nn > npos) nn = npos > min_delay ? npos - min_delay : 0;  // numpy slice truncation
    if (detector == 2 && nn > npos) nn = npos;
    p.noise_n = nn;
    p.n2 = N / 2; p.nbins = N / 2 + 1;
    p.kgroups = ceil_div(K, kR); p.mgroups = ceil_div(M, kR);
    if (N % 2 != 0 || N < 4 || M < 1 || K < 1) return false;
    if (N > kNT * kR || p.L > kNT * kR) return false;
    if (2 * p.n2 - 1 > kFftM) return false;
    static_assert(kNT >= kFftM / 16, "one thread per butterfly pair");
    p.sig_len = align_up((ceil_div(p.L, kR) + p.mgroups + 1) * kR, 2);
    p.ypad = K - 1 - (K - 1) / 2;
    p.ybuf_len = align_up((ceil_div(N, kR) + p.kgroups + 1) * kR, 2);
    // LDS layout: [sig | ybuf, sharing memory] [noise window] [coefficient tables] [misc] [reduction scratch]
    int off = 0;
    p.off_sig = off; p.off_y = off;
    off += align_up((p.sig_len > p.ybuf_len ? p.sig_len : p.ybuf_len) * 8, 16);
    p.off_nw = off; off += align_up((nn > 0 ? nn : 1) * 12, 16);   // float64 window + float32 copy
    p.off_ctap = off; off += align_up((p.kgroups + 1) * kR * 8, 16);
    p.off_cchirp = off; off += align_up((p.mgroups + 1) * kR * 8, 16);
    p.off_ctap32 = off; off += align_up((p.kgroups + 1) * kR * 4, 16);
    p.off_cchirp32 = off; off += align_up((p.mgroups + 1) * kR * 4, 16);
    p.off_misc = off; off += align_up((int)sizeof(Misc), 16);
    p.off_u = off;
    // scratch: 4 doubles per reduction slot (best, amp, sumsq, index), then the median counts [parts <= 4][nn][2] ints
    const int cnt_bytes = 4 * (nn > 0 ? nn : 1) * 2 * 4;
    p.smem_bytes = off + align_up(4 * 8 * (kNT / 64) + cnt_bytes, 16);
    p.smem_bytes_host = off + align_up(4 * 8 * kNT + cnt_bytes, 16);
    off = p.smem_bytes;
    p.off_fft = 0;
    p.fft9 = (2 * p.n2 - 1 <= kF9M) ? 1 : 0;
    p.spec_smem_bytes = p.fft9 ? kF9Slots * 8 : kFftM * 8;
    // envelope, smoothed envelope and reduction scratch lie inside the transform's buffer (hilb_phases.h): the
    // 4608-point form keeps the plain spectrum kernel's footprint plus nothing
    {
        const int hilb_end = kHilbOffRed + kHilbRedRows * 8 * (kF9NT / 64) + (int)sizeof(HilbMisc);
        const int f9_bytes = kF9Slots * 8;
        p.spec_hilb_smem_bytes = p.fft9 ? align_up(f9_bytes > hilb_end ? f9_bytes : hilb_end, 16) : kFftM * 8;
    }
    // Pair kernel + exact kernel: median detector, every lag inside one 4608-point cyclic convolution, both frames'
    // float32 FIR inputs inside the transform's buffer
    {
        const int kg3 = ceil_div(K, kB2R), mg3 = ceil_div(M, kB2R);
        const int nband = kB2Seg + M - 1;                                  // band_sig samples one segment reads
        p.b2_ywlen = align_up(nband + 16 + (p.kgroups + 1) * kR, 2);      // what up to 9 outputs per lane in batches of 8 taps read
        p.b2_bwlen = align_up(kB2Seg + 16 + (p.mgroups + 1) * kR, 2);
        (void)kg3; (void)mg3;
        int o = kF9Slots * 8;
        p.b2_off_ct32 = o;
        p.b2_off_red = o;  o += 2 * (kF9NT / 64) * 4 * 8;
        p.b2_off_misc = o; o += 16;
        p.b2_smem_bytes = o;
        p.b2_exact_smem_bytes = 4 * 8 * (p.b2_ywlen + p.b2_bwlen);        // four wavefronts, each with its two windows
        const bool fits = detector == 0 && p.L <= kF9M && p.ybuf_len <= kF9Slots && p.ybuf_len <= 8 * kF9NT && N <= kF9NT * kR &&
                          nn >= 1 && p.half + min_delay + nn <= p.L && nn <= kB2Seg && nn <= 256 &&
                          ceil_div(kB2Seg + M - 1, kB2R) <= 256 && p.b2_exact_smem_bytes <= 64 * 1024;
        p.band2 = fits ? 1 : 0;
    }
    return off <= 160 * 1024;
}

// Reversed taps and chirps, zero-padded to whole groups of kR plus one extra zero group.
inline void build_coef_tables(HostTables& t, int nb, int K, int M, const double* taps, const double* chirps) {
    const int kn = (ceil_div(K, kR) + 1) * kR, mn = (ceil_div(M, kR) + 1) * kR;
    t.taps_rev.assign((size_t)nb * kn, 0.0);
    t.chirp.assign((size_t)nb * mn, 0.0);
    for (int b = 0; b < nb; ++b) {
        for (int j = 0; j < K; ++j) t.taps_rev[(size_t)b * kn + j] = taps[(size_t)b * K + (K - 1 - j)];
        for (int j = 0; j < M; ++j) t.chirp[(size_t)b * mn + j] = chirps[(size_t)b * M + j];
    }
}

// Error bounds of the pair kernel's float32 matched filter (u = 2^-24), per band:
//   float32 FIR: |fl(band_sig[n]) - band_sig[n]| <= (K + 2) u ||taps||_1 max|y|  (y rounded to float32, K-term
//   fused multiply-add sum), which moves a lag by at most ||chirp||_1 times that;
//   float32 transforms: two 4608-point FFTs (passes of radix 9, 8, 8, 8) and the pointwise product move the
//   convolution by at most c u ||x||_2 max|G| in the 2-norm, hence every lag by as much, with c = 200: Higham,
//   Accuracy and Stability of Numerical Algorithms, Thm 24.2 gives log2(n) (mu + gamma_4 (sqrt 2 + mu)) ~ 82 u per
//   transform for correctly rounded twiddles, 2 * 82 u + 4 u for the product, rounded up.
inline void band2_bounds(BandParams& p, const double* taps, const double* chirps) {
    const double u = 5.9604644775390625e-8;
    for (int b = 0; b < p.nb && b < 3; ++b) {
        double t1 = 0.0, c1 = 0.0;
        for (int j = 0; j < p.K; ++j) t1 += std::fabs(taps[(size_t)b * p.K + j]);
        for (int j = 0; j < p.M; ++j) c1 += std::fabs(chirps[(size_t)b * p.M + j]);
        std::vector<std::complex<double>> g(kF9M, std::complex<double>(0, 0));
        for (int j = 0; j < p.M; ++j) g[j] = chirps[(size_t)b * p.M + j];
        fft_23(g);
        double gmax = 0.0;
        for (int k = 0; k < kF9M; ++k) gmax = std::max(gmax, std::abs(g[k]));
        p.eps_fir[b] = (double)(p.K + 2) * u * t1 * c1;
        p.eps_fft[b] = 200.0 * u * gmax;
    }
}

inline void build_tables(HostTables& t, const BandParams& p, const double* taps, const double* chirps) {
    const double pi = 3.14159265358979323846;
    const int n2 = p.n2, N = p.N;
    build_coef_tables(t, p.nb, p.K, p.M, taps, chirps);
    // Bluestein chirp w[n] = exp(-i*pi*n^2/n2); n^2 reduced mod 2*n2 in integers so the phase stays exact
    std::vector<std::complex<double>> w(n2);
    t.bw.assign((size_t)align_up(n2, 2), mk(0, 0));
    for (int n = 0; n < n2; ++n) {
        const long long r = ((long long)n * n) % (2LL * n2);
        const double ang = -pi * (double)r / (double)n2;
        w[n] = std::complex<double>(std::cos(ang), std::sin(ang));
        t.bw[n] = mk(w[n].real(), w[n].imag());
    }
    std::vector<std::complex<double>> bc(kFftM, std::complex<double>(0, 0));
    bc[0] = std::conj(w[0]);
    for (int n = 1; n < n2; ++n) { bc[n] = std::conj(w[n]); bc[kFftM - n] = std::conj(w[n]); }
    fft_pow2(bc);
    // chirp-filter spectrum in pass order, scaled by 1/M, stored so that lane u of the middle pass reads
    // elements (16u + 2jj, 16u + 2jj + 1) at [jj][u][2]
    const int nu = kFftM / 16;
    t.bhat.resize(kFftM);
    for (int pos = 0; pos < kFftM; ++pos) {
        const std::complex<double> v = bc[pass_order_freq(pos)] / (double)kFftM;
        const int u = pos / 16, j = pos % 16;
        t.bhat[(size_t)2 * ((j >> 1) * nu + u) + (j & 1)] = mk(v.real(), v.imag());
    }
    // radix-8 pass twiddles: for q in {1024, 128, 16}: W^i, W^2i, W^4i with W = exp(-2*pi*i/(8q)), i < q
    t.tw.clear();
    const int qs[3] = {1024, 128, 16};
    for (int qi = 0; qi < 3; ++qi)
        for (int m = 1; m <= 4; m *= 2)
            for (int i = 0; i < qs[qi]; ++i) {
                const double ang = -2.0 * pi * (double)(i * m) / (double)(8 * qs[qi]);
                t.tw.push_back(mk(std::cos(ang), std::sin(ang)));
            }
    // post table: A[k] = w[ka] (1 - i u_k)/2, B[k] = conj(w[kb]) (1 + i u_k)/2, u_k = exp(-2*pi*i*k/N),
    // ka = k mod n2, kb = (n2 - k) mod n2
    t.uk.resize((size_t)n2 + 1);
    for (int k = 0; k <= n2; ++k) {
        const double ang = -2.0 * pi * (double)k / (double)N;
        t.uk[k] = mk(std::cos(ang), std::sin(ang));
    }
    if (p.fft9) {
        // tables of the 4608-point variant
        std::vector<std::complex<double>> b9(kF9M, std::complex<double>(0, 0));
        b9[0] = std::conj(w[0]);
        for (int n = 1; n < n2; ++n) { b9[n] = std::conj(w[n]); b9[kF9M - n] = std::conj(w[n]); }
        fft_23(b9);
        t.f9bhat.resize(kF9M);
        for (int pos = 0; pos < kF9M; ++pos) {
            const std::complex<double> v = b9[pass_order_freq9(pos)] / (double)kF9M;
            t.f9bhat[pos] = mk(v.real(), v.imag());
        }
        t.f9tw9.resize(8 * 512);
        for (int m = 1; m <= 8; ++m)
            for (int i = 0; i < 512; ++i) {
                const double ang = -2.0 * pi * (double)(i * m) / (double)kF9M;
                t.f9tw9[(m - 1) * 512 + i] = mk(std::cos(ang), std::sin(ang));
            }
        auto r8 = [&](std::vector<float2>& dst, int q) {
            dst.resize(3 * q);
            for (int e = 0, m = 1; e < 3; ++e, m *= 2)
                for (int i = 0; i < q; ++i) {
                    const double ang = -2.0 * pi * (double)(i * m) / (double)(8 * q);
                    dst[e * q + i] = mk(std::cos(ang), std::sin(ang));
                }
        };
        r8(t.f9twa, 64);
        r8(t.f9twb, 8);
    }
    if (p.band2) {
        // reversed FIR taps in float32, each stored twice: one 8-byte scalar load is the operand pair of a packed
        // multiply-add over the two frames of a pair
        t.taps32x2.resize(t.taps_rev.size());
        for (size_t i = 0; i < t.taps_rev.size(); ++i) t.taps32x2[i] = mk(t.taps_rev[i], t.taps_rev[i]);
        // matched filter as a cyclic convolution of length 4608 with g[t] = chirp[M-1-t]:
        // (band_sig * g)[k] = sum_j chirp[j] band_sig[k - (M-1) + j] = scipy.signal.correlate(band_sig, chirp, 'full')[k]
        t.f9ghat.resize((size_t)p.nb * kF9M);
        for (int b = 0; b < p.nb; ++b) {
            std::vector<std::complex<double>> g(kF9M, std::complex<double>(0, 0));
            for (int j = 0; j < p.M; ++j) g[p.M - 1 - j] = chirps[(size_t)b * p.M + j];
            fft_23(g);
            for (int pos = 0; pos < kF9M; ++pos) {
                const std::complex<double> v = g[pass_order_freq9(pos)] / (double)kF9M;
                t.f9ghat[(size_t)b * kF9M + pos] = mk(v.real(), v.imag());
            }
        }
    }
    t.post.resize((size_t)2 * (n2 + 1));
    for (int k = 0; k <= n2; ++k) {
        const double ang = -2.0 * pi * (double)k / (double)N;
        const std::complex<double> u(std::cos(ang), std::sin(ang)), I(0, 1);
        const int ka = (k == n2) ? 0 : k, kb = (k == 0 || k == n2) ? 0 : n2 - k;
        const std::complex<double> A = w[ka] * (1.0 - I * u) * 0.5, Bc = std::conj(w[kb]) * (1.0 + I * u) * 0.5;
        t.post[2 * k] = mk(A.real(), A.imag());
        t.post[2 * k + 1] = mk(Bc.real(), Bc.imag());
    }
}

}  // namespace echo
