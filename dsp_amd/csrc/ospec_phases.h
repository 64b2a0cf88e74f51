// Spectrum of a long frame whose length no other kernel holds -- ODD lengths whose chirp-z convolution does not fit two
// LDS buffers (long_phases.h, long_spec_odd_kernel: N up to ~6700), even lengths whose half has a large prime factor --
// as that convolution with the transform IN PLACE in one buffer, float32.
//   X[k] = w[k] sum_n (x[n] w[n]) conj(w)[k - n],  w[n] = exp(-i pi n^2 / N),  0 <= k <= N/2  (np.fft.rfft of any length,
//   airSonar_optimized.py:710; N is odd for about half of all listening-window settings)
// as a cyclic convolution of length m >= N + N/2 (products of radices <= 10):
//   a[i] = x[i] w[i] (zero beyond N)  ->  forward passes, decimation in frequency, in place, output in pass order
//   a[e] *= bhat[e]  (DFT_m of the wrapped conj(w), stored in that pass order, divided by m)
//   the conjugate passes back in reverse order: natural order out  ->  X[k] = a[k] w[k]  ->  |X| or 20 log10 |X|
// One buffer of m complex floats with one pad per 8 elements: m <= ~18000, i.e. N up to ~12100, in a compute unit's LDS.
// Plain C++: phases of (tid, nt), so a host build can run them.
#pragma once
#include <complex>
#include <vector>

#include "rcf_phases.h"

namespace echo {

constexpr int kOsNT = 1024;                       // threads per workgroup

typedef cpx<float> zf;
ECHO_HD zf zfmk(float x, float y) { zf r; r.x = x; r.y = y; return r; }
ECHO_HD zf zfmul(zf a, zf b) { return zfmk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
ECHO_HD zf zfmulc(zf a, zf b) { return zfmk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)

// Plans the convolution for a frame length no other spectrum kernel holds (odd N beyond two LDS buffers; even N whose
// half has a prime factor the mixed-radix passes do not take -- the chirp-z form needs nothing of N): fills p.os_*;
// false when none fits.
inline bool plan_ospec(LongParams& p) {
    p.os_use = 0; p.os_m = 0; p.os_npass = 0; p.os_smem = 0;
    if (p.has_spec) return false;
    int best = 0, best_radix[kRcMaxPass], best_np = 0;
    long best_cost = 0;
    const int lo = p.N + p.nbins - 1;
    for (int m = lo; m <= lo + lo / 8; ++m) {
        if ((long)(m + m / 8 + 1) * 8 > 160 * 1024) break;
        int radix[kRcMaxPass], np = 0;
        if (!rc_factor(m, radix, &np)) continue;
        const long cost = (long)np * m;
        if (!best || cost < best_cost) { best = m; best_cost = cost; best_np = np; for (int i = 0; i < np; ++i) best_radix[i] = radix[i]; }
    }
    if (!best) return false;
    p.os_use = 1; p.os_m = best; p.os_npass = best_np;
    int roff = 0;
    for (int i = 0; i < best_np; ++i) { p.os_radix[i] = best_radix[i]; p.os_root_off[i] = roff; roff += best_radix[i]; }
    p.os_smem = (int)(((long)(best + best / 8 + 1) * 8 + 15) & ~15L);
    return true;
}

// frequency held at position e after the forward passes
inline int os_pass_freq(const LongParams& p, int e) {
    int S = p.os_m, w = 1, f = 0;
    for (int ps = 0; ps < p.os_npass; ++ps) {
        const int r = p.os_radix[ps], q = S / r;
        f += ((e / q) % r) * w;
        w *= r; S = q;
    }
    return f;
}

// Tables: tw = the pass twiddles (per pass the runs exp(-2 pi i (i k) / S), k = 1..r-1, i < S / r), roots (per pass
// the r roots of unity), bw[n] = w[n] (N entries), bhat[e] = DFT_m(b)[os_pass_freq(e)] / m with b[j] = conj(w)[|j|]
// for -(N-1) <= j <= N/2 wrapped to length m.
inline void ospec_tables(LongParams& p, std::vector<zf>& tw, std::vector<zf>& roots, std::vector<zf>& bw, std::vector<zf>& bhat) {
    const long double tau = 6.283185307179586476925286766559005768L;
    const int N = p.N, m = p.os_m;
    tw.clear(); roots.clear();
    for (int ps = 0, S = m; ps < p.os_npass; ++ps) {
        const int r = p.os_radix[ps], q = S / r;
        p.os_tw_off[ps] = (int)tw.size();
        for (int k = 1; k < r; ++k)
            for (int i = 0; i < q; ++i) {
                const long double a = -tau * (long double)((long)i * k % S) / (long double)S;
                tw.push_back(zfmk((float)cosl(a), (float)sinl(a)));
            }
        for (int t = 0; t < r; ++t) roots.push_back(zfmk((float)cosl(-tau * t / r), (float)sinl(-tau * t / r)));
        S = q;
    }
    std::vector<std::complex<long double>> w(N), b(m, std::complex<long double>(0, 0));
    bw.resize(N);
    for (int n = 0; n < N; ++n) {
        const long long r = ((long long)n * n) % (2LL * N);          // n^2 mod 2N keeps the phase exact
        const long double a = -(tau / 2) * (long double)r / (long double)N;
        w[n] = std::complex<long double>(cosl(a), sinl(a));
        bw[n] = zfmk((float)w[n].real(), (float)w[n].imag());
    }
    for (int j = 0; j < p.nbins; ++j) b[j] = std::conj(w[j]);
    for (int j = 1; j < N; ++j) b[m - j] = std::conj(w[j]);
    fft_smooth(b);
    bhat.resize(m);
    for (int e = 0; e < m; ++e) {
        const std::complex<long double> v = b[os_pass_freq(p, e)] / (long double)m;
        bhat[e] = zfmk((float)v.real(), (float)v.imag());
    }
}

ECHO_HD void ph_os_load(int tid, int nt, int blk, const LongParams& p, unsigned char* smem) {
    zf* f = reinterpret_cast<zf*>(smem);
    const double* x = p.spec_in + (size_t)blk * p.N;
    for (int n = tid; n < p.os_m; n += nt) {
        zf a = zfmk(0.f, 0.f);
        if (n < p.N) { const float v = (float)x[n]; const zf w = p.sbw[n]; a = zfmk(v * w.x, v * w.y); }
        f[zpos(n)] = a;
    }
}

template <int R, bool INV>
ECHO_HD void os_pass(int tid, int nt, int S, const zf* roots, const zf* tw, const LongParams& p, unsigned char* smem) {
    zf* f = reinterpret_cast<zf*>(smem);
    const int q = S / R;
    for (int t = tid; t < p.os_m / R; t += nt) {
        const int b = t / q, i = t - b * q, base = b * S + i;
        zf v[R];
        if (INV) {
            v[0] = f[zpos(base)];
#pragma unroll
            for (int k = 1; k < R; ++k) v[k] = zfmulc(f[zpos(base + k * q)], tw[(k - 1) * q + i]);
        } else {
#pragma unroll
            for (int j = 0; j < R; ++j) v[j] = f[zpos(base + j * q)];
        }
        gfft_bfly<float, R>(v, roots, INV);
        if (INV) {
#pragma unroll
            for (int j = 0; j < R; ++j) f[zpos(base + j * q)] = v[j];
        } else {
            f[zpos(base)] = v[0];
#pragma unroll
            for (int k = 1; k < R; ++k) f[zpos(base + k * q)] = zfmul(v[k], tw[(k - 1) * q + i]);
        }
    }
}

template <bool INV>
ECHO_HD void ph_os_pass(int tid, int nt, int ps, const LongParams& p, unsigned char* smem) {
    int S = p.os_m;
    for (int i = 0; i < ps; ++i) S /= p.os_radix[i];
    const zf* roots = p.os_roots + p.os_root_off[ps];
    const zf* tw = p.os_tw + p.os_tw_off[ps];
    switch (p.os_radix[ps]) {
        case 2: os_pass<2, INV>(tid, nt, S, roots, tw, p, smem); break;
        case 3: os_pass<3, INV>(tid, nt, S, roots, tw, p, smem); break;
        case 4: os_pass<4, INV>(tid, nt, S, roots, tw, p, smem); break;
        case 5: os_pass<5, INV>(tid, nt, S, roots, tw, p, smem); break;
        case 6: os_pass<6, INV>(tid, nt, S, roots, tw, p, smem); break;
        case 7: os_pass<7, INV>(tid, nt, S, roots, tw, p, smem); break;
        case 8: os_pass<8, INV>(tid, nt, S, roots, tw, p, smem); break;
        case 9: os_pass<9, INV>(tid, nt, S, roots, tw, p, smem); break;
        case 10: os_pass<10, INV>(tid, nt, S, roots, tw, p, smem); break;
        default: break;
    }
}

ECHO_HD void ph_os_mul(int tid, int nt, const LongParams& p, unsigned char* smem) {
    zf* f = reinterpret_cast<zf*>(smem);
    for (int e = tid; e < p.os_m; e += nt) f[zpos(e)] = zfmul(f[zpos(e)], p.os_bhat[e]);
}

ECHO_HD void ph_os_store(int tid, int nt, int blk, const LongParams& p, unsigned char* smem) {
    const zf* f = reinterpret_cast<const zf*>(smem);
    for (int k = tid; k < p.nbins; k += nt) {
        const zf x = zfmul(f[zpos(k)], p.sbw[k]);
        float mag = sqrtf(x.x * x.x + x.y * x.y);
        if (p.spec_db) mag = 20.0f * log10f(fmaxf(mag, 1e-12f));
        p.spec_out[(size_t)blk * p.nbins + k] = mag;
    }
}

#define ECHO_OSPEC_SEQUENCE(PHASE, SYNC, p, smem, blk, nt)                                                   \
    PHASE(ph_os_load(tid, nt, blk, p, smem)) SYNC                                                            \
    for (int ps_ = 0; ps_ < (p).os_npass; ++ps_) { PHASE(ph_os_pass<false>(tid, nt, ps_, p, smem)) SYNC }   \
    PHASE(ph_os_mul(tid, nt, p, smem)) SYNC                                                                  \
    for (int ps_ = (p).os_npass - 1; ps_ >= 0; --ps_) { PHASE(ph_os_pass<true>(tid, nt, ps_, p, smem)) SYNC } \
    PHASE(ph_os_store(tid, nt, blk, p, smem))

}  // namespace echo
