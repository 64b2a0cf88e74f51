// Mixed-radix complex FFT of arbitrary composite length n inside one workgroup's LDS (Stockham autosort, natural
// order in and out), float or double.  Used where the length has no large power-of-two structure: the Hilbert
// transform of the 4556-sample correlation (n = 2278 = 67*17*2, float64) and the spectrum of long frames
// (n = 4410 = 7*7*5*3*3*2, float32).  Each pass of radix r is evaluated output by output as an r-term sum with the
// r-th roots of unity (cost n*r complex multiply-adds per pass), four outputs of one butterfly per work item so an
// input element is read once per four outputs; the inter-pass twiddles W_n^(s*k*m) are applied when a pass stores
// its outputs.  Plain C++: phases of (tid, nt) as in band_phases.h.
#pragma once
#include <math.h>

#include <vector>

#include "band_phases.h"

namespace echo {

template <class T> struct cpx { T x, y; };

constexpr int kGfftMaxPass = 8;
constexpr int kGfftMaxRadix = 128;

struct GfftPlan {
    int n, npass;
    int radix[kGfftMaxPass];     // largest first: the first pass needs no twiddles
    int lp[kGfftMaxPass];        // product of the radices of the earlier passes
    int root_off[kGfftMaxPass];  // offset of the pass's r roots exp(-2*pi*i*t/r) in the roots table
    int nroots;
};

inline bool gfft_plan(GfftPlan& pl, int n) {
    pl.n = n; pl.npass = 0; pl.nroots = 0;
    if (n < 1) return false;
    int fac[32], nf = 0, rem = n;
    for (int p = 2; p * p <= rem; ++p)
        while (rem % p == 0) { if (nf == 32) return false; fac[nf++] = p; rem /= p; }
    if (rem > 1) { if (nf == 32) return false; fac[nf++] = rem; }
    // merge pairs of 2 into 4 (fewer passes); then order by decreasing radix
    int twos = 0, out[32], no = 0;
    for (int i = 0; i < nf; ++i) { if (fac[i] == 2) ++twos; else out[no++] = fac[i]; }
    for (; twos >= 2; twos -= 2) out[no++] = 4;
    if (twos) out[no++] = 2;
    for (int i = 0; i < no; ++i)
        for (int j = i + 1; j < no; ++j)
            if (out[j] > out[i]) { const int t = out[i]; out[i] = out[j]; out[j] = t; }
    if (no > kGfftMaxPass || (no > 0 && out[0] > kGfftMaxRadix)) return false;
    if (no == 0) { out[no++] = 1; }
    int lp = 1;
    for (int i = 0; i < no; ++i) {
        pl.radix[i] = out[i]; pl.lp[i] = lp; pl.root_off[i] = pl.nroots;
        pl.nroots += out[i]; lp *= out[i];
    }
    pl.npass = no;
    return true;
}

// roots[root_off[p] + t] = exp(-2*pi*i*t/radix[p]);  wn[e] = exp(-2*pi*i*e/n)
template <class T> inline void gfft_tables(const GfftPlan& pl, std::vector<cpx<T>>& roots, std::vector<cpx<T>>& wn) {
    const double pi = 3.14159265358979323846;
    roots.resize(pl.nroots);
    for (int p = 0; p < pl.npass; ++p)
        for (int t = 0; t < pl.radix[p]; ++t) {
            const double a = -2.0 * pi * (double)t / (double)pl.radix[p];
            roots[pl.root_off[p] + t] = cpx<T>{(T)cos(a), (T)sin(a)};
        }
    wn.resize(pl.n);
    for (int e = 0; e < pl.n; ++e) {
        const double a = -2.0 * pi * (double)e / (double)pl.n;
        wn[e] = cpx<T>{(T)cos(a), (T)sin(a)};
    }
}

ECHO_HD float gf_fma(float a, float b, float c) { return fmaf(a, b, c); }
ECHO_HD double gf_fma(double a, double b, double c) { return fma(a, b, c); }

// One pass.  in/out: LDS buffers of n elements (distinct).  roots: this pass's r roots (LDS or global).
// wn: global table of n twiddles, used to pre-multiply the outputs for the next pass.  inv: conjugate transform.
template <class T>
ECHO_HD void gfft_pass(int tid, int nt, const GfftPlan& pl, int ps, const cpx<T>* in, cpx<T>* out, const cpx<T>* roots,
                       const cpx<T>* wn, bool inv) {
    constexpr int Q = 4;
    const int r = pl.radix[ps], Lp = pl.lp[ps], m = pl.n / (Lp * r);
    const int nqb = (r + Q - 1) / Q, cols = Lp * m, items = nqb * cols;
    const bool last = ps + 1 >= pl.npass;
    const int r2 = last ? 1 : pl.radix[ps + 1];
    const int m2 = last ? 1 : pl.n / (pl.lp[ps + 1] * r2);
    for (int it = tid; it < items; it += nt) {
        const int qb = it / cols, c = it - qb * cols;
        const int k = c / m, j = c - k * m;
        const int q0 = qb * Q;
        int idx[Q], step[Q];
        cpx<T> acc[Q];
#pragma unroll
        for (int u = 0; u < Q; ++u) {
            idx[u] = 0; step[u] = q0 + u < r ? q0 + u : 0;
            acc[u].x = (T)0; acc[u].y = (T)0;
        }
        const cpx<T>* src = in + k * (m * r) + j;
        for (int s = 0; s < r; ++s) {
            const cpx<T> v = src[m * s];
#pragma unroll
            for (int u = 0; u < Q; ++u) {
                const cpx<T> w = roots[idx[u]];
                const T wy = inv ? -w.y : w.y;
                acc[u].x = gf_fma(v.x, w.x, acc[u].x); acc[u].x = gf_fma(-v.y, wy, acc[u].x);
                acc[u].y = gf_fma(v.x, wy, acc[u].y);  acc[u].y = gf_fma(v.y, w.x, acc[u].y);
                idx[u] += step[u];
                if (idx[u] >= r) idx[u] -= r;
            }
        }
#pragma unroll
        for (int u = 0; u < Q; ++u) {
            if (q0 + u >= r) continue;
            const int p = (k + Lp * (q0 + u)) * m + j;
            cpx<T> a = acc[u];
            if (!last) {
                // the next pass reads position p as (k2, s2, j2) and needs it multiplied by W_n^(s2*k2*m2)
                const int s2 = (p / m2) % r2, k2 = p / (m2 * r2);
                const cpx<T> w = wn[s2 * k2 * m2];
                const T wy = inv ? -w.y : w.y;
                const T re = a.x * w.x - a.y * wy, im = a.x * wy + a.y * w.x;
                a.x = re; a.y = im;
            }
            out[p] = a;
        }
    }
}

}  // namespace echo
