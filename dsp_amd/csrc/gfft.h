// Mixed-radix complex FFT of arbitrary composite length n inside one workgroup's LDS (Stockham autosort, natural
// order in and out), float or double.  Used where the length has no large power-of-two structure: the Hilbert
// transform of the 4556-sample correlation (n = 2278 = 67*17*2, float64) and the spectrum of long frames
// (n = 4410 = 7*7*5*3*3*2, float32).  Each pass of radix r is evaluated output by output as an r-term sum with the
// r-th roots of unity, outputs q and r-q sharing their sums (n*r/2 complex multiply-adds per pass); the inter-pass
// twiddles W_n^(s*k*m) are applied when a pass stores
// its outputs.  Plain C++: phases of (tid, nt) as in band_phases.h.
#pragma once
#include <math.h>

#include <vector>

#include "band_phases.h"

namespace echo {

// 2 * sizeof(T)-aligned: a complex double in LDS is then one 16-byte access (ds_read_b128, 256 bytes per clock) instead of a
// pair of 8-byte ones (ds_read2_b64, half that rate on gfx950)
template <class T> struct alignas(2 * sizeof(T)) cpx { T x, y; };

constexpr int kGfftMaxPass = 8;
constexpr int kGfftMaxRadix = 128;

struct GfftPlan {
    int n, npass;
    int radix[kGfftMaxPass];     // largest first: the first pass needs no twiddles
    int lp[kGfftMaxPass];        // product of the radices of the earlier passes
    int root_off[kGfftMaxPass];  // offset of the pass's r roots exp(-2*pi*i*t/r) in the roots table
    int pairs[kGfftMaxPass];     // conjugate output pairs per work item (1..3), chosen for nt threads
    int per_lane[kGfftMaxPass];  // 1: items are (pair block, butterfly) per lane; 0: (pair block, 64 butterflies) per wavefront
    int bfly[kGfftMaxPass];      // 1: the pass runs as in-register butterflies (radix <= kGfftMaxBfly), one per lane
    int nroots;
};
constexpr int kGfftMaxBfly = 10;

// merge_small: small prime factors are multiplied into radices of up to kGfftMaxBfly (3*3 -> 9, 2*5 -> 10, 2*2 -> 4 ...)
// and every pass of such a radix runs as in-register butterflies -- fewer passes, each input read once, about half the
// arithmetic of the r-term sums.  Off by default: the float64 transforms of the envelope branch keep the summation order
// their parity runs were made with.
inline bool gfft_plan(GfftPlan& pl, int n, int nt = 512, bool merge_small = false) {
    pl.n = n; pl.npass = 0; pl.nroots = 0;
    if (n < 1) return false;
    int fac[32], nf = 0, rem = n;
    for (int p = 2; p * p <= rem; ++p)
        while (rem % p == 0) { if (nf == 32) return false; fac[nf++] = p; rem /= p; }
    if (rem > 1) { if (nf == 32) return false; fac[nf++] = rem; }
    // merge pairs of 2 into 4 (fewer passes); then order by decreasing radix
    int twos = 0, out[32], no = 0;
    if (merge_small) {
        // largest factor first, each multiplied by the largest remaining factors that keep the product <= kGfftMaxBfly
        bool used[32];
        for (int i = 0; i < nf; ++i) used[i] = false;
        for (int i = nf - 1; i >= 0; --i) {
            if (used[i]) continue;
            used[i] = true;
            int prod = fac[i];
            if (prod <= kGfftMaxBfly)
                for (int j = i - 1; j >= 0; --j)
                    if (!used[j] && prod * fac[j] <= kGfftMaxBfly) { prod *= fac[j]; used[j] = true; }
            out[no++] = prod;
        }
    } else {
    for (int i = 0; i < nf; ++i) { if (fac[i] == 2) ++twos; else out[no++] = fac[i]; }
    for (; twos >= 2; twos -= 2) out[no++] = 4;
    if (twos) out[no++] = 2;
    }
    for (int i = 0; i < no; ++i)
        for (int j = i + 1; j < no; ++j)
            if (out[j] > out[i]) { const int t = out[i]; out[i] = out[j]; out[j] = t; }
    if (no > kGfftMaxPass || (no > 0 && out[0] > kGfftMaxRadix)) return false;
    if (no == 0) { out[no++] = 1; }
    int lp = 1;
    for (int i = 0; i < no; ++i) {
        pl.radix[i] = out[i]; pl.lp[i] = lp; pl.root_off[i] = pl.nroots;
        pl.nroots += out[i]; lp *= out[i];
        // pairs per wavefront item: fewest rounds over the nt/64 wavefronts, weighted by the work per item
        const int npair = out[i] / 2 + 1, nchunk = (n / out[i] + 63) / 64, nw = nt / 64 > 0 ? nt / 64 : 1;
        int best = 1, best_cost = 1 << 30;
        for (int P = 1; P <= 3; ++P) {
            const int items = ((npair + P - 1) / P) * nchunk;
            const int cost = ((items + nw - 1) / nw) * (P + 1);      // + 1: the input read and loop overhead
            if (cost < best_cost) { best_cost = cost; best = P; }
        }
        pl.pairs[i] = best; pl.per_lane[i] = 0;
        pl.bfly[i] = (merge_small && out[i] >= 2 && out[i] <= kGfftMaxBfly) ? 1 : 0;
        // few butterflies per pass (n/r well below a multiple of 64) leave lanes idle in the wavefront mapping:
        // then every lane takes its own (pair block, butterfly) and keeps its own root indices
        const int cols = n / out[i];
        if (cols * 4 < nchunk * 64 * 3) {
            best = 1; best_cost = 1 << 30;
            for (int P = 1; P <= 3; ++P) {
                const int items = ((npair + P - 1) / P) * cols;
                const int cost = ((items + nt - 1) / nt) * (2 * P + 1);   // index arithmetic per pair is vector work here
                if (cost < best_cost) { best_cost = cost; best = P; }
            }
            pl.pairs[i] = best; pl.per_lane[i] = 1;
        }
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

// One pass.  in/out: LDS buffers of n elements (distinct).  roots: this pass's r roots (LDS).  wn: global table of
// n twiddles, used to pre-multiply the outputs for the next pass.  inv: conjugate transform.
//
// A pass of radix r is the product of the r x r root matrix with the r x (n/r) matrix of its butterflies' inputs.
// Outputs q and r-q of a butterfly use conjugate roots, so they share four real sums over the inputs v_s and roots
// w = W_r^(s*q):  A = sum v.x w.x, B = sum v.y w.y, C = sum v.x w.y, D = sum v.y w.x;
//   out[q] = (A - B) + i (C + D),  out[r-q] = (A + B) + i (D - C)   (swapped for the inverse transform);
// q = 0 (and q = r/2 for even r) pair with themselves.  A wavefront takes P such pairs for 64 butterflies (one per
// lane): the root indices are the same for all its lanes, so their arithmetic is scalar and the root reads are
// LDS broadcasts; per input element a lane reads one input and issues 4P multiply-adds.
#if defined(__HIP_DEVICE_COMPILE__)
#define ECHO_WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#else
#define ECHO_WAVE_UNIFORM(x) (x)
#endif

template <class T, int P>
ECHO_HD void gfft_pass_p(int tid, int nt, const GfftPlan& pl, int ps, const cpx<T>* in, cpx<T>* out, const cpx<T>* roots,
                         const cpx<T>* wn, bool inv) {
    const int r = pl.radix[ps], Lp = pl.lp[ps], m = pl.n / (Lp * r);
    const int npair = r / 2 + 1, npb = (npair + P - 1) / P, cols = Lp * m, nchunk = (cols + 63) / 64;
    const bool last = ps + 1 >= pl.npass;
    const int m2 = last ? 1 : pl.n / (pl.lp[ps + 1] * pl.radix[ps + 1]);
    const int wave = ECHO_WAVE_UNIFORM(tid >> 6), lane = tid & 63, nw = nt >> 6;
    for (int wi = wave; wi < npb * nchunk; wi += nw) {
        const int pb = wi / nchunk, chunk = wi - pb * nchunk;
        const int craw = chunk * 64 + lane;
        const bool active = craw < cols;
        const int c = active ? craw : cols - 1;
        const int k = c / m, j = c - k * m;
        const int t0 = pb * P;
        int idx[P], step[P];
        T A[P], B[P], C[P], D[P];
        cpx<T> wa[P], wb[P];
#pragma unroll
        for (int u = 0; u < P; ++u) {
            step[u] = t0 + u < npair ? t0 + u : 0;
            A[u] = B[u] = C[u] = D[u] = (T)0;
            wa[u] = roots[0];                               // s = 0
            wb[u] = roots[step[u]];                         // s = 1 (step < r)
            idx[u] = 2 * step[u];                           // s = 2
            if (idx[u] >= r) idx[u] -= r;
        }
        // twiddle the next pass wants on this butterfly's outputs: position p = (k + Lp*q)*m + j is read there as
        // (k2, s2, j2) with k2 = k + Lp*q and s2 = j / m2, and needs W_n^(s2*k2*m2); fetched now, used after the loop
        const int s2m2 = last ? 0 : (j / m2) * m2;
        cpx<T> tw[2 * P];
#pragma unroll
        for (int u = 0; u < P; ++u)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int q = step[u], qe = (e && q) ? r - q : q;
                tw[2 * u + e] = last ? cpx<T>{(T)1, (T)0} : wn[s2m2 * (k + Lp * qe)];
            }
        const cpx<T>* src = in + k * (m * r) + j;
        // two elements in flight: the reads that refill one (input element and roots) are issued right after its
        // multiply-adds and have the other's multiply-adds to complete
        cpx<T> va = src[0], vb = src[r > 1 ? m : 0];
        int s = 0;
        for (; s + 1 < r; s += 2) {
#pragma unroll
            for (int u = 0; u < P; ++u) {
                A[u] = gf_fma(va.x, wa[u].x, A[u]); B[u] = gf_fma(va.y, wa[u].y, B[u]);
                C[u] = gf_fma(va.x, wa[u].y, C[u]); D[u] = gf_fma(va.y, wa[u].x, D[u]);
            }
            va = src[m * (s + 2 < r ? s + 2 : r - 1)];
#pragma unroll
            for (int u = 0; u < P; ++u) {
                wa[u] = roots[idx[u]];
                idx[u] += step[u];
                if (idx[u] >= r) idx[u] -= r;
            }
#pragma unroll
            for (int u = 0; u < P; ++u) {
                A[u] = gf_fma(vb.x, wb[u].x, A[u]); B[u] = gf_fma(vb.y, wb[u].y, B[u]);
                C[u] = gf_fma(vb.x, wb[u].y, C[u]); D[u] = gf_fma(vb.y, wb[u].x, D[u]);
            }
            vb = src[m * (s + 3 < r ? s + 3 : r - 1)];
#pragma unroll
            for (int u = 0; u < P; ++u) {
                wb[u] = roots[idx[u]];
                idx[u] += step[u];
                if (idx[u] >= r) idx[u] -= r;
            }
        }
        if (s < r) {                                        // odd radix: last element
#pragma unroll
            for (int u = 0; u < P; ++u) {
                A[u] = gf_fma(va.x, wa[u].x, A[u]); B[u] = gf_fma(va.y, wa[u].y, B[u]);
                C[u] = gf_fma(va.x, wa[u].y, C[u]); D[u] = gf_fma(va.y, wa[u].x, D[u]);
            }
        }
        if (!active) continue;
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const int q = t0 + u;
            if (q >= npair) continue;
            cpx<T> o[2];
            o[0].x = A[u] - B[u]; o[0].y = C[u] + D[u];       // root w
            o[1].x = A[u] + B[u]; o[1].y = D[u] - C[u];       // root conj(w)
            const int nout = (q == 0 || r - q == q) ? 1 : 2;
            for (int e = 0; e < nout; ++e) {
                cpx<T> a = o[inv ? 1 - e : e];
                if (!last) {
                    const cpx<T> t = tw[2 * u + e];
                    const T wy = inv ? -t.y : t.y;
                    const T re = a.x * t.x - a.y * wy, im = a.x * wy + a.y * t.x;
                    a.x = re; a.y = im;
                }
                out[(k + Lp * (e ? r - q : q)) * m + j] = a;
            }
        }
    }
}

// Per-lane mapping of the same pass: work item = (pair block, butterfly), consecutive lanes take consecutive
// butterflies and run into the next pair block, so all lanes are busy when a pass has few butterflies.  Root
// indices differ between lanes (vector arithmetic, LDS gathers of a few distinct addresses).
template <class T, int P>
ECHO_HD void gfft_pass_lane(int tid, int nt, const GfftPlan& pl, int ps, const cpx<T>* in, cpx<T>* out,
                            const cpx<T>* roots, const cpx<T>* wn, bool inv) {
    const int r = pl.radix[ps], Lp = pl.lp[ps], m = pl.n / (Lp * r);
    const int npair = r / 2 + 1, npb = (npair + P - 1) / P, cols = Lp * m, items = npb * cols;
    const bool last = ps + 1 >= pl.npass;
    const int m2 = last ? 1 : pl.n / (pl.lp[ps + 1] * pl.radix[ps + 1]);
    for (int it = tid; it < items; it += nt) {
        const int pb = it / cols, c = it - pb * cols;
        const int k = c / m, j = c - k * m;
        const int t0 = pb * P;
        unsigned idx[P], step[P];
        T A[P], B[P], C[P], D[P];
        cpx<T> wa[P], wb[P];
#pragma unroll
        for (int u = 0; u < P; ++u) {
            step[u] = t0 + u < npair ? (unsigned)(t0 + u) : 0u;
            A[u] = B[u] = C[u] = D[u] = (T)0;
            wa[u] = roots[0];
            wb[u] = roots[step[u]];
            idx[u] = 2 * step[u];
            idx[u] = idx[u] < idx[u] - (unsigned)r ? idx[u] : idx[u] - (unsigned)r;   // min(i, i - r) as unsigned = i mod r
        }
        const int s2m2 = last ? 0 : (j / m2) * m2;
        cpx<T> tw[2 * P];
#pragma unroll
        for (int u = 0; u < P; ++u)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int q = (int)step[u], qe = (e && q) ? r - q : q;
                tw[2 * u + e] = last ? cpx<T>{(T)1, (T)0} : wn[s2m2 * (k + Lp * qe)];
            }
        const cpx<T>* src = in + k * (m * r) + j;
        cpx<T> va = src[0], vb = src[r > 1 ? m : 0];
        int s = 0;
        for (; s + 1 < r; s += 2) {
#pragma unroll
            for (int u = 0; u < P; ++u) {
                A[u] = gf_fma(va.x, wa[u].x, A[u]); B[u] = gf_fma(va.y, wa[u].y, B[u]);
                C[u] = gf_fma(va.x, wa[u].y, C[u]); D[u] = gf_fma(va.y, wa[u].x, D[u]);
            }
            va = src[m * (s + 2 < r ? s + 2 : r - 1)];
#pragma unroll
            for (int u = 0; u < P; ++u) {
                wa[u] = roots[idx[u]];
                idx[u] += step[u];
                idx[u] = idx[u] < idx[u] - (unsigned)r ? idx[u] : idx[u] - (unsigned)r;
            }
#pragma unroll
            for (int u = 0; u < P; ++u) {
                A[u] = gf_fma(vb.x, wb[u].x, A[u]); B[u] = gf_fma(vb.y, wb[u].y, B[u]);
                C[u] = gf_fma(vb.x, wb[u].y, C[u]); D[u] = gf_fma(vb.y, wb[u].x, D[u]);
            }
            vb = src[m * (s + 3 < r ? s + 3 : r - 1)];
#pragma unroll
            for (int u = 0; u < P; ++u) {
                wb[u] = roots[idx[u]];
                idx[u] += step[u];
                idx[u] = idx[u] < idx[u] - (unsigned)r ? idx[u] : idx[u] - (unsigned)r;
            }
        }
        if (s < r) {
#pragma unroll
            for (int u = 0; u < P; ++u) {
                A[u] = gf_fma(va.x, wa[u].x, A[u]); B[u] = gf_fma(va.y, wa[u].y, B[u]);
                C[u] = gf_fma(va.x, wa[u].y, C[u]); D[u] = gf_fma(va.y, wa[u].x, D[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const int q = t0 + u;
            if (q >= npair) continue;
            cpx<T> o[2];
            o[0].x = A[u] - B[u]; o[0].y = C[u] + D[u];
            o[1].x = A[u] + B[u]; o[1].y = D[u] - C[u];
            const int nout = (q == 0 || r - q == q) ? 1 : 2;
            for (int e = 0; e < nout; ++e) {
                cpx<T> a = o[inv ? 1 - e : e];
                if (!last) {
                    const cpx<T> t = tw[2 * u + e];
                    const T wy = inv ? -t.y : t.y;
                    const T re = a.x * t.x - a.y * wy, im = a.x * wy + a.y * t.x;
                    a.x = re; a.y = im;
                }
                out[(k + Lp * (e ? r - q : q)) * m + j] = a;
            }
        }
    }
}

// ---- small radices as in-register butterflies.
// One lane per butterfly: the R inputs are read once, X[q] = sum_s v[s] W_R^(s q) is evaluated through the symmetric
// halves t[s] = v[s] + v[R-s], u[s] = v[s] - v[R-s]:
//   X[q], X[R-q] = (v[0] + sum_s cos(2 pi q s / R) t[s] (+ (-1)^q v[R/2])) -/+ i sum_s sin(2 pi q s / R) u[s]
// (forward; the inverse transform swaps the pair).  The cosines and sines come from the pass's root table
// (roots[k] = exp(-2 pi i k / R)), indexed at compile time.
template <int R> struct GfftFold {                      // (q s) mod R folded into 1 .. R/2 with the signs of cos and sin
    static constexpr int k(int q, int s) { return ((q * s) % R) <= R / 2 ? (q * s) % R : R - (q * s) % R; }
    static constexpr bool neg_sin(int q, int s) { return ((q * s) % R) > R / 2; }
};

template <class T, int R>
ECHO_HD void gfft_bfly(cpx<T> (&v)[R], const cpx<T>* roots, bool inv) {
    constexpr int H = (R - 1) / 2;                      // conjugate pairs
    constexpr bool even = R % 2 == 0;
    T c[H + 1], sn[H + 1];
#pragma unroll
    for (int k = 1; k <= H; ++k) { c[k] = roots[k].x; sn[k] = -roots[k].y; }      // cos, sin of 2 pi k / R
    cpx<T> t[H + 1], u[H + 1];
#pragma unroll
    for (int s = 1; s <= H; ++s) {
        t[s] = cpx<T>{v[s].x + v[R - s].x, v[s].y + v[R - s].y};
        u[s] = cpx<T>{v[s].x - v[R - s].x, v[s].y - v[R - s].y};
    }
    cpx<T> o[R];
    o[0] = v[0];
#pragma unroll
    for (int s = 1; s <= H; ++s) { o[0].x += t[s].x; o[0].y += t[s].y; }
    if (even) { o[0].x += v[R / 2].x; o[0].y += v[R / 2].y; }
#pragma unroll
    for (int q = 1; q <= H; ++q) {
        cpx<T> P = v[0], Q{(T)0, (T)0};
        if (even) {
            if (q % 2) { P.x -= v[R / 2].x; P.y -= v[R / 2].y; } else { P.x += v[R / 2].x; P.y += v[R / 2].y; }
        }
#pragma unroll
        for (int s = 1; s <= H; ++s) {
            const int k = GfftFold<R>::k(q, s);
            if (k == 0) {                               // q s is a multiple of R: cos = 1, sin = 0
                P.x += t[s].x; P.y += t[s].y;
                continue;
            }
            if (even && k == R / 2) {                   // cos = -1, sin = 0
                P.x -= t[s].x; P.y -= t[s].y;
                continue;
            }
            P.x = gf_fma(c[k], t[s].x, P.x); P.y = gf_fma(c[k], t[s].y, P.y);
            if (GfftFold<R>::neg_sin(q, s)) { Q.x = gf_fma(-sn[k], u[s].x, Q.x); Q.y = gf_fma(-sn[k], u[s].y, Q.y); }
            else                            { Q.x = gf_fma(sn[k], u[s].x, Q.x);  Q.y = gf_fma(sn[k], u[s].y, Q.y); }
        }
        // forward: X[q] = P - i Q, X[R-q] = P + i Q
        const cpx<T> a{P.x + Q.y, P.y - Q.x}, b{P.x - Q.y, P.y + Q.x};
        o[q] = inv ? b : a; o[R - q] = inv ? a : b;
    }
    if (even) {                                         // q = R/2: alternating sum
        cpx<T> a = v[0];
#pragma unroll
        for (int s = 1; s <= H; ++s) {
            if (s % 2) { a.x -= t[s].x; a.y -= t[s].y; } else { a.x += t[s].x; a.y += t[s].y; }
        }
        if ((R / 2) % 2) { a.x -= v[R / 2].x; a.y -= v[R / 2].y; } else { a.x += v[R / 2].x; a.y += v[R / 2].y; }
        o[R / 2] = a;
    }
#pragma unroll
    for (int q = 0; q < R; ++q) v[q] = o[q];
}

template <class T, int R>
ECHO_HD void gfft_pass_bfly(int tid, int nt, const GfftPlan& pl, int ps, const cpx<T>* in, cpx<T>* out,
                            const cpx<T>* roots, const cpx<T>* wn, bool inv) {
    const int Lp = pl.lp[ps], m = pl.n / (Lp * R), cols = Lp * m;
    const bool last = ps + 1 >= pl.npass;
    const int m2 = last ? 1 : pl.n / (pl.lp[ps + 1] * pl.radix[ps + 1]);
    for (int c = tid; c < cols; c += nt) {
        const int k = c / m, j = c - k * m;
        const cpx<T>* src = in + k * (m * R) + j;
        cpx<T> v[R];
#pragma unroll
        for (int s = 0; s < R; ++s) v[s] = src[s * m];
        // the twiddles the next pass wants on the outputs (see gfft_pass_p), fetched before the arithmetic
        const int s2m2 = last ? 0 : (j / m2) * m2;
        cpx<T> tw[R];
        if (!last) {
#pragma unroll
            for (int q = 0; q < R; ++q) tw[q] = wn[s2m2 * (k + Lp * q)];
        }
        gfft_bfly<T, R>(v, roots, inv);
#pragma unroll
        for (int q = 0; q < R; ++q) {
            cpx<T> a = v[q];
            if (!last) {
                const cpx<T> w = tw[q];
                const T wy = inv ? -w.y : w.y;
                a = cpx<T>{a.x * w.x - a.y * wy, a.x * wy + a.y * w.x};
            }
            out[(k + Lp * q) * m + j] = a;
        }
    }
}

// BFLY: the kernel's plans may hold butterfly passes (planned with merge_small).  A kernel whose plans never do leaves
// it off: the nine butterfly bodies would only raise its register allocation (the float64 envelope kernel spilled).
template <class T, bool BFLY = false>
ECHO_HD void gfft_pass(int tid, int nt, const GfftPlan& pl, int ps, const cpx<T>* in, cpx<T>* out, const cpx<T>* roots,
                       const cpx<T>* wn, bool inv) {
    if (BFLY && pl.bfly[ps]) {
        switch (pl.radix[ps]) {
            case 2: gfft_pass_bfly<T, 2>(tid, nt, pl, ps, in, out, roots, wn, inv); return;
            case 3: gfft_pass_bfly<T, 3>(tid, nt, pl, ps, in, out, roots, wn, inv); return;
            case 4: gfft_pass_bfly<T, 4>(tid, nt, pl, ps, in, out, roots, wn, inv); return;
            case 5: gfft_pass_bfly<T, 5>(tid, nt, pl, ps, in, out, roots, wn, inv); return;
            case 6: gfft_pass_bfly<T, 6>(tid, nt, pl, ps, in, out, roots, wn, inv); return;
            case 7: gfft_pass_bfly<T, 7>(tid, nt, pl, ps, in, out, roots, wn, inv); return;
            case 8: gfft_pass_bfly<T, 8>(tid, nt, pl, ps, in, out, roots, wn, inv); return;
            case 9: gfft_pass_bfly<T, 9>(tid, nt, pl, ps, in, out, roots, wn, inv); return;
            case 10: gfft_pass_bfly<T, 10>(tid, nt, pl, ps, in, out, roots, wn, inv); return;
            default: break;
        }
    }
    if (pl.per_lane[ps]) {
        switch (pl.pairs[ps]) {
            case 1: gfft_pass_lane<T, 1>(tid, nt, pl, ps, in, out, roots, wn, inv); break;
            case 2: gfft_pass_lane<T, 2>(tid, nt, pl, ps, in, out, roots, wn, inv); break;
            default: gfft_pass_lane<T, 3>(tid, nt, pl, ps, in, out, roots, wn, inv); break;
        }
        return;
    }
    switch (pl.pairs[ps]) {
        case 1: gfft_pass_p<T, 1>(tid, nt, pl, ps, in, out, roots, wn, inv); break;
        case 2: gfft_pass_p<T, 2>(tid, nt, pl, ps, in, out, roots, wn, inv); break;
        default: gfft_pass_p<T, 3>(tid, nt, pl, ps, in, out, roots, wn, inv); break;
    }
}

}  // namespace echo
