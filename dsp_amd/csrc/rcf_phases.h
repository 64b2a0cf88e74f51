// Matched filter of long frames as ONE real cyclic correlation per (frame, band), float64 (the single-transform form of
// cfft_phases.h for the case the chain meets: every lag the detector reads, k_begin <= k < L, out of one transform).
//
//   c[l] = sum_j chirp[j] * u[l + j],  u[i] = band_sig[k_begin - (M-1) + i],  0 <= l < nl = L - k_begin,
// is the cyclic correlation of length F2 = 2H >= nl + M - 1 (no lag wraps).  u is real, so it is transformed as the
// packed complex sequence z[n] = u[2n] + i u[2n+1] of length H (half the points of the two-sub-block transform of
// cfft_phases.h, which needs a cyclic length of 8192 complex points for the same lags):
//   Z = DFT_H(z)                                  in place, decimation in frequency, mixed radix (radices <= 10 as the
//                                                 in-register butterflies of gfft.h), output in pass order
//   X[k] = E[k] + w^k O[k],  E = (Z[k] + conj Z[H-k]) / 2,  O = -i (Z[k] - conj Z[H-k]) / 2,  w = exp(-2 pi i / F2)
//   Y[k] = X[k] * G[k],  G = conj(DFT_F2(chirp)) / H          (0 <= k <= H; pairs (k, H-k) are local to one thread)
//   Z'[k] = E'[k] + i O'[k],  E' = (Y[k] + conj Y[H-k]) / 2,  O' = conj(w^k) (Y[k] - conj Y[H-k]) / 2
//   z' = H * IDFT_H(Z')                           the conjugate transposes of the forward passes in reverse order:
//                                                 natural order out, c[2n] + i c[2n+1] in slot n
// The detector then runs on that layout (LongParams::det_view == 2) in the same workgroup.  LDS: H complex doubles with
// one 16-byte pad per 8 elements (H = 5400 for airSonar.py's sizes: 97 KB against the 147 KB of the 8192-point
// transform), which leaves room for a workgroup of another kernel on the compute unit.
// Plain C++: phases of (tid, nt), so a host build can run them.
#pragma once
#include <complex>
#include <vector>

#include "cfft_phases.h"

namespace echo {

#ifndef ECHO_RC_NT
#define ECHO_RC_NT 768
#endif
constexpr int kRcNT = ECHO_RC_NT;                 // threads per workgroup

// smallest H >= hmin that is a product of radices <= kGfftMaxBfly in at most kRcMaxPass passes; 0 if none below hmax
inline bool rc_factor(int H, int* radix, int* npass, int max_radix = kGfftMaxBfly) {
    GfftPlan pl;
    if (!gfft_plan(pl, H, kRcNT, true) || pl.npass > kRcMaxPass) return false;
    for (int i = 0; i < pl.npass; ++i) { if (!pl.bfly[i] || pl.radix[i] > max_radix) return false; radix[i] = pl.radix[i]; }
    *npass = pl.npass;
    return true;
}

constexpr int kRcSplitNT = 512;                   // threads of the block-wise kernel (two workgroups per compute unit)
constexpr int kRcSplitLds = 80 * 1024;            // its LDS bound: half of a compute unit's
constexpr int kRcSplitMaxR = 8;                   // its largest radix: four waves per SIMD leave 128 registers

// Plans the transforms for LongParams p (N, M, L, half set) over the lags k_begin .. L-1: fills p.rc_*; false when it
// does not apply.  split = false: ONE transform holds every lag (the detector can follow in the same workgroup).
// split = true: blocks of V = 2H - (M - 1) lags, each a transform small enough for two workgroups to share a compute
// unit's LDS; fewest blocks x passes x points wins.
inline bool plan_rcf(LongParams& p, int k_begin, bool split = false) {
    p.rc_use = 0; p.rc_H = 0; p.rc_npass = 0; p.rc_V = 0; p.rc_nblk = 0;
    const int nl = p.L - k_begin;
    if (nl < 1 || p.M < 512) return false;                   // short chirps keep the direct sum
    int best = 0, best_radix[kRcMaxPass], best_np = 0;
    long best_cost = 0;
    // even H: the pair (k, H-k) then has its fixed point at H/2
    int hlo, hhi;
    if (!split) { hlo = (nl + p.M - 1 + 1) / 2; hhi = hlo + hlo / 25 + 16; }      // within 4 % of the minimum
    else { hlo = (p.M + 63) / 2; hhi = (kRcSplitLds / 16 - 1) * 8 / 9; }          // (H + H/8 + 1) * 16 <= the bound
    for (int H = hlo + (hlo & 1); H <= hhi; H += 2) {
        int radix[kRcMaxPass], np = 0;
        const int V = 2 * H - (p.M - 1);
        // the block-wise kernel runs four waves per SIMD (128 registers): radices up to 8 fit without spilling
        if (V < 1 || !rc_factor(H, radix, &np, split ? kRcSplitMaxR : kGfftMaxBfly)) continue;
        const long cost = (long)((nl + V - 1) / V) * np * H;
        if (!best || cost < best_cost) {
            best = H; best_cost = cost; best_np = np;
            for (int i = 0; i < np; ++i) best_radix[i] = radix[i];
        }
    }
    if (!best) return false;
    const long bytes = (long)(best + best / 8 + 1) * 16;
    if (!split && (bytes + kDetExtra > 160 * 1024 || bytes + kCfTpExtra > 160 * 1024)) return false;
    if (split && bytes > kRcSplitLds) return false;
    p.rc_use = 1; p.rc_H = best; p.rc_npass = best_np; p.rc_k0 = k_begin;
    p.rc_V = split ? 2 * best - (p.M - 1) : nl;
    p.rc_nblk = (nl + p.rc_V - 1) / p.rc_V;
    int roff = 0;
    for (int i = 0; i < best_np; ++i) { p.rc_radix[i] = best_radix[i]; p.rc_root_off[i] = roff; roff += best_radix[i]; }
    p.rc_smem = (int)((bytes + 15) & ~15L);
    return true;
}

// frequency held at position e after the forward passes
inline int rc_pass_freq(const LongParams& p, int e) {
    int S = p.rc_H, w = 1, f = 0;
    for (int ps = 0; ps < p.rc_npass; ++ps) {
        const int r = p.rc_radix[ps], q = S / r;
        f += ((e / q) % r) * w;
        w *= r; S = q;
    }
    return f;
}

// host FFT of any length whose prime factors are small (tables only)
inline void fft_smooth(std::vector<std::complex<long double>>& a) {
    const size_t n = a.size();
    if (n <= 1) return;
    size_t r = n;
    for (size_t f = 2; f * f <= n; ++f) if (n % f == 0) { r = f; break; }
    const size_t m = n / r;
    std::vector<std::vector<std::complex<long double>>> sub(r, std::vector<std::complex<long double>>(m));
    for (size_t i = 0; i < n; ++i) sub[i % r][i / r] = a[i];
    if (m > 1) for (size_t j = 0; j < r; ++j) fft_smooth(sub[j]);
    const long double tau = 6.283185307179586476925286766559005768L;
    std::vector<std::complex<long double>> w(n);
    for (size_t t = 0; t < n; ++t) w[t] = std::complex<long double>(cosl(-tau * (long double)t / (long double)n), sinl(-tau * (long double)t / (long double)n));
    for (size_t k = 0; k < n; ++k) {
        std::complex<long double> acc(0, 0);
        for (size_t j = 0; j < r; ++j) acc += sub[j][k % m] * w[(j * k) % n];
        a[k] = acc;
    }
}

// Tables: wh = the pass twiddles, for pass ps with block size S, radix r, stride q = S / r the runs
// wh[tw_off[ps] + (m - 1) * q + i] = exp(-2 pi i (i m) / S), 1 <= m < r, i < q (one coalesced read per m);
// wk[k] = exp(-2 pi i k / 2H), k <= H/2; roots[root_off[p] + t] = exp(-2 pi i t / r_p);
// g[band][k] = conj(DFT_2H(chirp_band))[k] / H, k <= H; pos[k] = position of frequency k after the forward passes.
inline void rcf_tables(LongParams& p, int nb, const double* chirps, std::vector<zd>& wh, std::vector<zd>& wk,
                       std::vector<zd>& roots, std::vector<zd>& g, std::vector<int>& pos) {
    const long double tau = 6.283185307179586476925286766559005768L;
    const int H = p.rc_H, F2 = 2 * H;
    wh.clear(); wk.resize(H / 2 + 1);
    for (int ps = 0, S = H; ps < p.rc_npass; ++ps) {
        const int r = p.rc_radix[ps], q = S / r;
        p.rc_tw_off[ps] = (int)wh.size();
        for (int m = 1; m < r; ++m)
            for (int i = 0; i < q; ++i) {
                const long double a = -tau * (long double)((long)i * m % S) / (long double)S;
                wh.push_back(zmk((double)cosl(a), (double)sinl(a)));
            }
        S = q;
    }
    for (int k = 0; k <= H / 2; ++k) wk[k] = zmk((double)cosl(-tau * k / F2), (double)sinl(-tau * k / F2));
    roots.clear();
    for (int ps = 0; ps < p.rc_npass; ++ps)
        for (int t = 0; t < p.rc_radix[ps]; ++t)
            roots.push_back(zmk((double)cosl(-tau * t / p.rc_radix[ps]), (double)sinl(-tau * t / p.rc_radix[ps])));
    g.assign((size_t)nb * (H + 1), zmk(0, 0));
    for (int b = 0; b < nb; ++b) {
        std::vector<std::complex<long double>> h(F2, std::complex<long double>(0, 0));
        for (int j = 0; j < p.M && j < F2; ++j) h[j] = chirps[(size_t)b * p.M + j];
        fft_smooth(h);
        for (int k = 0; k <= H; ++k) {
            const std::complex<long double> c = std::conj(h[k]) / (long double)H;
            g[(size_t)b * (H + 1) + k] = zmk((double)c.real(), (double)c.imag());
        }
    }
    pos.assign(H, 0);
    for (int e = 0; e < H; ++e) pos[rc_pass_freq(p, e)] = e;
}

// z[n] = u[2n] + i u[2n+1]
ECHO_HD void ph_rc_load(int tid, int nt, int sub, int blk, const LongParams& p, unsigned char* smem) {
    zd* f = reinterpret_cast<zd*>(smem);
    const double* x = p.band_in + (size_t)blk * p.N;
    const int o = p.rc_k0 + sub * p.rc_V - (p.M - 1);
    for (int n = tid; n < p.rc_H; n += nt) {
        const int a = o + 2 * n, b = a + 1;
        f[zpos(n)] = zmk((a >= 0 && a < p.N) ? x[a] : 0.0, (b >= 0 && b < p.N) ? x[b] : 0.0);
    }
}

// One in-place pass of radix R on blocks of S elements (stride q = S / R).  Forward: butterfly, then output m times
// W_S^(i m); inverse: the conjugate twiddles first, then the conjugate butterfly.  W_S^(i m) = tw[(m - 1) q + i].
template <int R, bool INV>
ECHO_HD void rc_pass(int tid, int nt, int S, const zd* roots, const zd* tw, const LongParams& p, unsigned char* smem) {
    zd* f = reinterpret_cast<zd*>(smem);
    const int q = S / R;
    for (int t = tid; t < p.rc_H / R; t += nt) {
        const int b = t / q, i = t - b * q, base = b * S + i;
        zd v[R];
        if (INV) {
            v[0] = f[zpos(base)];
#pragma unroll
            for (int m = 1; m < R; ++m) v[m] = zmulc(f[zpos(base + m * q)], tw[(m - 1) * q + i]);
        } else {
#pragma unroll
            for (int j = 0; j < R; ++j) v[j] = f[zpos(base + j * q)];
        }
        gfft_bfly<double, R>(v, roots, INV);
        if (INV) {
#pragma unroll
            for (int j = 0; j < R; ++j) f[zpos(base + j * q)] = v[j];
        } else {
            f[zpos(base)] = v[0];
#pragma unroll
            for (int m = 1; m < R; ++m) f[zpos(base + m * q)] = zmul(v[m], tw[(m - 1) * q + i]);
        }
    }
}

// MAXR: the largest radix the kernel's plans hold (plan_rcf): larger butterflies are not compiled in, so they do not
// count towards its register allocation
template <bool INV, int MAXR = kGfftMaxBfly>
ECHO_HD void ph_rc_pass(int tid, int nt, int ps, const LongParams& p, unsigned char* smem) {
    int S = p.rc_H;
    for (int i = 0; i < ps; ++i) S /= p.rc_radix[i];
    const zd* roots = p.rc_roots + p.rc_root_off[ps];
    const zd* tw = p.rc_wh + p.rc_tw_off[ps];
    switch (p.rc_radix[ps]) {
        case 2: rc_pass<2, INV>(tid, nt, S, roots, tw, p, smem); break;
        case 3: rc_pass<3, INV>(tid, nt, S, roots, tw, p, smem); break;
        case 4: rc_pass<4, INV>(tid, nt, S, roots, tw, p, smem); break;
        case 5: rc_pass<5, INV>(tid, nt, S, roots, tw, p, smem); break;
        case 6: rc_pass<6, INV>(tid, nt, S, roots, tw, p, smem); break;
        case 7: rc_pass<7, INV>(tid, nt, S, roots, tw, p, smem); break;
        case 8: rc_pass<8, INV>(tid, nt, S, roots, tw, p, smem); break;
        case 9: if (MAXR >= 9) rc_pass<(MAXR >= 9 ? 9 : 2), INV>(tid, nt, S, roots, tw, p, smem); break;
        case 10: if (MAXR >= 10) rc_pass<(MAXR >= 10 ? 10 : 2), INV>(tid, nt, S, roots, tw, p, smem); break;
        default: break;
    }
}

// real-transform unpacking, product with G, packing for the inverse: pairs (k, H-k), each owned by one thread
ECHO_HD void ph_rc_split(int tid, int nt, int blk, const LongParams& p, unsigned char* smem) {
    zd* f = reinterpret_cast<zd*>(smem);
    const int H = p.rc_H;
    const zd* g = p.rc_g + (size_t)(blk % p.nb) * (H + 1);
    for (int k = tid; k <= H / 2; k += nt) {
        if (k == 0) {
            const int p0 = zpos(p.rc_pos[0]);
            const zd z = f[p0];
            const double y0 = (z.x + z.y) * g[0].x, yh = (z.x - z.y) * g[H].x;      // X[0], X[H] real; G[0], G[H] real
            f[p0] = zmk(0.5 * (y0 + yh), 0.5 * (y0 - yh));
            continue;
        }
        if (2 * k == H) {
            const int ph = zpos(p.rc_pos[k]);
            const zd z = f[ph];
            const zd y = zmul(zmk(z.x, -z.y), g[k]);                                   // X[H/2] = conj Z[H/2]
            f[ph] = zmk(y.x, -y.y);
            continue;
        }
        const int pk = zpos(p.rc_pos[k]), pc = zpos(p.rc_pos[H - k]);
        const zd zk = f[pk], zc = f[pc], w = p.rc_wk[k];
        const zd gk = g[k], gc = g[H - k];
        // E = (Zk + conj Zc) / 2, O = -i (Zk - conj Zc) / 2
        const zd E = zmk(0.5 * (zk.x + zc.x), 0.5 * (zk.y - zc.y));
        const zd D = zmk(0.5 * (zk.x - zc.x), 0.5 * (zk.y + zc.y));
        const zd O = zmk(D.y, -D.x);
        const zd wO = zmul(w, O);
        const zd Xk = zadd(E, wO);                                   // X[k]
        const zd Xc = zmk(E.x - wO.x, -(E.y - wO.y));                // X[H-k] = conj(E - w O)
        const zd Yk = zmul(Xk, gk), Yc = zmul(Xc, gc);
        // E' = (Yk + conj Yc) / 2, O' = conj(w) (Yk - conj Yc) / 2
        const zd Ep = zmk(0.5 * (Yk.x + Yc.x), 0.5 * (Yk.y - Yc.y));
        const zd Dp = zmk(0.5 * (Yk.x - Yc.x), 0.5 * (Yk.y + Yc.y));
        const zd Op = zmulc(Dp, w);
        // Z'[k] = E' + i O',  Z'[H-k] = conj(E') + i conj(O')
        f[pk] = zmk(Ep.x - Op.y, Ep.y + Op.x);
        f[pc] = zmk(Ep.x + Op.y, Op.x - Ep.y);
    }
}

// after the inverse passes slot n holds c[2n] + i c[2n+1]; clears the detector's scalars for a detector that follows
ECHO_HD void ph_rc_finish(int tid, int nt, const LongParams& p, unsigned char* smem) {
    if (tid == 0) det_misc_init(det_misc(p, smem));
}

// stand-alone: the lags of block sub to corr_out
ECHO_HD void ph_rc_store(int tid, int nt, int sub, int blk, const LongParams& p, unsigned char* smem) {
    const zd* f = reinterpret_cast<const zd*>(smem);
    double* out = p.corr_out + (size_t)blk * p.corr_stride - p.corr_k0;
    const int k0 = p.rc_k0 + sub * p.rc_V;
    const int nl = (p.L - k0) < p.rc_V ? (p.L - k0) : p.rc_V;
    for (int n = tid; 2 * n < nl; n += nt) {
        const zd v = f[zpos(n)];
        out[k0 + 2 * n] = v.x;
        if (2 * n + 1 < nl) out[k0 + 2 * n + 1] = v.y;
    }
}

#define ECHO_RC_TRANSFORM_R(PHASE, SYNC, p, smem, sub, blk, nt, MAXR)                        \
    PHASE(ph_rc_load(tid, nt, sub, blk, p, smem)) SYNC                                        \
    for (int ps_ = 0; ps_ < (p).rc_npass; ++ps_) { PHASE(ph_rc_pass<false, MAXR>(tid, nt, ps_, p, smem)) SYNC }   \
    PHASE(ph_rc_split(tid, nt, blk, p, smem)) SYNC                                            \
    for (int ps_ = (p).rc_npass - 1; ps_ >= 0; --ps_) { PHASE(ph_rc_pass<true, MAXR>(tid, nt, ps_, p, smem)) SYNC }
#define ECHO_RC_TRANSFORM(PHASE, SYNC, p, smem, sub, blk, nt) ECHO_RC_TRANSFORM_R(PHASE, SYNC, p, smem, sub, blk, nt, kGfftMaxBfly)

#define ECHO_RC_SEQUENCE(PHASE, SYNC, p, smem, sub, blk, nt, MAXR)                            \
    ECHO_RC_TRANSFORM_R(PHASE, SYNC, p, smem, sub, blk, nt, MAXR)                             \
    PHASE(ph_rc_store(tid, nt, sub, blk, p, smem))

#define ECHO_RC_DET_SEQUENCE(PHASE, SYNC, p, smem, blk, nt)                                   \
    ECHO_RC_TRANSFORM(PHASE, SYNC, p, smem, 0, blk, nt)                                       \
    PHASE(ph_rc_finish(tid, nt, p, smem)) SYNC                                                \
    ECHO_LDET_BODY(PHASE, SYNC, p, smem, blk, nt)

#define ECHO_RC_TP_SEQUENCE(PHASE, SYNC, p, smem, blk, nt)                                    \
    ECHO_RC_TRANSFORM(PHASE, SYNC, p, smem, 0, blk, nt)                                       \
    PHASE(ph_rc_finish(tid, nt, p, smem)) SYNC                                                \
    PHASE(if (tid < kLDetNT) ph_tp_scan_view(tid, kLDetNT, p, smem)) SYNC                     \
    PHASE(if (tid < kLDetNT) ph_tp_ref_fold(tid, kLDetNT, p, smem)) SYNC                      \
    PHASE(if (tid < kLDetNT) ph_tp_ref(tid, kLDetNT, p, smem)) SYNC                           \
    PHASE(if (tid < kLDetNT) ph_tp_second(tid, kLDetNT, p, smem)) SYNC                        \
    PHASE(if (tid < kLDetNT) ph_tp_second_fold(tid, kLDetNT, p, smem)) SYNC                   \
    PHASE(if (tid < kLDetNT) ph_tp_finish(tid, kLDetNT, blk, p, smem))

}  // namespace echo
