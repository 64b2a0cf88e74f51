// Hilbert transform of the correlation as ONE real cyclic convolution, float64 (envelope branch of _process_band_gpu,
// airSonar_optimized.py:431-475 -> scipy.signal.hilbert).  Included by env_phases.h.
//
// The imaginary part of scipy.signal.hilbert(x) for an even length L is the cyclic convolution of x with
//     g[d] = (2 / L) cot(pi d / L) for odd d,  0 for even d            (the inverse DFT of -i sgn(k)),
// and a cyclic convolution of length L is a linear one with the kernel laid out over -(L-1) .. L-1, hence a cyclic one
// of any length F2 >= 2 L - 1.  F2 = 9216 = 2 * 4608 serves every L the envelope kernel takes (L <= 4608), and x is
// real, so it is transformed as the packed complex sequence z[n] = x[2n] + i x[2n+1] of length H = 4608 = 9 * 8 * 8 * 8:
//     Z = DFT_H(z)                       in place, decimation in frequency, radix-9 pass then three radix-8 passes as
//                                        in-register butterflies; output in pass order (digit-reversed)
//     X[k] = E[k] + w^k O[k]             E, O from the pair (Z[k], conj Z[H-k]), w = exp(-2 pi i / F2)
//     Y[k] = i gam[k] X[k]               the kernel is real and odd, so its spectrum is imaginary; gam includes 1 / H
//     Z'[k] = E'[k] + i O'[k]            packing for the inverse, same pair
//     z' = H * IDFT_H(Z')                the conjugate passes in reverse order: natural order out,
//                                        H(x)[2n] + i H(x)[2n+1] in slot n
// (the last forward pass, the pair phase and the first inverse pass run as one phase, ph_envlin_mid)
// against two half-length transforms of length L / 2 = 2278 = 2 * 17 * 67 as 67- and 17-term sums per output: a fifth of
// the multiply-adds.  One buffer of H complex doubles with one 16-byte pad per 16 elements (78 KB): two workgroups
// share a compute unit's LDS.  Plain C++: phases of (tid, nt), so a host build can run them.
#pragma once

namespace echo {

constexpr int kLinH = 4608, kLinF2 = 2 * kLinH;
constexpr int kLinSlots = kLinH + kLinH / 16 + 1;
constexpr int kLinBytes = kLinSlots * 16;
constexpr int kLinBuf1 = 39168;            // byte offset of the second work buffer: behind slot zl(L / 2 - 1) for L <= 4608
constexpr int kLinTw1 = 8 * 512, kLinTw2 = kLinTw1 + 7 * 64, kLinTwN = kLinTw2 + 7 * 8;
ECHO_HD int zl(int n) { return n + (n >> 4); }
// position of frequency f after the forward passes (radices 9, 8, 8, 8: strides 512, 64, 8, 1)
ECHO_HD int envlin_pos(int f) { return (f % 9) * 512 + ((f / 9) & 7) * 64 + ((f / 72) & 7) * 8 + f / 576; }

using zc = cpx<double>;
ECHO_HD zc zc_mul(zc a, zc b) { return zc{fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x)}; }
ECHO_HD zc zc_mulc(zc a, zc b) { return zc{fma(a.x, b.x, a.y * b.y), fma(a.y, b.x, -(a.x * b.y))}; }   // a * conj(b)

// lane order of the middle phase: item t = (d2 = t & 3, d1 = (t >> 2) & 7, d0 = t >> 5) is the residue d0 + 9 d1 + 72 d2
inline int envlin_mid_residue(int t) { return (t >> 5) + 9 * ((t >> 2) & 7) + 72 * (t & 3); }
// wmid[d * 288 + t] = wk[k], gmid[d * 288 + t] = (gam[k], gam[H - k]) for k = residue(t) + 576 d: what the middle phase
// reads, in the order its lanes read it
inline void envlin_mid_tables(const std::vector<zc>& wk, const std::vector<double>& gam, std::vector<zc>& wmid,
                              std::vector<zc>& gmid) {
    wmid.assign(8 * 288, zc{0.0, 0.0}); gmid.assign(8 * 288, zc{0.0, 0.0});
    for (int t = 1; t < 288; ++t)
        for (int d = 0; d < 8; ++d) {
            const int k = envlin_mid_residue(t) + 576 * d;
            wmid[d * 288 + t] = wk[k];
            gmid[d * 288 + t] = zc{gam[k], gam[kLinH - k]};
        }
}
// Tables: tw = pass twiddles, for the pass on blocks of S elements with radix R and stride q = S / R the runs
// tw[off + (m - 1) q + i] = exp(-2 pi i (i m) / S), 1 <= m < R, i < q (the last pass has q = 1: none);
// wk[k] = exp(-2 pi i k / F2), k <= H;  gam[k] = Im(DFT_F2(kernel))[k] / H, k <= H.
inline void envlin_tables(int L, std::vector<zc>& tw, std::vector<zc>& wk, std::vector<double>& gam) {
    const long double tau = 6.283185307179586476925286766559005768L;
    tw.clear();
    const int radix[3] = {9, 8, 8};
    for (int ps = 0, S = kLinH; ps < 3; ++ps) {
        const int r = radix[ps], q = S / r;
        for (int m = 1; m < r; ++m)
            for (int i = 0; i < q; ++i) {
                const long double a = -tau * (long double)((long)i * m % S) / (long double)S;
                tw.push_back(zc{(double)cosl(a), (double)sinl(a)});
            }
        S = q;
    }
    wk.resize(kLinH + 1);
    for (int k = 0; k <= kLinH; ++k)
        wk[k] = zc{(double)cosl(-tau * k / kLinF2), (double)sinl(-tau * k / kLinF2)};
    // DFT_F2 of the odd kernel h[d] = g[d], h[F2 - d] = -g[d] (1 <= d < L): -2 i sum_d g[d] sin(2 pi k d / F2)
    std::vector<long double> sn(kLinF2), g(L, 0.0L);
    for (int t = 0; t < kLinF2; ++t) sn[t] = sinl(tau * (long double)t / (long double)kLinF2);
    for (int d = 1; d < L; d += 2) {
        const long double a = tau * 0.5L * (long double)d / (long double)L;
        g[d] = 2.0L / (long double)L * cosl(a) / sinl(a);
    }
    gam.assign(kLinH + 1, 0.0);
    for (int k = 0; k <= kLinH; ++k) {
        long double acc = 0.0L;
        for (int d = 1; d < L; d += 2) acc += g[d] * sn[(int)(((long)k * d) % kLinF2)];
        gam[k] = (double)(-2.0L * acc / (long double)kLinH);
    }
}

// ---- butterflies, in place, natural order in and out; INV: the conjugate transform
template <bool INV> ECHO_HD void envlin_r3(zc& a, zc& b, zc& c) {
    constexpr double C3 = 0.866025403784438646763723170753;
    const zc s{b.x + c.x, b.y + c.y}, d{b.x - c.x, b.y - c.y};
    const zc m{fma(-0.5, s.x, a.x), fma(-0.5, s.y, a.y)};
    a = zc{a.x + s.x, a.y + s.y};
    const zc p{fma(C3, d.y, m.x), fma(-C3, d.x, m.y)}, n{fma(-C3, d.y, m.x), fma(C3, d.x, m.y)};   // m -/+ i C3 d
    b = INV ? n : p; c = INV ? p : n;
}
template <bool INV> ECHO_HD void envlin_bf9(zc (&v)[9]) {
    // j = 3 a + b, k = c + 3 d:  W9^(j k) = W3^(a c) W9^(b c) W3^(b d)
    envlin_r3<INV>(v[0], v[3], v[6]);
    envlin_r3<INV>(v[1], v[4], v[7]);
    envlin_r3<INV>(v[2], v[5], v[8]);
    constexpr double c1 = 0.766044443118978035202392650555, s1 = 0.642787609686539326322643409907;
    constexpr double c2 = 0.173648177666930348851716626769, s2 = 0.984807753012208059366743024589;
    constexpr double c4 = -0.939692620785908384054109277325, s4 = 0.342020143325668733044099614683;
    const zc w1{c1, INV ? s1 : -s1}, w2{c2, INV ? s2 : -s2}, w4{c4, INV ? s4 : -s4};
    v[4] = zc_mul(v[4], w1); v[7] = zc_mul(v[7], w2);
    v[5] = zc_mul(v[5], w2); v[8] = zc_mul(v[8], w4);
    envlin_r3<INV>(v[0], v[1], v[2]);       // c = 0: X[0], X[3], X[6]
    envlin_r3<INV>(v[3], v[4], v[5]);       // c = 1: X[1], X[4], X[7]
    envlin_r3<INV>(v[6], v[7], v[8]);       // c = 2: X[2], X[5], X[8]
    const zc x1 = v[3], x2 = v[6], x3 = v[1], x5 = v[7], x6 = v[2], x7 = v[5];
    v[1] = x1; v[2] = x2; v[3] = x3; v[5] = x5; v[6] = x6; v[7] = x7;
}
template <bool INV> ECHO_HD void envlin_r4(zc& c0, zc& c1, zc& c2, zc& c3) {
    const zc s0{c0.x + c2.x, c0.y + c2.y}, s1{c0.x - c2.x, c0.y - c2.y}, s2{c1.x + c3.x, c1.y + c3.y};
    const zc d{c1.x - c3.x, c1.y - c3.y};
    const zc s3 = INV ? zc{-d.y, d.x} : zc{d.y, -d.x};                    // (c1 - c3) * (+/- i)
    c0 = zc{s0.x + s2.x, s0.y + s2.y}; c2 = zc{s0.x - s2.x, s0.y - s2.y};
    c1 = zc{s1.x + s3.x, s1.y + s3.y}; c3 = zc{s1.x - s3.x, s1.y - s3.y};
}
template <bool INV> ECHO_HD void envlin_bf8(zc (&v)[8]) {
    constexpr double R2 = 0.707106781186547524400844362105;
    zc a[4], b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] = zc{v[j].x + v[j + 4].x, v[j].y + v[j + 4].y};
        b[j] = zc{v[j].x - v[j + 4].x, v[j].y - v[j + 4].y};
    }
    if (INV) {
        b[1] = zc{(b[1].x - b[1].y) * R2, (b[1].x + b[1].y) * R2};         // * (1 + i) / sqrt 2
        b[2] = zc{-b[2].y, b[2].x};                                        // * i
        b[3] = zc{(-b[3].x - b[3].y) * R2, (b[3].x - b[3].y) * R2};        // * (-1 + i) / sqrt 2
    } else {
        b[1] = zc{(b[1].x + b[1].y) * R2, (b[1].y - b[1].x) * R2};         // * (1 - i) / sqrt 2
        b[2] = zc{b[2].y, -b[2].x};                                        // * -i
        b[3] = zc{(b[3].y - b[3].x) * R2, (-b[3].x - b[3].y) * R2};        // * (-1 - i) / sqrt 2
    }
    envlin_r4<INV>(a[0], a[1], a[2], a[3]);
    envlin_r4<INV>(b[0], b[1], b[2], b[3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[2 * r] = a[r]; v[2 * r + 1] = b[r]; }
}
template <int R, bool INV> ECHO_HD void envlin_bf(zc (&v)[R]) {
    if constexpr (R == 9) envlin_bf9<INV>(v);
    else if constexpr (R == 4) envlin_r4<INV>(v[0], v[1], v[2], v[3]);
    else envlin_bf8<INV>(v);
}

// One in-place pass of radix R on blocks of S elements (stride q = S / R).  Forward: butterfly, then output m times
// W_S^(i m); inverse: the conjugate twiddles first, then the conjugate butterfly.
// (H: length of the transform the pass belongs to.)
template <int H, int R, int S, bool INV>
ECHO_HD void ph_envlin_pass(int tid, int nt, const zc* tw, unsigned char* smem) {
    zc* f = reinterpret_cast<zc*>(smem);
    constexpr int q = S / R;
    for (int t = tid; t < H / R; t += nt) {
        const int b = t / q, i = t - b * q, base = b * S + i;
        zc v[R], w[R];
        if (q > 1) {
#pragma unroll
            for (int m = 1; m < R; ++m) w[m] = tw[(m - 1) * q + i];
        }
#pragma unroll
        for (int j = 0; j < R; ++j) v[j] = f[zl(base + j * q)];
        if (INV && q > 1) {
#pragma unroll
            for (int m = 1; m < R; ++m) v[m] = zc_mulc(v[m], w[m]);
        }
        envlin_bf<R, INV>(v);
        if (!INV && q > 1) {
#pragma unroll
            for (int m = 1; m < R; ++m) v[m] = zc_mul(v[m], w[m]);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) f[zl(base + j * q)] = v[j];
    }
}

// Real-transform unpacking, product with the kernel's spectrum and packing for the inverse on one pair (k, H - k),
// 0 < k < H, k != H / 2: zk = Z[k], zh = Z[H - k] in, Z'[k], Z'[H - k] out.
ECHO_HD void envlin_pair(zc& zk, zc& zh, zc w, double gk, double gc) {
    // E = (Zk + conj Zh) / 2, O = -i (Zk - conj Zh) / 2
    const zc E{0.5 * (zk.x + zh.x), 0.5 * (zk.y - zh.y)};
    const zc D{0.5 * (zk.x - zh.x), 0.5 * (zk.y + zh.y)};
    const zc O{D.y, -D.x};
    const zc wO = zc_mul(w, O);
    const zc Xk{E.x + wO.x, E.y + wO.y};                               // X[k]
    const zc Xc{E.x - wO.x, -(E.y - wO.y)};                            // X[H-k] = conj(E - w O)
    const zc Yk{-gk * Xk.y, gk * Xk.x}, Yc{-gc * Xc.y, gc * Xc.x};     // i gam X
    // E' = (Yk + conj Yc) / 2, O' = conj(w) (Yk - conj Yc) / 2
    const zc Ep{0.5 * (Yk.x + Yc.x), 0.5 * (Yk.y - Yc.y)};
    const zc Dp{0.5 * (Yk.x - Yc.x), 0.5 * (Yk.y + Yc.y)};
    const zc Op = zc_mulc(Dp, w);
    // Z'[k] = E' + i O',  Z'[H-k] = conj(E') + i conj(O')
    zk = zc{Ep.x - Op.y, Ep.y + Op.x};
    zh = zc{Ep.x + Op.y, Op.x - Ep.y};
}
// X[H/2] = conj Z[H/2]; Z'[H/2] = conj(i gam X)
ECHO_HD zc envlin_mid_point(zc z, const double* gam) {
    const double g = gam[kLinH / 2];
    return zc{g * z.y, -(g * z.x)};
}

// The last forward pass, the pair phase and the first inverse pass in one: the last pass's butterflies take 8 contiguous
// elements (no twiddles), and the 8 frequencies r + 576 d of the butterfly with residue r (element base
// (r % 9) * 512 + (r / 9 % 8) * 64 + (r / 72) * 8) have their partners H - f in the butterfly with residue 576 - r, at
// 7 - d.  A thread therefore takes both butterflies into registers, treats the 8 pairs, and runs the two conjugate
// butterflies before it stores: two barriers and two trips through LDS less than the three phases.  Residues 0 and 288
// pair with themselves (frequency 0 has no partner and vanishes with gam[0]; H / 2 is its own).
ECHO_HD int envlin_res_base(int r) { return (r % 9) * 512 + ((r / 9) & 7) * 64 + (r / 72) * 8; }
ECHO_HD void ph_envlin_mid(int tid, int nt, const zc* wk, const double* gam, const zc* wmid, const zc* gmid,
                           unsigned char* smem) {
    zc* f = reinterpret_cast<zc*>(smem);
    // The residues below 288 are those whose butterfly starts in the first half of its 64-element block (r = d0 +
    // 9 d1 + 72 d2 at element 512 d0 + 64 d1 + 8 d2, d2 < 4).  Consecutive lanes take consecutive d2, then d1, then d0:
    // butterflies 8 and 64 elements apart fall into different LDS banks, butterflies 512 apart into the same ones
    // (consecutive residues would put a wavefront's 16-byte reads nine to a bank).
    for (int t = 1 + tid; t < 288; t += nt) {
        const int d2 = t & 3, d1 = (t >> 2) & 7, d0 = t >> 5;
        const int r = d0 + 9 * d1 + 72 * d2;
        const int ba = d0 * 512 + d1 * 64 + d2 * 8, bb = envlin_res_base(576 - r);
        zc v[8], u[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = f[zl(ba + j)]; u[j] = f[zl(bb + j)]; }
        envlin_bf8<false>(v); envlin_bf8<false>(u);
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const zc g2 = gmid[d * 288 + t];
            envlin_pair(v[d], u[7 - d], wmid[d * 288 + t], g2.x, g2.y);
        }
        envlin_bf8<true>(v); envlin_bf8<true>(u);
#pragma unroll
        for (int j = 0; j < 8; ++j) { f[zl(ba + j)] = v[j]; f[zl(bb + j)] = u[j]; }
    }
    if (tid == nt - 1) {                                   // residue 288: d pairs with 7 - d
        const int ba = envlin_res_base(288);
        zc v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f[zl(ba + j)];
        envlin_bf8<false>(v);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int k = 288 + 576 * d;
            envlin_pair(v[d], v[7 - d], wk[k], gam[k], gam[kLinH - k]);
        }
        envlin_bf8<true>(v);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[zl(ba + j)] = v[j];
    }
    if (tid == nt - 65) {                                  // residue 0: d pairs with 8 - d; 0 vanishes, 4 is H / 2
        zc v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f[zl(j)];
        envlin_bf8<false>(v);
        v[0] = zc{0.0, 0.0};
        v[4] = envlin_mid_point(v[4], gam);
#pragma unroll
        for (int d = 1; d < 4; ++d) envlin_pair(v[d], v[8 - d], wk[576 * d], gam[576 * d], gam[kLinH - 576 * d]);
        envlin_bf8<true>(v);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[zl(j)] = v[j];
    }
}

}  // namespace echo
