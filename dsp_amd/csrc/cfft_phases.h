// Matched filter of long frames by FFT (float64): corr[k] = sum_j chirp[j] * band_sig[k - (M-1) + j]
// (scipy.signal.correlate(band_sig, chirp, 'full'), airSonar.py:224; 666.py two-peak variant likewise) for chirps of
// thousands of samples, where the direct sum of long_phases.h costs M multiply-adds per lag.
//
// Overlap-save with a cyclic length of F = 8192: a sub-block of V = F - M + 1 lags starting at lag kA is the cyclic
// correlation of u[i] = band_sig[kA - (M-1) + i], 0 <= i < F, with the zero-padded chirp; no lag of it wraps.  The
// chirp is real, so two sub-blocks share one complex transform: z = u1 + i*u2, IFFT(FFT(z) * G) = c1 + i*c2 with
// G = conj(FFT(chirp)) / F.  One workgroup of 1024 threads per (pair of sub-blocks, frame, band); the transform stays
// in LDS (8192 complex doubles, one 16-byte pad per 8 elements = 147,456 bytes):
//   load fused with the first radix-2 stage (stride 4096)   -> radix-8 passes with strides 512, 64, 8
//   -> 8 contiguous elements per thread: last forward stage, product with G (stored in pass order), first inverse stage
//   -> inverse radix-8 passes with strides 8, 64, 512       -> last radix-2 stage fused with the store of Re / Im.
// Forward passes are decimation in frequency, the inverse ones their conjugate transposes in reverse order, so no
// reordering pass exists.  Plain C++: phases of (tid, nt) like long_phases.h, so a host build can run them.
#pragma once
#include "long_phases.h"

namespace echo {

constexpr int kCfF = 8192;                       // cyclic length (complex points)
constexpr int kCfNT = 1024;                      // threads per workgroup: one radix-8 butterfly each
constexpr int kCfSlots = kCfF + kCfF / 8;        // padded LDS slots
constexpr int kCfSmem = kCfSlots * 16;
constexpr int kCfTpExtra = 3 * kLDetNT * 8 + 64;           // two-peak detector: per-thread partials and scalars
constexpr int kCfDetSmem = kCfSmem + (kDetExtra > kCfTpExtra ? kDetExtra : kCfTpExtra);   // with the detector's histograms and scalars behind the transform
constexpr int kCfTwLen = 3 * (512 + 64 + 8);     // pass twiddles: per stride q the runs W^i, W^2i, W^4i (i < q), W = exp(-2 pi i/(8q))

typedef cpx<double> zd;

ECHO_HD zd zmk(double x, double y) { zd r; r.x = x; r.y = y; return r; }
ECHO_HD zd zadd(zd a, zd b) { return zmk(a.x + b.x, a.y + b.y); }
ECHO_HD zd zsub(zd a, zd b) { return zmk(a.x - b.x, a.y - b.y); }
ECHO_HD zd zmul(zd a, zd b) { return zmk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
ECHO_HD zd zmulc(zd a, zd b) { return zmk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)
template <bool INV> ECHO_HD zd zmul_mi(zd a) { return INV ? zmk(-a.y, a.x) : zmk(a.y, -a.x); }   // a * (-i) or a * (+i)

template <bool INV> ECHO_HD void zdft4(zd& x0, zd& x1, zd& x2, zd& x3) {
    const zd t0 = zadd(x0, x2), t1 = zsub(x0, x2), t2 = zadd(x1, x3), t3 = zmul_mi<INV>(zsub(x1, x3));
    x0 = zadd(t0, t2); x2 = zsub(t0, t2); x1 = zadd(t1, t3); x3 = zsub(t1, t3);
}

// 8-point DFT, natural order in and out
template <bool INV> ECHO_HD void zdft8(zd (&v)[8]) {
    const double h = 0.70710678118654752440;
    zd a0 = zadd(v[0], v[4]), a1 = zadd(v[1], v[5]), a2 = zadd(v[2], v[6]), a3 = zadd(v[3], v[7]);
    zd b0 = zsub(v[0], v[4]), b1 = zsub(v[1], v[5]), b2 = zsub(v[2], v[6]), b3 = zsub(v[3], v[7]);
    if (!INV) { b1 = zmk(h * (b1.x + b1.y), h * (b1.y - b1.x));
                b2 = zmul_mi<false>(b2);
                b3 = zmk(h * (b3.y - b3.x), -h * (b3.x + b3.y)); }
    else      { b1 = zmk(h * (b1.x - b1.y), h * (b1.y + b1.x));
                b2 = zmul_mi<true>(b2);
                b3 = zmk(-h * (b3.x + b3.y), h * (b3.x - b3.y)); }
    zdft4<INV>(a0, a1, a2, a3);
    zdft4<INV>(b0, b1, b2, b3);
    v[0] = a0; v[2] = a1; v[4] = a2; v[6] = a3;
    v[1] = b0; v[3] = b1; v[5] = b2; v[7] = b3;
}

ECHO_HD int zpos(int e) { return e + (e >> 3); }

// frequency index held by element e after the forward passes (radices 2, 8, 8, 8, 8)
inline int cf_pass_freq(int e) {
    const int m0 = e / 4096, m1 = (e / 512) % 8, m2 = (e / 64) % 8, m3 = (e / 8) % 8, m4 = e % 8;
    return m0 + 2 * m1 + 16 * m2 + 128 * m3 + 1024 * m4;
}
ECHO_HD int cf_tw_offset(int q) { return q == 512 ? 0 : (q == 64 ? 3 * 512 : 3 * 512 + 3 * 64); }

// first lags of the two sub-blocks of a pair
ECHO_HD int cf_first_lag(const LongParams& p, int pair, int which) { return p.k_begin + (2 * pair + which) * p.cf_V; }

// u1 + i*u2 and the radix-2 stage: f[i] = a + b, f[i + F/2] = (a - b) * W_F^i
ECHO_HD void ph_cf_load(int tid, int nt, int pair, int blk, const LongParams& p, unsigned char* smem) {
    zd* f = reinterpret_cast<zd*>(smem);
    const double* x = p.band_in + (size_t)blk * p.N;
    const int kA1 = cf_first_lag(p, pair, 0), kA2 = cf_first_lag(p, pair, 1);
    const int o1 = kA1 - (p.M - 1), o2 = kA2 - (p.M - 1);
    const bool two = kA2 < p.L;
    for (int i = tid; i < kCfF / 2; i += nt) {
        const int n1a = o1 + i, n1b = n1a + kCfF / 2, n2a = o2 + i, n2b = n2a + kCfF / 2;
        zd a, b;
        a.x = (n1a >= 0 && n1a < p.N) ? x[n1a] : 0.0;
        b.x = (n1b >= 0 && n1b < p.N) ? x[n1b] : 0.0;
        a.y = (two && n2a >= 0 && n2a < p.N) ? x[n2a] : 0.0;
        b.y = (two && n2b >= 0 && n2b < p.N) ? x[n2b] : 0.0;
        f[zpos(i)] = zadd(a, b);
        f[zpos(i + kCfF / 2)] = zmul(zsub(a, b), p.cf_w2[i]);
    }
}

// radix-8 pass with stride q inside blocks of 8q elements.  Forward: butterfly, then twiddle W^(i*m) on output m;
// inverse: conjugate twiddle, then butterfly.
template <bool INV> ECHO_HD void ph_cf_r8(int tid, int nt, int q, const LongParams& p, unsigned char* smem) {
    zd* f = reinterpret_cast<zd*>(smem);
    const zd* tw = p.cf_tw + cf_tw_offset(q);
    for (int t = tid; t < kCfF / 8; t += nt) {
        const int i = t % q, base = (t / q) * 8 * q + i;
        const zd w1 = tw[i], w2 = tw[q + i], w4 = tw[2 * q + i];
        zd v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f[zpos(base + j * q)];
        const zd w3 = zmul(w1, w2), w5 = zmul(w1, w4), w6 = zmul(w2, w4), w7 = zmul(w3, w4);
        if (INV) {
            v[1] = zmulc(v[1], w1); v[2] = zmulc(v[2], w2); v[3] = zmulc(v[3], w3); v[4] = zmulc(v[4], w4);
            v[5] = zmulc(v[5], w5); v[6] = zmulc(v[6], w6); v[7] = zmulc(v[7], w7);
        }
        zdft8<INV>(v);
        if (!INV) {
            v[1] = zmul(v[1], w1); v[2] = zmul(v[2], w2); v[3] = zmul(v[3], w3); v[4] = zmul(v[4], w4);
            v[5] = zmul(v[5], w5); v[6] = zmul(v[6], w6); v[7] = zmul(v[7], w7);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) f[zpos(base + j * q)] = v[j];
    }
}

// last forward stage, product with G (pass order, 1/F folded in), first inverse stage
ECHO_HD void ph_cf_mid(int tid, int nt, int blk, const LongParams& p, unsigned char* smem) {
    zd* f = reinterpret_cast<zd*>(smem);
    const zd* g = p.cf_g + (size_t)(blk % p.nb) * kCfF;
    for (int t = tid; t < kCfF / 8; t += nt) {
        zd v[8], gg[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) gg[j] = g[8 * t + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f[9 * t + j];                  // zpos(8*t + j)
        zdft8<false>(v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = zmul(v[j], gg[j]);
        zdft8<true>(v);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[9 * t + j] = v[j];
    }
}

// last inverse stage y[i] = A + conj(W^i) B, y[i + F/2] = A - conj(W^i) B; real part -> first sub-block, imaginary -> second
ECHO_HD void ph_cf_store(int tid, int nt, int pair, int blk, const LongParams& p, unsigned char* smem) {
    const zd* f = reinterpret_cast<const zd*>(smem);
    const int kA1 = cf_first_lag(p, pair, 0), kA2 = cf_first_lag(p, pair, 1);
    double* out = p.corr_out + (size_t)blk * p.corr_stride - p.corr_k0;
    for (int i = tid; i < kCfF / 2; i += nt) {
        if (i >= p.cf_V) break;
        const zd A = f[zpos(i)], Bw = zmulc(f[zpos(i + kCfF / 2)], p.cf_w2[i]);
        const zd y0 = zadd(A, Bw), y1 = zsub(A, Bw);
        if (kA1 + i < p.L) out[kA1 + i] = y0.x;
        if (kA2 + i < p.L) out[kA2 + i] = y0.y;
        const int i1 = i + kCfF / 2;
        if (i1 < p.cf_V) {
            if (kA1 + i1 < p.L) out[kA1 + i1] = y1.x;
            if (kA2 + i1 < p.L) out[kA2 + i1] = y1.y;
        }
    }
}

// The last inverse stage in place, for a detector that continues in the same workgroup: slot zpos(i) then holds lag i
// of the first sub-block in its real part and of the second in its imaginary part (LongParams::det_view == 1).  Every
// thread rewrites the two slots it read.  Also clears the detector's scalars.
ECHO_HD void ph_cf_finish_inplace(int tid, int nt, const LongParams& p, unsigned char* smem) {
    zd* f = reinterpret_cast<zd*>(smem);
    for (int i = tid; i < kCfF / 2; i += nt) {
        const zd A = f[zpos(i)], Bw = zmulc(f[zpos(i + kCfF / 2)], p.cf_w2[i]);
        f[zpos(i)] = zadd(A, Bw);
        f[zpos(i + kCfF / 2)] = zsub(A, Bw);
    }
    if (tid == 0) det_misc_init(det_misc(p, smem));
}

#define ECHO_CF_TRANSFORM(PHASE, SYNC, p, smem, pair, blk, nt)                     \
    PHASE(ph_cf_load(tid, nt, pair, blk, p, smem)) SYNC                             \
    PHASE(ph_cf_r8<false>(tid, nt, 512, p, smem)) SYNC                              \
    PHASE(ph_cf_r8<false>(tid, nt, 64, p, smem)) SYNC                               \
    PHASE(ph_cf_r8<false>(tid, nt, 8, p, smem)) SYNC                                \
    PHASE(ph_cf_mid(tid, nt, blk, p, smem)) SYNC                                    \
    PHASE(ph_cf_r8<true>(tid, nt, 8, p, smem)) SYNC                                 \
    PHASE(ph_cf_r8<true>(tid, nt, 64, p, smem)) SYNC                                \
    PHASE(ph_cf_r8<true>(tid, nt, 512, p, smem)) SYNC

// matched filter and the median/MAD detector in one workgroup: the lags [k_begin, L) must fit one pair of sub-blocks
#define ECHO_CF_DET_SEQUENCE(PHASE, SYNC, p, smem, blk, nt)                         \
    ECHO_CF_TRANSFORM(PHASE, SYNC, p, smem, 0, blk, nt)                             \
    PHASE(ph_cf_finish_inplace(tid, nt, p, smem)) SYNC                              \
    ECHO_LDET_BODY(PHASE, SYNC, p, smem, blk, nt)

// matched filter and the two-peak detector (666.py:153-200) in one workgroup.  The detector's phases run on the first
// kLDetNT threads, so that its sums are formed in the order of the detector kernel
#define ECHO_CF_TP_SEQUENCE(PHASE, SYNC, p, smem, blk, nt)                                     \
    ECHO_CF_TRANSFORM(PHASE, SYNC, p, smem, 0, blk, nt)                                         \
    PHASE(ph_cf_finish_inplace(tid, nt, p, smem)) SYNC                                          \
    PHASE(if (tid < kLDetNT) ph_tp_scan_view(tid, kLDetNT, p, smem)) SYNC                       \
    PHASE(if (tid < kLDetNT) ph_tp_ref_fold(tid, kLDetNT, p, smem)) SYNC                        \
    PHASE(if (tid < kLDetNT) ph_tp_ref(tid, kLDetNT, p, smem)) SYNC                             \
    PHASE(if (tid < kLDetNT) ph_tp_second(tid, kLDetNT, p, smem)) SYNC                          \
    PHASE(if (tid < kLDetNT) ph_tp_second_fold(tid, kLDetNT, p, smem)) SYNC                     \
    PHASE(if (tid < kLDetNT) ph_tp_finish(tid, kLDetNT, blk, p, smem))

#define ECHO_CF_SEQUENCE(PHASE, SYNC, p, smem, pair, blk, nt)                      \
    PHASE(ph_cf_load(tid, nt, pair, blk, p, smem)) SYNC                             \
    PHASE(ph_cf_r8<false>(tid, nt, 512, p, smem)) SYNC                              \
    PHASE(ph_cf_r8<false>(tid, nt, 64, p, smem)) SYNC                               \
    PHASE(ph_cf_r8<false>(tid, nt, 8, p, smem)) SYNC                                \
    PHASE(ph_cf_mid(tid, nt, blk, p, smem)) SYNC                                    \
    PHASE(ph_cf_r8<true>(tid, nt, 8, p, smem)) SYNC                                 \
    PHASE(ph_cf_r8<true>(tid, nt, 64, p, smem)) SYNC                                \
    PHASE(ph_cf_r8<true>(tid, nt, 512, p, smem)) SYNC                               \
    PHASE(ph_cf_store(tid, nt, pair, blk, p, smem))

// number of sub-block pairs that cover the lags [k_begin, L)
ECHO_HD int cf_pairs(const LongParams& p) {
    const int nsb = (p.L - p.k_begin + p.cf_V - 1) / p.cf_V;
    return (nsb + 1) / 2;
}

}  // namespace echo
