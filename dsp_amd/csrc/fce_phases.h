// Full chain, band stage and envelope branch in one workgroup per (frame, band) (SonarWorker._process_band_gpu,
// airSonar_optimized.py:629-654): band_sig = fftconvolve(filtfilt(...), taps, 'same') (:193-194) by the sliding dot
// products of band_phases.h (the same sums as band_kernel: identical band_sig), corr = correlate(band_sig, chirp,
// 'full') (:639) as ONE real cyclic convolution with the reversed chirp, and gpu_envelope_detection(corr, 5) (:642) on
// it while it is still in LDS.  Included at the end of env_phases.h.
//
// corr[k] = sum_l band_sig[l] chirp[l - k + M - 1] is the linear convolution of band_sig (N samples) with the reversed
// chirp (M samples): L = N + M - 1 <= 4608 values, so a cyclic convolution of length 4608 holds it without wrap.  The
// signal is real: packed as z[n] = band_sig[2n] + i band_sig[2n+1] it is a complex transform of 2304 = 4 * 9 * 8 * 8
// points, in place in the slots the Hilbert transform of envlin_phases.h will use next (same padding), with the same
// pair phase between the forward and the inverse passes, the spectrum of the reversed chirp in the place of the Hilbert
// kernel's.  Against 275 multiply-adds per lag (1.25 M per frame and band) the transform pair costs about 0.1 M; the
// values agree with the direct sums to about 1e-16 of the correlation's maximum, uniformly over the lags.  That is
// enough for the envelope branch (its decisions hang on the large values), not for first_strong_peak's argmax over
// near-equal lags: the few (frame, band)s where the envelope branch finds no peak, and the record therefore comes from
// first_strong_peak (:657), are evaluated again by band_kernel with direct sums (see echo_abi.hip).
#pragma once

namespace echo {

constexpr int kFceH = 2304, kFceF2 = 2 * kFceH;
constexpr int kFceTw1 = 3 * 576, kFceTw2 = kFceTw1 + 8 * 64, kFceTwN = kFceTw2 + 7 * 8;
// position of frequency f after the forward passes (radices 4, 9, 8, 8: strides 576, 64, 8, 1)
ECHO_HD int fce_res_base(int r) { return (r & 3) * 576 + ((r >> 2) % 9) * 64 + (r / 36) * 8; }

// LDS of the fused kernel inside the envelope kernel's layout: the FIR input and the taps go where buffer 1 will be
inline bool plan_fce(EnvParams& p, int N, int M, int K) {
    if (!p.lin || p.L != N + M - 1 || p.L > kFceF2 || N < 2 * kR || K < 1 || K % 2 == 0) return false;
    p.N = N; p.K = K; p.kgroups = (K + kR - 1) / kR;
    p.ypad = K - 1 - (K - 1) / 2;
    p.ybuf_len = (((N + kR - 1) / kR + p.kgroups + 1) * kR + 1) & ~1;
    if ((N + kR - 1) / kR > kEnvNT) return false;                      // one group of kR outputs per thread
    p.off_ybuf = kLinBuf1;
    p.off_ctap = p.off_ybuf + p.ybuf_len * 8;
    if (p.off_ctap + (p.kgroups + 1) * kR * 8 > kLinBytes) return false;
    if (zl((N + 1) / 2 - 1) * 16 + 16 > kLinBuf1) return false;       // the packed band signal stays below the FIR input
    return true;
}

// tw: pass twiddles as in envlin_tables for the passes (S, R) = (2304, 4), (576, 9), (64, 8); wk[k] = exp(-2 pi i k / 4608),
// k <= 2304; g[band][k] = DFT_4608(reversed chirp of the band)[k] / 2304, k <= 2304.
// lane order of the middle phase: item t = (d2 = t & 3, d1 = (t >> 2) % 9, d0 = (t >> 2) / 9) is the residue
// d0 + 4 d1 + 36 d2
inline int fce_mid_residue(int t) { return (t >> 2) / 9 + 4 * ((t >> 2) % 9) + 36 * (t & 3); }
// wmid[d * 144 + t] = wk[k]; gmid[(band * 8 + d) * 144 + t][2] = g[band][k], g[band][H - k], k = residue(t) + 288 d
inline void fce_mid_tables(int nb, const std::vector<zc>& wk, const std::vector<zc>& g, std::vector<zc>& wmid,
                           std::vector<zc>& gmid) {
    wmid.assign(8 * 144, zc{0.0, 0.0}); gmid.assign((size_t)nb * 8 * 144 * 2, zc{0.0, 0.0});
    for (int t = 1; t < 144; ++t)
        for (int d = 0; d < 8; ++d) {
            const int k = fce_mid_residue(t) + 288 * d;
            wmid[d * 144 + t] = wk[k];
            for (int b = 0; b < nb; ++b) {
                gmid[(((size_t)b * 8 + d) * 144 + t) * 2] = g[(size_t)b * (kFceH + 1) + k];
                gmid[(((size_t)b * 8 + d) * 144 + t) * 2 + 1] = g[(size_t)b * (kFceH + 1) + kFceH - k];
            }
        }
}
inline void fce_tables(int M, int nb, const double* chirps, std::vector<zc>& tw, std::vector<zc>& wk, std::vector<zc>& g) {
    const long double tau = 6.283185307179586476925286766559005768L;
    tw.clear();
    const int radix[3] = {4, 9, 8};
    for (int ps = 0, S = kFceH; ps < 3; ++ps) {
        const int r = radix[ps], q = S / r;
        for (int m = 1; m < r; ++m)
            for (int i = 0; i < q; ++i) {
                const long double a = -tau * (long double)((long)i * m % S) / (long double)S;
                tw.push_back(zc{(double)cosl(a), (double)sinl(a)});
            }
        S = q;
    }
    wk.resize(kFceH + 1);
    std::vector<long double> cs(kFceF2), sn(kFceF2);
    for (int t = 0; t < kFceF2; ++t) {
        cs[t] = cosl(tau * (long double)t / (long double)kFceF2); sn[t] = sinl(tau * (long double)t / (long double)kFceF2);
    }
    for (int k = 0; k <= kFceH; ++k) wk[k] = zc{(double)cs[k], (double)(-sn[k])};
    g.assign((size_t)nb * (kFceH + 1), zc{0.0, 0.0});
    for (int b = 0; b < nb; ++b)
        for (int k = 0; k <= kFceH; ++k) {
            long double re = 0.0L, im = 0.0L;
            for (int j = 0; j < M; ++j) {                   // reversed chirp: sample j is chirp[M - 1 - j]
                const long double c = chirps[(size_t)b * M + (M - 1 - j)];
                const int t = (int)(((long)k * j) % kFceF2);
                re += c * cs[t]; im -= c * sn[t];
            }
            g[(size_t)b * (kFceH + 1) + k] = zc{(double)(re / kFceH), (double)(im / kFceH)};
        }
}

// ---- phases
// the IIR output with the FIR's zero padding, and this band's reversed taps
ECHO_HD void ph_fce_stage(int tid, int nt, int blk, const EnvParams& p, unsigned char* smem) {
    double* yb = reinterpret_cast<double*>(smem + p.off_ybuf);
    double* ct = reinterpret_cast<double*>(smem + p.off_ctap);
    const int band = blk % p.nb;
    const int ntap = (p.kgroups + 1) * kR;
    for (int i = tid; i < ntap; i += nt) ct[i] = p.taps_rev[(size_t)band * ntap + i];
    const double* y = p.y_in + (size_t)blk * p.N;
    constexpr int kBatch = 5;
    for (int i0 = tid; i0 < p.ybuf_len; i0 += kBatch * nt) {
        double v[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
            const int n = i0 + k * nt - p.ypad;
            const int nc = n < 0 ? 0 : (n >= p.N ? p.N - 1 : n);
            const double x = y[nc];
            v[k] = (n == nc) ? x : 0.0;
        }
#pragma unroll
        for (int k = 0; k < kBatch; ++k)
            if (i0 + k * nt < p.ybuf_len) yb[i0 + k * nt] = v[k];
    }
    if (tid == 0) {
        EnvMisc* m = env_misc(smem, p);
        m->undecided = 0; m->best_idx = -1; m->best_val = 0.0; m->n_final = 0; m->n_cand = 0; m->n_kept = 0;
    }
}
// band_sig[n] = sum_j taps_rev[j] ybuf[n + j] into the packed slots (sample n is component n & 1 of slot n / 2); zeros
// up to the transform's length
ECHO_HD void ph_fce_fir(int tid, int nt, const EnvParams& p, unsigned char* smem) {
    const double* yb = reinterpret_cast<const double*>(smem + p.off_ybuf);
    const double* ct = reinterpret_cast<const double*>(smem + p.off_ctap);
    double* z = reinterpret_cast<double*>(smem);
    for (int i = p.N + tid; i < kFceF2; i += nt) z[2 * zl(i >> 1) + (i & 1)] = 0.0;
    const int n0 = tid * kR;
    if (n0 >= p.N) return;
    double acc[kR];
    sliding_dot(yb + n0, ct, p.kgroups, acc);
#pragma unroll
    for (int r = 0; r < kR; ++r) {
        const int n = n0 + r;
        if (n < p.N) z[2 * zl(n >> 1) + (n & 1)] = acc[r];
    }
}
// first forward pass (radix 4 on the whole block); its inputs are the band signal itself, which leaves here as float32
// for the spectrum kernel: consecutive lanes hold consecutive sample pairs
ECHO_HD void ph_fce_pass0(int tid, int nt, int blk, const EnvParams& p, unsigned char* smem) {
    zc* f = reinterpret_cast<zc*>(smem);
    constexpr int q = kFceH / 4;
    float* bf = p.bandf_out ? p.bandf_out + (size_t)blk * p.N : nullptr;
    const bool al8 = (((size_t)blk * p.N) & 1) == 0;        // the row starts on an 8-byte boundary
    for (int i = tid; i < q; i += nt) {
        zc v[4], w[4];
#pragma unroll
        for (int m = 1; m < 4; ++m) w[m] = p.fce_tw[(m - 1) * q + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = f[zl(i + j * q)];
        if (bf) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = 2 * (i + j * q);
                if (n + 1 < p.N && al8) { float2 o; o.x = (float)v[j].x; o.y = (float)v[j].y; *reinterpret_cast<float2*>(bf + n) = o; }
                else { if (n < p.N) bf[n] = (float)v[j].x; if (n + 1 < p.N) bf[n + 1] = (float)v[j].y; }
            }
        }
        envlin_bf<4, false>(v);
#pragma unroll
        for (int m = 1; m < 4; ++m) v[m] = zc_mul(v[m], w[m]);
#pragma unroll
        for (int j = 0; j < 4; ++j) f[zl(i + j * q)] = v[j];
    }
}
// pair (k, H - k) of the 2304-point transform: unpack, multiply by the reversed chirp's spectrum, pack for the inverse
ECHO_HD void fce_pair(zc& zk, zc& zh, zc w, zc gk, zc gc) {
    const zc E{0.5 * (zk.x + zh.x), 0.5 * (zk.y - zh.y)};
    const zc D{0.5 * (zk.x - zh.x), 0.5 * (zk.y + zh.y)};
    const zc O{D.y, -D.x};
    const zc wO = zc_mul(w, O);
    const zc Xk{E.x + wO.x, E.y + wO.y};
    const zc Xc{E.x - wO.x, -(E.y - wO.y)};
    const zc Yk = zc_mul(Xk, gk), Yc = zc_mul(Xc, gc);
    const zc Ep{0.5 * (Yk.x + Yc.x), 0.5 * (Yk.y - Yc.y)};
    const zc Dp{0.5 * (Yk.x - Yc.x), 0.5 * (Yk.y + Yc.y)};
    const zc Op = zc_mulc(Dp, w);
    zk = zc{Ep.x - Op.y, Ep.y + Op.x};
    zh = zc{Ep.x + Op.y, Op.x - Ep.y};
}
// last forward pass, pair phase and first inverse pass in registers, as ph_envlin_mid: the butterfly with residue r
// (frequencies r + 288 d) has its partners in the butterfly with residue 288 - r, at 7 - d
ECHO_HD void ph_fce_mid(int tid, int nt, int blk, const EnvParams& p, unsigned char* smem) {
    zc* f = reinterpret_cast<zc*>(smem);
    const zc* wk = p.fce_wk;
    const zc* g = p.fce_g + (size_t)(blk % p.nb) * (kFceH + 1);
    const zc* wmid = p.fce_wmid;
    const zc* gmid = p.fce_gmid + (size_t)(blk % p.nb) * 8 * 144 * 2;
    // residues below 144 = butterflies in the first half of their 64-element block (r = d0 + 4 d1 + 36 d2 at element
    // 576 d0 + 64 d1 + 8 d2, d2 < 4); lanes run over d2, then d1, then d0 (LDS banks, as in ph_envlin_mid)
    for (int t = 1 + tid; t < 144; t += nt) {
        const int d2 = t & 3, q = t >> 2, d1 = q % 9, d0 = q / 9;
        const int r = d0 + 4 * d1 + 36 * d2;                // == fce_mid_residue(t)
        const int ba = d0 * 576 + d1 * 64 + d2 * 8, bb = fce_res_base(288 - r);
        zc v[8], u[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = f[zl(ba + j)]; u[j] = f[zl(bb + j)]; }
        envlin_bf8<false>(v); envlin_bf8<false>(u);
#pragma unroll
        for (int d = 0; d < 8; ++d)
            fce_pair(v[d], u[7 - d], wmid[d * 144 + t], gmid[(d * 144 + t) * 2], gmid[(d * 144 + t) * 2 + 1]);
        envlin_bf8<true>(v); envlin_bf8<true>(u);
#pragma unroll
        for (int j = 0; j < 8; ++j) { f[zl(ba + j)] = v[j]; f[zl(bb + j)] = u[j]; }
    }
    if (tid == nt - 1) {                                   // residue 144: d pairs with 7 - d
        const int ba = fce_res_base(144);
        zc v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f[zl(ba + j)];
        envlin_bf8<false>(v);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int k = 144 + 288 * d;
            fce_pair(v[d], v[7 - d], wk[k], g[k], g[kFceH - k]);
        }
        envlin_bf8<true>(v);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[zl(ba + j)] = v[j];
    }
    if (tid == nt - 65) {                                  // residue 0: d pairs with 8 - d; 0 and H / 2 (d = 4) stand alone
        zc v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f[zl(j)];
        envlin_bf8<false>(v);
        {   // X[0] = Re Z + Im Z, X[H] = Re Z - Im Z, both real, as are G[0] and G[H]
            const double y0 = (v[0].x + v[0].y) * g[0].x, yh = (v[0].x - v[0].y) * g[kFceH].x;
            v[0] = zc{0.5 * (y0 + yh), 0.5 * (y0 - yh)};
            const zc y = zc_mul(zc{v[4].x, -v[4].y}, g[kFceH / 2]);          // X[H/2] = conj Z[H/2]
            v[4] = zc{y.x, -y.y};
        }
#pragma unroll
        for (int d = 1; d < 4; ++d) fce_pair(v[d], v[8 - d], wk[288 * d], g[288 * d], g[kFceH - 288 * d]);
        envlin_bf8<true>(v);
#pragma unroll
        for (int j = 0; j < 8; ++j) f[zl(j)] = v[j];
    }
}
// slot m now holds corr[2m] + i corr[2m+1]: its statistics start, the slots behind it are cleared for the Hilbert
// transform, which finds its input in place, and every thread keeps the samples it will need again for the envelope in
// registers (the transform overwrites them, and two workgroups' buffers leave no LDS for a copy).  The correlation goes
// to global memory only when somebody wants it (corr_w: the chain's raw correlation for echo_chain_wave).
struct FceKeep { zc x[kEnvIterN]; };
ECHO_HD EnvPart1 ph_fce_corr_out(int tid, int nt, int blk, const EnvParams& p, unsigned char* smem, FceKeep& keep) {
    zc* z = reinterpret_cast<zc*>(smem);
    double* out = p.corr_w ? p.corr_w + (size_t)blk * p.L : nullptr;
    EnvPart1 a;
    a.v[0] = 0.0; a.v[1] = -INFINITY; a.v[2] = INFINITY; a.v[3] = 0.0; a.v[4] = 0.0; a.v[5] = 0.0; a.v[6] = 0.0;
#pragma unroll
    for (int t = 0; t < kEnvIterN; ++t) {
        const int m = tid + t * nt;
        if (m >= p.n) { keep.x[t] = zc{0.0, 0.0}; continue; }
        const zc x = z[zl(m)];
        keep.x[t] = x;
        if (out) *reinterpret_cast<zc*>(out + 2 * m) = x;
        const double a0 = fabs(x.x), a1 = fabs(x.y);
        a.v[3] += x.x + x.y; a.v[4] += x.x * x.x + x.y * x.y;
        a.v[5] = a0 > a.v[5] ? a0 : a.v[5]; a.v[5] = a1 > a.v[5] ? a1 : a.v[5];
    }
    for (int m = p.n + tid; m < kLinH; m += nt) z[zl(m)] = zc{0.0, 0.0};
    return a;
}
// envelope[i] = |x[i] + i H(x)[i]| into buffer 1, x from the registers
ECHO_HD void ph_fce_abs(int tid, int nt, const EnvParams& p, unsigned char* smem, const FceKeep& keep) {
    const zc* v = reinterpret_cast<const zc*>(smem);
    double* env = reinterpret_cast<double*>(env_buf(smem, p, 1));
#pragma unroll
    for (int t = 0; t < kEnvIterN; ++t) {
        const int m = tid + t * nt;
        if (m >= p.n) continue;
        const zc h = v[zl(m)];
        env[2 * m] = hypot(keep.x[t].x, h.x);
        env[2 * m + 1] = hypot(keep.x[t].y, h.y);
    }
}

#define ECHO_FCE_SEQUENCE(PHASE, SYNC, PART1_DECL, PART1_REF, KEEP_DECL, KEEP_REF, p, smem, blk, nt)         \
    do {                                                                                                     \
        PART1_DECL;                                                                                          \
        KEEP_DECL;                                                                                           \
        int cur_ = 0;                                                                                        \
        PHASE(ph_fce_stage(tid, nt, blk, p, smem)) SYNC                                                      \
        PHASE(ph_fce_fir(tid, nt, p, smem)) SYNC                                                             \
        if ((p).stop_after == 20) break;                                                                     \
        PHASE(ph_fce_pass0(tid, nt, blk, p, smem)) SYNC                                                      \
        PHASE(ph_envlin_pass<kFceH, 9, 576, false>(tid, nt, (p).fce_tw + kFceTw1, smem)) SYNC                \
        PHASE(ph_envlin_pass<kFceH, 8, 64, false>(tid, nt, (p).fce_tw + kFceTw2, smem)) SYNC                 \
        PHASE(ph_fce_mid(tid, nt, blk, p, smem)) SYNC                                                        \
        PHASE(ph_envlin_pass<kFceH, 8, 64, true>(tid, nt, (p).fce_tw + kFceTw2, smem)) SYNC                  \
        PHASE(ph_envlin_pass<kFceH, 9, 576, true>(tid, nt, (p).fce_tw + kFceTw1, smem)) SYNC                 \
        PHASE(ph_envlin_pass<kFceH, 4, 2304, true>(tid, nt, (p).fce_tw, smem)) SYNC                          \
        if ((p).stop_after == 21) break;                                                                     \
        PHASE(PART1_REF = ph_fce_corr_out(tid, nt, blk, p, smem, KEEP_REF)) SYNC                             \
        if ((p).stop_after == 6) break;                                                                      \
        ECHO_ENV_LIN_HILBERT(PHASE, SYNC, p, smem, blk, nt)                                                  \
        PHASE(ph_fce_abs(tid, nt, p, smem, KEEP_REF)) SYNC                                                   \
        cur_ ^= 1;                                                                                           \
        ECHO_ENV_TAIL(PHASE, SYNC, PART1_REF, p, smem, blk, nt);                                             \
    } while (0)

}  // namespace echo
