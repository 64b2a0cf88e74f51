// Per-(frame, band) workgroup algorithm, written as barrier-separated phases.
//
// Each phase is a function of (tid, nt): the work one thread of an nt-thread workgroup does between
// two workgroup barriers.  The HIP kernel calls the phases with __syncthreads() in between; a host
// build (tests/host_emul) runs the same functions for tid = 0..nt-1 in a loop, which is how the index
// arithmetic is unit-tested without a GPU.  The file is therefore plain C++ with no HIP intrinsics.
//
// Chain implemented (reference: airSonar_optimized.py):
//   y (zero-phase IIR output, float64) --FIR 'same'--> band_sig            :193-194
//   band_sig --sliding dot with chirp--> corr (scipy.signal.correlate)      :290-303
//   corr --first_strong_peak--> peak_idx, snr_db ; amplitude = max|corr|    :200-219, :666
//   band_sig --rfft (Bluestein on the packed half-length signal), dB-->     :710, :286-287
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define ECHO_HD __host__ __device__ __forceinline__
#else
#define ECHO_HD inline
struct float2 { float x, y; };
#endif

namespace echo {

constexpr int kR = 9;          // consecutive outputs per lane in the sliding dot products
constexpr int kFftM = 8192;    // Bluestein cyclic length (>= 2*(N/2)-1), 8*8*8*4*4
constexpr int kNT = 512;       // threads per workgroup of the band kernel

struct BandRes {               // per (frame, band) detector output, consumed by the fusion kernel
    double snr_db;
    double amplitude;
    int32_t peak_idx;          // -1 = None
    int32_t pad;
};

enum { MODE_RX_FIR = 0,        // input rx float32, FIR only (CuPy branch of gpu_bandpass)
       MODE_Y_FIR = 1,         // input IIR output float64, FIR -> band_sig (bandpass)
       MODE_BAND = 2,          // input band_sig float64 (standalone correlate / spectrum)
       MODE_CORR = 3 };        // input corr float64 (standalone first_strong_peak, mutates it)

struct BandParams {
    int N, M, L, half, nb, K;
    int min_delay, noise_n, detector;
    int n2, nbins;
    int mode;
    int do_corr, do_spec, spec_db;
    double snr_gate;
    // padded coefficient tables (device global), zero-filled to a multiple of kR
    int kgroups, mgroups;            // ceil(K/kR), ceil(M/kR)
    const double* taps_rev;          // [nb][kgroups*kR]  taps reversed
    const double* chirp;             // [nb][mgroups*kR]
    const float2* bw;                // [n2]      exp(-i*pi*n^2/n2)
    const float2* bhat;              // [kFftM]   FFT of the chirp filter, /kFftM, in pass order
    const float2* tw;                // [kFftM]   exp(-2*pi*i*k/kFftM)
    const float2* unpack;            // [n2+1]    exp(-2*pi*i*k/N)
    // LDS carve (byte offsets, 16-B aligned)
    int off_sig, sig_len;            // double[sig_len]: band_sig at [M-1, M-1+N), zeros elsewhere
    int off_u;                       // union: double ybuf[ybuf_len] | float2 fft[kFftM] | reduction scratch
    int ybuf_len, ypad;              // ybuf holds input at [ypad, ypad+N)
    int off_nw;                      // double[noise_n] noise window
    int off_misc;                    // small scalars
    int smem_bytes;
    // I/O (device global); any may be null depending on mode
    const float* rx_in;              // [B][N]
    const double* d_in;              // [B][nb][N] (y or band_sig)
    double* corr_io;                 // [B][nb][L]  in (MODE_CORR) / out (raw corr) / null
    double* band_out;                // [B][nb][N]
    float* spec_out;                 // [B][nb][nbins]
    BandRes* res_out;                // [B][nb]
    int write_corr_mutated;          // MODE_CORR: write the zeroed blind zone back
};

struct Misc {                        // lives at off_misc
    double best; double amp; double sumsq; double med0; double med1;
    int bidx; int pad;
};

ECHO_HD double* lds_d(unsigned char* s, int off) { return reinterpret_cast<double*>(s + off); }
ECHO_HD float2* lds_c(unsigned char* s, int off) { return reinterpret_cast<float2*>(s + off); }

// out[r] = sum_j c[j] * in[r + j], r < kR, j < ngroups*kR; the window of kR inputs lives in registers
// and rotates by static index, so each tap costs one LDS read per lane.
ECHO_HD void sliding_dot(const double* in, const double* __restrict__ c, int ngroups, double (&acc)[kR]) {
    double w[kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) { w[r] = in[r]; acc[r] = 0.0; }
    for (int g = 0; g < ngroups; ++g) {
        const double* cg = c + g * kR;
        const double* nx = in + (g + 1) * kR;
#pragma unroll
        for (int t = 0; t < kR; ++t) {
            const double ct = cg[t];
#pragma unroll
            for (int r = 0; r < kR; ++r) acc[r] = fma(w[(r + t) % kR], ct, acc[r]);
            w[t] = nx[t];
        }
    }
}

// ---------------------------------------------------------------- phase 0: stage the input in LDS
ECHO_HD void ph_load(int tid, int nt, int blk, const BandParams& p, unsigned char* smem) {
    double* sig = lds_d(smem, p.off_sig);
    double* yb = lds_d(smem, p.off_u);
    const int band = blk % p.nb, frame = blk / p.nb;
    (void)band;
    if (p.mode == MODE_RX_FIR || p.mode == MODE_Y_FIR) {
        for (int i = tid; i < p.ybuf_len; i += nt) {
            const int n = i - p.ypad;
            double v = 0.0;
            if (n >= 0 && n < p.N)
                v = (p.mode == MODE_RX_FIR) ? (double)p.rx_in[(size_t)frame * p.N + n]
                                            : p.d_in[(size_t)blk * p.N + n];
            yb[i] = v;
        }
        for (int i = tid; i < p.sig_len; i += nt)
            if (i < p.M - 1 || i >= p.M - 1 + p.N) sig[i] = 0.0;
    } else if (p.mode == MODE_BAND) {
        for (int i = tid; i < p.sig_len; i += nt) {
            const int n = i - (p.M - 1);
            sig[i] = (n >= 0 && n < p.N) ? p.d_in[(size_t)blk * p.N + n] : 0.0;
        }
    }
}

// ---------------------------------------------------------------- phase 1: FIR 'same'
// band_sig[n] = sum_j taps[j] * y[n + s - j], s = (K-1)/2  ==  sum_j' taps_rev[j'] * ybuf[n + j']
ECHO_HD void ph_fir(int tid, int nt, int blk, const BandParams& p, unsigned char* smem) {
    if (p.mode != MODE_RX_FIR && p.mode != MODE_Y_FIR) return;
    const double* yb = lds_d(smem, p.off_u);
    double* sig = lds_d(smem, p.off_sig);
    const int band = blk % p.nb;
    const int n0 = tid * kR;
    if (n0 >= p.N) return;
    double acc[kR];
    sliding_dot(yb + n0, p.taps_rev + (size_t)band * p.kgroups * kR, p.kgroups, acc);
#pragma unroll
    for (int r = 0; r < kR; ++r) {
        const int n = n0 + r;
        if (n < p.N) {
            sig[p.M - 1 + n] = acc[r];
            if (p.band_out) p.band_out[(size_t)blk * p.N + n] = acc[r];
        }
    }
}

// ---------------------------------------------------------------- phase 2: correlation + detector partials
// corr[k] = sum_j sig[k + j] * chirp[j]  (sig carries M-1 leading zeros), k < L.
// Per lane: blind-zone zeroing, max|corr|, signed max of pos = corr[half:] with first-index ties,
// noise-window samples, sum of squares of pos[:noise_n] (mean-power detector).
ECHO_HD void ph_corr(int tid, int nt, int blk, const BandParams& p, unsigned char* smem) {
    if (!p.do_corr) return;
    const double* sig = lds_d(smem, p.off_sig);
    double* nw = lds_d(smem, p.off_nw);
    double* red = lds_d(smem, p.off_u);            // [nt] best | [nt] amp | [nt] sumsq | int[nt] idx
    int* redi = reinterpret_cast<int*>(red + 3 * nt);
    const int band = blk % p.nb;
    const int k0 = tid * kR;
    double best = -INFINITY, amp = 0.0, sumsq = 0.0;
    int bidx = 0x7fffffff;
    if (k0 < p.L) {
        double acc[kR];
        if (p.mode == MODE_CORR) {
#pragma unroll
            for (int r = 0; r < kR; ++r) acc[r] = (k0 + r < p.L) ? p.corr_io[(size_t)blk * p.L + k0 + r] : 0.0;
        } else {
            sliding_dot(sig + k0, p.chirp + (size_t)band * p.mgroups * kR, p.mgroups, acc);
            if (p.corr_io) {
#pragma unroll
                for (int r = 0; r < kR; ++r)
                    if (k0 + r < p.L) p.corr_io[(size_t)blk * p.L + k0 + r] = acc[r];
            }
        }
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int k = k0 + r;
            if (k >= p.L) continue;
            double v = acc[r];
            const int q = k - p.half;
            if (q >= 0 && q < p.min_delay) {
                v = 0.0;
                if (p.mode == MODE_CORR && p.write_corr_mutated) p.corr_io[(size_t)blk * p.L + k] = 0.0;
            }
            const double av = fabs(v);
            amp = av > amp ? av : amp;
            if (q >= 0) {
                if (v > best) { best = v; bidx = q; }
                if (p.detector == 0) {
                    const int w = q - p.min_delay;
                    if (w >= 0 && w < p.noise_n) nw[w] = av;
                } else if (p.detector == 2) {
                    if (q < p.noise_n) sumsq += v * v;
                }
            }
        }
    }
    red[tid] = best; red[nt + tid] = amp; red[2 * nt + tid] = sumsq; redi[tid] = bidx;
}

ECHO_HD void combine(double& best, int& bidx, double b2, int i2) {
    if (b2 > best || (b2 == best && i2 < bidx)) { best = b2; bidx = i2; }
}

// phase 3a: 64 threads fold nt partials to 64; phase 3b: thread 0 folds 64 to 1.
ECHO_HD void ph_reduce_a(int tid, int nt, const BandParams& p, unsigned char* smem) {
    if (!p.do_corr || tid >= 64) return;
    double* red = lds_d(smem, p.off_u);
    int* redi = reinterpret_cast<int*>(red + 3 * nt);
    double best = red[tid], amp = red[nt + tid], ss = red[2 * nt + tid];
    int bidx = redi[tid];
    for (int j = tid + 64; j < nt; j += 64) {
        combine(best, bidx, red[j], redi[j]);
        amp = red[nt + j] > amp ? red[nt + j] : amp;
        ss += red[2 * nt + j];
    }
    red[tid] = best; red[nt + tid] = amp; red[2 * nt + tid] = ss; redi[tid] = bidx;
}

ECHO_HD void ph_reduce_b(int tid, int nt, const BandParams& p, unsigned char* smem) {
    if (!p.do_corr || tid != 0) return;
    double* red = lds_d(smem, p.off_u);
    int* redi = reinterpret_cast<int*>(red + 3 * nt);
    Misc* m = reinterpret_cast<Misc*>(smem + p.off_misc);
    double best = red[0], amp = red[nt], ss = red[2 * nt];
    int bidx = redi[0];
    for (int j = 1; j < 64 && j < nt; ++j) {
        combine(best, bidx, red[j], redi[j]);
        amp = red[nt + j] > amp ? red[nt + j] : amp;
        ss += red[2 * nt + j];
    }
    m->best = best; m->bidx = bidx; m->amp = amp; m->sumsq = ss;
}

// phase 4: median of the noise window by rank counting (np.median: mean of the two middle order statistics)
ECHO_HD void ph_rank(int tid, int nt, const BandParams& p, unsigned char* smem) {
    if (!p.do_corr || p.detector != 0) return;
    const double* nw = lds_d(smem, p.off_nw);
    Misc* m = reinterpret_cast<Misc*>(smem + p.off_misc);
    const int n = p.noise_n;
    for (int i = tid; i < n; i += nt) {
        const double v = nw[i];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double u = nw[j];
            rank += (u < v || (u == v && j < i)) ? 1 : 0;
        }
        if (rank == (n - 1) / 2) m->med0 = v;
        if (rank == n / 2) m->med1 = v;
    }
}

// phase 5: SNR, gate, result record
ECHO_HD void ph_finish(int tid, int nt, int blk, const BandParams& p, unsigned char* smem) {
    if (!p.do_corr || tid != 0 || !p.res_out) return;
    const Misc* m = reinterpret_cast<const Misc*>(smem + p.off_misc);
    BandRes r;
    r.amplitude = m->amp; r.pad = 0;
    const int npos = p.L - p.half;
    if (npos <= p.min_delay) { r.peak_idx = -1; r.snr_db = 0.0; p.res_out[blk] = r; return; }
    double noise_floor;
    if (p.detector == 0) { const double med = (m->med0 + m->med1) / 2.0; noise_floor = med * med; }
    else                 { noise_floor = m->sumsq / (double)p.noise_n; }
    const double peak_power = m->best * m->best;
    double snr = 0.0;
    if (noise_floor > 0.0) snr = 10.0 * log10(peak_power / noise_floor);
    if (snr < p.snr_gate) { r.peak_idx = -1; r.snr_db = 0.0; }
    else                  { r.peak_idx = m->bidx; r.snr_db = snr; }
    p.res_out[blk] = r;
}

// ---------------------------------------------------------------- spectrum: FFT building blocks
ECHO_HD float2 cmul(float2 a, float2 b) { float2 r; r.x = a.x * b.x - a.y * b.y; r.y = a.x * b.y + a.y * b.x; return r; }
ECHO_HD float2 cmulc(float2 a, float2 b) { float2 r; r.x = a.x * b.x + a.y * b.y; r.y = a.y * b.x - a.x * b.y; return r; }  // a*conj(b)
ECHO_HD float2 cadd(float2 a, float2 b) { float2 r; r.x = a.x + b.x; r.y = a.y + b.y; return r; }
ECHO_HD float2 csub(float2 a, float2 b) { float2 r; r.x = a.x - b.x; r.y = a.y - b.y; return r; }
template <bool INV> ECHO_HD float2 mul_mi(float2 a) {   // forward: a * (-i); inverse: a * (+i)
    float2 r; if (!INV) { r.x = a.y; r.y = -a.x; } else { r.x = -a.y; r.y = a.x; } return r;
}

template <bool INV> ECHO_HD void dft4(float2& x0, float2& x1, float2& x2, float2& x3) {
    const float2 t0 = cadd(x0, x2), t1 = csub(x0, x2), t2 = cadd(x1, x3), t3 = mul_mi<INV>(csub(x1, x3));
    x0 = cadd(t0, t2); x2 = csub(t0, t2); x1 = cadd(t1, t3); x3 = csub(t1, t3);
}

template <bool INV> ECHO_HD void dft8(float2 (&v)[8]) {
    const float h = 0.70710678118654752440f;
    float2 a0 = cadd(v[0], v[4]), a1 = cadd(v[1], v[5]), a2 = cadd(v[2], v[6]), a3 = cadd(v[3], v[7]);
    float2 b0 = csub(v[0], v[4]), b1 = csub(v[1], v[5]), b2 = csub(v[2], v[6]), b3 = csub(v[3], v[7]);
    float2 t;
    if (!INV) { t.x = h * (b1.x + b1.y); t.y = h * (b1.y - b1.x); b1 = t;
                b2 = mul_mi<false>(b2);
                t.x = h * (b3.y - b3.x); t.y = -h * (b3.x + b3.y); b3 = t; }
    else      { t.x = h * (b1.x - b1.y); t.y = h * (b1.y + b1.x); b1 = t;
                b2 = mul_mi<true>(b2);
                t.x = -h * (b3.x + b3.y); t.y = h * (b3.x - b3.y); b3 = t; }
    dft4<INV>(a0, a1, a2, a3);
    dft4<INV>(b0, b1, b2, b3);
    v[0] = a0; v[2] = a1; v[4] = a2; v[6] = a3;
    v[1] = b0; v[3] = b1; v[5] = b2; v[7] = b3;
}

// LDS address of FFT element i (hook for bank-conflict padding)
ECHO_HD int fa(int i) { return i; }

// Bluestein input: a[n] = (x[2n] + i x[2n+1]) * w[n], zero-padded to kFftM
ECHO_HD void ph_spec_load(int tid, int nt, const BandParams& p, unsigned char* smem) {
    if (!p.do_spec) return;
    const double* sig = lds_d(smem, p.off_sig) + (p.M - 1);
    float2* f = lds_c(smem, p.off_u);
    for (int n = tid; n < kFftM; n += nt) {
        float2 a; a.x = 0.f; a.y = 0.f;
        if (n < p.n2) {
            float2 z; z.x = (float)sig[2 * n]; z.y = (float)sig[2 * n + 1];
            a = cmul(z, p.bw[n]);
        }
        f[fa(n)] = a;
    }
}

// One radix-8 pass over sub-FFTs of size 8q.  Forward (decimation in frequency): butterfly then twiddle;
// inverse (its conjugate transpose): conjugate twiddle then butterfly.
template <bool INV> ECHO_HD void ph_fft_r8(int tid, int nt, int q, const BandParams& p, unsigned char* smem) {
    if (!p.do_spec) return;
    float2* f = lds_c(smem, p.off_u);
    const int tstep = kFftM / (8 * q);
    for (int u = tid; u < kFftM / 8; u += nt) {
        const int i = u % q, base = (u / q) * 8 * q + i;
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = f[fa(base + j * q)];
        if (INV) {
#pragma unroll
            for (int m = 1; m < 8; ++m) v[m] = cmulc(v[m], p.tw[i * m * tstep]);
        }
        dft8<INV>(v);
        if (!INV) {
#pragma unroll
            for (int m = 1; m < 8; ++m) v[m] = cmul(v[m], p.tw[i * m * tstep]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) f[fa(base + j * q)] = v[j];
    }
}

// Middle of the transform, per contiguous 16 elements, all in registers:
// forward radix-4 (q=4) and radix-4 (q=1), multiply by the chirp-filter spectrum, inverse radix-4 (q=1) and (q=4).
ECHO_HD void ph_fft_mid(int tid, int nt, const BandParams& p, unsigned char* smem) {
    if (!p.do_spec) return;
    float2* f = lds_c(smem, p.off_u);
    const int ts = kFftM / 16;
    for (int u = tid; u < kFftM / 16; u += nt) {
        const int base = u * 16;
        float2 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = f[fa(base + j)];
        // forward, q = 4: butterflies over (i, i+4, i+8, i+12), twiddle W16^(i*m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dft4<false>(v[i], v[i + 4], v[i + 8], v[i + 12]);
#pragma unroll
            for (int m = 1; m < 4; ++m)
                if (i) v[i + 4 * m] = cmul(v[i + 4 * m], p.tw[i * m * ts]);
        }
        // forward, q = 1
#pragma unroll
        for (int b = 0; b < 4; ++b) dft4<false>(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]);
        // pointwise product with the (pass-ordered, 1/M-scaled) chirp-filter spectrum
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = cmul(v[j], p.bhat[base + j]);
        // inverse, q = 1
#pragma unroll
        for (int b = 0; b < 4; ++b) dft4<true>(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]);
        // inverse, q = 4
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int m = 1; m < 4; ++m)
                if (i) v[i + 4 * m] = cmulc(v[i + 4 * m], p.tw[i * m * ts]);
            dft4<true>(v[i], v[i + 4], v[i + 8], v[i + 12]);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) f[fa(base + j)] = v[j];
    }
}

// Z[k] = c[k] * w[k]; X[k] = E + u[k] * (-i) * O with E,O = (Z[k] +- conj(Z[n2-k]))/2; magnitude (dB)
ECHO_HD void ph_spec_store(int tid, int nt, int blk, const BandParams& p, unsigned char* smem) {
    if (!p.do_spec) return;
    const float2* f = lds_c(smem, p.off_u);
    for (int k = tid; k < p.nbins; k += nt) {
        const int ka = (k == p.n2) ? 0 : k;
        const int kb = (k == 0 || k == p.n2) ? 0 : p.n2 - k;
        const float2 za = cmul(f[fa(ka)], p.bw[ka]);
        float2 zb = cmul(f[fa(kb)], p.bw[kb]);
        zb.y = -zb.y;
        float2 e, o;
        e.x = 0.5f * (za.x + zb.x); e.y = 0.5f * (za.y + zb.y);
        o.x = 0.5f * (za.x - zb.x); o.y = 0.5f * (za.y - zb.y);
        float2 mo; mo.x = o.y; mo.y = -o.x;                      // -i * O
        const float2 x = cadd(e, cmul(p.unpack[k], mo));
        float mag = sqrtf(x.x * x.x + x.y * x.y);
        if (p.spec_db) mag = 20.0f * log10f(fmaxf(mag, 1e-12f));
        p.spec_out[(size_t)blk * p.nbins + k] = mag;
    }
}

}  // namespace echo
