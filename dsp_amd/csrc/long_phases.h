// Long-frame path: the same chain as band_phases.h for frames that do not fit one workgroup's LDS, and the
// median/MAD first-crossing detector of the original parameter set.  Three kernels, each tiled over outputs:
//   FIR 'same'            y or rx  -> band_sig (float64, global)               airSonar.py:98-101
//   matched filter        band_sig -> corr[k_begin:] (float64, global)        airSonar.py:224
//   first_strong_peak     pos = corr[half:] -> first index above med + 6 MAD  airSonar.py:104-117
// Written as barrier-separated phases of (tid, nt) like band_phases.h, so a host build can run them thread by
// thread.  Plain C++ apart from the two LDS atomics.
#pragma once
#include "band_phases.h"
#include "gfft.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define ECHO_LDS_ADD(p, v) atomicAdd((p), (v))
#define ECHO_LDS_MIN(p, v) atomicMin((p), (v))
// hist[digit] += 1 for every active lane of the wavefront (all lanes must call).  The digits of one pass cluster
// (sign and exponent bits are nearly the same for every element) and LDS atomics on one address are served lane by
// lane, so digits shared by several lanes are combined into one atomic each; the peeling stops as soon as a digit
// turns out to be shared by fewer than four lanes (spread digits go through plain atomics, which do not collide).
__device__ __forceinline__ void wave_hist_add(int* hist, int digit, bool active) {
    unsigned long long todo = __ballot(active);
    const int lane = (int)__lane_id();
    for (int it = 0; it < 8 && todo; ++it) {
        const int leader = __ffsll((long long)todo) - 1;
        const int d0 = __builtin_amdgcn_readlane(digit, leader);
        const unsigned long long same = __ballot(digit == d0) & todo;
        const int n = (int)__popcll(same);
        if (lane == leader) atomicAdd(&hist[d0], n);
        todo &= ~same;
        if (n < 4) break;
    }
    if ((todo >> lane) & 1) atomicAdd(&hist[digit], 1);
}
#define ECHO_HIST_ADD(hist, digit, active) wave_hist_add((hist), (digit), (active))
// *counter += number of lanes of the wavefront with flag set (all lanes must call): one atomic per wavefront
#define ECHO_WAVE_COUNT_ADD(counter, flag)                                                    \
    do { const unsigned long long b_ = __ballot(flag);                                        \
         if (b_ && __lane_id() == 0) atomicAdd((counter), (int)__popcll(b_)); } while (0)
#define ECHO_LDS_FETCH_ADD(p, v) atomicAdd((p), (v))
// maximum of v over the wavefront, in every lane: DPP exchanges inside rows of 16 lanes, two shuffles across rows
__device__ __forceinline__ int wave_max_int(int v) {
    int t;
    t = __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true); v = t > v ? t : v;    // quad_perm [1,0,3,2]
    t = __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true); v = t > v ? t : v;    // quad_perm [2,3,0,1]
    t = __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true); v = t > v ? t : v;   // row_half_mirror
    t = __builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, true); v = t > v ? t : v;   // row_mirror
    t = __shfl_xor(v, 16); v = t > v ? t : v;
    t = __shfl_xor(v, 32); v = t > v ? t : v;
    return v;
}
#else
#define ECHO_LDS_FETCH_ADD(p, v) ((*(p) += (v)) - (v))
#define ECHO_LDS_ADD(p, v) (*(p) += (v))
#define ECHO_LDS_MIN(p, v) (*(p) = (v) < *(p) ? (v) : *(p))
#define ECHO_HIST_ADD(hist, digit, active) do { if (active) (hist)[(digit)] += 1; } while (0)
#define ECHO_WAVE_COUNT_ADD(counter, flag) do { if (flag) *(counter) += 1; } while (0)
#endif

namespace echo {

constexpr int kLFirNT = 256;                   // threads per FIR workgroup: kLFirNT*kR outputs per tile
constexpr int kLCorrNT = 768;                  // threads per correlation workgroup: kLCorrNT*kR outputs per tile
constexpr int kLDetNT = 512;                   // threads per detector workgroup (1024-thread workgroups ran one per CU)
constexpr int kLSpecNT = 512;                  // threads per spectrum workgroup
constexpr int kLPad = 64 * kR + 2 * kR;        // zero margin either side of the LDS signal: one wavefront's spread of
                                               // first outputs (64*kR) plus the window prefetch (2*kR)

constexpr int kRcMaxPass = 6;                  // passes of the real-transform matched filter (rcf_phases.h)

struct LongParams {
    int N, M, L, half, nb, K, min_delay, detector;
    int kgroups, mgroups, ypad;
    const double* taps_rev;          // [nb][(kgroups+1)*kR]
    const double* chirp;             // [nb][(mgroups+1)*kR]
    // FIR
    int fir_rx;                      // 1: input is rx float32 [B][N]; 0: input is float64 [B][nb][N]
    const float* rx_in;
    const double* d_in;
    double* band_out;                // [B][nb][N]
    float* bandf_out;                // optional float32 copy (spectrum input)
    // matched filter: corr[k], k_begin <= k < L, stored at corr_out[blk*corr_stride + k - corr_k0]
    const double* band_in;           // [B][nb][N]
    double* corr_out;
    int k_begin, corr_k0;
    long long corr_stride;
    // matched filter by FFT (cfft_phases.h), chosen for long chirps
    int cf_use, cf_V;                // cf_V: lags per sub-block = 8192 - M + 1
    const cpx<double>* cf_g;         // [nb][8192] conj(DFT(chirp)) / 8192 in pass order
    const cpx<double>* cf_w2;        // [4096] exp(-2 pi i n / 8192)
    const cpx<double>* cf_tw;        // pass twiddles
    // detector: pos[q] = det_io[blk*det_stride + det_off + q], 0 <= q < L - half; q < min_delay reads as zero
    double* det_io;
    long long det_stride;
    int det_off, write_mutated;
    // where the detector finds pos in LDS.  det_view 0: double pos[npos] at the start.  det_view 1: the matched
    // filter's transform, finished in place (cfft_phases.h), holds it: pos[q] for q >= det_q0 is component w of complex
    // slot i + i/8, with q - det_q0 = w * cf_V + i; below det_q0 lies the blind zone (zero).  det_lds_off: byte offset
    // of the histograms and scalars behind pos.
    // det_view 2: the single real transform of rcf_phases.h holds it: q - det_q0 = 2 i + w, component w of slot i + i/8.
    int det_view, det_q0, det_lds_off;
    // matched filter as real cyclic correlations (rcf_phases.h): block b of rc_nblk covers the rc_V lags from
    // rc_k0 + b * rc_V on (one block where the detector runs in the same workgroup)
    int rc_use, rc_H, rc_npass, rc_k0, rc_smem, rc_V, rc_nblk;
    int rc_radix[kRcMaxPass], rc_root_off[kRcMaxPass], rc_tw_off[kRcMaxPass];
    const cpx<double>* rc_wh;        // pass twiddles: per pass the runs exp(-2 pi i (i m) / S), m = 1..r-1, i < S / r
    const cpx<double>* rc_wk;        // [H/2 + 1] exp(-2 pi i k / 2H)
    const cpx<double>* rc_roots;     // per pass the r roots of unity
    const cpx<double>* rc_g;         // [nb][H + 1] conj(DFT_2H(chirp)) / H
    const int* rc_pos;               // [H] position of frequency k after the forward passes
    BandRes* res_out;                // [B][nb]
    // spectrum: rfft of the N-sample band signal as a packed complex FFT of length N/2 (gfft.h), float32
    int has_spec, n2, nbins, spec_db;
    GfftPlan splan;
    const cpx<float>* sroots;        // global: splan.nroots
    const cpx<float>* swn;           // global: n2 twiddles
    const cpx<float>* suk;           // global: n2 + 1, exp(-2*pi*i*k/N)
    const double* spec_in;           // [B][nb][N]
    float* spec_out;                 // [B][nb][nbins]
    // odd N: no half-length packing exists; the N-point transform is a chirp-z (Bluestein) convolution of length sb_m
    int spec_odd, sb_m;              // sb_m >= N + nbins - 1, a product of 2, 3, 5, 7 (splan is then the plan of sb_m)
    const cpx<float>* sbw;           // global: N chirp factors w[n] = exp(-i pi n^2 / N)
    const cpx<float>* sbhat;         // global: sb_m, DFT of the chirp filter conj(w) / sb_m
    // odd N beyond two LDS buffers: the same convolution with the transform in place (ospec_phases.h)
    int os_use, os_m, os_npass, os_smem;
    int os_radix[kRcMaxPass], os_root_off[kRcMaxPass], os_tw_off[kRcMaxPass];
    const cpx<float>* os_tw;         // pass twiddles
    const cpx<float>* os_roots;      // per pass the r roots of unity
    const cpx<float>* os_bhat;       // [os_m] DFT of the chirp filter in pass order, / os_m
    int det_stages, det_passes;      // development aids: run only the first stages / passes of the detector (0 = all)
    int noise_n;                     // two-peak detector: length of the noise region of pos
    double snr_gate;
    // LDS sizes
    int fir_smem, corr_smem, det_smem, spec_smem, tp_smem, lm_smem;
};

struct DetMisc {                     // detector scalars in LDS
    unsigned long long prefix;
    double med, mad, lo, hi;
    int rank, shift, first, cnt;     // cnt: elements left in the chosen bin
    int nzero, pad;                  // nzero: elements equal to +0.0 (counted in the first pass)
    // the shortcut (ph_fs_*): statistics of pos beyond the blind zone, and the state of the selection by one histogram
    int amax_hi, pad1;               // upper 32 bits of the double max |pos|
    int n_neg, n_zero, n_nan;        // negative, zero and NaN elements
    int fast;                        // 1: the shortcut's assumptions hold so far; 2: pos is all zero; 0: take the general path
    int bin, below, nlist, base;     // chosen bin, elements in lower bins, collected elements, (key >> kFsShift) of bin 0
    int pad2[2];
};
constexpr int kFsBins = 2048, kFsShift = 45, kFsList = 64;   // bins of 1/128 octave: 16 octaves below the maximum
constexpr int kDetExtra = 2 * 256 * 4 + 112 + kFsBins * 4 + 32 * 4 + kFsList * 8;   // detector LDS behind pos
static_assert(sizeof(DetMisc) <= 112, "DetMisc outgrew its LDS slot");

inline bool plan_long(LongParams& p, int N, int M, int K, int nb, int min_delay, int detector) {
    p.N = N; p.M = M; p.L = N + M - 1; p.half = p.L / 2; p.nb = nb; p.K = K;
    p.min_delay = min_delay; p.detector = detector;
    p.kgroups = (K + kR - 1) / kR; p.mgroups = (M + kR - 1) / kR;
    p.ypad = K - 1 - (K - 1) / 2;
    if (N < 4 || M < 1 || K < 1) return false;
    p.fir_smem = (kLFirNT * kR + (p.kgroups + 1) * kR) * 8 + (p.kgroups + 1) * kR * 8;
    p.corr_smem = (N + 2 * kLPad) * 8 + (p.mgroups + 1) * kR * 8;
    // the FFT form pays once the chirp is long: ~0.6 M multiply-adds per pair of sub-blocks against M per lag
    p.cf_use = (M >= 512 && M <= 6144) ? 1 : 0;
    p.cf_V = 8192 - M + 1;
    const int npos = p.L - p.half;
    p.det_view = 0; p.det_q0 = 0; p.det_lds_off = ((npos + 1) & ~1) * 8;
    p.det_smem = p.det_lds_off + kDetExtra;
    p.tp_smem = ((npos + 1) & ~1) * 8 + 3 * kLDetNT * 8 + 64;
    p.lm_smem = ((npos + 1) & ~1) * 8 + 3 * kLDetNT * 8 + kLDetNT * 4 + 64;
    p.n2 = N / 2; p.nbins = N / 2 + 1;
    p.spec_odd = N % 2; p.sb_m = 0;
    if (!p.spec_odd) {
        p.has_spec = gfft_plan(p.splan, p.n2, kLSpecNT, true) ? 1 : 0;   // small factors as in-register butterflies
        p.spec_smem = 2 * p.n2 * 8 + (p.has_spec ? p.splan.nroots * 8 : 0);
    } else {
        // outputs 0 .. nbins-1 from inputs 0 .. N-1: the chirp filter is needed at -(N-1) .. nbins-1
        p.has_spec = 0;
        for (int m = N + p.nbins - 1; m <= 2 * N + 64 && !p.has_spec; ++m) {
            int r = m;
            for (int f = 2; f <= 7; ++f) while (r % f == 0) r /= f;
            if (r == 1 && gfft_plan(p.splan, m, kLSpecNT, true)) { p.has_spec = 1; p.sb_m = m; }
        }
        p.spec_smem = 2 * p.sb_m * 8 + (p.has_spec ? p.splan.nroots * 8 : 0);
    }
    if (p.spec_smem > 160 * 1024) p.has_spec = 0;
    return p.fir_smem <= 160 * 1024 && p.corr_smem <= 160 * 1024 && p.det_smem <= 160 * 1024 &&
           p.lm_smem <= 160 * 1024;
}

// ---------------------------------------------------------------- spectrum of a long frame
// mag2db(np.fft.rfft(band_sig)) (airSonar.py:141-142): Z = FFT_n2(x[2m] + i x[2m+1]);
// X[k] = E + u_k O with E = (Z[k] + conj Z[n2-k]) / 2, O = -i (Z[k] - conj Z[n2-k]) / 2, u_k = exp(-2 pi i k / N).
// LDS: two buffers of n2 complex floats, then the roots.
ECHO_HD void ph_lspec_load(int tid, int nt, int blk, const LongParams& p, unsigned char* smem) {
    cpx<float>* z = reinterpret_cast<cpx<float>*>(smem);
    const double* x = p.spec_in + (size_t)blk * p.N;
    for (int m = tid; m < p.n2; m += nt) z[m] = cpx<float>{(float)x[2 * m], (float)x[2 * m + 1]};
    cpx<float>* r = z + 2 * p.n2;
    for (int i = tid; i < p.splan.nroots; i += nt) r[i] = p.sroots[i];
}
ECHO_HD void ph_lspec_pass(int tid, int nt, const LongParams& p, unsigned char* smem, int ps, int src) {
    cpx<float>* b0 = reinterpret_cast<cpx<float>*>(smem);
    const cpx<float>* r = b0 + 2 * p.n2 + p.splan.root_off[ps];
    gfft_pass<float, true>(tid, nt, p.splan, ps, b0 + (src ? p.n2 : 0), b0 + (src ? 0 : p.n2), r, p.swn, false);
}
ECHO_HD void ph_lspec_store(int tid, int nt, int blk, const LongParams& p, unsigned char* smem, int which) {
    const cpx<float>* Z = reinterpret_cast<const cpx<float>*>(smem) + (which ? p.n2 : 0);
    for (int k = tid; k < p.nbins; k += nt) {
        const int ka = k == p.n2 ? 0 : k, kb = (k == 0 || k == p.n2) ? 0 : p.n2 - k;
        const cpx<float> zk = Z[ka], zc = cpx<float>{Z[kb].x, -Z[kb].y};
        const cpx<float> u = p.suk[k];
        const float ex = 0.5f * (zk.x + zc.x), ey = 0.5f * (zk.y + zc.y);
        const float bx = 0.5f * (zk.x - zc.x), by = 0.5f * (zk.y - zc.y);
        const float ox = by, oy = -bx;                                     // -i (Z - conj Z') / 2
        const float xr = ex + (u.x * ox - u.y * oy), xi = ey + (u.x * oy + u.y * ox);
        float mag = sqrtf(xr * xr + xi * xi);
        if (p.spec_db) mag = 20.0f * log10f(fmaxf(mag, 1e-12f));
        p.spec_out[(size_t)blk * p.nbins + k] = mag;
    }
}

// Odd N (np.fft.rfft has no length restriction, airSonar_optimized.py:710): X[k] = w[k] sum_n (x[n] w[n]) conj(w)[k-n]
// with w[n] = exp(-i pi n^2 / N), the sum as a cyclic convolution of length sb_m through the same mixed-radix FFT.
// LDS: two buffers of sb_m complex floats, then the roots.
ECHO_HD void ph_lspec_odd_load(int tid, int nt, int blk, const LongParams& p, unsigned char* smem) {
    cpx<float>* z = reinterpret_cast<cpx<float>*>(smem);
    const double* x = p.spec_in + (size_t)blk * p.N;
    for (int n = tid; n < p.sb_m; n += nt) {
        cpx<float> a{0.f, 0.f};
        if (n < p.N) { const float v = (float)x[n]; const cpx<float> w = p.sbw[n]; a.x = v * w.x; a.y = v * w.y; }
        z[n] = a;
    }
    cpx<float>* r = z + 2 * p.sb_m;
    for (int i = tid; i < p.splan.nroots; i += nt) r[i] = p.sroots[i];
}
ECHO_HD void ph_lspec_odd_pass(int tid, int nt, const LongParams& p, unsigned char* smem, int ps, int src, bool inv) {
    cpx<float>* b0 = reinterpret_cast<cpx<float>*>(smem);
    const cpx<float>* r = b0 + 2 * p.sb_m + p.splan.root_off[ps];
    gfft_pass<float, true>(tid, nt, p.splan, ps, b0 + (src ? p.sb_m : 0), b0 + (src ? 0 : p.sb_m), r, p.swn, inv);
}
ECHO_HD void ph_lspec_odd_mul(int tid, int nt, const LongParams& p, unsigned char* smem, int which) {
    cpx<float>* z = reinterpret_cast<cpx<float>*>(smem) + (which ? p.sb_m : 0);
    for (int k = tid; k < p.sb_m; k += nt) {
        const cpx<float> a = z[k], b = p.sbhat[k];
        z[k] = cpx<float>{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
    }
}
ECHO_HD void ph_lspec_odd_store(int tid, int nt, int blk, const LongParams& p, unsigned char* smem, int which) {
    const cpx<float>* y = reinterpret_cast<const cpx<float>*>(smem) + (which ? p.sb_m : 0);
    for (int k = tid; k < p.nbins; k += nt) {
        const cpx<float> a = y[k], w = p.sbw[k];
        const float xr = a.x * w.x - a.y * w.y, xi = a.x * w.y + a.y * w.x;
        float mag = sqrtf(xr * xr + xi * xi);
        if (p.spec_db) mag = 20.0f * log10f(fmaxf(mag, 1e-12f));
        p.spec_out[(size_t)blk * p.nbins + k] = mag;
    }
}
#define ECHO_LSPEC_ODD_SEQUENCE(PHASE, SYNC, p, smem, blk, nt)                                   \
    do {                                                                                          \
        PHASE(ph_lspec_odd_load(tid, nt, blk, p, smem)) SYNC                                      \
        int cur_ = 0;                                                                             \
        for (int ps_ = 0; ps_ < (p).splan.npass; ++ps_) {                                         \
            PHASE(ph_lspec_odd_pass(tid, nt, p, smem, ps_, cur_, false)) SYNC                     \
            cur_ ^= 1;                                                                            \
        }                                                                                         \
        PHASE(ph_lspec_odd_mul(tid, nt, p, smem, cur_)) SYNC                                      \
        for (int ps_ = 0; ps_ < (p).splan.npass; ++ps_) {                                         \
            PHASE(ph_lspec_odd_pass(tid, nt, p, smem, ps_, cur_, true)) SYNC                      \
            cur_ ^= 1;                                                                            \
        }                                                                                         \
        PHASE(ph_lspec_odd_store(tid, nt, blk, p, smem, cur_))                                    \
    } while (0)

// Host: the tables of the long-frame spectrum.  Even N: roots, wn (length n2) and uk; odd N: roots, wn (length
// sb_m), the chirp factors bw and the filter spectrum bhat (a direct O(sb_m^2) transform in double, once per handle).
inline void long_spec_tables(const LongParams& p, std::vector<cpx<float>>& roots, std::vector<cpx<float>>& wn,
                             std::vector<cpx<float>>& uk, std::vector<cpx<float>>& bw, std::vector<cpx<float>>& bhat) {
    const double pi = 3.14159265358979323846;
    gfft_tables(p.splan, roots, wn);
    uk.assign((size_t)p.n2 + 1, cpx<float>{0.f, 0.f});
    for (int k = 0; k <= p.n2; ++k) {
        const double a = -2.0 * pi * (double)k / (double)p.N;
        uk[k] = cpx<float>{(float)cos(a), (float)sin(a)};
    }
    bw.clear(); bhat.clear();
    if (!p.spec_odd) return;
    const int N = p.N, m = p.sb_m;
    std::vector<double> wr(N), wi(N);
    bw.resize(N);
    for (int n = 0; n < N; ++n) {
        const long long r = ((long long)n * n) % (2LL * N);     // n^2 mod 2N keeps the phase exact
        const double a = -pi * (double)r / (double)N;
        wr[n] = cos(a); wi[n] = sin(a);
        bw[n] = cpx<float>{(float)wr[n], (float)wi[n]};
    }
    std::vector<double> br(m, 0.0), bi(m, 0.0), cr(m), ci(m);
    for (int j = 0; j < p.nbins; ++j) { br[j] = wr[j]; bi[j] = -wi[j]; }
    for (int j = 1; j < N; ++j) { br[m - j] = wr[j]; bi[m - j] = -wi[j]; }
    for (int e = 0; e < m; ++e) { const double a = -2.0 * pi * (double)e / (double)m; cr[e] = cos(a); ci[e] = sin(a); }
    bhat.resize(m);
    for (int k = 0; k < m; ++k) {
        double sr = 0.0, si = 0.0;
        long long idx = 0;
        for (int j = 0; j < m; ++j) {
            sr += br[j] * cr[idx] - bi[j] * ci[idx];
            si += br[j] * ci[idx] + bi[j] * cr[idx];
            idx += k; if (idx >= m) idx -= m;
        }
        bhat[k] = cpx<float>{(float)(sr / m), (float)(si / m)};
    }
}

// ---------------------------------------------------------------- FIR 'same', one tile of kLFirNT*kR outputs
// LDS: double in[nt*kR + (kgroups+1)*kR] holding input[t0 - ypad + i], then the reversed taps.
ECHO_HD void ph_lfir_load(int tid, int nt, int tile, int blk, const LongParams& p, unsigned char* smem) {
    double* in = reinterpret_cast<double*>(smem);
    const int len = nt * kR + (p.kgroups + 1) * kR;
    double* ct = in + len;
    const int band = blk % p.nb, frame = blk / p.nb;
    const int t0 = tile * nt * kR;
    for (int i = tid; i < len; i += nt) {
        const int n = t0 + i - p.ypad;
        double v = 0.0;
        if (n >= 0 && n < p.N) v = p.fir_rx ? (double)p.rx_in[(size_t)frame * p.N + n] : p.d_in[(size_t)blk * p.N + n];
        in[i] = v;
    }
    const int nc = (p.kgroups + 1) * kR;
    for (int i = tid; i < nc; i += nt) ct[i] = p.taps_rev[(size_t)band * nc + i];
}

// The outputs go back through LDS (the input window is dead once every lane has its sums) so that consecutive lanes
// store consecutive samples: a lane's own nine outputs are 72 bytes apart from its neighbour's.
ECHO_HD void ph_lfir_run(int tid, int nt, int tile, int blk, const LongParams& p, unsigned char* smem, double (&acc)[kR]) {
    const double* in = reinterpret_cast<const double*>(smem);
    const double* ct = in + nt * kR + (p.kgroups + 1) * kR;
    const int n0 = tile * nt * kR + tid * kR;
    if (n0 < p.N) sliding_dot(in + tid * kR, ct, p.kgroups, acc);
    else {
#pragma unroll
        for (int r = 0; r < kR; ++r) acc[r] = 0.0;
    }
}
ECHO_HD void ph_lfir_stage(int tid, int nt, unsigned char* smem, const double (&acc)[kR]) {
    double* o = reinterpret_cast<double*>(smem);
#pragma unroll
    for (int r = 0; r < kR; ++r) o[tid * kR + r] = acc[r];
}
ECHO_HD void ph_lfir_store(int tid, int nt, int tile, int blk, const LongParams& p, unsigned char* smem) {
    const double* o = reinterpret_cast<const double*>(smem);
    const int t0 = tile * nt * kR;
    for (int i = tid; i < nt * kR; i += nt) {
        const int n = t0 + i;
        if (n >= p.N) break;
        const double v = o[i];
        if (p.band_out) p.band_out[(size_t)blk * p.N + n] = v;
        if (p.bandf_out) p.bandf_out[(size_t)blk * p.N + n] = (float)v;
    }
}

// ---------------------------------------------------------------- matched filter, one tile of kLCorrNT*kR lags
// corr[k] = sum_j chirp[j] * band_sig[k - (M-1) + j].  LDS: double sig[kLPad + N + kLPad] (zero margins), then the
// chirp.  Each wavefront trims the tap range to the taps that overlap the frame for at least one of its lanes;
// the margins absorb the lanes whose own overlap is shorter.
ECHO_HD void ph_lcorr_load(int tid, int nt, int blk, const LongParams& p, unsigned char* smem) {
    double* sig = reinterpret_cast<double*>(smem);
    const int len = p.N + 2 * kLPad;
    double* cc = sig + len;
    const int band = blk % p.nb;
    for (int i = tid; i < len; i += nt) {
        const int n = i - kLPad;
        sig[i] = (n >= 0 && n < p.N) ? p.band_in[(size_t)blk * p.N + n] : 0.0;
    }
    const int nc = (p.mgroups + 1) * kR;
    for (int i = tid; i < nc; i += nt) cc[i] = p.chirp[(size_t)band * nc + i];
}

ECHO_HD void ph_lcorr_run(int tid, int nt, int tile, int blk, const LongParams& p, unsigned char* smem) {
    const double* sig = reinterpret_cast<const double*>(smem);
    const double* cc = sig + p.N + 2 * kLPad;
    const int kt0 = p.k_begin + tile * nt * kR;
    const int kw0 = kt0 + (tid / 64) * 64 * kR;          // first lag of this wavefront
    const int k0 = kt0 + tid * kR;
    if (kw0 >= p.L) return;
    int jlo = (p.M - 1) - (kw0 + 64 * kR - 1);
    jlo = jlo < 0 ? 0 : (jlo / kR) * kR;
    int jhi = p.N + (p.M - 1) - kw0;
    jhi = jhi > p.M ? p.M : jhi;
    const int ng = (jhi - jlo + kR - 1) / kR;
    double acc[kR];
    sliding_dot(sig + kLPad + k0 - (p.M - 1) + jlo, cc + jlo, ng, acc);
#pragma unroll
    for (int r = 0; r < kR; ++r) {
        const int k = k0 + r;
        if (k < p.L) p.corr_out[(size_t)blk * p.corr_stride + (k - p.corr_k0)] = acc[r];
    }
}

// ---------------------------------------------------------------- median/MAD first-crossing detector
// Order statistics by most-significant-digit radix selection on the order-preserving integer image of a double:
// eight passes of 8 bits, each a 256-bin LDS histogram of the elements that match the digits fixed so far.
constexpr unsigned long long kKeyPlusZero = 0x8000000000000000ull;      // ordered_key(+0.0)
ECHO_HD unsigned long long ordered_key(double v) {
    unsigned long long u;
    __builtin_memcpy(&u, &v, 8);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
ECHO_HD double key_value(unsigned long long k) {
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double v;
    __builtin_memcpy(&v, &u, 8);
    return v;
}

ECHO_HD double* det_pos(unsigned char* smem) { return reinterpret_cast<double*>(smem); }
ECHO_HD int* det_hist(const LongParams& p, unsigned char* smem) {
    return reinterpret_cast<int*>(smem + p.det_lds_off);
}
// pos[q] of this workgroup's (frame, band)
ECHO_HD double det_read(const LongParams& p, const unsigned char* smem, int q) {
    const double* d = reinterpret_cast<const double*>(smem);
    if (!p.det_view) return d[q];
    const int j = q - p.det_q0;
    if (j < 0) return 0.0;
    if (p.det_view == 2) { const int i = j >> 1; return d[2 * (i + (i >> 3)) + (j & 1)]; }
    const int w = j >= p.cf_V ? 1 : 0, i = j - (w ? p.cf_V : 0);
    return d[2 * (i + (i >> 3)) + w];
}
ECHO_HD DetMisc* det_misc(const LongParams& p, unsigned char* smem) {
    return reinterpret_cast<DetMisc*>(reinterpret_cast<unsigned char*>(det_hist(p, smem)) + 2 * 256 * 4);
}

ECHO_HD int* det_fhist(const LongParams& p, unsigned char* smem) {
    return reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(det_hist(p, smem)) + 2 * 256 * 4 + 112);
}
ECHO_HD int* det_wtot(const LongParams& p, unsigned char* smem) { return det_fhist(p, smem) + kFsBins; }
ECHO_HD double* det_list(const LongParams& p, unsigned char* smem) { return reinterpret_cast<double*>(det_wtot(p, smem) + 32); }
ECHO_HD void det_misc_init(DetMisc* m) {
    m->first = 0x7fffffff; m->med = 0.0; m->mad = 0.0;
    m->amax_hi = 0; m->n_neg = 0; m->n_zero = 0; m->n_nan = 0; m->fast = 0; m->nlist = 0;
}

ECHO_HD void ph_ldet_load(int tid, int nt, int blk, const LongParams& p, unsigned char* smem) {
    double* pos = det_pos(smem);
    const int npos = p.L - p.half;
    double* src = p.det_io + (size_t)blk * p.det_stride + p.det_off;
    for (int q = tid; q < npos; q += nt) {
        double v = 0.0;
        if (q >= p.min_delay) v = src[q];
        else if (p.write_mutated) src[q] = 0.0;
        pos[q] = v;
    }
    if (tid == 0) det_misc_init(det_misc(p, smem));
}

// Selection of the element of rank `rank` among key(i) = pos[i] (dev == 0) or |pos[i] - med| (dev == 1): per pass a
// 256-bin histogram of the next 8 key bits over the elements that match the bits fixed so far (two histograms in
// turn: the idle one is cleared while the other is read), then the bin that holds the rank.  As soon as that bin
// holds a single element the remaining bits are read off the element itself (ph_sel_find) -- for float64 data that
// is typically after four of the eight passes.
ECHO_HD double sel_value(const LongParams& p, const unsigned char* smem, int i, double med, int dev) {
    const double v = det_read(p, smem, i);
    return dev ? fabs(v - med) : v;
}
ECHO_HD void ph_sel_begin(int tid, const LongParams& p, unsigned char* smem, int rank) {
    if (tid < 512) det_hist(p, smem)[tid] = 0;
    if (tid != 0) return;
    DetMisc* m = det_misc(p, smem);
    m->prefix = 0; m->rank = rank; m->shift = 56; m->cnt = 0; m->nzero = 0;
}
ECHO_HD void ph_sel_count(int tid, int nt, const LongParams& p, unsigned char* smem, int dev, int cur) {
    DetMisc* m = det_misc(p, smem);
    int* hist = det_hist(p, smem) + 256 * cur;
    const int npos = p.L - p.half, shift = m->shift;
    const unsigned long long prefix = m->prefix;
    const double med = m->med;
    for (int i0 = 0; i0 < npos; i0 += nt) {                 // the same trip count for every lane of a wavefront
        const int i = i0 + tid;
        bool match = false, zero = false;
        int digit = 0;
        if (i < npos) {
            const unsigned long long k = ordered_key(sel_value(p, smem, i, med, dev));
            match = shift == 56 || (k >> (shift + 8)) == (prefix >> (shift + 8));
            digit = (int)((k >> shift) & 255);
            zero = k == kKeyPlusZero;
        }
        ECHO_HIST_ADD(hist, digit, match);
        if (shift == 56) ECHO_WAVE_COUNT_ADD(&m->nzero, zero);
    }
}
// The blind zone puts min_delay exact zeros into pos, so the median of pos is very often 0.0: when the bin chosen in
// the first pass holds nothing but +0.0 the selection is finished.
ECHO_HD void sel_zero_run(DetMisc* m, int shift_before, int d, int cnt) {
    if (shift_before == 56 && d == 0x80 && cnt == m->nzero && cnt > 0) { m->prefix = kKeyPlusZero; m->shift = -8; m->cnt = 1; }
}

// The bin that holds the rank.  Device: the first wavefront takes four bins per lane, a shuffle scan gives every
// lane the count below its bins, and the one lane whose bins straddle the rank finishes; lanes 256..511 clear the
// other histogram meanwhile.  Host: the same result by a serial scan.
ECHO_HD void ph_sel_pick(int tid, const LongParams& p, unsigned char* smem, int cur) {
    DetMisc* m = det_misc(p, smem);
    int* hist_all = det_hist(p, smem);
#if defined(__HIP_DEVICE_COMPILE__)
    if (tid >= 256 && tid < 512) hist_all[256 * (cur ^ 1) + (tid - 256)] = 0;
    if (tid >= 64) return;
    const int* hist = hist_all + 256 * cur;
    const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
    const int rank = m->rank, shift = m->shift;
    const int sum = h0 + h1 + h2 + h3;
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (tid >= off) inc += t;
    }
    const int exc = inc - sum;
    if (exc <= rank && rank < inc) {                       // one lane: the rank is below the number of counted elements
        int cum = exc, d = 4 * tid, cnt = h0;
        if (cum + h0 <= rank) { cum += h0; ++d; cnt = h1;
            if (cum + h1 <= rank) { cum += h1; ++d; cnt = h2;
                if (cum + h2 <= rank) { cum += h2; ++d; cnt = h3; } } }
        m->cnt = cnt;
        m->rank = rank - cum;
        m->prefix |= (unsigned long long)d << shift;
        m->shift = shift - 8;
        sel_zero_run(m, shift, d, cnt);
    }
#else
    if (tid != 0) return;
    const int* hist = hist_all + 256 * cur;
    int cum = 0, d = 0;
    for (; d < 255; ++d) {
        if (cum + hist[d] > m->rank) break;
        cum += hist[d];
    }
    m->cnt = hist[d];
    m->rank -= cum;
    m->prefix |= (unsigned long long)d << m->shift;
    m->shift -= 8;
    sel_zero_run(m, m->shift + 8, d, m->cnt);
    for (int i = 0; i < 256; ++i) hist_all[256 * (cur ^ 1) + i] = 0;   // the histogram of the next pass
#endif
}
// the chosen bin holds one element: it is the wanted one, take all of its bits
ECHO_HD void ph_sel_find(int tid, int nt, const LongParams& p, unsigned char* smem, int dev) {
    DetMisc* m = det_misc(p, smem);
    const int npos = p.L - p.half, shift = m->shift;        // bits above shift + 8 are fixed
    if (shift < 0) return;
    const unsigned long long prefix = m->prefix;
    const double med = m->med;
    for (int i = tid; i < npos; i += nt) {
        const unsigned long long k = ordered_key(sel_value(p, smem, i, med, dev));
        if ((k >> (shift + 8)) == (prefix >> (shift + 8))) m->prefix = k;      // exactly one element matches
    }
}
// after the eighth pass m->prefix is the key of the wanted element
ECHO_HD void ph_sel_store(int tid, const LongParams& p, unsigned char* smem, int which) {
    if (tid != 0) return;
    DetMisc* m = det_misc(p, smem);
    const double v = key_value(m->prefix);
    if (which == 0) m->lo = v; else m->hi = v;
}

// ---- the shortcut.  The blind zone puts min_delay exact zeros into pos and the correlation spreads evenly around
// zero, so the median of pos is 0.0 whenever the middle ranks fall into the zeros: counting the negative and the zero
// elements settles that without a selection.  The MAD is then the median of |pos|, and all but a few elements lie
// within 16 octaves below max|pos|: ONE histogram over bins of 1/128 octave (the top 19 bits of the double, offset so
// that max|pos| lands in the last bin; smaller elements go to bin 0) locates the bin of the wanted rank, the handful
// of elements in that bin is collected, and the rank inside the bin is read off by counting.  Exact like the general
// selection (it picks the same element); whenever an assumption fails -- NaN, median not in the zeros, the rank in
// bin 0, a crowded bin -- m->fast drops to 0 and the general path runs from the start.
ECHO_HD unsigned long long fs_bits(double a) { unsigned long long u; __builtin_memcpy(&u, &a, 8); return u; }
ECHO_HD int fs_bin(unsigned long long bits, int base) {
    const int b = (int)(bits >> kFsShift) - base;
    return b < 0 ? 0 : b;
}
ECHO_HD void ph_fs_stats(int tid, int nt, const LongParams& p, unsigned char* smem) {
    DetMisc* m = det_misc(p, smem);
    int* fh = det_fhist(p, smem);
    for (int i = tid; i < kFsBins; i += nt) fh[i] = 0;
    const int npos = p.L - p.half;
    int nneg = 0, nzero = 0, nnan = 0, amax = 0;             // amax: the upper word of the largest |pos|
#if defined(__HIP_DEVICE_COMPILE__)
    // counts as wavefront-wide population counts (scalar registers, nothing to reduce at the end); the trip count is
    // the same for every lane of a wavefront
    for (int q0 = p.min_delay + (tid & ~63); q0 < npos; q0 += 4 * nt) {   // four independent reads at a time
        double v[4];
        bool in[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int q = q0 + e * nt + (tid & 63);
            in[e] = q < npos; v[e] = in[e] ? det_read(p, smem, q) : 1.0;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            nneg += (int)__popcll(__ballot(v[e] < 0.0));
            nzero += (int)__popcll(__ballot(v[e] == 0.0));
            nnan += (int)__popcll(__ballot(v[e] != v[e]));
            const int hi = in[e] ? (int)(fs_bits(fabs(v[e])) >> 32) : 0;
            amax = hi > amax ? hi : amax;
        }
    }
    amax = wave_max_int(amax);
    if ((tid & 63) == 0) {
        if (nneg) atomicAdd(&m->n_neg, nneg);
        if (nzero) atomicAdd(&m->n_zero, nzero);
        if (nnan) atomicAdd(&m->n_nan, nnan);
        if (amax) atomicMax(&m->amax_hi, amax);
    }
#else
    for (int q = p.min_delay + tid; q < npos; q += nt) {
        const double v = det_read(p, smem, q);
        nneg += v < 0.0 ? 1 : 0; nzero += v == 0.0 ? 1 : 0; nnan += v != v ? 1 : 0;
        const int hi = (int)(fs_bits(fabs(v)) >> 32);
        amax = hi > amax ? hi : amax;
    }
    m->n_neg += nneg; m->n_zero += nzero; m->n_nan += nnan;
    if (amax > m->amax_hi) m->amax_hi = amax;
#endif
}
ECHO_HD void ph_fs_decide(int tid, const LongParams& p, unsigned char* smem, int two) {
    if (tid != 0) return;
    DetMisc* m = det_misc(p, smem);
    const int npos = p.L - p.half;
    const int blind = p.min_delay < npos ? p.min_delay : npos;
    const int zeros = m->n_zero + blind;
    const int r0 = two ? npos / 2 - 1 : npos / 2, r1 = npos / 2;
    const bool med_zero = m->n_nan == 0 && r0 >= m->n_neg && r1 < m->n_neg + zeros;
    if (!med_zero) { m->fast = 0; return; }
    m->med = 0.0;
    if (zeros == npos) { m->fast = 2; m->lo = 0.0; m->hi = 0.0; return; }   // silence: |pos| is all zero
    if (m->amax_hi == 0) { m->fast = 0; return; }                          // nothing but denormals: the general path
    m->fast = 1;
    m->base = (m->amax_hi >> (kFsShift - 32)) - (kFsBins - 1);
    det_fhist(p, smem)[0] = blind;                                          // the blind zone
}
ECHO_HD void ph_fs_hist(int tid, int nt, const LongParams& p, unsigned char* smem) {
    const DetMisc* m = det_misc(p, smem);
    int* fh = det_fhist(p, smem);
    const int npos = p.L - p.half, base = m->base;
    for (int q = p.min_delay + tid; q < npos; q += nt)
        ECHO_LDS_ADD(&fh[fs_bin(fs_bits(fabs(det_read(p, smem, q))), base)], 1);
}
// the bin that holds the rank: totals per wavefront, then every thread scans its own run of bins
ECHO_HD void ph_fs_scan_a(int tid, int nt, const LongParams& p, unsigned char* smem) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int* fh = det_fhist(p, smem);
    const int per = (kFsBins + nt - 1) / nt;                 // nt need not divide the number of bins
    int c = 0;
    for (int e = 0; e < per; ++e) c += tid * per + e < kFsBins ? fh[tid * per + e] : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((tid & 63) == 0) det_wtot(p, smem)[tid >> 6] = c;
#endif
    if (tid == 0) det_misc(p, smem)->nlist = 0;
}
ECHO_HD void fs_chosen(DetMisc* m, int bin, int below, int cnt) {
    m->bin = bin; m->below = below; m->cnt = cnt;
    if (bin == 0 || cnt > kFsList) m->fast = 0;
}
ECHO_HD void ph_fs_scan_b(int tid, int nt, const LongParams& p, unsigned char* smem, int rank) {
    DetMisc* m = det_misc(p, smem);
    const int* fh = det_fhist(p, smem);
#if defined(__HIP_DEVICE_COMPILE__)
    const int per = (kFsBins + nt - 1) / nt, lane = tid & 63, wave = tid >> 6;
    int c = 0;
    for (int e = 0; e < per; ++e) c += tid * per + e < kFsBins ? fh[tid * per + e] : 0;
    int inc = c;
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
    const int* wt = det_wtot(p, smem);
    int exc = inc - c;
    for (int w = 0; w < wave; ++w) exc += wt[w];
    if (exc <= rank && rank < exc + c) {                    // one thread
        int cum = exc;
        for (int e = 0; e < per && tid * per + e < kFsBins; ++e) {
            const int h = fh[tid * per + e];
            if (rank < cum + h) { fs_chosen(m, tid * per + e, cum, h); break; }
            cum += h;
        }
    }
#else
    if (tid != 0) return;
    int cum = 0, b = 0;
    for (; b < kFsBins - 1; ++b) { if (rank < cum + fh[b]) break; cum += fh[b]; }
    fs_chosen(m, b, cum, fh[b]);
#endif
}
ECHO_HD void ph_fs_collect(int tid, int nt, const LongParams& p, unsigned char* smem) {
    DetMisc* m = det_misc(p, smem);
    double* list = det_list(p, smem);
    const int npos = p.L - p.half, base = m->base, bin = m->bin;
    for (int q = p.min_delay + tid; q < npos; q += 4 * nt) {   // four independent reads at a time
        double a[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = q + e * nt < npos ? fabs(det_read(p, smem, q + e * nt)) : -1.0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (a[e] >= 0.0 && fs_bin(fs_bits(a[e]), base) == bin) {
                const int slot = ECHO_LDS_FETCH_ADD(&m->nlist, 1);
                if (slot < kFsList) list[slot] = a[e];
            }
    }
}
// the element of rank (rank - below) among the collected ones: the one with at most that many smaller and more than
// that many smaller-or-equal elements (equal elements all write the same value)
ECHO_HD void ph_fs_pick(int tid, const LongParams& p, unsigned char* smem, int rank, int which) {
    DetMisc* m = det_misc(p, smem);
    const double* list = det_list(p, smem);
    const int n = m->nlist < kFsList ? m->nlist : kFsList, k = rank - m->below;
    if (tid >= n) return;
    const double x = list[tid];
    int less = 0, leq = 0;
    for (int j = 0; j < n; ++j) { const double y = list[j]; less += y < x ? 1 : 0; leq += y <= x ? 1 : 0; }
    if (less <= k && k < leq) { if (which == 0) m->lo = x; else m->hi = x; }
}

// np.median: the middle element, or the mean of the two middle elements ((a + b) / 2)
ECHO_HD void ph_ldet_set_med(int tid, const LongParams& p, unsigned char* smem, int two) {
    if (tid != 0) return;
    DetMisc* m = det_misc(p, smem);
    m->med = two ? (m->lo + m->hi) / 2.0 : m->lo;
}
ECHO_HD void ph_ldet_set_mad(int tid, const LongParams& p, unsigned char* smem, int two) {
    if (tid != 0) return;
    DetMisc* m = det_misc(p, smem);
    m->mad = (two ? (m->lo + m->hi) / 2.0 : m->lo) + 1e-9;
}
ECHO_HD void ph_ldet_cross(int tid, int nt, const LongParams& p, unsigned char* smem) {
#pragma clang fp contract(off)   // med + 6 * mad rounds twice, as numpy does, also where the file allows contraction
    DetMisc* m = det_misc(p, smem);
    const int npos = p.L - p.half;
    const double six_mad = 6.0 * m->mad;
    const double thr = m->med + six_mad;
    int first = 0x7fffffff;
    for (int q = tid; q < npos && first == 0x7fffffff; q += 4 * nt) {   // four independent reads at a time
        double v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = q + e * nt < npos ? det_read(p, smem, q + e * nt) : thr;
#pragma unroll
        for (int e = 3; e >= 0; --e) if (v[e] > thr) first = q + e * nt;
    }
    if (first != 0x7fffffff) ECHO_LDS_MIN(&m->first, first);
}
ECHO_HD void ph_ldet_finish(int tid, int blk, const LongParams& p, unsigned char* smem) {
    if (tid != 0 || !p.res_out) return;
    const DetMisc* m = det_misc(p, smem);
    BandRes r;
    r.snr_db = 0.0; r.amplitude = 0.0; r.pad = 0;
    r.peak_idx = m->first == 0x7fffffff ? -1 : m->first;
    p.res_out[blk] = r;
}

// ---------------------------------------------------------------- two-peak detector (666.py:153-200)
// pos = corr[half:]; ref = argmax(pos) is the direct path; pos2 = pos with pos2[ref] = 0; peak = argmax(pos2);
// noise floor = mean(pos2[:noise_n]^2); snr = 10 log10(pos2[peak]^2 / noise floor) (0 when the floor is 0); None when
// pos2 is all zero or snr < gate; delay = peak - ref.  amplitude = max|corr| over every lag (666.py:439).
// LDS: double pos[npos], then scratch double a[nt], double b[nt], int i[nt] (as doubles), then TpMisc.
struct TpMisc { double best, amp, noise, best2; int ref, peak, nonzero, pad; };
ECHO_HD double* tp_scratch(const LongParams& p, unsigned char* smem) {
    return reinterpret_cast<double*>(smem + p.det_lds_off);
}
ECHO_HD TpMisc* tp_misc(const LongParams& p, unsigned char* smem, int nt) {
    return reinterpret_cast<TpMisc*>(tp_scratch(p, smem) + 3 * nt);
}
// pass 1: stage pos, per-thread argmax of pos (first index on ties) and max|corr| over all lags
ECHO_HD void ph_tp_load(int tid, int nt, int blk, const LongParams& p, unsigned char* smem) {
    double* pos = det_pos(smem);
    const double* corr = p.det_io + (size_t)blk * p.det_stride;
    const int npos = p.L - p.half;
    double best = -INFINITY, amp = 0.0;
    int bi = 0x7fffffff;
    for (int k = tid; k < p.L; k += nt) {
        const double v = corr[k], av = fabs(v);
        amp = av > amp ? av : amp;
        if (k >= p.half) {
            const int q = k - p.half;
            pos[q] = v;
            if (v > best) { best = v; bi = q; }        // ascending q per thread: the first index wins ties
        }
    }
    (void)npos;
    double* sc = tp_scratch(p, smem);
    sc[tid] = best; sc[nt + tid] = amp; reinterpret_cast<int*>(sc + 2 * nt)[tid] = bi;
}
// pass 1 where the correlation is in LDS already (LongParams::det_view): the same per-thread sequences as ph_tp_load
ECHO_HD void ph_tp_scan_view(int tid, int nt, const LongParams& p, unsigned char* smem) {
    double best = -INFINITY, amp = 0.0;
    int bi = 0x7fffffff;
    for (int k0 = tid; k0 < p.L; k0 += 4 * nt) {            // four LDS reads in flight; folded in ascending k
        double v4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = k0 + e * nt < p.L ? det_read(p, smem, k0 + e * nt - p.half) : 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k0 + e * nt;
            if (k >= p.L) break;
            const double v = v4[e], av = fabs(v);
            amp = av > amp ? av : amp;
            if (k >= p.half && v > best) { best = v; bi = k - p.half; }
        }
    }
    double* sc = tp_scratch(p, smem);
    sc[tid] = best; sc[nt + tid] = amp; reinterpret_cast<int*>(sc + 2 * nt)[tid] = bi;
}
ECHO_HD void tp_better(double& b, int& i, double b2, int i2) { if (b2 > b || (b2 == b && i2 < i)) { b = b2; i = i2; } }
// The per-thread partials are folded in two steps: 32 threads each fold a run of nt/32 entries into its first entry,
// then thread 0 folds the 32 run heads (nt is a multiple of 32).
constexpr int kTpFold = 32;
ECHO_HD void ph_tp_ref_fold(int tid, int nt, const LongParams& p, unsigned char* smem) {
    if (tid >= kTpFold) return;
    double* sc = tp_scratch(p, smem);
    int* si = reinterpret_cast<int*>(sc + 2 * nt);
    const int run = nt / kTpFold, t0 = tid * run;
    double best = sc[t0], amp = sc[nt + t0]; int bi = si[t0];
    for (int t = t0 + 1; t < t0 + run; ++t) { tp_better(best, bi, sc[t], si[t]); amp = sc[nt + t] > amp ? sc[nt + t] : amp; }
    sc[t0] = best; sc[nt + t0] = amp; si[t0] = bi;
}
ECHO_HD void ph_tp_ref(int tid, int nt, const LongParams& p, unsigned char* smem) {
    if (tid != 0) return;
    const double* sc = tp_scratch(p, smem);
    const int* si = reinterpret_cast<const int*>(sc + 2 * nt);
    TpMisc* m = tp_misc(p, smem, nt);
    const int run = nt / kTpFold;
    double best = sc[0], amp = sc[nt]; int bi = si[0];
    for (int t = run; t < nt; t += run) { tp_better(best, bi, sc[t], si[t]); amp = sc[nt + t] > amp ? sc[nt + t] : amp; }
    m->best = best; m->ref = bi; m->amp = amp;
}
// pass 2 over pos2: argmax, any nonzero, noise power
ECHO_HD void ph_tp_second(int tid, int nt, const LongParams& p, unsigned char* smem) {
    const TpMisc* m = tp_misc(p, smem, nt);
    const int npos = p.L - p.half, ref = m->ref;
    double best = -INFINITY, noise = 0.0; int bi = 0x7fffffff, nz = 0;
    for (int q0 = tid; q0 < npos; q0 += 4 * nt) {           // four LDS reads in flight; folded in ascending q
        double v4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v4[e] = q0 + e * nt < npos ? det_read(p, smem, q0 + e * nt) : 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int q = q0 + e * nt;
            if (q >= npos) break;
            const double v = q == ref ? 0.0 : v4[e];
            if (v > best) { best = v; bi = q; }
            nz |= (v != 0.0) ? 1 : 0;
            if (q < p.noise_n) noise += v * v;
        }
    }
    double* sc = tp_scratch(p, smem);
    int* si = reinterpret_cast<int*>(sc + 2 * nt);
    sc[tid] = best; sc[nt + tid] = noise; si[tid] = (bi << 1) | nz;      // bi < 2^30
}
ECHO_HD void ph_tp_second_fold(int tid, int nt, const LongParams& p, unsigned char* smem) {
    if (tid >= kTpFold) return;
    double* sc = tp_scratch(p, smem);
    int* si = reinterpret_cast<int*>(sc + 2 * nt);
    const int run = nt / kTpFold, t0 = tid * run;
    double best = sc[t0], noise = sc[nt + t0]; int bi = si[t0] >> 1, nz = si[t0] & 1;
    for (int t = t0 + 1; t < t0 + run; ++t) { tp_better(best, bi, sc[t], si[t] >> 1); noise += sc[nt + t]; nz |= si[t] & 1; }
    sc[t0] = best; sc[nt + t0] = noise; si[t0] = (bi << 1) | nz;
}
ECHO_HD void ph_tp_finish(int tid, int nt, int blk, const LongParams& p, unsigned char* smem) {
    if (tid != 0) return;
    const double* sc = tp_scratch(p, smem);
    const int* si = reinterpret_cast<const int*>(sc + 2 * nt);
    const TpMisc* m = tp_misc(p, smem, nt);
    const int run = nt / kTpFold;
    double best = sc[0], noise = sc[nt]; int bi = si[0] >> 1, nz = si[0] & 1;
    for (int t = run; t < nt; t += run) { tp_better(best, bi, sc[t], si[t] >> 1); noise += sc[nt + t]; nz |= si[t] & 1; }
    BandRes r;
    r.amplitude = m->amp; r.snr_db = 0.0; r.peak_idx = -1; r.pad = 0;
    const int npos = p.L - p.half;
    if (npos > 0 && nz) {
        const double floor_ = p.noise_n > 0 ? noise / (double)p.noise_n : 0.0;
        const double snr = floor_ > 0.0 ? 10.0 * log10(best * best / floor_) : 0.0;
        if (!(snr < p.snr_gate)) { r.snr_db = snr; r.peak_idx = bi - m->ref; r.pad = 1; }
    }
    if (p.res_out) p.res_out[blk] = r;
}

// ---------------------------------------------------------------- median / mean-power detector on long frames
// first_strong_peak of airSonar_optimized.py:200-219 (noise floor = median(|pos[min_delay : min_delay + n]|)^2) and of
// realtime_sonar.py:127-142 (mean(pos[:n]^2)) for frames that do not fit the single-workgroup kernel of
// band_phases.h; same results as ph_corr .. ph_finish there.  det_io points at the whole correlation (L lags):
// amplitude = max|corr| after the blind zone has been zeroed.  p.noise_n is the window length after NumPy's slice
// truncation.  LDS: double pos[npos], scratch double best[nt], amp[nt], sumsq[nt], int idx[nt], then LmMisc.
struct LmMisc { double best, amp, sumsq, med0, med1; int bidx, pad; };
ECHO_HD double* lm_scratch(const LongParams& p, unsigned char* smem) {
    return reinterpret_cast<double*>(smem + (((p.L - p.half) + 1) & ~1) * 8);
}
ECHO_HD LmMisc* lm_misc(const LongParams& p, unsigned char* smem, int nt) {
    return reinterpret_cast<LmMisc*>(reinterpret_cast<unsigned char*>(lm_scratch(p, smem) + 3 * nt) + nt * 4);
}
ECHO_HD void ph_lm_load(int tid, int nt, int blk, const LongParams& p, unsigned char* smem) {
    double* pos = det_pos(smem);
    double* corr = p.det_io + (size_t)blk * p.det_stride;
    double best = -INFINITY, amp = 0.0, sumsq = 0.0;
    int bi = 0x7fffffff;
    for (int k = tid; k < p.L; k += nt) {
        double v = corr[k];
        if (k >= p.half) {
            const int q = k - p.half;
            if (q < p.min_delay) { v = 0.0; if (p.write_mutated) corr[k] = 0.0; }
            pos[q] = v;
            if (v > best) { best = v; bi = q; }                 // ascending q per thread: the first index wins ties
            if (p.detector == 2 && q < p.noise_n) sumsq += v * v;
        }
        const double av = fabs(v);
        amp = av > amp ? av : amp;
    }
    double* sc = lm_scratch(p, smem);
    sc[tid] = best; sc[nt + tid] = amp; sc[2 * nt + tid] = sumsq; reinterpret_cast<int*>(sc + 3 * nt)[tid] = bi;
}
ECHO_HD void ph_lm_fold(int tid, int nt, const LongParams& p, unsigned char* smem) {
    if (tid >= kTpFold) return;
    double* sc = lm_scratch(p, smem);
    int* si = reinterpret_cast<int*>(sc + 3 * nt);
    const int run = nt / kTpFold, t0 = tid * run;
    double best = sc[t0], amp = sc[nt + t0], ss = sc[2 * nt + t0]; int bi = si[t0];
    for (int t = t0 + 1; t < t0 + run; ++t) {
        tp_better(best, bi, sc[t], si[t]);
        amp = sc[nt + t] > amp ? sc[nt + t] : amp;
        ss += sc[2 * nt + t];
    }
    sc[t0] = best; sc[nt + t0] = amp; sc[2 * nt + t0] = ss; si[t0] = bi;
}
ECHO_HD void ph_lm_best(int tid, int nt, const LongParams& p, unsigned char* smem) {
    if (tid != 0) return;
    const double* sc = lm_scratch(p, smem);
    const int* si = reinterpret_cast<const int*>(sc + 3 * nt);
    LmMisc* m = lm_misc(p, smem, nt);
    const int run = nt / kTpFold;
    double best = sc[0], amp = sc[nt], ss = sc[2 * nt]; int bi = si[0];
    for (int t = run; t < nt; t += run) {
        tp_better(best, bi, sc[t], si[t]);
        amp = sc[nt + t] > amp ? sc[nt + t] : amp;
        ss += sc[2 * nt + t];
    }
    m->best = best; m->bidx = bi; m->amp = amp; m->sumsq = ss; m->med0 = 0.0; m->med1 = 0.0;
}
// np.median of the noise window: element v is the k-th order statistic iff #(u < v) <= k < #(u < v) + #(u == v)
ECHO_HD void ph_lm_median(int tid, int nt, const LongParams& p, unsigned char* smem) {
    if (p.detector != 0) return;
    const double* nw = det_pos(smem) + p.min_delay;
    LmMisc* m = lm_misc(p, smem, nt);
    const int n = p.noise_n;
    const int k0 = (n - 1) / 2, k1 = n / 2;
    for (int i = tid; i < n; i += nt) {
        const double v = fabs(nw[i]);
        int less = 0, eq = 0;
        for (int j = 0; j < n; ++j) { const double u = fabs(nw[j]); less += (u < v) ? 1 : 0; eq += (u == v) ? 1 : 0; }
        if (less <= k0 && k0 < less + eq) m->med0 = v;
        if (less <= k1 && k1 < less + eq) m->med1 = v;
    }
}
ECHO_HD void ph_lm_finish(int tid, int nt, int blk, const LongParams& p, unsigned char* smem) {
    if (tid != 0 || !p.res_out) return;
    const LmMisc* m = lm_misc(p, smem, nt);
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
#define ECHO_LMED_SEQUENCE(PHASE, SYNC, p, smem, blk, nt)              \
    PHASE(ph_lm_load(tid, nt, blk, p, smem)) SYNC                       \
    PHASE(ph_lm_fold(tid, nt, p, smem)) SYNC                            \
    PHASE(ph_lm_best(tid, nt, p, smem)) SYNC                            \
    PHASE(ph_lm_median(tid, nt, p, smem)) SYNC                          \
    PHASE(ph_lm_finish(tid, nt, blk, p, smem))

// The barrier-separated sequence of one detector workgroup; SYNC is __syncthreads() on the device and the end of a
// for-all-threads loop on the host.
#define ECHO_LDET_SEQUENCE(PHASE, SYNC, p, smem, blk, nt)                                               \
    do {                                                                                                \
        const int npos_ = (p).L - (p).half;                                                             \
        if (npos_ <= (p).min_delay) {                                                                   \
            PHASE(if (tid == 0 && (p).res_out) { BandRes r_; r_.snr_db = 0.0; r_.amplitude = 0.0;       \
                                                  r_.pad = 0; r_.peak_idx = -1; (p).res_out[blk] = r_; }) \
            break;                                                                                      \
        }                                                                                               \
        PHASE(ph_ldet_load(tid, nt, blk, p, smem)) SYNC                                                 \
        ECHO_LDET_BODY(PHASE, SYNC, p, smem, blk, nt);                                                  \
    } while (0)

// the detector on a pos that is in LDS already (det_view, det_lds_off say where), scalars initialised
#define ECHO_LDET_BODY(PHASE, SYNC, p, smem, blk, nt)                                                   \
    do {                                                                                                \
        const int npos_ = (p).L - (p).half;                                                             \
        const int two_ = (npos_ % 2 == 0) ? 1 : 0;                                                      \
        const bool all_ = !((p).det_stages > 0 && (p).det_stages < 3) && !((p).det_passes > 0 && (p).det_passes < 8); \
        if (all_) {                                                                                     \
            PHASE(ph_fs_stats(tid, nt, p, smem)) SYNC                                                   \
            PHASE(ph_fs_decide(tid, p, smem, two_)) SYNC                                                \
            if (det_misc(p, smem)->fast == 1) {                                                         \
                PHASE(ph_fs_hist(tid, nt, p, smem)) SYNC                                                \
                for (int w_ = 0; w_ <= two_; ++w_) {                                                    \
                    const int rank_ = two_ ? npos_ / 2 - 1 + w_ : npos_ / 2;                            \
                    PHASE(ph_fs_scan_a(tid, nt, p, smem)) SYNC                                          \
                    PHASE(ph_fs_scan_b(tid, nt, p, smem, rank_)) SYNC                                   \
                    if (det_misc(p, smem)->fast != 1) break;                                            \
                    PHASE(ph_fs_collect(tid, nt, p, smem)) SYNC                                         \
                    PHASE(ph_fs_pick(tid, p, smem, rank_, w_)) SYNC                                     \
                }                                                                                       \
            }                                                                                           \
        }                                                                                               \
        if (all_ && det_misc(p, smem)->fast != 0) {                                                     \
            PHASE(ph_ldet_set_mad(tid, p, smem, two_)) SYNC                                             \
        } else {                                                                                        \
            const int nsel_ = (p).det_stages > 0 && (p).det_stages < 3 ? (p).det_stages - 1 : 2;        \
            const int npass_ = (p).det_passes > 0 && (p).det_passes < 8 ? (p).det_passes : 8;           \
            for (int dev_ = 0; dev_ < nsel_; ++dev_) {                                                  \
                for (int w_ = 0; w_ <= two_; ++w_) {                                                    \
                    const int rank_ = two_ ? npos_ / 2 - 1 + w_ : npos_ / 2;                            \
                    PHASE(ph_sel_begin(tid, p, smem, rank_)) SYNC                                       \
                    for (int pass_ = 0; pass_ < npass_; ++pass_) {                                      \
                        PHASE(ph_sel_count(tid, nt, p, smem, dev_, pass_ & 1)) SYNC                     \
                        PHASE(ph_sel_pick(tid, p, smem, pass_ & 1)) SYNC                                \
                        if (det_misc(p, smem)->cnt == 1) break;                                         \
                    }                                                                                   \
                    PHASE(ph_sel_find(tid, nt, p, smem, dev_)) SYNC                                     \
                    PHASE(ph_sel_store(tid, p, smem, w_)) SYNC                                          \
                }                                                                                       \
                PHASE(if (dev_ == 0) ph_ldet_set_med(tid, p, smem, two_);                               \
                      else ph_ldet_set_mad(tid, p, smem, two_)) SYNC                                    \
            }                                                                                           \
        }                                                                                               \
        PHASE(ph_ldet_cross(tid, nt, p, smem)) SYNC                                                     \
        PHASE(ph_ldet_finish(tid, blk, p, smem))                                                        \
    } while (0)

}  // namespace echo
