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
struct alignas(16) float4 { float x, y, z, w; };
#endif

// Pointer to LDS data read one 8-byte element at a time.  On the device this is a volatile LDS-address-space
// pointer: it stops the compiler from merging neighbouring reads into two-element reads, which run at half the
// LDS rate on gfx950.  On the host it is a plain pointer.
#if defined(__HIP_DEVICE_COMPILE__)
typedef const volatile __attribute__((address_space(3))) double* lds_cptr_t;
#define ECHO_LDS_CPTR(p) ((lds_cptr_t)(p))
#else
typedef const double* lds_cptr_t;
#define ECHO_LDS_CPTR(p) (p)
#endif

namespace echo {

struct B2Det;

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
    int do_corr, spec_db;
    double snr_gate;
    // padded coefficient tables (device global), zero-filled to a multiple of kR plus one extra zero group
    int kgroups, mgroups;            // ceil(K/kR), ceil(M/kR)
    const double* taps_rev;          // [nb][(kgroups+1)*kR]  taps reversed
    const double* chirp;             // [nb][(mgroups+1)*kR]
    const float2* bw;                // [n2 rounded up to even]  exp(-i*pi*n^2/n2)
    const float2* bhat;              // [8][kFftM/16][2]  FFT of the chirp filter, /kFftM, pass order, lane-major
    const float2* tw;                // radix-8 pass twiddles: for q in {1024,128,16}: [3][q] = W^i, W^2i, W^4i, W = exp(-2*pi*i/(8q))
    const float2* post;              // [n2+1][2]  A[k] = w[k](1 - i u_k)/2, B[k] = conj(w[n2-k])(1 + i u_k)/2, u_k = exp(-2*pi*i*k/N)
    int off_ctap, off_cchirp;        // LDS copies of this band's coefficient tables
    int off_ctap32;                  // float32 copy of the reversed taps
    int off_cchirp32;                // float32 copy of the chirp (its values are float32 by construction)
    int corr_fast;                   // chain only: float32 correlation where only max|corr| is needed (see ph_corr)
    // LDS carve (byte offsets, 16-B aligned)
    int off_sig, sig_len;            // double[sig_len]: band_sig at [M-1, M-1+N), zeros elsewhere
    int off_y;                       // double ybuf[ybuf_len]: the FIR input; shares the memory of sig (the FIR keeps its
                                     // outputs in registers until every lane has read its inputs)
    int off_u;                       // reduction scratch + median counts, last region of the layout
    int smem_bytes_host;             // the host emulation keeps one reduction slot per thread instead of per wavefront
    int off_fft;                     // spectrum kernel: float2 fft[kFftM]
    int spec_smem_bytes;
    int ybuf_len, ypad;              // ybuf holds input at [ypad, ypad+N)
    int off_nw;                      // double[noise_n] noise window, then float[noise_n] (rounded copy)
    int off_misc;                    // small scalars
    int smem_bytes;
    // I/O (device global); any may be null depending on mode
    const float* rx_in;              // [B][N]
    const double* d_in;              // [B][nb][N] (y or band_sig)
    double* corr_io;                 // [B][nb][L]  in (MODE_CORR) / out (raw corr) / null
    double* band_out;                // [B][nb][N]
    float* spec_out;                 // [B][nb][nbins]
    float* bandf_out;                // [B][nb][N] float32 copy of band_sig for the spectrum kernel (may be null)
    const float* specf_in;           // spectrum kernel input, float32 [B][nb][N] (or null)
    const double* specd_in;          // spectrum kernel input, float64 [B][nb][N] (or null)
    BandRes* res_out;                // [B][nb]
    const float2* uk;                // [n2+1] exp(-2*pi*i*k/N) (envelope branch of the spectrum kernel)
    // 4608-point variant of the cyclic convolution (fft9_phases.h), used when 2*n2 - 1 <= 4608
    int fft9;
    const float2* f9tw9;             // [8][512]  exp(-2*pi*i*(i*m)/4608), m = 1..8
    const float2* f9twa;             // [3][64]   W^i, W^2i, W^4i, W = exp(-2*pi*i/512)
    const float2* f9twb;             // [3][8]    W^i, W^2i, W^4i, W = exp(-2*pi*i/64)
    const float2* f9bhat;            // [4608]    FFT of the chirp filter / 4608, pass order
    int spec_hilb_smem_bytes;        // LDS of the spectrum kernel with the envelope branch
    double* sq_out;                  // [B][nb] signal_quality (envelope branch; null = spectrum only)
    int write_corr_mutated;          // MODE_CORR: write the zeroed blind zone back
    // Pair kernel of the chain (band2_kernel): float32 FFT matched filter over all lags, exact float64 re-evaluation
    // of the noise window and of every lag that can be the maximum
    int band2;                       // configuration fits the pair kernel
    const float2* f9ghat;            // [nb][4608] FFT of the reversed chirp / 4608, pass order of the 4608-point transform
    const float2* taps32x2;          // [nb][(kgroups + 1) * kR] reversed FIR taps in float32, each as the pair (c, c)
    double eps_fir[3];               // per band: bound on the float32 FIR's contribution to a lag's error, per unit max|y|
    double eps_fft[3];               // per band: bound on the float32 transforms' contribution, per unit ||band_sig pair||_2
    int b2_off_ct32, b2_off_red, b2_off_misc, b2_smem_bytes;   // LDS of the pair kernel behind the transform buffer
    int b2_ywlen, b2_bwlen;          // exact kernel: doubles of the y and band_sig windows of one segment
    int b2_exact_smem_bytes;
    struct B2Det* det_out;           // [B][nb] float32 maxima and candidate lags, pair kernel -> exact kernel
    // Full chain behind the fused envelope kernel: only the (frame, band)s whose envelope branch found no peak need
    // first_strong_peak (airSonar_optimized.py:657); a workgroup leaves at once when skip_gate[blk * skip_stride] >= 0.
    const int32_t* skip_gate;
    int skip_stride;
};

struct Misc {                        // lives at off_misc
    double best; double amp; double sumsq; double med0; double med1;
    int bidx; int pad;
};

// Fixed-order fold of a double over the 64 lanes of a wavefront, every lane ends with the total; op 0 sum, 1 max, 2 min.
// Neighbours, pairs, the mirrored half row and the mirrored row by DPP moves (the same sets as a butterfly over
// lane ^ 1, 2, 4, 8, and the operations commute), then the four rows' totals read as scalars: (r0 + r1) + (r2 + r3).
// Shuffles through the LDS crossbar (__shfl_xor) cost an LDS round trip per 32-bit half and step.  The host build has no
// wavefronts: it returns its argument and keeps one partial per thread.
ECHO_HD double wave_fold_op(int op, double a, double b) { return op == 0 ? a + b : (op == 1 ? (a > b ? a : b) : (a < b ? a : b)); }
#if defined(__HIP_DEVICE_COMPILE__)
template <int CTRL> __device__ __forceinline__ double wave_fold_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_fold_lane(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane),
                            __builtin_amdgcn_readlane(__double2loint(v), lane));
}
#endif
ECHO_HD double wave_fold(int op, double a) {
#if defined(__HIP_DEVICE_COMPILE__)
    a = wave_fold_op(op, a, wave_fold_dpp<0xB1>(a));      // quad_perm [1,0,3,2]
    a = wave_fold_op(op, a, wave_fold_dpp<0x4E>(a));      // quad_perm [2,3,0,1]
    a = wave_fold_op(op, a, wave_fold_dpp<0x141>(a));     // row_half_mirror
    a = wave_fold_op(op, a, wave_fold_dpp<0x140>(a));     // row_mirror
    const double r0 = wave_fold_lane(a, 0), r1 = wave_fold_lane(a, 16), r2 = wave_fold_lane(a, 32), r3 = wave_fold_lane(a, 48);
    a = wave_fold_op(op, wave_fold_op(op, r0, r1), wave_fold_op(op, r2, r3));
#endif
    return a;
}

ECHO_HD double* lds_d(unsigned char* s, int off) { return reinterpret_cast<double*>(s + off); }
// reduction slots of a workgroup: one per wavefront on the device, one per thread in the host emulation
ECHO_HD int red_slots(int nt) {
#if defined(__HIP_DEVICE_COMPILE__)
    return nt / 64;
#else
    return nt;
#endif
}
ECHO_HD float2* lds_c(unsigned char* s, int off) { return reinterpret_cast<float2*>(s + off); }

// One group of kR taps: acc[r] += sum_t w[(r+t)%kR] * c[t]; slot t of the register window is refilled with the
// sample kR positions ahead as soon as its last use is behind, so the window rotates by static index.
ECHO_HD void dot_block(double (&w)[kR], const double (&c)[kR], const double* nx, double (&acc)[kR]) {
    // volatile: keeps one 8-byte LDS read per element; merged two-element reads run at half the LDS rate on gfx950
    lds_cptr_t vx = ECHO_LDS_CPTR(nx);
#pragma unroll
    for (int t = 0; t < kR; ++t) {
#pragma unroll
        for (int r = 0; r < kR; ++r) acc[r] = fma(w[(r + t) % kR], c[t], acc[r]);
        w[t] = vx[t];
    }
}

// out[r] = sum_j c[j] * in[r + j], r < kR, j < ngroups*kR.  `in` and `c` are LDS arrays; c is padded with one
// extra (zero) group so the coefficient prefetch never runs past it.  Coefficients for the next group are
// fetched while the current group's kR*kR multiply-adds issue.
ECHO_HD void sliding_dot(const double* in, const double* c, int ngroups, double (&acc)[kR]) {
    double w[kR], c0[kR], c1[kR];
    lds_cptr_t vc = ECHO_LDS_CPTR(c);
#pragma unroll
    for (int r = 0; r < kR; ++r) { w[r] = in[r]; c0[r] = c[r]; acc[r] = 0.0; }
    int g = 0;
    for (; g + 1 < ngroups; g += 2) {
#pragma unroll
        for (int r = 0; r < kR; ++r) c1[r] = vc[(g + 1) * kR + r];
        dot_block(w, c0, in + (g + 1) * kR, acc);
#pragma unroll
        for (int r = 0; r < kR; ++r) c0[r] = vc[(g + 2) * kR + r];
        dot_block(w, c1, in + (g + 2) * kR, acc);
    }
    if (g < ngroups) dot_block(w, c0, in + (g + 1) * kR, acc);
}

// The same sliding dot with R consecutive outputs per lane (R independent accumulation chains): identical operation
// order per output, hence identical bits, for any R.  c must be zero-padded to (ngroups + 1) * R entries, `in` readable
// up to index (ngroups + 1) * R - 1.
template <int R> ECHO_HD void dot_block_r(double (&w)[R], const double (&c)[R], const double* nx, double (&acc)[R]) {
    lds_cptr_t vx = ECHO_LDS_CPTR(nx);
#pragma unroll
    for (int t = 0; t < R; ++t) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = fma(w[(r + t) % R], c[t], acc[r]);
        w[t] = vx[t];
    }
}
template <int R> ECHO_HD void sliding_dot_r(const double* in, const double* c, int ngroups, double (&acc)[R]) {
    double w[R], c0[R], c1[R];
    lds_cptr_t vc = ECHO_LDS_CPTR(c);
#pragma unroll
    for (int r = 0; r < R; ++r) { w[r] = in[r]; c0[r] = c[r]; acc[r] = 0.0; }
    int g = 0;
    for (; g + 1 < ngroups; g += 2) {
#pragma unroll
        for (int r = 0; r < R; ++r) c1[r] = vc[(g + 1) * R + r];
        dot_block_r<R>(w, c0, in + (g + 1) * R, acc);
#pragma unroll
        for (int r = 0; r < R; ++r) c0[r] = vc[(g + 2) * R + r];
        dot_block_r<R>(w, c1, in + (g + 2) * R, acc);
    }
    if (g < ngroups) dot_block_r<R>(w, c0, in + (g + 1) * R, acc);
}

// float32 signal and coefficients (the pair kernel's FIR)
ECHO_HD void dot_block_ff(float (&w)[kR], const float (&c)[kR], const float* nx, float (&acc)[kR]) {
#pragma unroll
    for (int t = 0; t < kR; ++t) {
#pragma unroll
        for (int r = 0; r < kR; ++r) acc[r] = fmaf(w[(r + t) % kR], c[t], acc[r]);
        w[t] = nx[t];
    }
}
ECHO_HD void sliding_dot_ff(const float* in, const float* c, int ngroups, float (&acc)[kR]) {
    float w[kR], c0[kR], c1[kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) { w[r] = in[r]; c0[r] = c[r]; acc[r] = 0.f; }
    int g = 0;
    for (; g + 1 < ngroups; g += 2) {
#pragma unroll
        for (int r = 0; r < kR; ++r) c1[r] = c[(g + 1) * kR + r];
        dot_block_ff(w, c0, in + (g + 1) * kR, acc);
#pragma unroll
        for (int r = 0; r < kR; ++r) c0[r] = c[(g + 2) * kR + r];
        dot_block_ff(w, c1, in + (g + 2) * kR, acc);
    }
    if (g < ngroups) dot_block_ff(w, c0, in + (g + 1) * kR, acc);
}

// float32 variant: the window comes from the float64 LDS signal (converted on load), coefficients from a float copy.
ECHO_HD void dot_block_f(float (&w)[kR], const float (&c)[kR], const double* nx, float (&acc)[kR]) {
    lds_cptr_t vx = ECHO_LDS_CPTR(nx);
#pragma unroll
    for (int t = 0; t < kR; ++t) {
#pragma unroll
        for (int r = 0; r < kR; ++r) acc[r] = fmaf(w[(r + t) % kR], c[t], acc[r]);
        w[t] = (float)vx[t];
    }
}

ECHO_HD void sliding_dot_f(const double* in, const float* c, int ngroups, float (&acc)[kR]) {
    float w[kR], c0[kR], c1[kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) { w[r] = (float)in[r]; c0[r] = c[r]; acc[r] = 0.f; }
    int g = 0;
    for (; g + 1 < ngroups; g += 2) {
#pragma unroll
        for (int r = 0; r < kR; ++r) c1[r] = c[(g + 1) * kR + r];
        dot_block_f(w, c0, in + (g + 1) * kR, acc);
#pragma unroll
        for (int r = 0; r < kR; ++r) c0[r] = c[(g + 2) * kR + r];
        dot_block_f(w, c1, in + (g + 2) * kR, acc);
    }
    if (g < ngroups) dot_block_f(w, c0, in + (g + 1) * kR, acc);
}

// ---------------------------------------------------------------- phase 0: stage the input in LDS
ECHO_HD void ph_load(int tid, int nt, int blk, const BandParams& p, unsigned char* smem) {
    double* sig = lds_d(smem, p.off_sig);
    double* yb = lds_d(smem, p.off_y);
    const int band = blk % p.nb, frame = blk / p.nb;
    // this band's coefficient tables -> LDS (read back as wave-uniform broadcasts by the sliding dots)
    if (p.mode == MODE_RX_FIR || p.mode == MODE_Y_FIR) {
        double* ct = lds_d(smem, p.off_ctap);
        const int n = (p.kgroups + 1) * kR;
        float* cf = reinterpret_cast<float*>(smem + p.off_ctap32);
        for (int i = tid; i < n; i += nt) { const double v = p.taps_rev[(size_t)band * n + i]; ct[i] = v; cf[i] = (float)v; }
    }
    if (p.do_corr && p.mode != MODE_CORR) {
        double* cc = lds_d(smem, p.off_cchirp);
        const int n = (p.mgroups + 1) * kR;
        float* cf = reinterpret_cast<float*>(smem + p.off_cchirp32);
        for (int i = tid; i < n; i += nt) { const double v = p.chirp[(size_t)band * n + i]; cc[i] = v; cf[i] = (float)v; }
    }
    if (p.mode == MODE_RX_FIR || p.mode == MODE_Y_FIR) {
        // all global loads of a batch are issued before the first LDS store, so their latencies overlap
        constexpr int kBatch = 5;
        for (int i0 = tid; i0 < p.ybuf_len; i0 += kBatch * nt) {
            double v[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const int n = i0 + k * nt - p.ypad;
                const int nc = n < 0 ? 0 : (n >= p.N ? p.N - 1 : n);
                const double x = (p.mode == MODE_RX_FIR) ? (double)p.rx_in[(size_t)frame * p.N + nc]
                                                         : p.d_in[(size_t)blk * p.N + nc];
                v[k] = (n == nc) ? x : 0.0;
            }
#pragma unroll
            for (int k = 0; k < kBatch; ++k)
                if (i0 + k * nt < p.ybuf_len) yb[i0 + k * nt] = v[k];
        }
    } else if (p.mode == MODE_BAND) {
        constexpr int kBatch = 5;
        for (int i0 = tid; i0 < p.sig_len; i0 += kBatch * nt) {
            double v[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; ++k) {
                const int n = i0 + k * nt - (p.M - 1);
                const int nc = n < 0 ? 0 : (n >= p.N ? p.N - 1 : n);
                const double x = p.d_in[(size_t)blk * p.N + nc];
                v[k] = (n == nc) ? x : 0.0;
            }
#pragma unroll
            for (int k = 0; k < kBatch; ++k)
                if (i0 + k * nt < p.sig_len) sig[i0 + k * nt] = v[k];
        }
    }
}

// ---------------------------------------------------------------- phase 1: FIR 'same'
// band_sig[n] = sum_j taps[j] * y[n + s - j], s = (K-1)/2  ==  sum_j' taps_rev[j'] * ybuf[n + j']
// Returns the lane's kR outputs in acc (kept in registers across the barrier that separates the last read of ybuf
// from the first write of sig, which share memory).
ECHO_HD void ph_fir(int tid, int nt, int blk, const BandParams& p, unsigned char* smem, double (&acc)[kR]) {
    if (p.mode != MODE_RX_FIR && p.mode != MODE_Y_FIR) return;
    const double* yb = lds_d(smem, p.off_y);
    const int n0 = tid * kR;
    if (n0 >= p.N) return;
    // Chain fast path: band_sig[n] below nexact only reaches the float32 half of the correlation and the float32
    // spectrum, so those lanes run the FIR in float32 too (float64 is kept wherever the detector can see it).
    const int nexact = p.half + p.min_delay - (p.M - 1);
    const bool fast = p.corr_fast && p.mode == MODE_Y_FIR && !p.band_out && !p.corr_io && p.detector == 0;
    if (fast && n0 + kR <= nexact) {
        float accf[kR];
        sliding_dot_f(yb + n0, reinterpret_cast<const float*>(smem + p.off_ctap32), p.kgroups, accf);
#pragma unroll
        for (int r = 0; r < kR; ++r) acc[r] = (double)accf[r];
    } else {
        sliding_dot(yb + n0, lds_d(smem, p.off_ctap), p.kgroups, acc);
    }
}
ECHO_HD void ph_fir_store(int tid, int nt, int blk, const BandParams& p, unsigned char* smem, const double (&acc)[kR]) {
    if (p.mode != MODE_RX_FIR && p.mode != MODE_Y_FIR) return;
    double* sig = lds_d(smem, p.off_sig);
    for (int i = tid; i < p.sig_len; i += nt)
        if (i < p.M - 1 || i >= p.M - 1 + p.N) sig[i] = 0.0;
    const int n0 = tid * kR;
    if (n0 >= p.N) return;
#pragma unroll
    for (int r = 0; r < kR; ++r) {
        const int n = n0 + r;
        if (n < p.N) {
            sig[p.M - 1 + n] = acc[r];
            if (p.band_out) p.band_out[(size_t)blk * p.N + n] = acc[r];
            if (p.bandf_out) p.bandf_out[(size_t)blk * p.N + n] = (float)acc[r];
        }
    }
}

ECHO_HD void combine(double& best, int& bidx, double b2, int i2) {
    if (b2 > best || (b2 == best && i2 < bidx)) { best = b2; bidx = i2; }
}

// ---------------------------------------------------------------- phase 2: correlation + detector partials
// corr[k] = sum_j sig[k + j] * chirp[j]  (sig carries M-1 leading zeros), k < L.
// Per lane: blind-zone zeroing, max|corr|, signed max of pos = corr[half:] with first-index ties,
// noise-window samples, sum of squares of pos[:noise_n] (mean-power detector).
ECHO_HD void ph_corr(int tid, int nt, int blk, const BandParams& p, unsigned char* smem) {
    if (!p.do_corr) return;
    const double* sig = lds_d(smem, p.off_sig);
    double* nw = lds_d(smem, p.off_nw);
    float* nwf = reinterpret_cast<float*>(nw + p.noise_n);
    double* red = lds_d(smem, p.off_u);            // [slots] best | [slots] amp | [slots] sumsq | int[slots] idx
    double best = -INFINITY, amp = 0.0, sumsq = 0.0;
    int bidx = 0x7fffffff;
    // Fast split (chain, median detector): only pos[min_delay:] = corr[half+min_delay:] feeds the argmax, the noise
    // floor and the SNR, so only those outputs are evaluated in float64.  corr[:half] contributes nothing but
    // max|corr| (amplitude enters the confidence through min(amp/0.1, 1) and is compared at 1e-4), so the first
    // half of the workgroup evaluates it in float32 at twice the rate; the blind zone is zero by definition and is
    // not evaluated at all.
    const bool fast = p.corr_fast && p.mode != MODE_CORR && !p.corr_io && p.detector == 0;
    const int kstart = p.half + p.min_delay;
    const int nlow = nt / 2;                               // threads [0, nlow): float32 part, [nlow, nt): float64 part
    if (fast && (p.half + kR - 1) / kR <= nlow && (p.L - kstart + kR - 1) / kR <= nt - nlow) {
        if (tid < nlow) {
            const int k0 = tid * kR;
            if (k0 < p.half) {
                float acc[kR];
                sliding_dot_f(sig + k0, reinterpret_cast<const float*>(smem + p.off_cchirp32), p.mgroups, acc);
#pragma unroll
                for (int r = 0; r < kR; ++r)
                    if (k0 + r < p.half) { const double av = fabs((double)acc[r]); amp = av > amp ? av : amp; }
            }
        } else {
            const int k0 = kstart + (tid - nlow) * kR;
            if (k0 < p.L) {
                double acc[kR];
                sliding_dot(sig + k0, lds_d(smem, p.off_cchirp), p.mgroups, acc);
#pragma unroll
                for (int r = 0; r < kR; ++r) {
                    const int k = k0 + r;
                    if (k >= p.L) continue;
                    const double v = acc[r], av = fabs(v);
                    const int q = k - p.half, w = q - p.min_delay;
                    amp = av > amp ? av : amp;
                    if (v > best) { best = v; bidx = q; }
                    if (w < p.noise_n) { nw[w] = av; nwf[w] = (float)av; }
                }
            }
        }
        if (p.min_delay > 0 && tid == 0 && 0.0 > best) { best = 0.0; bidx = 0; }   // the zeroed blind zone: pos[0] = 0
    } else {
    const int k0 = tid * kR;
    if (k0 < p.L) {
        double acc[kR];
        if (p.mode == MODE_CORR) {
#pragma unroll
            for (int r = 0; r < kR; ++r) acc[r] = (k0 + r < p.L) ? p.corr_io[(size_t)blk * p.L + k0 + r] : 0.0;
        } else {
            sliding_dot(sig + k0, lds_d(smem, p.off_cchirp), p.mgroups, acc);
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
                    if (w >= 0 && w < p.noise_n) { nw[w] = av; nwf[w] = (float)av; }
                } else if (p.detector == 2) {
                    if (q < p.noise_n) sumsq += v * v;
                }
            }
        }
    }
    }
    // one partial per wavefront on the device (butterfly over the 64 lanes), one per thread on the host
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double b2 = __shfl_xor(best, o); const int i2 = __shfl_xor(bidx, o);
        combine(best, bidx, b2, i2);
        const double a2 = __shfl_xor(amp, o); amp = a2 > amp ? a2 : amp;
        sumsq += __shfl_xor(sumsq, o);
    }
    const int slots = nt / 64, slot = tid / 64;
    const bool writer = (tid & 63) == 0;
#else
    const int slots = nt, slot = tid;
    const bool writer = true;
#endif
    if (writer) {
        int* redi = reinterpret_cast<int*>(red + 3 * slots);
        red[slot] = best; red[slots + slot] = amp; red[2 * slots + slot] = sumsq; redi[slot] = bidx;
    }
}

// phase 3: thread 0 folds the per-wavefront partials
ECHO_HD void ph_reduce(int tid, int nt, const BandParams& p, unsigned char* smem) {
    if (!p.do_corr || tid != 0) return;
    const int slots = red_slots(nt);
    const double* red = lds_d(smem, p.off_u);
    const int* redi = reinterpret_cast<const int*>(red + 3 * slots);
    Misc* m = reinterpret_cast<Misc*>(smem + p.off_misc);
    double best = red[0], amp = red[slots], ss = red[2 * slots];
    int bidx = redi[0];
    for (int j = 1; j < slots; ++j) {
        combine(best, bidx, red[j], redi[j]);
        amp = red[slots + j] > amp ? red[slots + j] : amp;
        ss += red[2 * slots + j];
    }
    m->best = best; m->bidx = bidx; m->amp = amp; m->sumsq = ss;
}

// phase 4: median of the noise window (np.median: mean of the two middle order statistics) by counting.
// Element v is the k-th order statistic iff #(u < v) <= k < #(u < v) + #(u == v).  Counting runs on the float-rounded
// copy (rounding is monotonic, so a strict float inequality decides the float64 one); an element whose float value
// is shared with another is recounted in float64 when its float rank interval contains a wanted rank.  The count
// loop is split over nt / n thread groups whose partial counts meet in LDS (after the reduction scratch).
ECHO_HD int rank_parts(int nt, int n) { int parts = n > 0 ? nt / n : 1; return parts < 1 ? 1 : (parts > 4 ? 4 : parts); }

ECHO_HD void ph_rank_count(int tid, int nt, const BandParams& p, unsigned char* smem) {
    if (!p.do_corr || p.detector != 0) return;
    const int n = p.noise_n;
    if (n <= 0) return;
    const float* nwf = reinterpret_cast<const float*>(lds_d(smem, p.off_nw) + n);
    int* cnt = reinterpret_cast<int*>(lds_d(smem, p.off_u) + 4 * red_slots(nt));   // [parts][n][2]
    const int parts = rank_parts(nt, n);
    const int chunk = (n + parts - 1) / parts;
    for (int t = tid; t < parts * n; t += nt) {
        const int i = t % n, part = t / n;
        const int j0 = part * chunk, j1 = (j0 + chunk < n) ? j0 + chunk : n;
        const float v = nwf[i];
        int less = 0, eq = 0;
#pragma unroll 8
        for (int j = j0; j < j1; ++j) {
            const float u = nwf[j];
            less += (u < v) ? 1 : 0;
            eq += (u == v) ? 1 : 0;
        }
        cnt[(part * n + i) * 2] = less; cnt[(part * n + i) * 2 + 1] = eq;
    }
}

ECHO_HD void ph_rank_pick(int tid, int nt, const BandParams& p, unsigned char* smem) {
    if (!p.do_corr || p.detector != 0) return;
    const double* nw = lds_d(smem, p.off_nw);
    const int* cnt = reinterpret_cast<const int*>(lds_d(smem, p.off_u) + 4 * red_slots(nt));
    Misc* m = reinterpret_cast<Misc*>(smem + p.off_misc);
    const int n = p.noise_n;
    if (n <= 0) { if (tid == 0) { m->med0 = 0.0; m->med1 = 0.0; } return; }
    const int parts = rank_parts(nt, n);
    const int k0 = (n - 1) / 2, k1 = n / 2;
    for (int i = tid; i < n; i += nt) {
        int less = 0, eq = 0;
        for (int part = 0; part < parts; ++part) { less += cnt[(part * n + i) * 2]; eq += cnt[(part * n + i) * 2 + 1]; }
        if (k1 < less || k0 >= less + eq) continue;          // neither wanted rank inside this element's interval
        const double v = nw[i];
        if (eq > 1) {                                         // float tie: exact recount
            less = 0; eq = 0;
            for (int j = 0; j < n; ++j) { const double u = nw[j]; less += (u < v) ? 1 : 0; eq += (u == v) ? 1 : 0; }
        }
        if (less <= k0 && k0 < less + eq) m->med0 = v;
        if (less <= k1 && k1 < less + eq) m->med1 = v;
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
ECHO_HD float2 mk2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
ECHO_HD float2 cmul(float2 a, float2 b) { return mk2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
ECHO_HD float2 cmulc(float2 a, float2 b) { return mk2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a*conj(b)
ECHO_HD float2 cadd(float2 a, float2 b) { return mk2(a.x + b.x, a.y + b.y); }
ECHO_HD float2 csub(float2 a, float2 b) { return mk2(a.x - b.x, a.y - b.y); }
template <bool INV> ECHO_HD float2 mul_mi(float2 a) {   // forward: a * (-i); inverse: a * (+i)
    return INV ? mk2(-a.y, a.x) : mk2(a.y, -a.x);
}

template <bool INV> ECHO_HD void dft4(float2& x0, float2& x1, float2& x2, float2& x3) {
    const float2 t0 = cadd(x0, x2), t1 = csub(x0, x2), t2 = cadd(x1, x3), t3 = mul_mi<INV>(csub(x1, x3));
    x0 = cadd(t0, t2); x2 = csub(t0, t2); x1 = cadd(t1, t3); x3 = csub(t1, t3);
}

template <bool INV> ECHO_HD void dft8(float2 (&v)[8]) {
    const float h = 0.70710678118654752440f;
    float2 a0 = cadd(v[0], v[4]), a1 = cadd(v[1], v[5]), a2 = cadd(v[2], v[6]), a3 = cadd(v[3], v[7]);
    float2 b0 = csub(v[0], v[4]), b1 = csub(v[1], v[5]), b2 = csub(v[2], v[6]), b3 = csub(v[3], v[7]);
    if (!INV) { b1 = mk2(h * (b1.x + b1.y), h * (b1.y - b1.x));
                b2 = mul_mi<false>(b2);
                b3 = mk2(h * (b3.y - b3.x), -h * (b3.x + b3.y)); }
    else      { b1 = mk2(h * (b1.x - b1.y), h * (b1.y + b1.x));
                b2 = mul_mi<true>(b2);
                b3 = mk2(-h * (b3.x + b3.y), h * (b3.x - b3.y)); }
    dft4<INV>(a0, a1, a2, a3);
    dft4<INV>(b0, b1, b2, b3);
    v[0] = a0; v[2] = a1; v[4] = a2; v[6] = a3;
    v[1] = b0; v[3] = b1; v[5] = b2; v[7] = b3;
}

// LDS position of FFT element i.  Bits 1..4 of the index are XORed with bits of i>>5 so that every access pattern
// of the transform is bank-conflict free: 32 consecutive elements (strided passes), 16 elements of two adjacent
// 128-blocks (q = 16 pass) and 16 contiguous elements per lane (middle pass).  Bit 0 is untouched, so an even
// element and its successor stay adjacent and can move as one 16-byte access.
ECHO_HD int fa(int i) {
    const int h = i >> 5;
    return i ^ (((h & 7) << 1) | (((h >> 2) & 1) << 4));
}
ECHO_HD float4 ld4(const float2* f, int e) { return *reinterpret_cast<const float4*>(f + e); }
ECHO_HD void st4(float2* f, int e, float2 a, float2 b) {
    float4 v; v.x = a.x; v.y = a.y; v.z = b.x; v.w = b.y; *reinterpret_cast<float4*>(f + e) = v;
}

// Bluestein input: a[n] = (x[2n] + i x[2n+1]) * w[n], zero-padded to kFftM; two elements per lane, x read
// straight from global memory (float32 copy written by the band kernel, or float64 band_sig)
ECHO_HD void ph_spec_load(int tid, int nt, int blk, const BandParams& p, unsigned char* smem) {
    float2* f = lds_c(smem, p.off_fft);
    for (int n = 2 * tid; n < kFftM; n += 2 * nt) {
        float2 a0 = mk2(0.f, 0.f), a1 = mk2(0.f, 0.f);
        if (n < p.n2) {
            const float4 w = ld4(p.bw, n);
            const bool two = n + 1 < p.n2;
            float2 x0, x1 = mk2(0.f, 0.f);
            if (p.specf_in) {
                const float2* x = reinterpret_cast<const float2*>(p.specf_in + (size_t)blk * p.N) + n;
                x0 = x[0]; if (two) x1 = x[1];
            } else {
                const double* x = p.specd_in + (size_t)blk * p.N + 2 * n;
                x0 = mk2((float)x[0], (float)x[1]); if (two) x1 = mk2((float)x[2], (float)x[3]);
            }
            a0 = cmul(x0, mk2(w.x, w.y));
            if (two) a1 = cmul(x1, mk2(w.z, w.w));
        }
        st4(f, fa(n), a0, a1);
    }
}

// Radix-8 butterfly with twiddles W^m, m = 1..7, built from W, W^2, W^4.
// Forward (decimation in frequency): butterfly then twiddle; inverse (its conjugate transpose): conjugate
// twiddle then butterfly.
template <bool INV> ECHO_HD void bfly8(float2 (&v)[8], float2 w1, float2 w2, float2 w4) {
    const float2 w3 = cmul(w1, w2), w5 = cmul(w1, w4), w6 = cmul(w2, w4), w7 = cmul(w3, w4);
    if (INV) {
        v[1] = cmulc(v[1], w1); v[2] = cmulc(v[2], w2); v[3] = cmulc(v[3], w3); v[4] = cmulc(v[4], w4);
        v[5] = cmulc(v[5], w5); v[6] = cmulc(v[6], w6); v[7] = cmulc(v[7], w7);
    }
    dft8<INV>(v);
    if (!INV) {
        v[1] = cmul(v[1], w1); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], w3); v[4] = cmul(v[4], w4);
        v[5] = cmul(v[5], w5); v[6] = cmul(v[6], w6); v[7] = cmul(v[7], w7);
    }
}

ECHO_HD int tw_offset(int q) { return q == 1024 ? 0 : (q == 128 ? 3 * 1024 : 3 * 1024 + 3 * 128); }

// Twiddles of one radix-8 pass for the two butterflies a thread owns.  They do not depend on the data, so the
// kernel loads them one phase ahead of their use and the global-load latency hides behind the barrier.
struct Tw3 { float4 w1, w2, w4; };
ECHO_HD Tw3 fft_tw_load(int tid, int q, const BandParams& p) {
    const float2* t1 = p.tw + tw_offset(q);
    const int i = 2 * (tid % (q >> 1));
    Tw3 t; t.w1 = ld4(t1, i); t.w2 = ld4(t1 + q, i); t.w4 = ld4(t1 + 2 * q, i);
    return t;
}

// One radix-8 pass over sub-FFTs of size 8q; each of the kFftM/16 threads owns butterflies i and i+1 of one
// sub-FFT, so every LDS access moves two adjacent elements (16 bytes).
template <bool INV> ECHO_HD void ph_fft_r8(int tid, int q, const BandParams& p, unsigned char* smem, const Tw3& t) {
    if (tid >= kFftM / 16) return;
    float2* f = lds_c(smem, p.off_fft);
    const int hq = q >> 1;
    const int i = 2 * (tid % hq), base = (tid / hq) * 8 * q + i;
    float2 va[8], vb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 x = ld4(f, fa(base + j * q));
        va[j] = mk2(x.x, x.y); vb[j] = mk2(x.z, x.w);
    }
    bfly8<INV>(va, mk2(t.w1.x, t.w1.y), mk2(t.w2.x, t.w2.y), mk2(t.w4.x, t.w4.y));
    bfly8<INV>(vb, mk2(t.w1.z, t.w1.w), mk2(t.w2.z, t.w2.w), mk2(t.w4.z, t.w4.w));
#pragma unroll
    for (int j = 0; j < 8; ++j) st4(f, fa(base + j * q), va[j], vb[j]);
}

// exp(-2*pi*i*k/16) for the k = i*m products of the in-register radix-4 stage
ECHO_HD float2 w16(int k) {
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    switch (k) {
        case 1: return mk2(c1, -s1);
        case 2: return mk2(h, -h);
        case 3: return mk2(s1, -c1);
        case 4: return mk2(0.f, -1.f);
        case 6: return mk2(-h, -h);
        case 9: return mk2(-c1, s1);
        default: return mk2(1.f, 0.f);
    }
}

// Middle of the transform, per contiguous 16 elements, all in registers:
// forward radix-4 (q=4) and radix-4 (q=1), multiply by the chirp-filter spectrum, inverse radix-4 (q=1) and (q=4).
struct Bh8 { float4 v[8]; };
ECHO_HD Bh8 fft_bh_load(int tid, const BandParams& p) {
    Bh8 b;
    const int nu = kFftM / 16, u = tid < nu ? tid : 0;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) b.v[jj] = ld4(p.bhat, 2 * (jj * nu + u));
    return b;
}

ECHO_HD void ph_fft_mid(int tid, const BandParams& p, unsigned char* smem, const Bh8& bh8) {
    if (tid >= kFftM / 16) return;
    float2* f = lds_c(smem, p.off_fft);
    {
        const int u = tid;
        const int e0 = 16 * u, hi = e0 & ~31, lo = e0 & 31, g = fa(e0) ^ e0;
        float2 v[16];
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const float4 x = ld4(f, hi | ((lo + 2 * jj) ^ g));
            v[2 * jj] = mk2(x.x, x.y); v[2 * jj + 1] = mk2(x.z, x.w);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {          // forward, q = 4: (i, i+4, i+8, i+12), twiddle W16^(i*m)
            dft4<false>(v[i], v[i + 4], v[i + 8], v[i + 12]);
#pragma unroll
            for (int m = 1; m < 4; ++m)
                if (i) v[i + 4 * m] = cmul(v[i + 4 * m], w16(i * m));
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) dft4<false>(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]);   // q = 1
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {       // pointwise product with the chirp-filter spectrum
            const float4 bh = bh8.v[jj];
            v[2 * jj] = cmul(v[2 * jj], mk2(bh.x, bh.y));
            v[2 * jj + 1] = cmul(v[2 * jj + 1], mk2(bh.z, bh.w));
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) dft4<true>(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]);    // inverse q = 1
#pragma unroll
        for (int i = 0; i < 4; ++i) {          // inverse, q = 4
#pragma unroll
            for (int m = 1; m < 4; ++m)
                if (i) v[i + 4 * m] = cmulc(v[i + 4 * m], w16(i * m));
            dft4<true>(v[i], v[i + 4], v[i + 8], v[i + 12]);
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) st4(f, hi | ((lo + 2 * jj) ^ g), v[2 * jj], v[2 * jj + 1]);
    }
}

// X[k] = c[k] * A[k] + conj(c[n2-k]) * B[k]: the Bluestein post-multiply and the real-FFT unpacking folded into
// one table (Z[k] = c[k] w[k]; X[k] = (Z[k] + conj Z[n2-k])/2 - i u_k (Z[k] - conj Z[n2-k])/2); magnitude (dB)
ECHO_HD void ph_spec_store(int tid, int nt, int blk, const BandParams& p, unsigned char* smem) {
    const float2* f = lds_c(smem, p.off_fft);
    for (int k = tid; k < p.nbins; k += nt) {
        const int ka = (k == p.n2) ? 0 : k;
        const int kb = (k == 0 || k == p.n2) ? 0 : p.n2 - k;
        const float4 ab = ld4(p.post, 2 * k);
        const float2 ca = f[fa(ka)];
        float2 cb = f[fa(kb)]; cb.y = -cb.y;
        const float2 x = cadd(cmul(ca, mk2(ab.x, ab.y)), cmul(cb, mk2(ab.z, ab.w)));
        float mag = sqrtf(x.x * x.x + x.y * x.y);
        if (p.spec_db) mag = 20.0f * log10f(fmaxf(mag, 1e-12f));
        p.spec_out[(size_t)blk * p.nbins + k] = mag;
    }
}

}  // namespace echo
