// One workgroup per (frame, band): FIR -> correlation + peak statistics -> Bluestein rfft -> dB spectrum,
// with every intermediate resident in LDS.  The algorithm is in band_phases.h (shared with the host
// emulation used by the unit tests); this file only sequences the phases between workgroup barriers.
#include <hip/hip_runtime.h>

#include "echo_internal.h"

namespace echo {

// Diagnostic build only (-DECHO_DIAG_STAMPS, `make diag`): thread 0 of the first blocks records the shader clock
// after every phase into a buffer of its own.  The production library contains none of this.
#ifdef ECHO_DIAG_STAMPS
constexpr int kStampBlocks = 4096, kStampSlots = 24;
__device__ unsigned long long g_stamps[kStampBlocks * kStampSlots];
#define STAMP(k) do { if (tid == 0 && blk < kStampBlocks) g_stamps[blk * kStampSlots + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

__global__ __launch_bounds__(kNT) void band_kernel(BandParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
    if (p.skip_gate && p.skip_gate[(size_t)blk * p.skip_stride] >= 0) return;      // uniform over the workgroup
    STAMP(0);
    ph_load(tid, kNT, blk, p, smem);
    __syncthreads(); STAMP(1);
    double fir_acc[kR];
    ph_fir(tid, kNT, blk, p, smem, fir_acc);
    __syncthreads();                          // every lane has read its FIR inputs: sig may overwrite ybuf
    ph_fir_store(tid, kNT, blk, p, smem, fir_acc);
    __syncthreads(); STAMP(2);
    if (p.do_corr) {
        ph_corr(tid, kNT, blk, p, smem);
        __syncthreads(); STAMP(3);
        ph_reduce(tid, kNT, p, smem);
        ph_rank_count(tid, kNT, p, smem);     // reads the noise window, writes counts beyond the reduction scratch
        __syncthreads(); STAMP(4);
        ph_rank_pick(tid, kNT, p, smem);
        __syncthreads(); STAMP(5);
        ph_finish(tid, kNT, blk, p, smem);
        STAMP(6);
    }
}

// Spectrum: one workgroup per (frame, band); Bluestein rfft of the band signal in a 64-KB LDS buffer, two
// workgroups per CU.  Data-independent table loads are issued one phase ahead of their use, in front of a barrier.
__global__ __launch_bounds__(kNT, 4) void spec_kernel(BandParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
    STAMP(7);
    Tw3 tw = fft_tw_load(tid, 1024, p);
    ph_spec_load(tid, kNT, blk, p, smem);
    __syncthreads(); STAMP(8);
    ph_fft_r8<false>(tid, 1024, p, smem, tw);
    tw = fft_tw_load(tid, 128, p);
    __syncthreads(); STAMP(9);
    ph_fft_r8<false>(tid, 128, p, smem, tw);
    tw = fft_tw_load(tid, 16, p);
    __syncthreads(); STAMP(10);
    ph_fft_r8<false>(tid, 16, p, smem, tw);
    __syncthreads(); STAMP(11);
    {
        const Bh8 bh = fft_bh_load(tid, p);
        ph_fft_mid(tid, p, smem, bh);
    }
    tw = fft_tw_load(tid, 16, p);
    __syncthreads(); STAMP(12);
    ph_fft_r8<true>(tid, 16, p, smem, tw);
    tw = fft_tw_load(tid, 128, p);
    __syncthreads(); STAMP(13);
    ph_fft_r8<true>(tid, 128, p, smem, tw);
    tw = fft_tw_load(tid, 1024, p);
    __syncthreads(); STAMP(14);
    ph_fft_r8<true>(tid, 1024, p, smem, tw);
    __syncthreads(); STAMP(15);
    ph_spec_store(tid, kNT, blk, p, smem);
    STAMP(16);
}

// The 8192-point cyclic convolution of the Bluestein transform, in place in LDS (entered after a barrier, leaves
// after one): forward passes, register-resident middle with the chirp-filter product, inverse passes.
__device__ __forceinline__ void bluestein_round(int tid, const BandParams& p, unsigned char* smem) {
    Tw3 tw = fft_tw_load(tid, 1024, p);
    ph_fft_r8<false>(tid, 1024, p, smem, tw);
    tw = fft_tw_load(tid, 128, p);
    __syncthreads();
    ph_fft_r8<false>(tid, 128, p, smem, tw);
    tw = fft_tw_load(tid, 16, p);
    __syncthreads();
    ph_fft_r8<false>(tid, 16, p, smem, tw);
    __syncthreads();
    {
        const Bh8 bh = fft_bh_load(tid, p);
        ph_fft_mid(tid, p, smem, bh);
    }
    tw = fft_tw_load(tid, 16, p);
    __syncthreads();
    ph_fft_r8<true>(tid, 16, p, smem, tw);
    tw = fft_tw_load(tid, 128, p);
    __syncthreads();
    ph_fft_r8<true>(tid, 128, p, smem, tw);
    tw = fft_tw_load(tid, 1024, p);
    __syncthreads();
    ph_fft_r8<true>(tid, 1024, p, smem, tw);
    __syncthreads();
}

// The same with the 4608-point cyclic convolution (fft9_phases.h), 576 threads.
__device__ __forceinline__ void bluestein9_round(int tid, const BandParams& p, unsigned char* smem) {
    ph9_r9<false>(tid, p, smem);
    __syncthreads();
    ph9_r8<false>(tid, 64, p, smem);
    __syncthreads();
    ph9_r8<false>(tid, 8, p, smem);
    __syncthreads();
    ph9_mid(tid, p, smem);
    __syncthreads();
    ph9_r8<true>(tid, 8, p, smem);
    __syncthreads();
    ph9_r8<true>(tid, 64, p, smem);
    __syncthreads();
    ph9_r9<true>(tid, p, smem);
    __syncthreads();
}


// ---------------------------------------------------------------------------------------------------------------
// Fast path of the fused chain (named mode, median detector): two kernels.
//
// band2_kernel, one 576-thread workgroup per (pair of frames, band), float32: FIR -> band_sig of both frames as the
//   real and imaginary part of one complex signal (correlation with a real chirp is real-linear, so one complex
//   4608-point cyclic convolution yields both correlations, every lag) -> float32 maxima (signed over
//   pos[min_delay:], absolute over every lag outside the blind zone) and the list of lags whose float32 value lies
//   within 2 eps of them, eps being a bound on the float32 path's error.
// exact_kernel, one wavefront per (frame, band), float64: the noise window and every listed lag, exactly: FIR outputs
//   from the float64 y, then the chirp dot products, both as sequential fused multiply-add sums (the operation order
//   of the single-frame kernel).  The true extrema are among the re-evaluated lags, so peak index, SNR and amplitude
//   are those of an all-float64 evaluation and do not depend on the partner frame.
// Reference: airSonar_optimized.py:193-194 (FIR), :290-303 (correlate), :200-219 (first_strong_peak), :666 (amplitude).

// Block reduction of four values per thread (op 1: max, op 0: sum); every thread gets the four results.
// `area` alternates between calls so that one barrier per call suffices.
__device__ __forceinline__ void b2_reduce4(double (&v)[4], const int (&ops)[4], double* red, int area, int tid) {
    constexpr int kWaves = kF9NT / 64;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = wave_fold(ops[q] ? 1 : 0, v[q]);     // DPP moves and lane reads, no LDS round trips
    double* r = red + area * kWaves * 4;
    if ((tid & 63) == 0)
        for (int q = 0; q < 4; ++q) r[(tid >> 6) * 4 + q] = v[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double a = r[q];
        for (int w = 1; w < kWaves; ++w) { const double x = r[w * 4 + q]; a = ops[q] ? (x > a ? x : a) : a + x; }
        v[q] = a;
    }
}

// An opaque copy of a per-lane value: index arithmetic derived from it cannot be hoisted above this point, which keeps
// the addresses of a later phase out of the registers of an earlier one.
#define ECHO_OPAQUE(x) asm volatile("" : "+v"(x))

// Packed float32 arithmetic over the two frames of a pair: element x of a v2f belongs to the first frame, y to the
// second (v_pk_fma_f32: two multiply-adds per instruction).
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2f lds_read_v2f(const float2* q) {
    typedef const volatile __attribute__((address_space(3))) unsigned long long* lds_u64_t;
    const unsigned long long u = *(lds_u64_t)q;              // one ds_read_b64, never merged with its neighbour
    return __builtin_bit_cast(v2f, u);
}

// acc[r] = sum_j in[r + j] * c[j] for both frames at once, j < ngroups * kR in ascending order (the order of the
// single-frame kernel's float32 sums).  in: interleaved (frame A, frame B) samples in LDS, lane stride kR elements of
// 8 bytes (conflict-free); c: the (c, c) pairs in device memory, indexed uniformly (scalar loads).
__device__ __forceinline__ void sliding_dot_pk(const float2* in, const float2* __restrict__ c, int ngroups, v2f (&acc)[kR]) {
    v2f w[kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) { w[r] = lds_read_v2f(in + r); acc[r] = v2f{0.f, 0.f}; }
#pragma unroll 1
    for (int g = 0; g < ngroups; ++g) {
        const float2* nx = in + (g + 1) * kR;
        v2f cg[kR];
#pragma unroll
        for (int t = 0; t < kR; ++t) { const float2 q = c[g * kR + t]; cg[t] = v2f{q.x, q.y}; }
#pragma unroll
        for (int t = 0; t < kR; ++t) {
#pragma unroll
            for (int r = 0; r < kR; ++r) acc[r] = __builtin_elementwise_fma(w[(r + t) % kR], cg[t], acc[r]);
            w[t] = lds_read_v2f(nx + t);
        }
    }
}

__global__ __launch_bounds__(kF9NT, 5) void band2_kernel(BandParams p, int B) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int tid = threadIdx.x;
    const int blk = blockIdx.x;
    (void)blk;
    const int band = blockIdx.x % p.nb, pair = blockIdx.x / p.nb;
    STAMP(0);
    const bool have1 = 2 * pair + 1 < B;
    float2* fft = reinterpret_cast<float2*>(smem);
    float2* ybuf = reinterpret_cast<float2*>(smem);                   // both frames' FIR inputs, interleaved, inside the transform buffer
    double* red = reinterpret_cast<double*>(smem + p.b2_off_red);
    int* ncand = reinterpret_cast<int*>(smem + p.b2_off_misc);        // [2]
    const int kstart = p.half + p.min_delay;
    if (tid < 2) ncand[tid] = 0;
    // ---- y of both frames, rounded to float32, into LDS: sample n of (frame A, frame B) at ybuf[n + ypad]
    double ymax = 0.0;
    {
        const double* y0 = p.d_in + ((size_t)(2 * pair) * p.nb + band) * p.N;
        const double* y1 = y0 + (size_t)p.nb * p.N;
        constexpr int kBatch = 8;                            // ybuf_len <= 8 * 576 for the frame lengths this kernel takes: one round trip
        for (int i0 = tid; i0 < p.ybuf_len; i0 += kBatch * kF9NT) {
            double t0[kBatch], t1[kBatch];
#pragma unroll
            for (int e = 0; e < kBatch; ++e) {
                const int i = i0 + e * kF9NT, n = i - p.ypad;
                const bool in = i < p.ybuf_len && n >= 0 && n < p.N;
                t0[e] = in ? y0[n] : 0.0;
                t1[e] = (in && have1) ? y1[n] : 0.0;
            }
#pragma unroll
            for (int e = 0; e < kBatch; ++e)
                if (i0 + e * kF9NT < p.ybuf_len) {
                    ybuf[i0 + e * kF9NT] = mk2((float)t0[e], (float)t1[e]);
                    const double a0 = fabs(t0[e]), a1 = fabs(t1[e]);
                    ymax = a0 > ymax ? a0 : ymax; ymax = a1 > ymax ? a1 : ymax;
                }
        }
    }
    __syncthreads();
    STAMP(1);
    // ---- float32 FIR of both frames: acc[r] = (band_sig_A[n0 + r], band_sig_B[n0 + r])
    v2f acc[kR];
    {
        const int n0 = tid * kR;
        if (n0 < p.N) sliding_dot_pk(ybuf + n0, p.taps32x2 + (size_t)band * ((p.kgroups + 1) * kR), p.kgroups, acc);
        else {
#pragma unroll
            for (int r = 0; r < kR; ++r) acc[r] = v2f{0.f, 0.f};
        }
    }
    __syncthreads();                                       // every lane has read its inputs: the buffer may be overwritten
    ECHO_OPAQUE(tid);
    STAMP(2);
    // ---- x[n] = band_sig_A[n] + i band_sig_B[n] into the transform buffer
    double ss = 0.0;
    {
        const int n0 = tid * kR;
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int n = n0 + r;
            if (n >= kF9M) continue;
            const bool in = n < p.N;
            const float a = in ? acc[r].x : 0.f, b = in ? acc[r].y : 0.f;
            fft[pos9(n)] = mk2(a, b);
            ss += (double)a * (double)a + (double)b * (double)b;
        }
    }
    double eps;
    {
        double r4[4] = {ymax, ss, 0.0, 0.0};
        const int ops[4] = {1, 0, 0, 0};
        b2_reduce4(r4, ops, red, 0, tid);                  // its barrier also publishes the transform input
        eps = p.eps_fir[band] * r4[0] + p.eps_fft[band] * sqrt(r4[1]);
    }
    ECHO_OPAQUE(tid);
    // float32 band signals to the spectrum kernel, read back from the buffer so that consecutive lanes store
    // consecutive samples
    if (p.bandf_out) {
        float* o0 = p.bandf_out + ((size_t)(2 * pair) * p.nb + band) * p.N;
        float* o1 = o0 + (size_t)p.nb * p.N;
        for (int n = tid; n < p.N; n += kF9NT) {
            const float2 x = fft[pos9(n)];
            o0[n] = x.x;
            if (have1) o1[n] = x.y;
        }
        __syncthreads();                                   // the transform overwrites the buffer in place
    }
    // ---- cyclic convolution with the reversed chirp: corr_A + i corr_B, every lag; the last inverse pass leaves its
    // outputs (lags tid + 512 j) in registers, where the maxima are taken
    ECHO_OPAQUE(tid);
    STAMP(3);
    float2 cv[9];
    {
        BandParams q = p;
        q.off_fft = 0; q.f9bhat = p.f9ghat + (size_t)band * kF9M;
        ph9_r9<false>(tid, q, smem);
        __syncthreads();
        ph9_r8<false>(tid, 64, q, smem);
        __syncthreads();
        ph9_r8<false>(tid, 8, q, smem);
        __syncthreads();
        ph9_mid(tid, q, smem);
        __syncthreads();
        ph9_r8<true>(tid, 8, q, smem);
        __syncthreads();
        ph9_r8<true>(tid, 64, q, smem);
        __syncthreads();
        ph9_r9_inv_regs(tid, q, smem, cv);
    }
    STAMP(4);
    // ---- float32 maxima, then the lags that can hold the true ones
    float mx[2] = {-INFINITY, -INFINITY}, am[2] = {0.f, 0.f};
    if (tid < 512) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int k = tid + 512 * j, qd = k - p.half;
            if (k >= p.L || (qd >= 0 && qd < p.min_delay)) continue;     // beyond the last lag; blind zone: zero by definition
            am[0] = fmaxf(am[0], fabsf(cv[j].x)); am[1] = fmaxf(am[1], fabsf(cv[j].y));
            if (k >= kstart) { mx[0] = fmaxf(mx[0], cv[j].x); mx[1] = fmaxf(mx[1], cv[j].y); }
        }
    }
    {
        double r4[4] = {(double)mx[0], (double)mx[1], (double)am[0], (double)am[1]};
        const int ops[4] = {1, 1, 1, 1};
        b2_reduce4(r4, ops, red, 1, tid);
        // thresholds in float32, rounded down
        const double e2 = 2.0 * eps;
        mx[0] = __double2float_rd(r4[0] - e2); mx[1] = __double2float_rd(r4[1] - e2);
        am[0] = __double2float_rd(r4[2] - e2); am[1] = __double2float_rd(r4[3] - e2);
        if (tid < 2 && (tid == 0 || have1)) {
            B2Det* d = p.det_out + ((size_t)(2 * pair + tid) * p.nb + band);
            d->eps = eps; d->max32 = (float)r4[tid]; d->amp32 = (float)r4[2 + tid];
        }
    }
    const int nz_lo = kstart, nz_hi = kstart + p.noise_n;   // evaluated exactly anyway
    // A sample that is not finite (in either frame of the pair) spreads through the transform into every lag of both
    // frames, and the bound is then NaN or infinite: nothing is known about either frame, and both get the overflowing
    // list that makes the exact kernel evaluate every lag from the frame's own float64 signal -- a frame's record never
    // depends on what its partner holds.
    const bool unusable = !(eps < (double)INFINITY);
    if (unusable) {
        if (tid < 2) ncand[tid] = kB2MaxCand + 1;
    } else if (tid < 512) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int k = tid + 512 * j, qd = k - p.half;
            if (k >= p.L || (qd >= 0 && qd < p.min_delay)) continue;
            if (k >= nz_lo && k < nz_hi) continue;
            const float2 c = cv[j];
            const bool c0 = (k >= kstart && c.x >= mx[0]) || fabsf(c.x) >= am[0];
            const bool c1 = have1 && ((k >= kstart && c.y >= mx[1]) || fabsf(c.y) >= am[1]);
            if (c0) { const int slot = atomicAdd(&ncand[0], 1); if (slot < kB2MaxCand) p.det_out[(size_t)(2 * pair) * p.nb + band].cand[slot] = k; }
            if (c1) { const int slot = atomicAdd(&ncand[1], 1); if (slot < kB2MaxCand) p.det_out[(size_t)(2 * pair + 1) * p.nb + band].cand[slot] = k; }
        }
    }
    __syncthreads();
    if (tid < 2 && (tid == 0 || have1)) p.det_out[(size_t)(2 * pair + tid) * p.nb + band].ncand = ncand[tid];
    STAMP(5);
}

// ---- exact_kernel: one wavefront per (frame, band), four independent wavefronts per workgroup, each with its own
// LDS windows; a wavefront synchronises with itself only (its LDS operations execute in program order).
constexpr int kExWaves = 4, kExNT = 64 * kExWaves;
struct ExView { const double *gt64, *gc64; double *yw, *bw, *cv; };

__device__ __forceinline__ void ex_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// The wavefront exchanges of the exact kernel.  Device pass only: the host pass of this file never runs a kernel body and
// gets declarations that return their arguments.
#if defined(__HIP_DEVICE_COMPILE__)
// Value of lane ^ O: a DPP move where one control expresses the exchange (inside a quad, across the halves of a row of
// 16 lanes), a shuffle through the LDS crossbar otherwise.  O must be a compile-time constant after unrolling.
__device__ __forceinline__ double ex_xor(double v, int o) {
    switch (o) {
        case 1: return wave_fold_dpp<0xB1>(v);               // quad_perm [1,0,3,2]
        case 2: return wave_fold_dpp<0x4E>(v);               // quad_perm [2,3,0,1]
        case 8: return wave_fold_dpp<0x128>(v);              // row_ror:8
        default: return __shfl_xor(v, o);
    }
}
template <int CTRL> __device__ __forceinline__ int ex_dpp_int(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, false); }
// Fold of an int over the wavefront, every lane gets the result (MAX: maximum, else minimum): four DPP exchanges inside
// the rows of 16 lanes, then the four rows' results as scalar lane reads.
template <bool MAX> __device__ __forceinline__ int ex_fold_int(int v) {
    int t;
    t = ex_dpp_int<0xB1>(v); v = MAX ? (t > v ? t : v) : (t < v ? t : v);
    t = ex_dpp_int<0x4E>(v); v = MAX ? (t > v ? t : v) : (t < v ? t : v);
    t = ex_dpp_int<0x141>(v); v = MAX ? (t > v ? t : v) : (t < v ? t : v);
    t = ex_dpp_int<0x140>(v); v = MAX ? (t > v ? t : v) : (t < v ? t : v);
    const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16),
              r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    const int a = MAX ? (r1 > r0 ? r1 : r0) : (r1 < r0 ? r1 : r0), b = MAX ? (r3 > r2 ? r3 : r2) : (r3 < r2 ? r3 : r2);
    return MAX ? (b > a ? b : a) : (b < a ? b : a);
}
// The detector partials of a wavefront folded into every lane: signed maximum with the lower index among equals
// (a total order, so the result does not depend on the order of the exchanges) and max|corr|.
template <int CTRL> __device__ __forceinline__ void ex_fold_step(double& b, int& bi, double& am) {
    const double b2 = wave_fold_dpp<CTRL>(b); const int i2 = ex_dpp_int<CTRL>(bi);
    combine(b, bi, b2, i2);
    const double a2 = wave_fold_dpp<CTRL>(am); am = a2 > am ? a2 : am;
}
__device__ __forceinline__ void ex_fold_det(double& b, int& bi, double& am) {
    ex_fold_step<0xB1>(b, bi, am); ex_fold_step<0x4E>(b, bi, am); ex_fold_step<0x141>(b, bi, am); ex_fold_step<0x140>(b, bi, am);
    double rb = wave_fold_lane(b, 0), ra = wave_fold_lane(am, 0);
    int ri = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
    for (int row = 16; row < 64; row += 16) {
        combine(rb, ri, wave_fold_lane(b, row), __builtin_amdgcn_readlane(bi, row));
        const double a2 = wave_fold_lane(am, row); ra = a2 > ra ? a2 : ra;
    }
    b = rb; bi = ri; am = ra;
}
#else
__device__ inline double ex_xor(double v, int) { return v; }
template <bool MAX> __device__ inline int ex_fold_int(int v) { return v; }
__device__ inline void ex_fold_det(double&, int&, double&) {}
#endif

// One 8-byte read of LDS that the compiler neither merges with its neighbours nor widens to a flat access.
typedef const volatile __attribute__((address_space(3))) double lds_cv_f64;
__device__ __forceinline__ double lds_read_f64(const double* p) { return *(lds_cv_f64*)p; }

// acc[r] = sum_j x[r + j] * c[j], j < n rounded up to a multiple of T (c is zero beyond n, x finite), as sequential
// fused multiply-add sums in ascending j.  x: LDS window, read in batches of R + T - 1 samples for R * T multiply-adds;
// c: the coefficient table in device memory, indexed uniformly across the wavefront (scalar loads).  The window reads
// are volatile so that they stay single 8-byte reads (ds_read_b64: 256 bytes per clock); merged into pairs
// (ds_read2_b64) the same bytes take twice the LDS cycles.
template <int R, int T> __device__ __forceinline__ void ex_dots(const double* x, const double* __restrict__ c, int n,
                                                                  double (&acc)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
#pragma unroll 1
    for (int j = 0; j < n; j += T) {
        double xv[R + T - 1], cc[T];
#pragma unroll
        for (int t = 0; t < R + T - 1; ++t) xv[t] = lds_read_f64(x + j + t);
#pragma unroll
        for (int t = 0; t < T; ++t) cc[t] = c[j + t];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = fma(xv[r + t], cc[t], acc[r]);
    }
}

// Exact float64 evaluation of one segment of lags [k0, k0 + nl), nl <= kB2Seg, of one (frame, band) by one wavefront:
// stage y, FIR -> band_sig window (zero outside [0, N), as the 'same' crop leaves it), chirp dot products into v.cv
// (which overlays the y window), and the fold into the running signed maximum of pos[min_delay:] (first index on
// ties) and the running max|corr| outside the blind zone.  RF / RC: consecutive FIR outputs / lags per lane (sliding dot products: one LDS read of the signal
// per RF or RC multiply-adds); both are odd, so that the 8-byte reads of consecutive lanes fall into different LDS
// banks; the summation order per output is the same for every RF / RC.
template <int RF, int RC>
__device__ __forceinline__ void exact_segment(const BandParams& p, const ExView& v, const double* y, int k0, int nl,
                                              int lane, double& best, int& bidx, double& amp) {
    const int kstart = p.half + p.min_delay;
    const int nb0 = k0 - (p.M - 1);                          // band_sig index of window element 0
    const int nband = nl + p.M - 1;
    const int nfir = (nband + RF - 1) / RF, ncl = (nl + RC - 1) / RC;
    ex_wave_sync();                                          // the previous segment's reads of cv are done
    for (int i = lane; i < p.b2_ywlen; i += 64) {
        const int n = nb0 - p.ypad + i;
        v.yw[i] = (n >= 0 && n < p.N) ? y[n] : 0.0;
    }
    ex_wave_sync();
    for (int l = lane; l < nfir; l += 64) {
        double a[RF];
        ex_dots<RF, 8>(v.yw + RF * l, v.gt64, p.K, a);
#pragma unroll
        for (int r = 0; r < RF; ++r) { const int n = nb0 + RF * l + r; v.bw[RF * l + r] = (n >= 0 && n < p.N) ? a[r] : 0.0; }
    }
    for (int i = RF * nfir + lane; i < p.b2_bwlen; i += 64) v.bw[i] = 0.0;
    ex_wave_sync();
    for (int l = lane; l < ncl; l += 64) {
        double a[RC];
        ex_dots<RC, 8>(v.bw + RC * l, v.gc64, p.M, a);
#pragma unroll
        for (int r = 0; r < RC; ++r) if (RC * l + r < nl) v.cv[RC * l + r] = a[r];
    }
    ex_wave_sync();
    double b = -INFINITY, am = 0.0; int bi = 0x7fffffff;
    for (int i = lane; i < nl; i += 64) {
        const int k = k0 + i, qd = k - p.half;
        if (qd >= 0 && qd < p.min_delay) continue;           // blind zone: zero by definition
        const double x = v.cv[i], a = fabs(x);
        am = a > am ? a : am;
        if (k >= kstart && x > b) { b = x; bi = qd; }
    }
    ex_fold_det(b, bi, am);
    combine(best, bidx, b, bi);
    amp = am > amp ? am : amp;
}

__global__ __launch_bounds__(kExNT) void exact_kernel(BandParams p, int nfb) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wavefront-uniform: table reads become scalar loads
    const int fb = blockIdx.x * kExWaves + wave;             // (frame, band) of this wavefront
    if (fb >= nfb) return;                                   // wavefronts never wait for each other
    const int band = fb % p.nb;
    ExView v;
    {
        double* d = reinterpret_cast<double*>(smem) + (size_t)wave * (p.b2_ywlen + p.b2_bwlen);
        const int nt_ = (p.kgroups + 1) * kR, nc_ = (p.mgroups + 1) * kR;
        v.gt64 = p.taps_rev + (size_t)band * nt_; v.gc64 = p.chirp + (size_t)band * nc_;
        v.yw = d; v.cv = d; v.bw = d + p.b2_ywlen;
    }
    const double* y = p.d_in + (size_t)fb * p.N;
    // the listed lags, read once: lane i holds entry i
    const B2Det* det = p.det_out + fb;
    const int nc = det->ncand;
    const int mine = (lane < nc && lane < kB2MaxCand) ? det->cand[lane] : 0x7fffffff;
    // pos[0] = 0 in the blind zone takes part in the argmax
    double best = p.min_delay > 0 ? 0.0 : -INFINITY, amp = 0.0; int bidx = p.min_delay > 0 ? 0 : 0x7fffffff;
    // ---- the noise window, exactly
    const int kstart = p.half + p.min_delay, n = p.noise_n;
    exact_segment<9, 5>(p, v, y, kstart, n, lane, best, bidx, amp);
    double x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { const int i = lane * 4 + e; x[e] = i < n ? fabs(v.cv[i]) : INFINITY; }
    // ---- the listed lags (segments that start at a listed lag cover the listed lags behind it); an overflowing list
    // means nothing is known, and every lag is evaluated
    if (nc > kB2MaxCand) {
        for (int k0 = 0; k0 < p.L; k0 += kB2Seg)
            exact_segment<9, 5>(p, v, y, k0, p.L - k0 < kB2Seg ? p.L - k0 : kB2Seg, lane, best, bidx, amp);
    } else {
        int done_below = 0;                                    // lags below it have been evaluated
        for (;;) {
            const int lo = ex_fold_int<false>(mine >= done_below ? mine : 0x7fffffff);
            if (lo == 0x7fffffff) break;
            // how far the listed lags reach from lo inside one segment
            const int hi = ex_fold_int<true>((mine >= lo && mine < lo + kB2Seg) ? mine : lo);
            if (hi - lo >= 16) exact_segment<9, 5>(p, v, y, lo, hi - lo + 1, lane, best, bidx, amp);
            else exact_segment<5, 1>(p, v, y, lo, hi - lo + 1, lane, best, bidx, amp);
            done_below = hi + 1;
        }
    }
    // ---- median of the noise window (np.median: mean of the two middle order statistics): the window, padded with
    // +inf to 256 values, four per lane, sorted by a bitonic network in registers.  Compare and select (the values are
    // magnitudes or +inf: equal values are the same bits, so either choice among equals gives the same sequence); the
    // minimum / maximum instructions would first have to canonicalise each operand.
#pragma unroll
    for (int k = 2; k <= 256; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = lane * 4 + e;
                    const double o = ex_xor(x[e], j >> 2);
                    const bool want_min = ((i & j) == 0) == ((i & k) == 0);
                    x[e] = ((o < x[e]) == want_min) ? o : x[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e & j) continue;
                    const int i = lane * 4 + e;
                    const bool asc = (i & k) == 0;
                    const double a = x[e], b = x[e | j];
                    const bool sw = (b < a) == asc;
                    x[e] = sw ? b : a; x[e | j] = sw ? a : b;
                }
            }
        }
    }
    // ranks (n - 1) / 2 and n / 2 of the sorted sequence
    const int r0 = (n - 1) / 2, r1 = n / 2;
    double m0 = 0.0, m1 = 0.0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double a = __shfl(x[e], r0 >> 2), b = __shfl(x[e], r1 >> 2);
        if ((r0 & 3) == e) m0 = a;
        if ((r1 & 3) == e) m1 = b;
    }
    // ---- SNR, gate, record (first_strong_peak, airSonar_optimized.py:209-219)
    if (lane == 0 && p.res_out) {
        BandRes r;
        r.amplitude = amp; r.pad = 0;
        const double med = (m0 + m1) / 2.0, noise_floor = med * med;
        const double peak_power = best * best;
        double snr = 0.0;
        if (noise_floor > 0.0) snr = 10.0 * log10(peak_power / noise_floor);
        if (snr < p.snr_gate) { r.peak_idx = -1; r.snr_db = 0.0; }
        else                  { r.peak_idx = bidx; r.snr_db = snr; }
        p.res_out[fb] = r;
    }
}


__global__ __launch_bounds__(kF9NT) void spec9_kernel(BandParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
    ph9_load(tid, kF9NT, blk, p, smem);
    __syncthreads();
    bluestein9_round(tid, p, smem);
    ph9_store(tid, kF9NT, blk, p, smem);
}

__global__ __launch_bounds__(kF9NT) void spec9_hilbert_kernel(BandParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
    ph9_load(tid, kF9NT, blk, p, smem);
    __syncthreads();
    bluestein9_round(tid, p, smem);
    ph_hilb_mid<true>(tid, kF9NT, blk, p, smem);
    __syncthreads();
    bluestein9_round(tid, p, smem);
    ph_hilb_env<true>(tid, kF9NT, blk, p, smem);
    __syncthreads();
    ph_hilb_smooth(tid, kF9NT, blk, p, smem);
    __syncthreads();
    ph_hilb_dev(tid, kF9NT, blk, p, smem);
    __syncthreads();
    ph_hilb_finish(tid, kF9NT, blk, p, smem);
}

// Spectrum plus the envelope branch of the band signal (hilb_phases.h): dB spectrum and signal_quality.
__global__ __launch_bounds__(kNT, 4) void spec_hilbert_kernel(BandParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, blk = blockIdx.x;
    ph_spec_load(tid, kNT, blk, p, smem);
    __syncthreads();
    bluestein_round(tid, p, smem);
    ph_hilb_mid<false>(tid, kNT, blk, p, smem);
    __syncthreads();
    bluestein_round(tid, p, smem);
    ph_hilb_env<false>(tid, kNT, blk, p, smem);
    __syncthreads();
    ph_hilb_smooth(tid, kNT, blk, p, smem);
    __syncthreads();
    ph_hilb_dev(tid, kNT, blk, p, smem);
    __syncthreads();
    ph_hilb_finish(tid, kNT, blk, p, smem);
}

#ifdef ECHO_DIAG_STAMPS
extern "C" int echo_diag_read_stamps(unsigned long long* host, int n) {
    if (n > kStampBlocks * kStampSlots) n = kStampBlocks * kStampSlots;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -4;
}
extern "C" int echo_diag_stamp_shape(int* blocks, int* slots) { *blocks = kStampBlocks; *slots = kStampSlots; return 0; }
#endif

int band_kernel_prepare(int smem_bytes, int spec_smem_bytes, int spec_hilb_smem_bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(band_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(spec_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, spec_smem_bytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(spec9_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, spec_smem_bytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(spec_hilbert_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, spec_hilb_smem_bytes);
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(spec9_hilbert_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, spec_hilb_smem_bytes);
    return e == hipSuccess ? 0 : -4;
}

int band2_kernel_prepare(int smem_bytes, int exact_smem_bytes) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(band2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            smem_bytes) != hipSuccess) return -4;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(exact_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               exact_smem_bytes) == hipSuccess ? 0 : -4;
}

int launch_band2_pair(const BandParams& p, int B, hipStream_t s) {
    if (B <= 0) return 0;
    if (!p.band2 || !p.d_in || !p.f9ghat || !p.det_out) return -1;
    hipLaunchKernelGGL(band2_kernel, dim3((unsigned)((B + 1) / 2) * p.nb), dim3(kF9NT), p.b2_smem_bytes, s, p, B);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_band2_exact(const BandParams& p, int B, hipStream_t s) {
    if (B <= 0) return 0;
    if (!p.band2 || !p.d_in || !p.res_out || !p.det_out) return -1;
    const int nfb = B * p.nb;
    hipLaunchKernelGGL(exact_kernel, dim3((unsigned)((nfb + kExWaves - 1) / kExWaves)), dim3(kExNT), p.b2_exact_smem_bytes, s, p, nfb);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_spec(const BandParams& p, int B, hipStream_t s) {
    if (B <= 0) return 0;
    if (p.fft9) hipLaunchKernelGGL(spec9_kernel, dim3((unsigned)B * p.nb), dim3(kF9NT), p.spec_smem_bytes, s, p);
    else hipLaunchKernelGGL(spec_kernel, dim3((unsigned)B * p.nb), dim3(kNT), p.spec_smem_bytes, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_spec_hilbert(const BandParams& p, int B, hipStream_t s) {
    if (B <= 0) return 0;
    if (!p.specf_in || !p.sq_out || !p.spec_out || !p.uk || !hilb_fits(p.N, p.n2)) return -1;
    if (p.fft9) hipLaunchKernelGGL(spec9_hilbert_kernel, dim3((unsigned)B * p.nb), dim3(kF9NT), p.spec_hilb_smem_bytes, s, p);
    else hipLaunchKernelGGL(spec_hilbert_kernel, dim3((unsigned)B * p.nb), dim3(kNT), p.spec_hilb_smem_bytes, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_band(const BandParams& p, int B, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(band_kernel, dim3((unsigned)B * p.nb), dim3(kNT), p.smem_bytes, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace echo
