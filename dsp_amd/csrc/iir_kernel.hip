// Zero-phase IIR (scipy.signal.filtfilt, method='pad', padtype='odd') for a batch of frames.
//
// Reference call: filtfilt(b, a, sig) at airSonar_optimized.py:193 with the (b, a) of
// iirfilter(order, ..., output='ba') (:172): a direct-form-II-transposed recurrence of 2*order states run
// forward then backward over the odd-extended frame.  The (b, a) form of a 12th-order band-pass is
// ill-conditioned (about 4e-8 relative spread between algebraically equal realisations in float64), so
// matching the reference's peak indices requires the reference's own rounding sequence: this kernel
// evaluates exactly
//     y    = z[0] + b[0]*x
//     z[k] = (z[k+1] + x*b[k+1]) - y*a[k+1]
//     z[last] = x*b[last] - y*a[last]
// in float64 with one rounding per multiply and per add (this file is compiled with -ffp-contract=off).
// The recurrence is sequential in time, so the mapping is one wavefront lane per (frame, band): a wave
// owns 64 consecutive frames of one band, coefficients are wave-uniform (scalar registers).
//
// Memory: rx is read through an LDS transpose tile (coalesced 128-B row segments in, lane-per-row out);
// the forward pass streams its output to an HBM scratch laid out [sample][lane] (one 512-B row per store);
// the backward pass reads that scratch in reverse with a register double buffer and writes the result
// through an LDS tile as 256-B row segments of y[frame][band][n].
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "echo_internal.h"

namespace echo {

template <int NC, bool BODD0>
struct Df2t {
    double z[NC - 1];
    __device__ __forceinline__ void init(const double* __restrict__ zi, double x0) {
#pragma unroll
        for (int k = 0; k < NC - 1; ++k) z[k] = zi[k] * x0;
    }
    __device__ __forceinline__ double step(double x, const double (&b)[NC], const double (&a)[NC]) {
        const double y = z[0] + b[0] * x;
#pragma unroll
        for (int k = 0; k < NC - 2; ++k) {
            double t;
            if (BODD0 && ((k + 1) & 1)) t = z[k + 1];          // b[k+1] == 0 exactly: z + x*0 == z
            else                        t = z[k + 1] + x * b[k + 1];
            z[k] = t - y * a[k + 1];
        }
        z[NC - 2] = x * b[NC - 1] - y * a[NC - 1];
        return y;
    }
};

constexpr int XS = 68;      // LDS row stride (floats) of the rx tile: 64 samples + 4 pad (16-B aligned rows, conflict-free b128)
constexpr int YS = 34;      // LDS row stride (doubles) of the output tile: 32 samples + 2 pad (16-B aligned rows)
constexpr int CH = 32;      // samples per streamed chunk

// The waves of a workgroup are independent (each owns 16 frames of one band and its own LDS tiles), so they never
// wait for each other: only the compiler has to be kept from moving LDS accesses across the hand-over points.  A
// wave's LDS operations execute in program order.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Three independent waves per workgroup, each with its own 52,224 bytes of LDS tiles: 156,672 bytes admit one
// workgroup per compute unit, whose waves land on different SIMDs.  (With one-wave workgroups the dispatcher stacks
// several of these issue-bound waves on one SIMD while others stay empty.)  A 16384-frame pass of three bands is 768
// waves = 256 workgroups, one per compute unit.
constexpr int LW = 3;
constexpr int kLaneWaveLds = 2 * 64 * XS * 4 + 64 * YS * 8;

template <int NC, bool BODD0>
__global__ __launch_bounds__(64 * LW) void iir_filtfilt_kernel(IirParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lane_lds[];
    const int wave = threadIdx.x >> 6;
    float (*xt)[64 * XS] = reinterpret_cast<float (*)[64 * XS]>(lane_lds + wave * kLaneWaveLds);
    double* yt = reinterpret_cast<double*>(lane_lds + wave * kLaneWaveLds + 2 * 64 * XS * 4);
    const int lane = threadIdx.x & 63;
    // waves id, id+8, id+16 are the bands of one frame tile
    const int id = blockIdx.x * LW + wave;
    const int grp = id / (8 * p.nb), rem = id % (8 * p.nb);
    const int band = rem / 8, tile = grp * 8 + rem % 8;
    if (tile >= p.ntiles) return;
    const int N = p.N, P = p.P, NE = N + 2 * P;
    const int f0 = tile * 64;

    double b[NC], a[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) { b[k] = p.b[band * NC + k]; a[k] = p.a[band * NC + k]; }
    const double* zi = p.zi + band * (NC - 1);

    // Cooperative, branch-free load of columns [c0, c0+32) of the tile's 64 rows: lane -> (row it*2 + lane/32,
    // column lane%32), one 128-B row segment per half wave.  Columns past the frame end are clamped (their values
    // are never consumed); rows past the batch end repeat the last frame.
    const int lrow = lane >> 5, lcol = lane & 31;
    float pre[CH];
    auto load_chunk = [&](int c0) {
        int c = c0 + lcol;
        c = c < 0 ? 0 : (c > N - 1 ? N - 1 : c);
#pragma unroll
        for (int it = 0; it < CH; ++it) {
            int f = f0 + it * 2 + lrow;
            f = f < p.B ? f : p.B - 1;
            pre[it] = p.rx[(size_t)f * N + c];
        }
    };
    auto store_chunk = [&](float* dst, int colofs) {
#pragma unroll
        for (int it = 0; it < CH; ++it) dst[(it * 2 + lrow) * XS + colofs + lcol] = pre[it];
    };

    double* fwd = p.fwd + ((size_t)(band * p.ntiles + tile) * NE) * 64 + lane;
    Df2t<NC, BODD0> f;
    // Only the two 32-step blocks below are unrolled; everything else stays a rolled loop so the hot code
    // (about 20 KB) fits the instruction cache.

    // ---- forward pass: left odd extension from x[0..P]
    load_chunk(0);  store_chunk(xt[0], 0);
    load_chunk(CH); store_chunk(xt[0], CH);
    wave_sync();
    const float* xr = xt[0] + lane * XS;
    {
        const float x0 = xr[0];
        f.init(zi, (double)(2.0f * x0 - xr[P]));
#pragma unroll 1
        for (int k = 0; k < P; ++k) {
            const float e = 2.0f * x0 - xr[P - k];
            *fwd = f.step((double)e, b, a);
            fwd += 64;
        }
    }
    wave_sync();
    // ---- forward pass: the frame itself, 32-sample chunks held in registers, next chunk prefetched
    const int nfull = N / CH, ntail = N - nfull * CH;
    load_chunk(0); store_chunk(xt[0], 0);
    wave_sync();
#pragma unroll 1
    for (int c = 0; c < nfull; ++c) {
        const int cur = c & 1;
        float xc[CH];
        {
            const float4* src = reinterpret_cast<const float4*>(xt[cur] + lane * XS);
#pragma unroll
            for (int v = 0; v < CH / 4; ++v) {
                const float4 t = src[v];
                xc[4 * v] = t.x; xc[4 * v + 1] = t.y; xc[4 * v + 2] = t.z; xc[4 * v + 3] = t.w;
            }
        }
        load_chunk((c + 1) * CH);            // past the frame end the columns clamp; the tail chunk reads what it needs
#pragma unroll
        for (int t = 0; t < CH; ++t) { *fwd = f.step((double)xc[t], b, a); fwd += 64; }
        store_chunk(xt[cur ^ 1], 0);
        wave_sync();
    }
    {
        const float* xc = xt[nfull & 1] + lane * XS;
#pragma unroll 1
        for (int t = 0; t < ntail; ++t) { *fwd = f.step((double)xc[t], b, a); fwd += 64; }
    }
    wave_sync();
    // ---- forward pass: right odd extension from x[N-64 .. N-1]
    load_chunk(N - 64); store_chunk(xt[0], 0);
    load_chunk(N - 32); store_chunk(xt[0], CH);
    wave_sync();
    double ylast = 0.0;
    {
        const float xl = xr[63];
#pragma unroll 1
        for (int k = 0; k < P; ++k) {
            const float e = 2.0f * xl - xr[62 - k];
            ylast = f.step((double)e, b, a);
            *fwd = ylast;
            fwd += 64;
        }
    }
    // fwd now points one row past the last forward sample (row NE); a lane re-reads only what it wrote itself,
    // so program order is all the ordering the backward pass needs.

    // ---- backward pass over the forward output, newest sample first.  The last P steps (left extension) only
    // produce samples that filtfilt crops, so the pass stops at the first frame sample.
    f.init(zi, ylast);
    const double* bwd = fwd - 64;                       // row NE-1; row i is bwd[(i - (NE-1)) * 64]
    double* yrow = yt + lane * YS;
    const int srow = lane >> 4, scol = (lane & 15) * 2;  // flush mapping: 4 rows x 16 double2 per instruction
    auto flush = [&](int n0) {          // write tile columns [n0, n0+32) of 64 rows as 16-byte stores
        wave_sync();
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int r = it * 4 + srow;
            const int fr = f0 + r, n = n0 + scol;
            if (fr < p.B && n < N) {
                const double2 v = *reinterpret_cast<const double2*>(yt + r * YS + scol);
                double* dst = p.y + ((size_t)fr * p.nb + band) * N + n;
                if (p.y_aligned16 && n + 1 < N) *reinterpret_cast<double2*>(dst) = v;
                else { dst[0] = v.x; if (n + 1 < N) dst[1] = v.y; }
            }
        }
        wave_sync();
    };
    // right extension: rows NE-1 .. P+N, no output
#pragma unroll 1
    for (int k = 0; k < P; ++k) { (void)f.step(*bwd, b, a); bwd -= 64; }
    // tail chunk: samples N-1 .. nfull*32
#pragma unroll 1
    for (int t = ntail - 1; t >= 0; --t) { yrow[t] = f.step(*bwd, b, a); bwd -= 64; }
    if (ntail > 0) flush(nfull * CH);
    // full chunks, 32 rows prefetched into registers one chunk ahead
    double cur[CH], nxt[CH];
    auto loadc = [&](double (&buf)[CH], const double* top, bool live) {
#pragma unroll
        for (int j = 0; j < CH; ++j) buf[j] = live ? top[-(ptrdiff_t)j * 64] : 0.0;
    };
    if (nfull > 0) loadc(cur, bwd, true);
#pragma unroll 1
    for (int c = nfull - 1; c >= 0; --c) {
        loadc(nxt, bwd - (ptrdiff_t)CH * 64, c > 0);
#pragma unroll
        for (int j = 0; j < CH; ++j) yrow[CH - 1 - j] = f.step(cur[j], b, a);
        flush(c * CH);
        bwd -= (ptrdiff_t)CH * 64;
#pragma unroll
        for (int j = 0; j < CH; ++j) cur[j] = nxt[j];
    }
}

// ------------------------------------------------------------------------------------------------------------
// Quad-lane variant: the 2*order states of one (frame, band) are spread over the 4 lanes of a quad (NS = states/4
// per lane), so a wave owns 16 frames of one band and the sequential instruction stream per sample shrinks from
// ~42 to ~24 instructions.  The arithmetic per state is unchanged, operation for operation:
//     y    = z[0] + b[0]*x                         (lane 0 of the quad, broadcast by DPP quad_perm)
//     z[k] = (z[k+1] + x*b[k+1]) - y*a[k+1]        (z[k+1] of the neighbouring lane arrives by DPP; z[2*order] = 0)
// Multiplying x by a coefficient that is exactly zero and adding it changes nothing, so no special case is needed.
// Small batches leave most SIMDs idle in the lane-per-frame kernel; this one trades 2x the total issue slots for
// 4x the waves, and at large batches both are bound by the forward-scratch traffic to HBM instead.
__device__ __forceinline__ double dpp_quad_bcast0(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // mov_dpp: every lane is written, so no previous value of the destination is needed (with update_dpp's `old` operand
    // the compiler first copies the source into the destination, one more instruction per half)
    lo = __builtin_amdgcn_mov_dpp(lo, 0x00, 0xf, 0xf, false);          // quad_perm [0,0,0,0]
    hi = __builtin_amdgcn_mov_dpp(hi, 0x00, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_quad_next(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, 0xF9, 0xf, 0xf, false);          // quad_perm [1,2,3,3]
    hi = __builtin_amdgcn_mov_dpp(hi, 0xF9, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// BZ: the numerator's odd coefficients are exactly zero (a Butterworth band-pass) and NS is even, so that the states
// with a zero coefficient sit in the same slots (j even <-> state index odd) of every lane: their x * 0 and the addition
// of that zero are skipped (z + 0 == z), 4 of the 23 instructions per sample for NS = 4.
template <int NS, bool BZ = false>
struct Df2tQuad {
    double z[NS];
    double bl[NS], al[NS], b0;
    bool last;     // sub-lane 3: there is no state beyond its last one
    double keep;   // 0 for that sub-lane, 1 for the others
    const double* zi;
    template <int NC> __device__ __forceinline__ void setup(const IirParams& p, int band, int sub) {
        b0 = p.b[band * NC];
        last = sub == 3; keep = last ? 0.0 : 1.0;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            bl[j] = p.b[band * NC + NS * sub + j + 1];
            al[j] = p.a[band * NC + NS * sub + j + 1];
        }
        zi = p.zi + band * (NC - 1) + NS * sub;
    }
    __device__ __forceinline__ void init(double x0) {
#pragma unroll
        for (int j = 0; j < NS; ++j) z[j] = zi[j] * x0;
    }
    // The order of the statements is the intended issue order (the wave issues in order and a float64 result is
    // ready ~18 cycles after issue): everything that does not need this sample's y goes first, so the only exposed
    // dependences are y -> broadcast -> y*a -> subtract.  sched_barrier keeps the compiler from sinking the early
    // work back behind the broadcast.
    __device__ __forceinline__ double step(double x) {
        double px[NS], t[NS];
        const double yb = b0 * x;
#pragma unroll
        for (int j = 0; j < NS; ++j) px[j] = (BZ && j % 2 == 0) ? 0.0 : x * bl[j];
        const double zn = dpp_quad_next(z[0]);
        const double yc = z[0] + yb;
#pragma unroll
        for (int j = 0; j + 1 < NS; ++j) t[j] = (BZ && j % 2 == 0) ? z[j + 1] : z[j + 1] + px[j];
        // zn * 1 + px rounds exactly as zn + px; the sub-lane without a next state multiplies by 0 instead of selecting
        t[NS - 1] = __builtin_fma(zn, keep, px[NS - 1]);
        __builtin_amdgcn_sched_barrier(0);
        const double y = dpp_quad_bcast0(yc);
        double py[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) py[j] = y * al[j];
#pragma unroll
        for (int j = 0; j < NS; ++j) z[j] = t[j] - py[j];
        __builtin_amdgcn_sched_barrier(0);
        return y;
    }
};

// Row-wide variant for small batches (streaming ticks): one state per lane, a sequence per DPP row of 16 lanes
// (lanes beyond the 2*order states carry zero coefficients and stay zero), so a wave owns 4 frames and a sample costs
// ~13 instructions instead of ~27: the tick latency, which is the length of this dependent chain, drops accordingly.
// row_shl:1 with bound_ctrl hands every lane its neighbour's state and the last lane a zero; row_newbcast:0 spreads y.
// Four times the instruction issue per frame of the quad mapping: only for batches that leave the chip idle anyway.
__device__ __forceinline__ double dpp_row_bcast0(double v) {
    // row_newbcast:0 (every lane is written) is the one DPP control the float64 data path takes: one 64-bit move
    return __builtin_amdgcn_update_dpp(v, v, 0x150, 0xf, 0xf, false);
}
__device__ __forceinline__ double dpp_row_next(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x101, 0xf, 0xf, true);    // row_shl:1, out-of-row reads give 0
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x101, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

template <int NC>
struct Df2tRow {
    double z, bl, al, b0;
    const double* zi;
    bool live;
    __device__ __forceinline__ void setup(const IirParams& p, int band, int sub) {
        live = sub < NC - 1;
        const int k = live ? sub : 0;
        b0 = p.b[band * NC];
        bl = live ? p.b[band * NC + k + 1] : 0.0;
        al = live ? p.a[band * NC + k + 1] : 0.0;
        zi = p.zi + band * (NC - 1) + k;
    }
    __device__ __forceinline__ void init(double x0) { z = live ? zi[0] * x0 : 0.0; }
    // a sample is ~15 instructions of 4+ cycles each on the 16-lane SIMD: the wave is bound by its own issue, so the
    // count of instructions is what matters here, not the length of the dependent chain
    __device__ __forceinline__ double step(double x) {
        const double yb = b0 * x;
        const double px = x * bl;
        const double zn = dpp_row_next(z);
        const double yc = z + yb;
        const double t = zn + px;
        __builtin_amdgcn_sched_barrier(0);
        const double y = dpp_row_bcast0(yc);
        const double py = y * al;
        z = t - py;
        __builtin_amdgcn_sched_barrier(0);
        return y;
    }
};

constexpr int QW = 4;       // waves per workgroup: one per SIMD of the compute unit that receives the workgroup

// One-wave workgroups leave the choice of SIMD to the dispatcher, which was seen to stack two or three of these
// latency-chains on one SIMD after some predecessors (0.79 ms instead of 0.59 ms per 4096 frames); a four-wave
// workgroup is spread over the four SIMDs of its compute unit.
template <int NC, int LPS, bool BZ> struct IirFilt;
template <int NC, bool BZ> struct IirFilt<NC, 4, BZ> {
    Df2tQuad<(NC - 1) / 4, BZ && ((NC - 1) / 4) % 2 == 0> f;
    __device__ __forceinline__ void setup(const IirParams& p, int band, int sub) { f.template setup<NC>(p, band, sub); }
};
template <int NC, bool BZ> struct IirFilt<NC, 16, BZ> {
    Df2tRow<NC> f;
    __device__ __forceinline__ void setup(const IirParams& p, int band, int sub) { f.setup(p, band, sub); }
};

// LPS lanes per (frame, band): 4 (quad mapping, 16 frames per wave) or 16 (row mapping, 4 frames per wave)
template <int NC, int LPS, bool BZ = false>
__global__ __launch_bounds__(64 * QW) void iir_filtfilt_quad_kernel(IirParams p) {
    constexpr int QF = 64 / LPS;                // frames per wave
    __shared__ __attribute__((aligned(16))) float xt_all[QW][2][QF * XS];
    __shared__ __attribute__((aligned(16))) double yt_all[QW][QF * YS];
    if (p.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (p.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (p.prio == 3) __builtin_amdgcn_s_setprio(3);
    const int wave = threadIdx.x >> 6;
    float (*xt)[QF * XS] = xt_all[wave];
    double* yt = yt_all[wave];
    const int lane = threadIdx.x & 63, u = lane / LPS, sub = lane % LPS;
    const int id = blockIdx.x * QW + wave;
    const int grp = id / (8 * p.nb), rem = id % (8 * p.nb);
    const int band = rem / 8, tile = grp * 8 + rem % 8;
    if (tile >= p.ntiles) return;
    const int N = p.N, P = p.P, NE = N + 2 * P;
    const int f0 = tile * QF;

    IirFilt<NC, LPS, BZ> filt;
    filt.setup(p, band, sub);
    auto& f = filt.f;
    auto init = [&](double x0) { f.init(x0); };

    // cooperative load of columns [c0, c0+32) of the 16 rows: lane -> (row it*2 + lane/32, column lane%32)
    const int lrow = lane >> 5, lcol = lane & 31;
    constexpr int LI = QF / 2;
    float pre[LI];
    auto load_chunk = [&](int c0) {
        int c = c0 + lcol;
        c = c < 0 ? 0 : (c > N - 1 ? N - 1 : c);
#pragma unroll
        for (int it = 0; it < LI; ++it) {
            int fr = f0 + it * 2 + lrow;
            fr = fr < p.B ? fr : p.B - 1;
            pre[it] = p.rx[(size_t)fr * N + c];
        }
    };
    auto store_chunk = [&](float* dst, int colofs) {
#pragma unroll
        for (int it = 0; it < LI; ++it) dst[(it * 2 + lrow) * XS + colofs + lcol] = pre[it];
    };

    double* fwd = p.fwd + ((size_t)(band * p.ntiles + tile) * NE) * QF + u;   // row i at fwd[i * QF]
    // every lane of a quad holds the same y after the broadcast and stores it to the same address: no divergence

    // ---- forward pass: left odd extension from x[0..P]
    load_chunk(0);  store_chunk(xt[0], 0);
    load_chunk(CH); store_chunk(xt[0], CH);
    wave_sync();
    const float* xr = xt[0] + u * XS;
    {
        const float x0 = xr[0];
        init((double)(2.0f * x0 - xr[P]));
#pragma unroll 1
        for (int k = 0; k < P; ++k) {
            const float e = 2.0f * x0 - xr[P - k];
            const double y = f.step((double)e);
            *fwd = y;
            fwd += QF;
        }
    }
    wave_sync();
    // ---- forward pass: the frame, 32-sample chunks in registers, next chunk prefetched
    const int nfull = N / CH, ntail = N - nfull * CH;
    load_chunk(0); store_chunk(xt[0], 0);
    wave_sync();
#pragma unroll 1
    for (int c = 0; c < nfull; ++c) {
        const int cur = c & 1;
        float xc[CH];
        {
            const float4* src = reinterpret_cast<const float4*>(xt[cur] + u * XS);
#pragma unroll
            for (int v = 0; v < CH / 4; ++v) {
                const float4 t = src[v];
                xc[4 * v] = t.x; xc[4 * v + 1] = t.y; xc[4 * v + 2] = t.z; xc[4 * v + 3] = t.w;
            }
        }
        load_chunk((c + 1) * CH);
#pragma unroll
        for (int t = 0; t < CH; ++t) {
            const double y = f.step((double)xc[t]);
            fwd[t * QF] = y;
        }
        fwd += CH * QF;
        store_chunk(xt[cur ^ 1], 0);
        wave_sync();
    }
    {
        const float* xc = xt[nfull & 1] + u * XS;
#pragma unroll 1
        for (int t = 0; t < ntail; ++t) {
            const double y = f.step((double)xc[t]);
            *fwd = y;
            fwd += QF;
        }
    }
    wave_sync();
    // ---- forward pass: right odd extension from x[N-64 .. N-1]
    load_chunk(N - 64); store_chunk(xt[0], 0);
    load_chunk(N - 32); store_chunk(xt[0], CH);
    wave_sync();
    double ylast = 0.0;
    {
        const float xl = xr[63];
#pragma unroll 1
        for (int k = 0; k < P; ++k) {
            const float e = 2.0f * xl - xr[62 - k];
            ylast = f.step((double)e);
            *fwd = ylast;
            fwd += QF;
        }
    }
    // The four lanes of a quad re-read what lane 0 of the same quad stored: same wave, program order.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- backward pass, newest sample first; stops at the first frame sample (the rest is cropped)
    init(ylast);
    const double* bwd = fwd - QF;
    double* yrow = yt + u * YS;
    const int srow = lane >> 4, scol = (lane & 15) * 2;
    auto flush = [&](int n0) {          // tile columns [n0, n0+32) of 16 rows as 16-byte stores
        wave_sync();
        double2 v[QF / 4];
#pragma unroll
        for (int it = 0; it < QF / 4; ++it) v[it] = *reinterpret_cast<const double2*>(yt + (it * 4 + srow) * YS + scol);
#pragma unroll
        for (int it = 0; it < QF / 4; ++it) {
            const int fr = f0 + it * 4 + srow, n = n0 + scol;
            if (fr < p.B && n < N) {
                double* dst = p.y + ((size_t)fr * p.nb + band) * N + n;
                if (p.y_aligned16 && n + 1 < N) *reinterpret_cast<double2*>(dst) = v[it];
                else { dst[0] = v[it].x; if (n + 1 < N) dst[1] = v[it].y; }
            }
        }
        wave_sync();
    };
#pragma unroll 1
    for (int k = 0; k < P; ++k) { (void)f.step(*bwd); bwd -= QF; }
#pragma unroll 1
    for (int t = ntail - 1; t >= 0; --t) { const double y = f.step(*bwd); yrow[t] = y; bwd -= QF; }
    if (ntail > 0) flush(nfull * CH);
    // Full chunks.  The forward output comes back from HBM (the scratch is far larger than the caches), so rows are
    // requested two half-chunks (32 samples, about 2 us of arithmetic) before they are consumed: three register
    // buffers of 16 rows rotate through a loop unrolled by three.  Half-chunks keep the kernel at <= 184 registers,
    // which lets it share a SIMD with two workgroups' worth of band-kernel waves when it runs as a prefetch.
    constexpr int HB = CH / 2;
    double bufA[HB], bufB[HB], bufC[HB];
    const double* row0 = bwd - (ptrdiff_t)(nfull * CH - 1) * QF;       // row of frame sample 0
    auto loadh = [&](double (&buf)[HB], int h) {                        // rows of half-chunk h, newest first; h < 0 clamps
        const double* top = row0 + (ptrdiff_t)((h < 0 ? 0 : h) * HB + HB - 1) * QF;
#pragma unroll
        for (int j = 0; j < HB; ++j) buf[j] = top[-(ptrdiff_t)j * QF];
    };
    auto runh = [&](const double (&buf)[HB], int h) {
        double* dst = yrow + (h & 1) * HB;
#pragma unroll
        for (int j = 0; j < HB; ++j) dst[HB - 1 - j] = f.step(buf[j]);
        if ((h & 1) == 0) flush((h >> 1) * CH);                         // lower half done: the 32-sample tile is complete
    };
    const int H = 2 * nfull;
    if (H > 0) { loadh(bufA, H - 1); loadh(bufB, H - 2); }
#pragma unroll 1
    for (int h = H - 1; h >= 0; h -= 3) {
        loadh(bufC, h - 2); runh(bufA, h);
        if (h - 1 < 0) break;
        loadh(bufA, h - 3); runh(bufB, h - 1);
        if (h - 2 < 0) break;
        loadh(bufB, h - 4); runh(bufC, h - 2);
    }
}

template __global__ void iir_filtfilt_quad_kernel<13, 4>(IirParams);
template __global__ void iir_filtfilt_quad_kernel<17, 4>(IirParams);
template __global__ void iir_filtfilt_quad_kernel<13, 16>(IirParams);
template __global__ void iir_filtfilt_quad_kernel<17, 16>(IirParams);
template __global__ void iir_filtfilt_quad_kernel<17, 4, true>(IirParams);

// p.ntiles = ceil(B / (64 / lps)) tiles of 64 / lps frames; lps = 4 or 16
int launch_iir_quad(const IirParams& p, int ncoef, int lps, bool bodd0, hipStream_t s) {
    const int groups = (p.ntiles + 7) / 8;
    dim3 grid(groups * 8 * p.nb / QW), block(64 * QW);
    if (ncoef == 17 && lps == 4 && bodd0) hipLaunchKernelGGL((iir_filtfilt_quad_kernel<17, 4, true>), grid, block, 0, s, p);
    else if (ncoef == 13 && lps == 4)       hipLaunchKernelGGL((iir_filtfilt_quad_kernel<13, 4>), grid, block, 0, s, p);
    else if (ncoef == 17 && lps == 4)  hipLaunchKernelGGL((iir_filtfilt_quad_kernel<17, 4>), grid, block, 0, s, p);
    else if (ncoef == 13 && lps == 16) hipLaunchKernelGGL((iir_filtfilt_quad_kernel<13, 16>), grid, block, 0, s, p);
    else if (ncoef == 17 && lps == 16) hipLaunchKernelGGL((iir_filtfilt_quad_kernel<17, 16>), grid, block, 0, s, p);
    else return -2;
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

#define IIR_INST(NC, BZ) template __global__ void iir_filtfilt_kernel<NC, BZ>(IirParams);
IIR_INST(13, true) IIR_INST(13, false) IIR_INST(17, true) IIR_INST(17, false)

// once per process and device: the lane kernel's LDS request is above the default limit
int iir_kernel_prepare() {
    const int smem = LW * kLaneWaveLds;
    bool ok = true;
    ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&iir_filtfilt_kernel<13, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) == hipSuccess;
    ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&iir_filtfilt_kernel<13, false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) == hipSuccess;
    ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&iir_filtfilt_kernel<17, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) == hipSuccess;
    ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&iir_filtfilt_kernel<17, false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) == hipSuccess;
    return ok ? 0 : -4;
}

int launch_iir(const IirParams& p, int ncoef, bool bodd0, hipStream_t s) {
    const int groups = (p.ntiles + 7) / 8;
    const int smem = LW * kLaneWaveLds;
    dim3 grid((groups * 8 * p.nb + LW - 1) / LW), block(64 * LW);
    if (ncoef == 13 && bodd0)       hipLaunchKernelGGL((iir_filtfilt_kernel<13, true>), grid, block, smem, s, p);
    else if (ncoef == 13)           hipLaunchKernelGGL((iir_filtfilt_kernel<13, false>), grid, block, smem, s, p);
    else if (ncoef == 17 && bodd0)  hipLaunchKernelGGL((iir_filtfilt_kernel<17, true>), grid, block, smem, s, p);
    else if (ncoef == 17)           hipLaunchKernelGGL((iir_filtfilt_kernel<17, false>), grid, block, smem, s, p);
    else return -2;
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace echo
