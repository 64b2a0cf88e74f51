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

constexpr int XS = 65;      // LDS row stride (floats) of the rx tile: 64 samples + 1 pad
constexpr int YS = 33;      // LDS row stride (doubles) of the output tile: 32 samples + 1 pad
constexpr int CH = 32;      // samples per streamed chunk
constexpr int PF = 16;      // backward-pass prefetch depth (samples)

template <int NC, bool BODD0>
__global__ __launch_bounds__(64) void iir_filtfilt_kernel(IirParams p) {
    __shared__ float xt[2][64 * XS];
    __shared__ double yt[64 * YS];
    const int lane = threadIdx.x;
    // blocks id, id+8, id+16 (same XCD under round-robin placement) are the bands of one frame tile,
    // so the three reads of an rx tile meet in one L2
    const int id = blockIdx.x;
    const int grp = id / (8 * p.nb), rem = id % (8 * p.nb);
    const int band = rem / 8, tile = grp * 8 + rem % 8;
    if (tile >= p.ntiles) return;
    const int N = p.N, P = p.P, NE = N + 2 * P;
    const int f0 = tile * 64;

    double b[NC], a[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) { b[k] = p.b[band * NC + k]; a[k] = p.a[band * NC + k]; }
    const double* zi = p.zi + band * (NC - 1);

    // cooperative load of columns [c0, c0+ncol) of the tile's 64 rows into registers (ncol = 32 or 64)
    auto row_ptr = [&](int r) { int f = f0 + r; if (f >= p.B) f = p.B - 1; return p.rx + (size_t)f * N; };
    float pre[CH];
    auto load_chunk = [&](int c0) {          // 32 columns: lane -> (row = it*2 + lane/32, col = lane%32)
#pragma unroll
        for (int it = 0; it < CH; ++it) {
            const int r = it * 2 + (lane >> 5), c = c0 + (lane & 31);
            pre[it] = (c >= 0 && c < N) ? row_ptr(r)[c] : 0.f;
        }
    };
    auto store_chunk = [&](float* dst, int colofs) {
#pragma unroll
        for (int it = 0; it < CH; ++it) {
            const int r = it * 2 + (lane >> 5), c = lane & 31;
            dst[r * XS + colofs + c] = pre[it];
        }
    };

    double* fwd = p.fwd + ((size_t)(band * p.ntiles + tile) * NE) * 64 + lane;
    Df2t<NC, BODD0> f;

    // ---- forward pass: left odd extension from x[0..P]
    load_chunk(0);  store_chunk(xt[0], 0);
    load_chunk(CH); store_chunk(xt[0], CH);
    __syncthreads();
    const float* xr = xt[0] + lane * XS;
    const float x0 = xr[0];
    int i = 0;
    {
        const float e0 = 2.0f * x0 - xr[P];
        f.init(zi, (double)e0);
    }
    for (int k = 0; k < P; ++k, ++i) {
        const float e = 2.0f * x0 - xr[P - k];
        fwd[(size_t)i * 64] = f.step((double)e, b, a);
    }
    __syncthreads();
    // ---- forward pass: the frame itself, 32-sample chunks, next chunk prefetched into registers
    const int nchunks = (N + CH - 1) / CH;
    load_chunk(0); store_chunk(xt[0], 0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int cur = c & 1;
        if (c + 1 < nchunks) load_chunk((c + 1) * CH);
        const float* xc = xt[cur] + lane * XS;
        const int cnt = (N - c * CH) < CH ? (N - c * CH) : CH;
        if (cnt == CH) {
#pragma unroll 4
            for (int t = 0; t < CH; ++t, ++i) fwd[(size_t)i * 64] = f.step((double)xc[t], b, a);
        } else {
            for (int t = 0; t < cnt; ++t, ++i) fwd[(size_t)i * 64] = f.step((double)xc[t], b, a);
        }
        if (c + 1 < nchunks) store_chunk(xt[cur ^ 1], 0);
        __syncthreads();
    }
    // ---- forward pass: right odd extension from x[N-64 .. N-1]
    load_chunk(N - 64); store_chunk(xt[0], 0);
    load_chunk(N - 32); store_chunk(xt[0], CH);
    __syncthreads();
    double ylast = 0.0;
    {
        const float xl = xr[63];
        for (int k = 0; k < P; ++k, ++i) {
            const float e = 2.0f * xl - xr[62 - k];
            ylast = f.step((double)e, b, a);
            fwd[(size_t)i * 64] = ylast;
        }
    }
    // all of this lane's forward stores must be visible to its own later loads: same lane, same address,
    // program order suffices.

    // ---- backward pass over the forward output, newest sample first
    f.init(zi, ylast);
    double bufA[PF], bufB[PF];
    auto loadg = [&](double (&buf)[PF], int top) {     // buf[j] = fwd[top - j], j < PF (clamped at 0)
#pragma unroll
        for (int j = 0; j < PF; ++j) { const int ii = top - j; buf[j] = fwd[(size_t)(ii < 0 ? 0 : ii) * 64]; }
    };
    double* yrow = yt + lane * YS;
    auto flush = [&](int n0) {          // write tile columns [n0, n0+32) of 64 rows
        __syncthreads();
#pragma unroll 4
        for (int it = 0; it < CH; ++it) {
            const int r = it * 2 + (lane >> 5), c = lane & 31;
            const int fr = f0 + r, n = n0 + c;
            if (fr < p.B && n < N) p.y[((size_t)fr * p.nb + band) * N + n] = yt[r * YS + c];
        }
        __syncthreads();
    };
    auto consume = [&](double (&buf)[PF], int top) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int ii = top - j;
            if (ii < 0) break;
            const double y = f.step(buf[j], b, a);
            const int n = ii - P;
            if (n >= 0 && n < N) {
                yrow[n & (CH - 1)] = y;
                if ((n & (CH - 1)) == 0) flush(n);
            }
        }
    };
    int top = NE - 1;
    loadg(bufA, top);
    while (top >= 0) {
        loadg(bufB, top - PF);
        consume(bufA, top);
        top -= PF;
        if (top < 0) break;
        loadg(bufA, top - PF);
        consume(bufB, top);
        top -= PF;
    }
}

#define IIR_INST(NC, BZ) template __global__ void iir_filtfilt_kernel<NC, BZ>(IirParams);
IIR_INST(13, true) IIR_INST(13, false) IIR_INST(17, true) IIR_INST(17, false)

int launch_iir(const IirParams& p, int ncoef, bool bodd0, hipStream_t s) {
    const int groups = (p.ntiles + 7) / 8;
    dim3 grid(groups * 8 * p.nb), block(64);
    if (ncoef == 13 && bodd0)       hipLaunchKernelGGL((iir_filtfilt_kernel<13, true>), grid, block, 0, s, p);
    else if (ncoef == 13)           hipLaunchKernelGGL((iir_filtfilt_kernel<13, false>), grid, block, 0, s, p);
    else if (ncoef == 17 && bodd0)  hipLaunchKernelGGL((iir_filtfilt_kernel<17, true>), grid, block, 0, s, p);
    else if (ncoef == 17)           hipLaunchKernelGGL((iir_filtfilt_kernel<17, false>), grid, block, 0, s, p);
    else return -2;
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace echo
