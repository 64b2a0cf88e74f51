// Bluestein cyclic convolution of length 4608 = 9 * 8 * 8 * 8 for frames whose packed half-length n2 satisfies
// 2*n2 - 1 <= 4608 (the optimized parameter set: n2 = 2141): 0.53 of the arithmetic of the 8192-point version in
// band_phases.h, same structure: forward decimation-in-frequency passes, a register-resident last stage fused with
// the product by the chirp-filter spectrum and the first inverse stage, inverse decimation-in-time passes; no
// reordering pass (the filter spectrum is stored in pass order).
//   pass 1: radix 9, stride 512 (512 butterflies)            pass 2: radix 8, stride 64 inside each 512-block
//   pass 3: radix 8, stride 8                                  middle: radix 8 on 8 contiguous elements, x bhat, inverse
// 576 threads: one radix-8 butterfly each (the radix-9 passes use 512 of them).
// LDS position of element e is e + e/8 (one 8-byte pad per 8 elements): every pass then reads and writes 8-byte
// elements without bank conflicts inside each group of 16 lanes (strides of 1, 9 and 72 slots).
#pragma once
#include "band_phases.h"

namespace echo {

constexpr int kF9M = 4608;
constexpr int kF9NT = 576;
constexpr int kF9Slots = kF9M + kF9M / 8;          // 5184 float2 = 41472 bytes

ECHO_HD int pos9(int e) { return e + (e >> 3); }

// log2 for the dB conversion: the hardware instruction on the device (1 ulp; the spectra are compared at 1e-4 of the
// spectral peak and 0.05 dB near it), the library function on the host.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float echo_log2f(float v) { return __builtin_amdgcn_logf(v); }
// sqrt(a^2 + b^2) for magnitudes far from overflow and underflow (signals of order 1): the hardware square root
__device__ __forceinline__ float echo_abs2f(float a, float b) { return __builtin_amdgcn_sqrtf(a * a + b * b); }
#else
inline float echo_log2f(float v) { return log2f(v); }
inline float echo_abs2f(float a, float b) { return hypotf(a, b); }
#endif
// 20 log10(max(sqrt(m2), 1e-12)) from the squared magnitude
ECHO_HD float echo_db_from_sq(float m2) { return 3.01029995663981195f * echo_log2f(fmaxf(m2, 1e-24f)); }

// One complex element from the transform buffer in LDS.  On the device the read is a volatile 8-byte LDS read, which
// keeps it a single ds_read_b64 (256 bytes per clock); two neighbouring reads merged into ds_read2_b64 move the same
// bytes in twice the LDS cycles.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float2 ld9(const float2* q) {
    typedef const volatile __attribute__((address_space(3))) unsigned long long* lds_u64_t;
    const unsigned long long u = *(lds_u64_t)q;
    return __builtin_bit_cast(float2, u);
}
#else
inline float2 ld9(const float2* q) { return *q; }
#endif

// Pair kernel (band2_kernel): sizes shared by the kernel and the host-side plan
constexpr int kB2R = 3;                 // lags (or FIR outputs) per lane in the exact float64 rounds
constexpr int kB2Seg = 243;             // lags per exact segment: 81 lanes x 3
constexpr int kB2MaxCand = 30;          // lags listed for exact evaluation per frame; beyond it every lag is evaluated
struct B2Misc {                         // exact kernel: running results of one (frame, band), in LDS
    double best, amp, med0, med1;
    int bidx, pad;
};
struct B2Det {                          // pair kernel -> exact kernel, per (frame, band), in device memory
    double eps;                         // bound on |float32 lag - exact lag|
    float max32, amp32;                 // float32 maxima (signed over pos[min_delay:], absolute over every lag)
    int ncand, pad;                     // lags listed; beyond kB2MaxCand the list is incomplete
    int cand[kB2MaxCand];
};

// frequency index held at position p after the forward passes
inline int pass_order_freq9(int p) {
    const int m1 = p / 512, m2 = (p / 64) % 8, m3 = (p / 8) % 8, m4 = p % 8;
    return m1 + 9 * m2 + 72 * m3 + 576 * m4;
}

template <bool INV> ECHO_HD void dft3(float2& a, float2& b, float2& c) {
    const float s3 = 0.86602540378443864676f;
    const float2 s = cadd(b, c), d = csub(b, c);
    const float2 m = mk2(a.x - 0.5f * s.x, a.y - 0.5f * s.y);
    // forward: -i * s3 * d ; inverse: +i * s3 * d
    const float2 r = INV ? mk2(-s3 * d.y, s3 * d.x) : mk2(s3 * d.y, -s3 * d.x);
    a = cadd(a, s); b = cadd(m, r); c = csub(m, r);
}

// 9-point DFT, n = 3*n1 + n2 in, k = k1 + 3*k2 out (natural order both sides)
template <bool INV> ECHO_HD void dft9(float2 (&v)[9]) {
    // exp(-2*pi*i*k/9), k = 1, 2, 4
    const float2 w1 = mk2(0.76604444311897803520f, INV ? 0.64278760968653932632f : -0.64278760968653932632f);
    const float2 w2 = mk2(0.17364817766693034885f, INV ? 0.98480775301220805937f : -0.98480775301220805937f);
    const float2 w4 = mk2(-0.93969262078590838405f, INV ? 0.34202014332566873304f : -0.34202014332566873304f);
    float2 t[3][3];
#pragma unroll
    for (int n2 = 0; n2 < 3; ++n2) {
        float2 a = v[n2], b = v[3 + n2], c = v[6 + n2];
        dft3<INV>(a, b, c);
        t[n2][0] = a; t[n2][1] = b; t[n2][2] = c;
    }
    t[1][1] = cmul(t[1][1], w1); t[1][2] = cmul(t[1][2], w2);
    t[2][1] = cmul(t[2][1], w2); t[2][2] = cmul(t[2][2], w4);
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {
        float2 a = t[0][k1], b = t[1][k1], c = t[2][k1];
        dft3<INV>(a, b, c);
        v[k1] = a; v[k1 + 3] = b; v[k1 + 6] = c;
    }
}

// Bluestein input a[n] = (x[2n] + i x[2n+1]) w[n], zero-padded to 4608
ECHO_HD void ph9_load(int tid, int nt, int blk, const BandParams& p, unsigned char* smem) {
    float2* f = lds_c(smem, p.off_fft);
    for (int n = tid; n < kF9M; n += nt) {
        float2 a = mk2(0.f, 0.f);
        if (n < p.n2) {
            float2 x;
            if (p.specf_in) x = reinterpret_cast<const float2*>(p.specf_in + (size_t)blk * p.N)[n];
            else { const double* xd = p.specd_in + (size_t)blk * p.N + 2 * n; x = mk2((float)xd[0], (float)xd[1]); }
            a = cmul(x, p.bw[n]);
        }
        f[pos9(n)] = a;
    }
}

// radix-9 pass, stride 512; twiddles f9tw9[(m-1)*512 + i] = exp(-2*pi*i*(i*m)/4608), m = 1..8
template <bool INV> ECHO_HD void ph9_r9(int tid, const BandParams& p, unsigned char* smem) {
    if (tid >= 512) return;
    float2* f = lds_c(smem, p.off_fft);
    float2 v[9], w[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) w[m] = p.f9tw9[m * 512 + tid];
#pragma unroll
    for (int j = 0; j < 9; ++j) v[j] = ld9(f + pos9(tid + 512 * j));
    if (INV) {
#pragma unroll
        for (int m = 1; m < 9; ++m) v[m] = cmulc(v[m], w[m - 1]);
    }
    dft9<INV>(v);
    if (!INV) {
#pragma unroll
        for (int m = 1; m < 9; ++m) v[m] = cmul(v[m], w[m - 1]);
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) f[pos9(tid + 512 * j)] = v[j];
}

// the last inverse pass with its outputs left in registers: v[j] is element tid + 512 j of the result (tid < 512)
ECHO_HD void ph9_r9_inv_regs(int tid, const BandParams& p, unsigned char* smem, float2 (&v)[9]) {
    if (tid >= 512) {
#pragma unroll
        for (int j = 0; j < 9; ++j) v[j] = mk2(0.f, 0.f);
        return;
    }
    const float2* f = lds_c(smem, p.off_fft);
    float2 w[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) w[m] = p.f9tw9[m * 512 + tid];
#pragma unroll
    for (int j = 0; j < 9; ++j) v[j] = ld9(f + pos9(tid + 512 * j));
#pragma unroll
    for (int m = 1; m < 9; ++m) v[m] = cmulc(v[m], w[m - 1]);
    dft9<true>(v);
}

// radix-8 pass with stride q (64 or 8) inside each 512-block; twiddles tw[0..q) = W, tw[q..2q) = W^2, tw[2q..3q) = W^4,
// W = exp(-2*pi*i/(8q))^i
template <bool INV> ECHO_HD void ph9_r8(int tid, int q, const BandParams& p, unsigned char* smem) {
    if (tid >= kF9M / 8) return;
    float2* f = lds_c(smem, p.off_fft);
    const float2* tw = q == 64 ? p.f9twa : p.f9twb;
    const int blk512 = tid >> 6, r = tid & 63;
    const int i = r % q, base = blk512 * 512 + (r / q) * 8 * q + i;
    float2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ld9(f + pos9(base + j * q));
    bfly8<INV>(v, tw[i], tw[q + i], tw[2 * q + i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[pos9(base + j * q)] = v[j];
}

// last forward stage, product with the chirp-filter spectrum (pass order, scaled by 1/4608), first inverse stage
ECHO_HD void ph9_mid(int tid, const BandParams& p, unsigned char* smem) {
    if (tid >= kF9M / 8) return;
    float2* f = lds_c(smem, p.off_fft);
    float2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = ld9(f + 9 * tid + j);         // pos9(8*tid + j)
    dft8<false>(v);
    const float4* bh = reinterpret_cast<const float4*>(p.f9bhat + 8 * tid);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const float4 b = bh[jj];
        v[2 * jj] = cmul(v[2 * jj], mk2(b.x, b.y));
        v[2 * jj + 1] = cmul(v[2 * jj + 1], mk2(b.z, b.w));
    }
    dft8<true>(v);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[9 * tid + j] = v[j];
}

// X[k] from the convolution output (as ph_spec_store, positions through pos9)
ECHO_HD void ph9_store(int tid, int nt, int blk, const BandParams& p, unsigned char* smem) {
    const float2* f = lds_c(smem, p.off_fft);
    for (int k = tid; k < p.nbins; k += nt) {
        const int ka = (k == p.n2) ? 0 : k;
        const int kb = (k == 0 || k == p.n2) ? 0 : p.n2 - k;
        const float4 ab = ld4(p.post, 2 * k);
        const float2 ca = f[pos9(ka)];
        float2 cb = f[pos9(kb)]; cb.y = -cb.y;
        const float2 x = cadd(cmul(ca, mk2(ab.x, ab.y)), cmul(cb, mk2(ab.z, ab.w)));
        // 20 log10(max(|X|, 1e-12)) = 10 log10(max(|X|^2, 1e-24)): no square root on the dB path
        const float m2 = x.x * x.x + x.y * x.y;
        const float mag = p.spec_db ? echo_db_from_sq(m2) : sqrtf(m2);
        p.spec_out[(size_t)blk * p.nbins + k] = mag;
    }
}

}  // namespace echo
