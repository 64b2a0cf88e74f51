// Internal declarations shared by the translation units of libechohip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "band_phases.h"
#include "hilb_phases.h"
#include "long_phases.h"
#include "env_phases.h"

namespace echo {

struct IirParams {
    const float* rx;      // [B][N]
    double* y;            // [B][nb][N]   zero-phase IIR output
    double* fwd;          // [nb][ntiles][N+2P][64] forward-pass scratch
    const double* b;      // [nb][NC]
    const double* a;      // [nb][NC]
    const double* zi;     // [nb][NC-1]
    int B, N, P, nb, ntiles;
    int y_aligned16;      // y base and every row start are 16-byte aligned (N even): 16-byte stores allowed
    int prio;             // 0..3: wave priority the kernel asks for (s_setprio); the recurrence is a latency chain
};

struct FuseParams {
    const BandRes* res;   // [B][nb]
    void* out;            // EchoResult[B]
    double* fused;        // [B] (may be null)
    uint8_t* valid_any;   // [B] (may be null)
    int B, nb, fs, range_offset, confidence, fusion, detector;
    double sound_speed;
    double fw[3];
    // full chain (envelope branch): per (frame, band) envelope results and signal quality; null = named chain
    const EnvRes* env;
    const double* sq;
    double* quality_out;  // optional [B][nb][2]: signal_quality, corr_quality
};

struct KalmanParams {
    double* x; double* p; uint8_t* init;      // [S]
    const double* z; const uint8_t* valid;    // [T][S]
    double* out;                              // [T][S] (may be null)
    void* rec;                                // EchoResult[T*S] (may be null): .kalman is filled
    int S, T;
    double q, r;
};

int launch_iir(const IirParams& p, int ncoef, bool bodd0, hipStream_t s);       // one lane per (frame, band), 64-frame tiles
int iir_kernel_prepare();                                                          // raises the lane kernel's LDS limit
int launch_iir_quad(const IirParams& p, int ncoef, int lps, bool bodd0, hipStream_t s);   // lps lanes per (frame, band): 4 or 16; tiles of 64/lps frames; bodd0: b[odd] == 0
int launch_band(const BandParams& p, int B, hipStream_t s);
int band_kernel_prepare(int smem_bytes, int spec_smem_bytes, int spec_hilb_smem_bytes);
int band2_kernel_prepare(int smem_bytes, int exact_smem_bytes);
int launch_band2_pair(const BandParams& p, int B, hipStream_t s);     // chain fast path: float32 matched filter, pairs of frames per workgroup
int launch_band2_exact(const BandParams& p, int B, hipStream_t s);    // chain fast path: exact float64 re-evaluation, one workgroup per (frame, band)
int launch_spec(const BandParams& p, int B, hipStream_t s);
int launch_spec_hilbert(const BandParams& p, int B, hipStream_t s);   // spectrum + signal_quality of the band signal
int long_kernels_prepare(const LongParams& p);
int launch_long_fir(const LongParams& p, int B, hipStream_t s);        // B*nb <= 65535 per launch
int launch_long_corr(const LongParams& p, int B, hipStream_t s);         // direct sum, or the FFT form when p.cf_use
int cfft_kernel_prepare();
int launch_long_corr_fft(const LongParams& p, int B, hipStream_t s);
bool long_corr_det_fits(const LongParams& p);
int launch_long_corr_det(const LongParams& p, int B, hipStream_t s);
// the same as one real transform per (frame, band) (rcf_phases.h), when LongParams::rc_use
int rcf_kernel_prepare(const LongParams& p);
bool long_rcorr_det_fits(const LongParams& p);
bool long_rcorr_fits(const LongParams& p);
int launch_long_rcorr_det(const LongParams& p, int B, hipStream_t s);
int launch_long_rcorr(const LongParams& p, int B, hipStream_t s);
int launch_long_detect(const LongParams& p, int B, hipStream_t s);
int launch_long_spec(const LongParams& p, int B, hipStream_t s);
// odd frame lengths beyond launch_long_spec's two LDS buffers (ospec_phases.h), when LongParams::os_use
int ospec_kernel_prepare(const LongParams& p);
int launch_long_spec_odd_ip(const LongParams& p, int B, hipStream_t s);
int env_kernel_prepare(const EnvParams& p);
int launch_corr_env(const EnvParams& p, int B, hipStream_t s);
int fce_kernel_prepare(const EnvParams& p);
int launch_fce(const EnvParams& p, int B, hipStream_t s);              // FIR + correlation + envelope branch (fce_phases.h)
int launch_unpack_env(const EnvRes* res, int32_t* peak, double* snr, double* quality, int32_t* npeaks, int n,
                      hipStream_t s);
int launch_synth(float* rx, const float* tx_mix, int B, int N, int M, int fs, long long first, unsigned long long seed,
                 float c_sound, hipStream_t s);
int launch_pack_res(const int32_t* peak, const double* snr, const double* amp, BandRes* res, int n, hipStream_t s);
int launch_unpack_res(const BandRes* res, int32_t* peak, double* snr, double* amp, int n, hipStream_t s);
int launch_fuse(const FuseParams& p, hipStream_t s);
int launch_kalman(const KalmanParams& p, hipStream_t s);
int launch_argmax(const double* x, int32_t* out, int rows, int n, hipStream_t s);
int launch_mag2db(const float* x, float* out, size_t n, hipStream_t s);
int launch_band_confidence(const double* snr, const double* amp, const int32_t* band, double* out, int n, int kind,
                           const double* fw, hipStream_t s);
int launch_normalize_conf(const double* conf, double* out, int rows, int n, hipStream_t s);
int launch_iir_order0(const float* rx, double* y, const double* b, int B, int N, int nb, hipStream_t s);
int launch_xcorr(const double* a, const double* b, double* out, int rows, int n, int m, hipStream_t s);   // -2: b too long for LDS
// rows_kernel.hip: per-row kernels for signals of any length.  Return 0, -2 (does not fit) or -4 (launch failed).
int launch_fsp_rows(double* corr, int rows, int len, int min_delay, int noise_n, double gate_db, int32_t* peak_idx,
                    double* snr_db, hipStream_t s);
int launch_dft_rows(const double* x, const void* roots /* double2[n]: exp(-2 pi i t / n) */, float* spec, int db, int rows,
                    int n, hipStream_t s);

}  // namespace echo
