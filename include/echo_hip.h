/*
 * echo_hip.h -- C ABI of libechohip.so: batched echo-ranging on MI355X (gfx950).
 *
 * Every entry point replaces one function of the reference's accelerator seam (module-level
 * `gpu_*` / `bandpass` / `first_strong_peak` ... functions that take host NumPy arrays, try CuPy and
 * fall back to NumPy/SciPy).  Paths below are relative to the reference repository.
 *
 * Conventions
 *   - All data pointers are DEVICE pointers owned by the caller, row-major, contiguous, batch-major:
 *     rx[B][N], band_sig[B][nb][N], corr[B][nb][L], spec_db[B][nb][N/2+1].
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous on it.
 *   - Return value: 0 on success, negative ECHO_ERR_* otherwise; echo_strerror() names it.  Data-dependent
 *     "no echo" is not an error: peak_idx = -1, snr_db = 0, valid = 0 (reference: `return None, 0.0`).
 *   - Threads and streams.  The tables of a handle are immutable after echo_create; everything else it owns changes
 *     inside calls: the scratch grows on demand (hipFree + hipMalloc), envelope plans are made on first use, the
 *     prefetch buffers, timing events and internal lanes come and go.  The reference calls its per-band functions from
 *     three pool threads at once (airSonar_optimized.py:717-719), so every entry point that touches that state takes a
 *     lock owned by the handle for the duration of the call: calls on one handle from several threads are serialised,
 *     never interleaved, and need no lock of the caller's.  (echo_destroy is the exception: no other call on the handle
 *     may be running or follow.)  On the device a handle's scratch serves one call at a time: calls issued on one
 *     stream are ordered by it; a call on another stream than the previous call on that handle first makes its stream
 *     wait for the work queued on the earlier one (one event, no host wait).  Calls on one handle therefore never
 *     overlap on the device; to overlap batches use one handle per stream (echo_chain_ordered).  Inputs and outputs
 *     belong to the caller: two calls that write the same output race as any two kernels would.
 *     echo_chain_noalloc is outside this device-side ordering (its scratch is per GPU, not per handle; see there).
 *   - A handle belongs to the device that was current in echo_create: calls with another device current return
 *     ECHO_ERR_INVALID.
 *   - Samples are finite numbers (AudioIO.record delivers int16 / 2**15, airSonar_optimized.py:573).  Frames holding
 *     Inf or NaN are outside the contract: the zero-phase IIR then yields a different Inf/NaN pattern than
 *     scipy.signal.filtfilt (it evaluates z + 0 * y where SciPy evaluates z alone), and the record of such a frame is
 *     unspecified.  Every other frame of the batch is unaffected: a frame's results never depend on its neighbours.
 */
#ifndef ECHO_HIP_H
#define ECHO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ECHO_MAX_BANDS 3

enum {
    ECHO_OK = 0,
    ECHO_ERR_INVALID = -1,      /* bad argument / size combination                        */
    ECHO_ERR_UNSUPPORTED = -2,  /* configuration does not fit this build's kernels        */
    ECHO_ERR_NOMEM = -3,        /* device allocation failed                               */
    ECHO_ERR_HIP = -4,          /* a HIP runtime call or kernel launch failed             */
    ECHO_ERR_NODEVICE = -5,     /* no gfx950 device visible                               */
    ECHO_ERR_TOO_LARGE = -6     /* echo_chain_noalloc: batch beyond the fixed scratch     */
};

/* detector / confidence / fusion variants (the three first_strong_peak generations the reference ships) */
enum { ECHO_DET_MEDIAN = 0,   /* airSonar_optimized.py:200-219 */
       ECHO_DET_MAD = 1,      /* airSonar.py:104-117           */
       ECHO_DET_MEANPOW = 2,  /* realtime_sonar.py:127-142     */
       ECHO_DET_TWOPEAK = 3   /* 666.py:153-200: peak_idx is the delay echo - direct path and may be <= 0;
                                 validity is carried by valid_mask (chain) or snr_db > 0 (first_strong_peak) */ };
enum { ECHO_CONF_CLAMPED = 0, /* airSonar_optimized.py:222-243 */
       ECHO_CONF_RAW = 1,     /* realtime_sonar.py:393         */
       ECHO_CONF_NONE = 2 };
enum { ECHO_FUSE_WEIGHTED_EPS = 0, /* airSonar_optimized.py:733-738 */
       ECHO_FUSE_WEIGHTED = 1,     /* realtime_sonar.py:336-339     */
       ECHO_FUSE_MEAN = 2,         /* airSonar.py:203-205           */
       ECHO_FUSE_SNRNORM = 3 };    /* 666.py:372-385: weights = snr / max(snr) * 100 + 1e-9 */

/* Host-side description of one sonar configuration; replaces the module-global `cfg` plus the
 * results of design_filters()/generate_chirps() (airSonar_optimized.py:65-113, :157-183).
 * Table pointers are HOST pointers, copied during echo_create. */
typedef struct EchoConfig {
    int32_t fs;            /* Config.FS                                                   */
    int32_t n_rx;          /* samples per frame  N = int(FS*LISTEN_LEN)                   */
    int32_t n_chirp;       /* chirp samples      M = int(FS*CHIRP_LEN)                    */
    int32_t n_bands;       /* len(BANDS), 1..ECHO_MAX_BANDS                               */
    int32_t iir_ncoef;     /* len(b) == len(a) == 2*order+1                               */
    int32_t n_taps;        /* FIR length                                                  */
    int32_t min_delay;     /* blind-zone samples int(FS*CHIRP_LEN*1.2)                    */
    int32_t noise_n;       /* noise-window samples                                        */
    int32_t detector;      /* ECHO_DET_*                                                  */
    int32_t confidence;    /* ECHO_CONF_*                                                 */
    int32_t fusion;        /* ECHO_FUSE_*                                                 */
    int32_t range_offset;  /* delay = peak_idx - range_offset (M-1 in airSonar_optimized) */
    double snr_gate_db;    /* 6.0                                                         */
    double sound_speed;    /* calculate_sound_speed(temperature)                          */
    double kalman_q, kalman_r;
    double freq_weights[ECHO_MAX_BANDS];
    const double* b;       /* [n_bands][iir_ncoef]   a0-normalised numerator              */
    const double* a;       /* [n_bands][iir_ncoef]                                        */
    const double* zi;      /* [n_bands][iir_ncoef-1] lfilter_zi(b, a)                     */
    const double* taps;    /* [n_bands][n_taps]                                           */
    const double* chirps;  /* [n_bands][n_chirp]                                          */
} EchoConfig;

/* Per-frame result record (64 bytes): what run() logs, emits and appends to the CSV
 * (airSonar_optimized.py:733-747), one row per frame. */
typedef struct EchoResult {
    float   dist[ECHO_MAX_BANDS];      /* metres; 0 where the band is invalid             */
    float   conf[ECHO_MAX_BANDS];      /* raw band confidence                             */
    float   snr_db[ECHO_MAX_BANDS];
    int32_t peak_idx[ECHO_MAX_BANDS];  /* index into pos = corr[half:], -1 = None         */
    float   fused_dist;                /* confidence-weighted distance                    */
    float   avg_conf;                  /* mean of normalised confidences                  */
    float   kalman;                    /* Kalman-smoothed distance of the frame's stream  */
    uint32_t valid_mask;               /* bit j: band j produced a range                  */
} EchoResult;

typedef struct EchoHandle EchoHandle;

/* Set-up: replaces SonarWorker.__init__'s design_filters/generate_chirps/`cp.asarray(taps)` preload
 * (airSonar_optimized.py:598-619).  Uses the current HIP device.
 * Sizes: frames with an odd n_rx run on the tiled kernels, their spectra as a chirp-z convolution in a workgroup's LDS
 * (in two buffers up to n_rx of about 6600, with the transform in place up to about 12,100; long even frames whose half
 * has a prime factor the mixed-radix passes do not take go the same way); longer frames without an FFT plan get their
 * spectra as direct float64 sums, O(n_rx^2) per frame and band.  Correlations of up to 4608 lags (n_rx + n_chirp - 1; the reference's constants give 4556) with the
 * median or mean-power detector run on single-workgroup kernels; anything longer, and the MAD and two-peak detectors,
 * on tiled kernels (every detector; echo_chain_full and echo_envelope_detection are then ECHO_ERR_UNSUPPORTED).
 * echo_chain_full takes every even correlation length of the single-workgroup kernels: from 3072 lags on the Hilbert
 * transform is one real cyclic convolution of length 9216, whatever the factors of the length (4562 = 2 * 2281 included);
 * shorter ones take half-length transforms (Bluestein convolutions where the half has a large prime factor).
 * iir_ncoef = 1 with n_taps = 1 is the identity filter ([1], [1], [1]) design_filters falls back to (:178-182); zi may
 * then be NULL.  ECHO_ERR_UNSUPPORTED from echo_create: IIR order other than 0, 6 or 8 (iir_ncoef 1 / 13 / 17), or a frame so long that one
 * frame's detector input exceeds a workgroup's 160 KB of LDS (about 20,000 lags). */
int  echo_create(const EchoConfig* cfg, EchoHandle** out);
void echo_destroy(EchoHandle* h);
/* Pre-size the handle's scratch for batches of up to `max_frames` so later calls allocate nothing. */
int  echo_reserve(EchoHandle* h, int max_frames);
const char* echo_strerror(int code);
/* Library/build identification: "echohip <version> gfx950". */
const char* echo_version(void);

/* gpu_bandpass, CuPy branch: FIR only, `cp.convolve(signal, taps, 'same')` (airSonar_optimized.py:305-313).
 * rx[B][N] float32 -> out[B][nb][N] float64. */
int echo_fir_bank(EchoHandle* h, const float* rx, double* out, int B, void* stream);

/* bandpass / gpu_bandpass CPU branch: filtfilt(b, a, sig) then fftconvolve(y, taps, 'same')
 * (airSonar_optimized.py:185-197, :324-327).  rx[B][N] float32 -> band_sig[B][nb][N] float64. */
int echo_bandpass_bank(EchoHandle* h, const float* rx, double* band_sig, int B, void* stream);

/* mag2db(np.fft.rfft(band_sig)) (airSonar_optimized.py:710, :286-287; gpu_spectrum :332-343 with db=0).
 * band_sig[B][nb][N] float64 -> spec[B][nb][N/2+1] float32.  db != 0: 20*log10(max(|X|,1e-12)); db == 0: |X|. */
int echo_spectrum(EchoHandle* h, const double* band_sig, float* spec, int db, int B, void* stream);

/* gpu_spectrum on individual signals (airSonar_optimized.py:332-343; mag2db(np.fft.rfft(x)) :710 with db != 0):
 * x[rows][N] float64, N the handle's frame length -> spec[rows][N/2+1] float32.  The transform does not depend on the
 * band, so any number of rows is taken. */
int echo_spectrum_rows(EchoHandle* h, const double* x, float* spec, int db, int rows, void* stream);

/* gpu_spectrum / mag2db(np.fft.rfft(x)) for signals of ANY length n (airSonar_optimized.py:332-343, :710; np.fft.rfft
 * has no length restriction): x[rows][n] float64 -> spec[rows][n/2+1] float32, |X| or its dB as echo_spectrum.  No
 * handle: a length the FFT kernels hold (even n up to about 19,000 whose half has small prime factors or fits the
 * Bluestein kernels, odd n up to about 12,100) gets a plan of its own on first use, kept per (device, n) until the process
 * ends; every other length up to 2^18 runs as direct float64 sums (O(n^2) per row, exact to rounding).  Thread-safe.
 * ECHO_ERR_UNSUPPORTED: n beyond 2^18 without an FFT plan. */
int echo_spectrum_any(const double* x, float* spec, int db, int rows, int n, void* stream);

/* gpu_mag2db / mag2db on an arbitrary array (airSonar_optimized.py:345-357). */
int echo_mag2db(const float* x, float* out, size_t n, void* stream);

/* gpu_correlate: scipy.signal.correlate(a, b, 'full') against the band's chirp
 * (airSonar_optimized.py:290-303).  band_sig[B][nb][N] -> corr[B][nb][N+M-1], float64. */
int echo_correlate(EchoHandle* h, const double* band_sig, double* corr, int B, void* stream);

/* gpu_correlate for arbitrary operands, as the reference calls it per band (airSonar_optimized.py:290-303, calls
 * :639, :712): out[row][k] = scipy.signal.correlate(a[row], b, 'full')[k], a[rows][n], b[m], out[rows][n+m-1], float64,
 * direct sums in the order of echo_correlate.  ECHO_ERR_UNSUPPORTED when b does not fit a workgroup's LDS (m > ~8000). */
int echo_xcorr(const double* a, const double* b, double* out, int rows, int n, int m, void* stream);

/* first_strong_peak (airSonar_optimized.py:200-219; variants per cfg.detector).  Mutates corr exactly as the
 * reference does (blind zone corr[half : half+min_delay] zeroed).  Also returns amplitude = max|corr| of the
 * mutated array (airSonar_optimized.py:666) when `amplitude` is non-NULL. */
int echo_first_strong_peak(EchoHandle* h, double* corr, int32_t* peak_idx, double* snr_db,
                           double* amplitude, int B, void* stream);

/* first_strong_peak(corr, fs, min_delay_samples) with the caller's own arguments (airSonar_optimized.py:200-219): rows
 * of any length `len` (pos = corr[len/2:]), any blind zone min_delay (= min_delay_samples, or int(fs * CHIRP_LEN * 1.2)
 * when the caller passes None), noise window noise_n = int(SNR_NOISE_MS * fs) samples behind it (truncated at the end of
 * pos like the reference's slice), gate snr_gate_db (6.0 in the reference).  corr[rows][len] float64 is mutated as the
 * reference mutates it (pos[:min_delay] = 0) unless pos.size <= min_delay, which gives (-1, 0.0) and leaves the row
 * alone (:206-207).  peak_idx[rows] = np.argmax(pos) (first index on ties; -1 = None), snr_db[rows].  No handle.
 * ECHO_ERR_UNSUPPORTED: a noise window of more than 8192 samples. */
int echo_first_strong_peak_rows(double* corr, int rows, int len, int min_delay, int noise_n, double snr_gate_db,
                                int32_t* peak_idx, double* snr_db, void* stream);

/* gpu_envelope_detection(corr, 'hilbert', smooth_window=5) and the envelope-path peak choice of _process_band_gpu
 * (airSonar_optimized.py:431-475, :642-654; calculate_signal_quality :477-490; scipy.signal.hilbert,
 * uniform_filter1d, find_peaks(height = mean + 2 std, distance = L // 50, prominence = std)).
 * corr[B][nb][L] is the raw correlation and is not modified.  Outputs, each optional (NULL to skip):
 * envelope[B][nb][L] smoothed envelope; peak_idx[B][nb] = valid_peaks[argmax(envelope[valid_peaks])] as an index into
 * corr, or -1 when no peak lies at or beyond min_delay (the caller then falls back to first_strong_peak, :657);
 * snr_db[B][nb] (:652-654); quality[B][nb] = corr_quality; n_peaks[B][nb] = len(corr_peaks);
 * peaks[B][nb][ECHO_ENV_MAX_PEAKS] = corr_peaks in ascending order, padded with -1. */
#define ECHO_ENV_MAX_PEAKS 64
int echo_envelope_detection(EchoHandle* h, const double* corr, double* envelope, int32_t* peak_idx, double* snr_db,
                            double* quality, int32_t* n_peaks, int32_t* peaks, int B, void* stream);

/* gpu_envelope_detection(signal, 'hilbert', smooth_window) on arbitrary rows (airSonar_optimized.py:431-475), e.g. the
 * band-signal call with smooth_window = 3 (:636, :708): x[rows][len] float64, len even; smooth_window odd, 1..9.
 * Outputs as echo_envelope_detection, per row; peak_idx / snr_db apply `min_delay` the way :647-654 do (pass 0 when
 * only envelope, peaks and quality are wanted).  Rows of 3072 to 4608 samples run the Hilbert transform as one real cyclic
 * convolution of length 9216; other lengths as two half-length transforms (Bluestein convolutions where the half has a
 * large prime factor).  ECHO_ERR_UNSUPPORTED: odd or very short len, even smooth_window, or a
 * transform that does not fit one workgroup's LDS (len beyond about 4700).  Allocates its plan on first use. */
int echo_envelope_rows(EchoHandle* h, const double* x, int len, int smooth_window, int min_delay, double* envelope,
                       int32_t* peak_idx, double* snr_db, double* quality, int32_t* n_peaks, int32_t* peaks, int rows,
                       void* stream);

/* gpu_argmax (airSonar_optimized.py:359-368): first index of the maximum of each row of x[rows][n]. */
int echo_argmax(const double* x, int32_t* out, int rows, int n, void* stream);

/* calculate_band_confidence + range conversion + normalize_confidences + np.average fusion
 * (airSonar_optimized.py:222-262, :662-667, :733-738).  Inputs [B][nb]; outputs per frame. */
int echo_confidence_fuse(EchoHandle* h, const int32_t* peak_idx, const double* snr_db, const double* amplitude,
                         EchoResult* out, int B, void* stream);

/* calculate_band_confidence elementwise (airSonar_optimized.py:222-243; realtime_sonar.py:393 when cfg.confidence is
 * ECHO_CONF_RAW): conf[i] from snr_db[i], amplitude[i] and band_idx[i] (0..2, selects the frequency weight). */
int echo_band_confidence(EchoHandle* h, const double* snr_db, const double* amplitude, const int32_t* band_idx,
                         double* conf, int n, void* stream);

/* normalize_confidences (airSonar_optimized.py:245-262) on rows of n confidences (the valid bands of a frame): each row
 * scaled to sum to 100, or 100 / n each when its sum is not positive. */
int echo_normalize_confidences(const double* conf, double* out, int rows, int n, void* stream);

/* ScalarKalman(q, r).update with explicit noise parameters (airSonar_optimized.py:265-283); otherwise as
 * echo_kalman_update, which uses the handle's configuration. */
int echo_kalman_update_qr(double* x, double* p, uint8_t* init, const double* z, const uint8_t* valid, double* out,
                          int S, int T, double q, double r, void* stream);

/* ScalarKalman.update for S independent streams (airSonar_optimized.py:265-283): T time steps,
 * z[t*S + s]; streams whose valid[t*S+s] == 0 are skipped at that step (run() only updates when a band
 * produced a range, :733).  State x/p/init live in caller-owned device arrays of length S. */
int echo_kalman_update(EchoHandle* h, double* x, double* p, uint8_t* init, const double* z, const uint8_t* valid,
                       double* out, int S, int T, void* stream);

/* The fused per-frame chain of SonarWorker.run() for B = n_streams * T frames, time-major
 * (frame i = stream i % n_streams at time i / n_streams): bandpass -> {rfft dB spectrum, correlate ->
 * first_strong_peak -> confidence} -> fusion -> Kalman (airSonar_optimized.py:629-676, :703-747, envelope
 * stages excluded).  spec_db may be NULL (no spectra stored); kalman_* may be NULL (record.kalman = fused).
 * Large calls (from 12288 frames) split themselves into chunks of 4096 frames that run on three streams
 * of the handle's own, forked from and joined to `stream` by events, with only the Kalman stages of the chunks in
 * order: the results are the bytes of the same call run pass after pass, the caller sees one call on one stream.
 * (ECHO_LANES=0 in the environment at echo_create keeps such calls on the caller's stream.)  The same holds for
 * echo_chain_full, echo_chain_wave and echo_chain_ordered. */
int echo_chain(EchoHandle* h, const float* rx, EchoResult* out, float* spec_db,
               double* kalman_x, double* kalman_p, uint8_t* kalman_init,
               int B, int n_streams, void* stream);

/* The whole of SonarWorker._process_band_gpu per band (airSonar_optimized.py:629-676), envelope branch included:
 * signal_quality from the Hilbert envelope of band_sig (:636), gpu_envelope_detection of the correlation (:642),
 * the envelope peak at or beyond min_delay as peak_idx (an index into corr) with the envelope SNR (:647-654), else
 * the first_strong_peak result (:657); amplitude = max|corr| of whichever corr that path leaves (:666); confidence
 * scaled by 0.5 + 0.5 * (signal_quality + corr_quality) / 2 (:670-671); then fusion and Kalman as echo_chain.
 * quality (optional) receives [B][nb][2] = signal_quality, corr_quality.  Other arguments as echo_chain.
 * For the reference's sizes (correlations of 3072 to 4608 lags) the band stage and the envelope branch are one kernel:
 * the correlation and the Hilbert transform run as real cyclic convolutions (values within about 1e-16 of the
 * correlation's maximum of the direct sums), which decides the envelope peaks exactly as SciPy does on every frame
 * checked (1.8 million frame-bands against the reference's chain); where the envelope branch finds no peak and the
 * record comes from first_strong_peak, that (frame, band) is evaluated by direct sums as echo_chain does. */
int echo_chain_full(EchoHandle* h, const float* rx, EchoResult* out, float* spec_db, double* quality,
                    double* kalman_x, double* kalman_p, uint8_t* kalman_init,
                    int B, int n_streams, void* stream);

/* echo_chain / echo_chain_full (full != 0) for callers that keep several batches in flight on several streams (one
 * handle per stream: a handle's scratch serves one call at a time).  Everything up to the fusion depends on the batch
 * alone; only the Kalman stage continues the state left by the batch before (ScalarKalman is sequential in time,
 * airSonar_optimized.py:265-283, :739).  The call makes that stage -- and nothing else -- wait for
 * kalman_wait_event (a hipEvent_t, or NULL for the first batch) and records kalman_done_event (hipEvent_t or NULL)
 * behind it, so batches submitted in order on different streams update one Kalman state in that order while their
 * other kernels overlap.  The Kalman pointers are required.  No counterpart in the reference (its loop is serial). */
int echo_chain_ordered(EchoHandle* h, const float* rx, EchoResult* out, float* spec_db, double* kalman_x,
                       double* kalman_p, uint8_t* kalman_init, int B, int n_streams, int full, void* stream,
                       void* kalman_wait_event, void* kalman_done_event);

/* One pass of SonarWorker.run()'s loop body with everything it hands on (airSonar_optimized.py:703-757): the records
 * of echo_chain_full (what the log line :741-743, distanceSig :745 and the CSV row :746-747 are made of) plus the
 * waveSig arrays of every frame: spec_db[B][nb][N/2+1] ('band_spectra', :710, :752) and corr[B][nb][N+M-1], the raw
 * correlations ('correlations', :712, :753; float64, unmodified: the envelope path of _process_band_gpu does not write
 * into its corr, and process_band_with_output keeps its own; they equal echo_correlate's direct sums to about 1e-16 of
 * each row's maximum, and exactly where the record comes from first_strong_peak).  Both are required; other arguments
 * as echo_chain_full. */
int echo_chain_wave(EchoHandle* h, const float* rx, EchoResult* out, float* spec_db, double* corr, double* quality,
                    double* kalman_x, double* kalman_p, uint8_t* kalman_init, int B, int n_streams, void* stream);

/* echo_chain without any allocation (no counterpart in the reference).  For callers that cannot call hipMalloc: code
 * under HIP graph capture, tight serving loops, harnesses that forbid allocation.  Same arguments and results as
 * echo_chain; the scratch of the pass (IIR forward output, IIR result, float32 band signal, detector records, fused
 * ranges) lives in a file-scope device array of ECHO_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds
 * until the process ends.  B is limited to what fits: echo_noalloc_max_frames(h) (256 frames for the reference's
 * constants N = 4282, three bands, order 6); beyond it ECHO_ERR_TOO_LARGE and nothing runs.  The call never invokes
 * hipMalloc/hipFree, does not grow the handle's scratch, leaves a pending echo_chain_prefetch pending (the next
 * echo_chain with that rx still finds it) and records no event of the handle's, so it may be issued on a stream that is
 * being captured into a HIP graph.  It must be issued with the handle's device current.  Every call on a GPU shares that GPU's array: calls that
 * overlap on one GPU (different streams or handles) race; calls on one stream are ordered and safe. */
#define ECHO_NOALLOC_SCRATCH_BYTES (72u << 20)
int echo_chain_noalloc(EchoHandle* h, const float* rx, EchoResult* out, float* spec_db,
                       double* kalman_x, double* kalman_p, uint8_t* kalman_init,
                       int B, int n_streams, void* stream);
/* Largest B echo_chain_noalloc accepts for this handle's configuration (>= 0), or a negative ECHO_ERR_*. */
int echo_noalloc_max_frames(EchoHandle* h);

/* Synthetic rx frames generated in device memory (no counterpart in the reference, which records from a sound card;
 * the result has the form AudioIO.record delivers, airSonar_optimized.py:573: int16 samples / 2**15 as float32).
 * Frame first_frame + i is a pure function of (seed, first_frame + i): an echo of tx_mix[n_chirp] (the transmitted mix
 * / 32768, generate_chirps :157-165) at a range drawn from U[1.5, 14] m with amplitude U[0.05, 0.4], a second echo at
 * half amplitude in 10 % of the frames, Gaussian noise of sigma U[0.002, 0.02].  rx[B][n_rx] is written. */
int echo_synth_frames(EchoHandle* h, const float* tx_mix, float* rx, int B, long long first_frame,
                      unsigned long long seed, void* stream);

/* Pipelining across batches (no counterpart in the reference).  The zero-phase IIR is a sequential recurrence that
 * leaves most of the chip idle while the other kernels of the chain are arithmetic-bound, so the IIR of the next
 * batch can run under the kernels of the current one.  echo_chain_prefetch starts it on a stream owned by the
 * handle; the next echo_chain / echo_chain_full call with the same rx and B uses the result instead of filtering
 * again (any other chain call discards it).  The caller guarantees that rx[B][N] is complete in device memory at the
 * time of the call -- the prefetch is not ordered after work on a caller stream -- and unchanged until that chain
 * call has been issued.  B must fit one internal pass (16384 frames; 4096 for the long-frame configurations),
 * otherwise ECHO_ERR_UNSUPPORTED.  echo_prefetch_ms: device time of the most
 * recent prefetched IIR. */
int echo_chain_prefetch(EchoHandle* h, const float* rx, int B);
int echo_prefetch_ms(EchoHandle* h, float* iir_ms);

/* Streaming mode (realtime_sonar.py:300-350: one frame per stream per tick, Kalman state carried between ticks):
 * echo_graph_create records one echo_chain call with exactly these arguments into a HIP graph (stream capture on
 * `stream`, which must not be the null stream; nothing executes during the capture).  echo_graph_launch replays it:
 * per tick the caller refreshes the contents of `rx` (same device buffer) and launches.  The handle's scratch is
 * reserved for B frames before the capture; B must not exceed one internal pass (16384 frames).  Other calls on the
 * handle must not run concurrently with a replay (shared scratch).  A graph holds pointers into the handle's scratch:
 * while one exists, a call that would have to enlarge the scratch (more frames than any call before) returns
 * ECHO_ERR_INVALID -- size the handle with echo_reserve first -- and graphs are destroyed before their handle. */
typedef struct EchoGraph EchoGraph;
int echo_graph_create(EchoHandle* h, const float* rx, EchoResult* out, float* spec_db,
                      double* kalman_x, double* kalman_p, uint8_t* kalman_init,
                      int B, int n_streams, void* stream, EchoGraph** graph);
int echo_graph_launch(EchoGraph* graph, void* stream);
void echo_graph_destroy(EchoGraph* graph);

/* Kernel timing with HIP events recorded on the call's own stream: echo_set_profiling(h, slots) arms the next
 * `slots` echo_chain calls (0 disarms); echo_last_kernel_ms waits for them and returns the average device time
 * in milliseconds of the IIR kernel, the band kernel, the spectrum kernel and the fusion(+Kalman) kernels over the armed calls
 * (first internal pass of each call), then re-arms from slot 0.  No counterpart in the reference. */
int echo_set_profiling(EchoHandle* h, int slots);
/* As echo_set_profiling, but only every `every`-th chain call is timed (the first one is): the event records between
 * the kernels keep a call's kernels from following each other back to back, which costs a few per cent of a
 * pipelined loop's throughput when every call is timed. */
int echo_set_profiling_every(EchoHandle* h, int slots, int every);
int echo_last_kernel_ms(EchoHandle* h, float* iir_ms, float* band_ms, float* spec_ms, float* fuse_ms);
/* The same averages as an array, with the band stage split where it runs as two kernels: ms[0..5] = IIR, band stage,
 * spectrum, fusion(+Kalman), pair kernel (float32 matched filter), exact kernel (float64 re-evaluation); the last two
 * are 0 when the band stage is the single-frame kernel.  Entries beyond 5 are set to 0. */
int echo_kernel_times(EchoHandle* h, float* ms, int n);
/* Scheduling diagnostic, no counterpart in the reference: the number of internal passes of this handle (its internal
 * lanes included) whose IIR ran on the device's shared, compute-unit-confined stream since echo_create.  The library
 * takes that route by itself for long-frame configurations and for echo_chain_full when the call is one pass of at
 * least 2048 frames and a batch of another handle is still in flight on the device; results do not depend on it.
 * -1 for a null handle. */
long long echo_confined_iir_passes(EchoHandle* h);

#ifdef __cplusplus
}
#endif
#endif /* ECHO_HIP_H */
