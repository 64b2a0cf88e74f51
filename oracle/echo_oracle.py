"""CPU oracle for the echo-ranging hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product (``dsp_amd``) never does: it fails loudly when the HIP library is missing.

What it is: a restatement, one frame at a time, of the per-frame chain of the reference's
``airSonar_optimized.py`` (CPU branch: ``GPU_AVAILABLE = False``), plus the variant detectors of
``airSonar.py`` and ``realtime_sonar.py``.  The arithmetic itself lives in third-party wheels the
reference imports without pinning (``scipy.signal.filtfilt/fftconvolve/correlate/hilbert/find_peaks``,
``scipy.ndimage.uniform_filter1d``, ``numpy.fft``); the oracle calls the same routines in the same order
(scipy 1.15.3 / numpy 2.2.6 in this image) and additionally spells out, in ``filtfilt_df2t``, the
published algorithm of ``filtfilt`` that the GPU kernel follows operation by operation.

Parity pin: the reference has no tests or fixtures.  The oracle is pinned by golden vectors produced
by importing the reference itself (``tools/make_goldens.py`` -> ``tests/golden/*.npz``), checked in
``tests/test_oracle_golden.py``.

``cfg`` arguments are duck-typed: any object with the attributes of ``dsp_amd.config.SonarConfig``.
"""
from __future__ import annotations

import numpy as np
from scipy.signal import correlate, fftconvolve, filtfilt, find_peaks, hilbert, lfilter_zi
from scipy.ndimage import uniform_filter1d

DET_MEDIAN, DET_MAD, DET_MEANPOW, DET_TWOPEAK = 0, 1, 2, 3
CONF_CLAMPED, CONF_RAW, CONF_NONE = 0, 1, 2
FUSE_WEIGHTED_EPS, FUSE_WEIGHTED, FUSE_MEAN, FUSE_SNRNORM = 0, 1, 2, 3


# --------------------------------------------------------------------------- filtering
def bandpass(sig, filt):
    """airSonar_optimized.py:185-197 (CPU branch; same code at :324-327): zero-phase IIR then FIR 'same'."""
    b, a, taps = filt
    y = filtfilt(b, a, sig)
    return fftconvolve(y, taps, mode='same')


def lfilter_df2t(b, a, x, zi):
    """Direct-form-II-transposed recurrence exactly as scipy.signal.lfilter evaluates it for float64
    (one multiply and one add per term, no fused multiply-add, this association order):
        y[n]  = z[0] + b[0]*x[n]
        z[k]  = (z[k+1] + x[n]*b[k+1]) - y[n]*a[k+1]      k = 0 .. na-3
        z[na-2] = x[n]*b[na-1] - y[n]*a[na-1]
    Bit-identical to scipy on this image (tests/test_oracle_golden.py)."""
    na = len(a)
    z = np.array(zi, dtype=np.float64)
    y = np.empty(len(x), dtype=np.float64)
    for i in range(len(x)):
        xn = x[i]
        yn = z[0] + b[0] * xn
        for k in range(na - 2):
            z[k] = z[k + 1] + xn * b[k + 1] - yn * a[k + 1]
        z[na - 2] = xn * b[na - 1] - yn * a[na - 1]
        y[i] = yn
    return y


def filtfilt_df2t(b, a, x):
    """scipy.signal.filtfilt(b, a, x) with its defaults (padtype='odd', padlen=3*max(len(a),len(b)),
    method='pad'), written out.  The odd extension is evaluated in the dtype of x (float32 for rx
    frames); everything after it is float64.  This is the algorithm the IIR kernel implements."""
    b = np.asarray(b, dtype=np.float64); a = np.asarray(a, dtype=np.float64)
    x = np.asarray(x)
    padlen = 3 * max(len(a), len(b))
    left = 2 * x[0] - x[padlen:0:-1]
    right = 2 * x[-1] - x[-2:-padlen - 2:-1]
    ext = np.concatenate((left, x, right))
    zi = lfilter_zi(b, a)
    y = lfilter_df2t(b, a, ext.astype(np.float64), zi * ext[0])
    y2 = lfilter_df2t(b, a, y[::-1], zi * y[-1])
    return y2[::-1][padlen:-padlen]


def fir_same_direct(y, taps):
    """np.convolve(y, taps)[(K-1)//2 : (K-1)//2 + N] -- what fftconvolve(..., 'same') equals up to FFT rounding."""
    k = len(taps)
    full = np.convolve(np.asarray(y, dtype=np.float64), np.asarray(taps, dtype=np.float64))
    s = (k - 1) // 2
    return full[s:s + len(y)]


# --------------------------------------------------------------------------- spectrum / correlation
def mag2db(x):
    """airSonar_optimized.py:286-287."""
    return 20 * np.log10(np.maximum(np.abs(x), 1e-12))


def spectrum_db(band_sig):
    """airSonar_optimized.py:710: mag2db(np.fft.rfft(band_sig))."""
    return mag2db(np.fft.rfft(band_sig))


def gpu_spectrum(signal):
    """airSonar_optimized.py:332-343 CPU branch."""
    return np.abs(np.fft.rfft(signal))


def gpu_correlate(a, b):
    """airSonar_optimized.py:290-303 CPU branch: scipy.signal.correlate(a, b, 'full')."""
    return correlate(a, b, 'full')


def gpu_argmax(signal):
    """airSonar_optimized.py:359-368 CPU branch."""
    return np.argmax(signal)


# --------------------------------------------------------------------------- detectors
def first_strong_peak(corr, fs, cfg, min_delay_samples=None):
    """airSonar_optimized.py:200-219.  Mutates ``corr`` (blind zone zeroed through the view)."""
    if min_delay_samples is None:
        min_delay_samples = int(fs * cfg.CHIRP_LEN * 1.2)
    half = corr.size // 2
    pos = corr[half:]
    if pos.size <= min_delay_samples:
        return None, 0.0
    pos[:min_delay_samples] = 0
    noise_samples = int(cfg.SNR_NOISE_MS * fs)
    noise_floor = np.median(np.abs(pos[min_delay_samples:min_delay_samples + noise_samples])) ** 2
    peak_idx = gpu_argmax(pos)
    peak_power = pos[peak_idx] ** 2
    if noise_floor > 0:
        snr_db = 10 * np.log10(peak_power / noise_floor)
    else:
        snr_db = 0.0
    if snr_db < 6.0:
        return None, 0.0
    return peak_idx, snr_db


def first_strong_peak_meanpow(corr, fs, cfg, min_delay_samples=None):
    """realtime_sonar.py:127-142: noise floor = mean(pos[:20 %]**2) including the zeroed blind zone."""
    if min_delay_samples is None:
        min_delay_samples = int(fs * cfg.CHIRP_LEN * 1.2)
    half = corr.size // 2
    pos = corr[half:]
    if pos.size <= min_delay_samples:
        return None, 0.0
    pos[:min_delay_samples] = 0
    noise_samples = int(len(pos) * 0.2)
    noise_floor = float(np.mean(pos[:noise_samples] ** 2))
    peak_idx = int(np.argmax(pos))
    peak_power = pos[peak_idx] ** 2
    snr_db = 10 * np.log10(peak_power / noise_floor) if noise_floor > 0 else 0.0
    if snr_db < 6.0:
        return None, 0.0
    return peak_idx, snr_db


def first_strong_peak_mad(corr, fs, min_delay_samples=500):
    """airSonar.py:104-117: first index above median + 6*MAD; returns int or None."""
    half = corr.size // 2
    pos = corr[half:]
    if pos.size <= min_delay_samples:
        return None
    pos[:min_delay_samples] = 0
    med = np.median(pos)
    mad = np.median(np.abs(pos - med)) + 1e-9
    noise_thr = med + 6 * mad
    idxs = np.where(pos > noise_thr)[0]
    if idxs.size == 0:
        return None
    return int(idxs[0])


def first_strong_peak_twopeak(corr, fs):
    """666.py:153-200: the strongest sample of pos = corr[half:] is the direct path (time zero); it is zeroed in a
    copy and the strongest remaining sample is the echo.  Noise floor = mean power of the first 20 % of that copy,
    6 dB gate.  Returns (delay_samples = echo index - direct index, snr_db) or (None, 0.0)."""
    half = corr.size // 2
    pos = corr[half:]
    if pos.size == 0:
        return None, 0.0
    ref_idx = np.argmax(pos)
    pos2 = pos.copy()
    pos2[ref_idx] = 0
    if np.all(pos2 == 0):
        return None, 0.0
    peak_idx = np.argmax(pos2)
    noise_region = int(len(pos2) * 0.2)
    noise_floor = np.mean(pos2[:noise_region] ** 2) if noise_region > 0 else 0.0
    peak_power = pos2[peak_idx] ** 2
    snr_db = 10 * np.log10(peak_power / noise_floor) if noise_floor > 0 else 0.0
    if snr_db < 6.0:
        return None, 0.0
    return int(peak_idx - ref_idx), float(snr_db)


# --------------------------------------------------------------------------- confidence / fusion / Kalman
def calculate_band_confidence(snr, amplitude, band_idx, freq_weights=(0.8, 1.0, 0.9)):
    """airSonar_optimized.py:222-243."""
    snr_weight = min(snr / 10.0, 1.0)
    amp_weight = min(amplitude / 0.1, 1.0)
    freq_weight = freq_weights[band_idx]
    confidence = (snr_weight * 0.5 + amp_weight * 0.3 + freq_weight * 0.2)
    return min(confidence, 1.0)


def band_confidence_raw(snr, amplitude, band_idx, freq_weights=(0.8, 1.0, 0.9)):
    """realtime_sonar.py:393."""
    return min((snr / 10) * 0.5 + (amplitude / 0.1) * 0.3 + freq_weights[band_idx] * 0.2, 1.0)


def normalize_confidences(confidences):
    """airSonar_optimized.py:245-262."""
    confidences = np.array(confidences)
    total = np.sum(confidences)
    if total > 0:
        normalized = (confidences / total) * 100.0
    else:
        normalized = np.full_like(confidences, 100.0 / len(confidences))
    return normalized


class ScalarKalman:
    """airSonar_optimized.py:265-283."""

    def __init__(self, q=0.005, r=0.1):
        self.x = None
        self.p = 1.0
        self.q = q
        self.r = r

    def update(self, z):
        if self.x is None:
            self.x = z
            return z
        self.p += self.q
        k = self.p / (self.p + self.r)
        self.x += k * (z - self.x)
        self.p *= (1 - k)
        return self.x


def fuse(results, cfg):
    """Fusion over the bands that produced a range.  ``results`` = [(distance, confidence, snr)] of valid bands.
    airSonar_optimized.py:733-738 (weights + 1e-9), realtime_sonar.py:336-339 (no epsilon),
    airSonar.py:203-205 (plain mean).  Returns (weighted_dist, avg_confidence, conf_norm) or None."""
    if not results:
        return None
    distances, confidences, snrs = zip(*results)
    if cfg.FUSION == FUSE_MEAN:
        return float(np.mean(distances)), 0.0, np.zeros(len(distances))
    if cfg.FUSION == FUSE_SNRNORM:                       # 666.py:372-385
        snr_array = np.array(snrs)
        normalized = snr_array / np.max(snr_array) * 100 if np.max(snr_array) > 0 else np.zeros_like(snr_array)
        weights = normalized + 1e-9
        return float(np.average(distances, weights=weights)), float(np.mean(normalized)), normalized
    conf_norm = normalize_confidences(confidences)
    weights = conf_norm / 100.0
    if cfg.FUSION == FUSE_WEIGHTED_EPS:
        weights = weights + 1e-9
    weighted_dist = np.average(distances, weights=weights)
    return float(weighted_dist), float(np.mean(conf_norm)), conf_norm


# --------------------------------------------------------------------------- envelope branch
def gpu_hilbert(signal):
    """airSonar_optimized.py:392-419 CPU branch."""
    return hilbert(signal)


def gpu_envelope(signal):
    """airSonar_optimized.py:421-424."""
    return np.abs(np.asarray(gpu_hilbert(signal)))


def calculate_signal_quality(signal, envelope):
    """airSonar_optimized.py:477-490."""
    signal_power = np.mean(signal ** 2)
    noise_estimate = np.std(signal - np.mean(signal))
    snr = 10 * np.log10(signal_power / (noise_estimate ** 2 + 1e-10))
    envelope_smooth = np.std(np.diff(envelope)) / (np.std(envelope) + 1e-10)
    dynamic_range = (np.max(envelope) - np.min(envelope)) / (np.mean(envelope) + 1e-10)
    snr_score = np.clip(snr / 20.0, 0, 1)
    smooth_score = np.clip(1.0 - envelope_smooth, 0, 1)
    dynamic_score = np.clip(dynamic_range / 2.0, 0, 1)
    quality = (snr_score * 0.5 + smooth_score * 0.3 + dynamic_score * 0.2)
    return float(np.clip(quality, 0, 1))


def gpu_envelope_detection(signal, smooth_window=5):
    """airSonar_optimized.py:431-475, method='hilbert'."""
    envelope = gpu_envelope(signal)
    if smooth_window > 1:
        envelope = uniform_filter1d(envelope, size=smooth_window)
    threshold = np.mean(envelope) + 2 * np.std(envelope)
    peaks, _ = find_peaks(envelope, height=threshold, distance=len(envelope) // 50,
                          prominence=np.std(envelope))
    quality = calculate_signal_quality(signal, envelope)
    return envelope, peaks, quality


# --------------------------------------------------------------------------- per-band / per-frame chain
def process_band(rx, chirp_sig, filt, band_idx, cfg, mode="named"):
    """One band of one frame.  mode="full" is ``SonarWorker._process_band_gpu``
    (airSonar_optimized.py:629-676) including the envelope branch; mode="named" is the same function
    with the envelope stages left out, i.e. its ``first_strong_peak`` path (:657-667) and base confidence.
    Returns dict with every intermediate the parity tests compare."""
    fs = cfg.FS
    out = {}
    band_sig = bandpass(rx, filt)
    out["band_sig"] = band_sig
    out["spec_db"] = spectrum_db(band_sig)
    corr = gpu_correlate(band_sig, chirp_sig)
    out["corr_raw"] = corr.copy()
    min_delay = cfg.min_delay
    peak_idx = None; snr = 0.0
    used_envelope = False
    signal_quality = corr_quality = 0.0
    if mode == "full":
        _, _, signal_quality = gpu_envelope_detection(band_sig, smooth_window=3)
        corr_envelope, corr_peaks, corr_quality = gpu_envelope_detection(corr, smooth_window=5)
        out["corr_envelope"] = corr_envelope
        out["corr_peaks"] = corr_peaks
        valid_peaks = corr_peaks[corr_peaks >= min_delay]
        if len(valid_peaks) > 0:
            peak_idx = valid_peaks[np.argmax(corr_envelope[valid_peaks])]
            signal_power = corr_envelope[peak_idx] ** 2
            noise_power = np.mean(corr_envelope[:min_delay] ** 2) + 1e-10
            snr = 10 * np.log10(signal_power / noise_power)
            used_envelope = True
    if not used_envelope:
        if cfg.DETECTOR == DET_MEDIAN:
            peak_idx, snr = first_strong_peak(corr, fs, cfg, min_delay)
        elif cfg.DETECTOR == DET_MEANPOW:
            peak_idx, snr = first_strong_peak_meanpow(corr, fs, cfg, min_delay)
        elif cfg.DETECTOR == DET_TWOPEAK:
            peak_idx, snr = first_strong_peak_twopeak(corr, fs)
        else:
            peak_idx = first_strong_peak_mad(corr, fs, min_delay)
            snr = 0.0
    out["corr_after"] = corr
    out["signal_quality"] = signal_quality
    out["corr_quality"] = corr_quality
    out["used_envelope"] = used_envelope
    out["amplitude"] = float(np.max(np.abs(corr)))
    if peak_idx is None:
        out.update(peak_idx=-1, snr_db=0.0, distance=None, confidence=0.0)
        return out
    delay = peak_idx - cfg.range_offset
    distance = delay / fs * cfg.sound_speed / 2
    amplitude = out["amplitude"]
    if cfg.CONFIDENCE == CONF_CLAMPED:
        conf = calculate_band_confidence(snr, amplitude, band_idx, cfg.FREQ_WEIGHTS)
    elif cfg.CONFIDENCE == CONF_RAW:
        conf = band_confidence_raw(snr, amplitude, band_idx, cfg.FREQ_WEIGHTS)
    else:
        conf = 0.0
    if mode == "full":
        conf = conf * (0.5 + 0.5 * (signal_quality + corr_quality) / 2.0)
    out.update(peak_idx=int(peak_idx), snr_db=float(snr), distance=float(distance), confidence=float(conf))
    return out


def process_frame(rx, chirps, filters, cfg, mode="named"):
    """All bands + fusion of one frame (airSonar_optimized.py:717-739, without threads or the duplicated work)."""
    bands = [process_band(rx, chirps[i], filters[i], i, cfg, mode) for i in range(len(filters))]
    results = [(b["distance"], b["confidence"], b["snr_db"]) for b in bands if b["distance"] is not None]
    fused = fuse(results, cfg)
    return bands, fused


def chain(rx_batch, chirps, filters, cfg, n_streams=None, mode="named", keep=()):
    """Batched driver: frames are time-major, frame i belongs to stream i % n_streams at time i // n_streams.
    Returns a dict of arrays shaped like the GPU result record."""
    nb = len(filters)
    B = len(rx_batch)
    S = n_streams or B
    res = {
        "peak_idx": np.full((B, nb), -1, dtype=np.int32),
        "snr_db": np.zeros((B, nb)), "amplitude": np.zeros((B, nb)),
        "distance": np.zeros((B, nb)), "confidence": np.zeros((B, nb)),
        "valid": np.zeros((B, nb), dtype=bool),
        "fused_dist": np.zeros(B), "avg_conf": np.zeros(B), "valid_any": np.zeros(B, dtype=bool),
        "kalman": np.zeros(B),
    }
    kept = {k: [] for k in keep}
    kfs = [ScalarKalman(cfg.KALMAN_Q, cfg.KALMAN_R) for _ in range(S)]
    for i in range(B):
        bands, fused = process_frame(rx_batch[i], chirps, filters, cfg, mode)
        for j, b in enumerate(bands):
            res["peak_idx"][i, j] = b["peak_idx"]
            res["snr_db"][i, j] = b["snr_db"]
            res["amplitude"][i, j] = b["amplitude"]
            res["confidence"][i, j] = b["confidence"]
            if b["distance"] is not None:
                res["distance"][i, j] = b["distance"]; res["valid"][i, j] = True
        for k in keep:
            kept[k].append(np.stack([b[k] for b in bands]))
        kf = kfs[i % S]
        if fused is not None:
            res["fused_dist"][i] = fused[0]; res["avg_conf"][i] = fused[1]; res["valid_any"][i] = True
            res["kalman"][i] = kf.update(fused[0])
        else:
            res["kalman"][i] = kf.x if kf.x is not None else 0.0
    for k in keep:
        res[k] = np.stack(kept[k])
    return res
