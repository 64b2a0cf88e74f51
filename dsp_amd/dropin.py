"""Drop-in spellings of the reference's module-level DSP functions, batched.

Same names, argument meaning and "never raise for a data condition" behaviour as
airSonar_optimized.py:185-368, with one difference that is the point of this package: the leading
dimension may be a batch of frames.  NumPy in -> NumPy out (upload/compute/download, like the reference's
CuPy wrappers); torch CUDA tensors in -> torch CUDA tensors out (no host round trip).

A process-wide default engine is created lazily on the current GPU; ``use_engine`` swaps it.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from .config import SonarConfig, optimized_config
from .design import calculate_sound_speed, design_filters as _design_filters, generate_chirps as _generate_chirps  # noqa: F401
from .engine import BatchedKalman, EchoEngine, records_to_numpy

_engine: Optional[EchoEngine] = None


def get_engine(cfg: Optional[SonarConfig] = None) -> EchoEngine:
    global _engine
    if _engine is None or (cfg is not None and cfg != _engine.cfg):
        _engine = EchoEngine(cfg or optimized_config())
    return _engine


def use_engine(engine: EchoEngine):
    global _engine
    _engine = engine


def _is_np(x):
    return isinstance(x, np.ndarray)


def _back(t, like_np, squeeze):
    if squeeze:
        t = t[0]
    return t.cpu().numpy() if like_np else t


def design_filters(fs=None):
    """airSonar_optimized.py:167-183: list of (b, a, taps), one per band."""
    return get_engine().tables.filters


def generate_chirps(fs=None, duration=None):
    """airSonar_optimized.py:157-165: (tx_pcm int16, [chirp float32])."""
    t = get_engine().tables
    return t.tx_pcm, t.chirps_f32


def _band_of(filt, e: EchoEngine) -> int:
    for j, f in enumerate(e.tables.filters):
        if filt is f or (len(filt) == 3 and np.array_equal(filt[2], f[2]) and np.array_equal(filt[1], f[1])):
            return j
    raise ValueError("filt is not one of this engine's design_filters() entries")


def bandpass_bank(rx):
    """All bands at once: rx [B,N] (or [N]) -> band_sig [B,nb,N] (or [nb,N])."""
    e = get_engine()
    sq = rx.ndim == 1
    out = e.bandpass_bank(rx[None] if sq else rx)
    return _back(out, _is_np(rx), sq)


def bandpass(sig, filt):
    """airSonar_optimized.py:185-197 for one band of design_filters(); sig [N] or [B,N]."""
    e = get_engine()
    j = _band_of(filt, e)
    sq = sig.ndim == 1
    out = e.bandpass_bank(sig[None] if sq else sig)[:, j]
    return _back(out, _is_np(sig), sq)


gpu_bandpass = bandpass


def gpu_correlate(a, b=None):
    """airSonar_optimized.py:290-303.  a = band_sig [..., nb, N]; each band is correlated with its own chirp
    (the reference passes the band's chirp as b; b is accepted and ignored when a carries all bands)."""
    e = get_engine()
    sq = a.ndim == 2
    out = e.correlate(a[None] if sq else a)
    return _back(out, _is_np(a), sq)


def gpu_spectrum(signal):
    """airSonar_optimized.py:332-343: |rfft| of band_sig [..., nb, N]."""
    e = get_engine()
    sq = signal.ndim == 2
    out = e.spectrum(signal[None] if sq else signal, db=False)
    return _back(out, _is_np(signal), sq)


def spectrum_db(band_sig):
    """mag2db(np.fft.rfft(band_sig)), airSonar_optimized.py:710."""
    e = get_engine()
    sq = band_sig.ndim == 2
    out = e.spectrum(band_sig[None] if sq else band_sig, db=True)
    return _back(out, _is_np(band_sig), sq)


def mag2db(x):
    """airSonar_optimized.py:286-287 (float32 on the device)."""
    e = get_engine()
    if _is_np(x):
        return e.mag2db(np.abs(x).astype(np.float32)).cpu().numpy()
    return e.mag2db(x.abs() if x.is_complex() else x)


gpu_mag2db = mag2db


def gpu_argmax(signal):
    """airSonar_optimized.py:359-368; the last axis is reduced."""
    e = get_engine()
    out = e.argmax(signal)
    if _is_np(signal):
        out = out.cpu().numpy()
        return int(out[0]) if signal.ndim == 1 else out
    return out


def first_strong_peak(corr, fs=None, min_delay_samples=None):
    """airSonar_optimized.py:200-219.  corr [L] for band 0, or [nb,L], or [B,nb,L].
    Scalar call returns (peak_idx | None, snr_db); batched calls return (int32 array with -1 for None, snr array).
    The blind zone of ``corr`` is zeroed in place, as in the reference."""
    e = get_engine()
    cfg = e.cfg
    if min_delay_samples is not None and int(min_delay_samples) != cfg.min_delay:
        raise ValueError("min_delay_samples differs from the engine's configuration")
    is_np = _is_np(corr)
    shape = corr.shape
    if corr.ndim == 1:
        full = np.zeros((1, cfg.n_bands, shape[0])) if is_np else torch.zeros((1, cfg.n_bands, shape[0]),
                                                                               dtype=torch.float64, device=e.device)
        full[0, 0] = corr
    elif corr.ndim == 2:
        full = corr[None]
    else:
        full = corr
    dev = e._dev(full, torch.float64)
    peak, snr, _ = e.first_strong_peak(dev)
    if is_np:
        mut = dev.cpu().numpy()
        if corr.ndim == 1:
            corr[...] = mut[0, 0]
        else:
            corr[...] = mut.reshape(shape)
        peak = peak.cpu().numpy(); snr = snr.cpu().numpy()
    elif dev.data_ptr() != corr.data_ptr():
        corr.copy_(dev[0, 0] if corr.ndim == 1 else dev.reshape(shape))
    if corr.ndim == 1:
        p = int(peak[0, 0]); s = float(snr[0, 0])
        return (None, 0.0) if p < 0 else (p, s)
    if corr.ndim == 2:
        return peak[0], snr[0]
    return peak, snr


def gpu_envelope_detection(signal, method="hilbert", smooth_window=5):
    """airSonar_optimized.py:431-475 for correlation-length signals (the call at :642): Hilbert envelope,
    ``uniform_filter1d(smooth_window)``, ``find_peaks(height = mean + 2 std, distance = len // 50, prominence = std)``
    and ``calculate_signal_quality``.  ``signal``: [L] for one band (band 0's slot is used; the stage does not depend
    on the band), or [nb, L], or [B, nb, L].  Returns ``(envelope, peaks, quality)`` like the reference for a 1-D
    input; for batched input ``peaks`` is a list (per row) of index arrays.  Only method='hilbert' with
    smooth_window=5 runs on the GPU (the band-signal call with window 3 only feeds signal_quality and lives inside
    ``process_frames(..., mode="full")``)."""
    if method != "hilbert" or smooth_window != 5:
        raise ValueError("gpu_envelope_detection: the GPU path implements method='hilbert', smooth_window=5")
    e = get_engine()
    like_np = _is_np(signal)
    x = torch.as_tensor(signal, dtype=torch.float64)
    nd = x.dim()
    nb, L = e.cfg.n_bands, e.cfg.n_corr
    if nd == 1:
        x3 = x.reshape(1, 1, L).expand(1, nb, L)
    elif nd == 2:
        x3 = x.reshape(1, nb, L)
    else:
        x3 = x
    r = e.envelope_detection(x3.contiguous())
    env, q, pk = r["envelope"], r["quality"], r["peaks"].cpu().numpy()
    peaks = [[row[row >= 0].astype(np.int64) for row in frame] for frame in pk]
    if like_np:
        env, q = env.cpu().numpy(), q.cpu().numpy()
    if nd == 1:
        return env[0, 0], peaks[0][0], float(q[0, 0])
    if nd == 2:
        return env[0], peaks[0], q[0]
    return env, peaks, q


def calculate_band_confidence(snr, amplitude, band_idx):
    """airSonar_optimized.py:222-243, elementwise on arrays (host arithmetic: three multiplies per element;
    the device path is ``EchoEngine.confidence_fuse``)."""
    fw = np.asarray(get_engine().cfg.FREQ_WEIGHTS)[band_idx]
    snr = np.asarray(snr, dtype=np.float64); amplitude = np.asarray(amplitude, dtype=np.float64)
    conf = np.minimum(snr / 10.0, 1.0) * 0.5 + np.minimum(amplitude / 0.1, 1.0) * 0.3 + fw * 0.2
    return np.minimum(conf, 1.0)


def normalize_confidences(confidences, valid=None):
    """airSonar_optimized.py:245-262 over the last axis; ``valid`` masks bands that produced no range."""
    c = np.array(confidences, dtype=np.float64)
    if valid is None:
        valid = np.ones_like(c, dtype=bool)
    c = np.where(valid, c, 0.0)
    total = c.sum(axis=-1, keepdims=True)
    n = np.maximum(valid.sum(axis=-1, keepdims=True), 1)
    return np.where(valid, np.where(total > 0, c / np.where(total > 0, total, 1.0) * 100.0, 100.0 / n), 0.0)


class ScalarKalman:
    """airSonar_optimized.py:265-283 with the same scalar interface, backed by a one-stream BatchedKalman."""

    def __init__(self, q=None, r=None):
        e = get_engine()
        if (q is not None and q != e.cfg.KALMAN_Q) or (r is not None and r != e.cfg.KALMAN_R):
            raise ValueError("q/r differ from the engine's configuration")
        self._k = BatchedKalman(e, 1)

    @property
    def x(self):
        return float(self._k.x[0]) if int(self._k.init[0]) else None

    def update(self, z):
        return float(self._k.update(np.array([z], dtype=np.float64))[0])


def process_frames(rx, n_streams=None, kalman: Optional[BatchedKalman] = None, want_spectra=True, mode="named"):
    """The whole per-frame chain of SonarWorker.run() (airSonar_optimized.py:703-747) for a batch; mode="full"
    includes the Hilbert-envelope branch of _process_band_gpu (:636-654, :670-671).
    Returns (structured records, spec_db or None)."""
    e = get_engine()
    rec, spec = e.chain(rx, n_streams=n_streams, kalman=kalman, want_spectra=want_spectra, mode=mode)
    if _is_np(rx):
        return records_to_numpy(rec), (spec.cpu().numpy() if spec is not None else None)
    return rec, spec
