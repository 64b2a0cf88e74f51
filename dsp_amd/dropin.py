"""Drop-in spellings of the reference's module-level DSP functions.

Same names, argument order and call shapes as airSonar_optimized.py:157-368, :431-490: what
``SonarWorker._process_band_gpu`` (:629-676) and ``process_band_with_output`` (:703-715) pass -- one band's 1-D
signal, a chirp, a ``(b, a, taps)`` tuple, scalars -- works unchanged, and every function also takes leading batch
dimensions.  NumPy in -> NumPy out (upload / compute / download, like the reference's CuPy wrappers); torch CUDA
tensors in -> torch CUDA tensors out.  All arithmetic runs in the HIP library; a mis-shaped argument raises
``EchoError`` instead of reaching a kernel.

A process-wide default engine is created lazily on the current GPU; ``use_engine`` swaps it.

Threads: the reference submits its per-band functions from ``ThreadPoolExecutor(max_workers=3)``
(airSonar_optimized.py:717-719), so every function here may be called from several threads at once.  The engine cache
below is guarded by a lock, and the library serialises the calls on one handle itself (include/echo_hip.h, "Threads and
streams"); each call works on arrays of its own, as in the reference.
"""
from __future__ import annotations

import threading
from dataclasses import replace
from typing import Optional

import numpy as np
import torch
from scipy.signal import lfilter_zi

from ._lib import EchoError
from .config import SonarConfig, optimized_config
from .design import SonarTables, calculate_sound_speed, design_filters as _design_filters, generate_chirps as _generate_chirps  # noqa: F401
from .engine import BatchedKalman, EchoEngine, records_to_numpy

_engine: Optional[EchoEngine] = None
_filter_engines = {}          # engines for (b, a, taps) tuples that are not one of the default engine's bands
_cache_lock = threading.RLock()   # guards the two caches above: three pool threads may arrive at a cold cache together


def get_engine(cfg: Optional[SonarConfig] = None) -> EchoEngine:
    global _engine
    e = _engine
    if e is not None and (cfg is None or cfg == e.cfg):
        return e
    with _cache_lock:
        if _engine is None or (cfg is not None and cfg != _engine.cfg):
            use_engine(EchoEngine(cfg or optimized_config()))
        return _engine


def use_engine(engine: EchoEngine):
    global _engine
    with _cache_lock:
        if engine is not _engine:
            # dropped, not closed: another thread may be inside a call on one of them; an engine closes itself when
            # the last reference to it goes
            _filter_engines.clear()
        _engine = engine


def _is_np(x):
    return not isinstance(x, torch.Tensor)


def _ret(t, like_np):
    return t.cpu().numpy() if like_np else t


def _rows(x, e: EchoEngine, dtype, what):
    """Any-rank input -> ([rows, n] device tensor, leading shape)."""
    t = e._dev(x, dtype)
    if t.dim() < 1 or t.shape[-1] < 1:
        raise EchoError(f"{what}: expected a signal along the last axis, got shape {list(t.shape)}")
    lead = tuple(t.shape[:-1])
    return t.reshape(-1, t.shape[-1]), lead


# --------------------------------------------------------------------------- set-up
def design_filters(fs=None):
    """airSonar_optimized.py:167-183: list of (b, a, taps), one per band, for sample rate ``fs`` (default: the
    engine's)."""
    e = get_engine()
    if fs is None or fs == e.cfg.FS:
        return e.tables.filters
    return _design_filters(e.cfg, fs=fs)


def generate_chirps(fs=None, duration=None):
    """airSonar_optimized.py:157-165: (tx_pcm int16, [chirp float32])."""
    e = get_engine()
    if (fs is None or fs == e.cfg.FS) and (duration is None or duration == e.cfg.CHIRP_LEN):
        return e.tables.tx_pcm, e.tables.chirps_f32
    return _generate_chirps(e.cfg, fs=fs, duration=duration)


# --------------------------------------------------------------------------- filtering
def _band_of(filt, e: EchoEngine):
    if len(filt) != 3:
        raise EchoError("filt must be a (b, a, taps) tuple")
    for j, f in enumerate(e.tables.filters):
        if filt is f or all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(filt, f)):
            return j
    return None


def _engine_for_filter(filt, e: EchoEngine) -> EchoEngine:
    """A one-band engine around a caller-supplied (b, a, taps): the same kernels with these coefficients.  The library
    takes zero-phase IIRs of order 6 and 8 (13 / 17 coefficients) and the identity filter ([1], [1], [1]) that
    design_filters falls back to (airSonar_optimized.py:178-182); anything else raises EchoError."""
    b, a, taps = (np.atleast_1d(np.asarray(x, dtype=np.float64)) for x in filt)
    key = (b.tobytes(), a.tobytes(), taps.tobytes())
    with _cache_lock:
        return _filter_engines.get(key) or _make_filter_engine(key, b, a, taps, e)


def _make_filter_engine(key, b, a, taps, e: EchoEngine) -> EchoEngine:
    na = max(len(a), len(b))
    bb = np.zeros(na); aa = np.zeros(na)
    bb[:len(b)] = b; aa[:len(a)] = a
    if aa[0] == 0:
        raise EchoError("filt: a[0] must not be zero")
    if aa[0] != 1.0:                                   # scipy.signal.lfilter normalises by a[0]
        bb = bb / aa[0]; aa = aa / aa[0]
    zi = lfilter_zi(bb, aa) if na > 1 else np.zeros(0)
    cfg1 = replace(e.cfg, BANDS=(e.cfg.BANDS[0],), IIR_ORDER=(na - 1) // 2, FIR_TAPS=len(taps))
    t0 = e.tables
    tables = SonarTables(cfg=cfg1, b=bb[None], a=aa[None], zi=zi[None], taps=taps[None], chirps=t0.chirps[:1],
                         tx_pcm=t0.tx_pcm, filters=[(b, a, taps)], chirps_f32=t0.chirps_f32[:1])
    if len(_filter_engines) >= 8:                      # bounded cache; the evicted engine closes with its last user
        _filter_engines.pop(next(iter(_filter_engines)))
    eng = EchoEngine(cfg1, e.device, tables=tables)
    _filter_engines[key] = eng
    return eng


def bandpass_bank(rx):
    """All bands at once: rx [B,N] (or [N]) -> band_sig [B,nb,N] (or [nb,N])."""
    e = get_engine()
    rows, lead = _rows(rx, e, torch.float32, "bandpass_bank")
    out = e.bandpass_bank(rows)
    return _ret(out.reshape(lead + tuple(out.shape[1:])), _is_np(rx))


def bandpass(sig, filt):
    """airSonar_optimized.py:185-197: filtfilt(b, a, sig) then fftconvolve(y, taps, 'same').  ``sig`` [N] (the
    reference's call) or [..., N]; ``filt`` one of design_filters()'s entries or any (b, a, taps) the library holds."""
    e = get_engine()
    rows, lead = _rows(sig, e, torch.float32, "bandpass")
    j = _band_of(filt, e)
    if j is None:
        out = _engine_for_filter(filt, e).bandpass_bank(rows)[:, 0]
    else:
        out = e.bandpass_bank(rows)[:, j]
    return _ret(out.reshape(lead + (out.shape[-1],)), _is_np(sig))


gpu_bandpass = bandpass


# --------------------------------------------------------------------------- correlation / spectrum
def gpu_correlate(a, b=None):
    """airSonar_optimized.py:290-303: scipy.signal.correlate(a, b, 'full').  ``a`` [n] with ``b`` [m] (the reference's
    per-band call, :639, :712) or ``a`` [..., n] with one ``b`` for every row.  With ``b`` omitted ``a`` is a whole
    bank [..., nb, N] and every band is correlated with its own chirp."""
    e = get_engine()
    if b is None:
        t = e._dev(a, torch.float64)
        if t.dim() < 2:
            raise EchoError("gpu_correlate(a): without b, a must be a bank [..., n_bands, n_rx]")
        lead = tuple(t.shape[:-2])
        out = e.correlate(t.reshape((-1,) + tuple(t.shape[-2:])))
        return _ret(out.reshape(lead + tuple(out.shape[1:])), _is_np(a))
    rows, lead = _rows(a, e, torch.float64, "gpu_correlate")
    bt = e._dev(b, torch.float64)
    if bt.dim() != 1:
        raise EchoError(f"gpu_correlate: b must be one-dimensional, got shape {list(bt.shape)}")
    out = e.xcorr(rows, bt)
    return _ret(out.reshape(lead + (out.shape[-1],)), _is_np(a))


def _spectrum(signal, db, what):
    e = get_engine()
    rows, lead = _rows(signal, e, torch.float64, what)
    # the engine's own frame length runs on its tables; any other length on a plan of its own (echo_spectrum_any)
    out = e.spectrum_rows(rows, db=db) if rows.shape[1] == e.cfg.n_rx else e.spectrum_any(rows, db=db)
    return _ret(out.reshape(lead + (out.shape[-1],)), _is_np(signal))


def gpu_spectrum(signal):
    """airSonar_optimized.py:332-343: |rfft(signal)|, signal [n] or [..., n], any n (float32 result)."""
    return _spectrum(signal, False, "gpu_spectrum")


def spectrum_db(band_sig):
    """mag2db(np.fft.rfft(band_sig)), airSonar_optimized.py:710; any length."""
    return _spectrum(band_sig, True, "spectrum_db")


def mag2db(x):
    """airSonar_optimized.py:286-287 (float32 on the device)."""
    e = get_engine()
    if _is_np(x):
        return e.mag2db(np.abs(np.asarray(x)).astype(np.float32)).cpu().numpy()
    return e.mag2db(x.abs() if x.is_complex() else x)


gpu_mag2db = mag2db


def gpu_argmax(signal):
    """airSonar_optimized.py:359-368; the last axis is reduced."""
    e = get_engine()
    out = e.argmax(signal)
    if _is_np(signal):
        out = out.cpu().numpy()
        return int(out[0]) if np.ndim(signal) == 1 else out
    return out


# --------------------------------------------------------------------------- detectors
def first_strong_peak(corr, fs=None, min_delay_samples=None):
    """airSonar_optimized.py:200-219 with the reference's own arguments: ``corr`` [L] (one band, the reference's call) or
    [..., L], any L; ``fs`` (default: the engine's) sets the noise window int(SNR_NOISE_MS * fs) and, when
    ``min_delay_samples`` is None, the blind zone int(fs * CHIRP_LEN * 1.2).  The scalar call returns
    (peak_idx | None, snr_db); batched calls return (int32 array with -1 for None, snr array) with the leading axes.
    The blind zone of ``corr`` is zeroed in place, as in the reference (not when pos.size <= min_delay_samples)."""
    e = get_engine()
    cfg = e.cfg
    fs = cfg.FS if fs is None else fs
    md = int(fs * cfg.CHIRP_LEN * 1.2) if min_delay_samples is None else int(min_delay_samples)
    if md < 0:
        raise EchoError("min_delay_samples must not be negative")     # (a negative slice bound means something else in NumPy)
    noise_n = int(cfg.SNR_NOISE_MS * fs)
    is_np = _is_np(corr)
    if is_np and not isinstance(corr, np.ndarray):
        raise EchoError("first_strong_peak writes into corr: pass a NumPy array or a torch tensor")
    if corr.ndim < 1 or corr.shape[-1] < 1:
        raise EchoError(f"corr must be [L] or [..., L], got {list(corr.shape)}")
    shape = tuple(corr.shape)
    dev = e._dev(corr, torch.float64)                    # a contiguous float64 tensor on the device is taken as it is
    in_place = (not is_np) and dev.data_ptr() == corr.data_ptr()
    peak, snr = e.first_strong_peak_rows(dev.reshape(-1, shape[-1]), md, noise_n, cfg.SNR_GATE_DB)
    if is_np:
        corr[...] = dev.cpu().numpy()
        peak = peak.cpu().numpy(); snr = snr.cpu().numpy()
    elif not in_place:
        corr.copy_(dev)
    if len(shape) == 1:
        p = int(peak[0]); sn = float(snr[0])
        return (None, 0.0) if p < 0 else (p, sn)
    return peak.reshape(shape[:-1]), snr.reshape(shape[:-1])


def gpu_envelope_detection(signal, method="hilbert", smooth_window=5):
    """airSonar_optimized.py:431-475: Hilbert envelope, ``uniform_filter1d(smooth_window)``,
    ``find_peaks(height = mean + 2 std, distance = len // 50, prominence = std)`` and ``calculate_signal_quality``.
    ``signal``: [len] -- the band signal with smooth_window 3 (:636, :708) or the correlation with 5 (:642) -- or
    [..., len].  Returns ``(envelope, peaks, quality)`` like the reference for a 1-D input; for batched input
    ``envelope`` and ``quality`` carry the leading axes and ``peaks`` is a flat list (row-major) of index arrays.
    Only method='hilbert' (the one the reference's live path uses) is implemented."""
    if method != "hilbert":
        raise EchoError("gpu_envelope_detection: only method='hilbert' is implemented")
    e = get_engine()
    like_np = _is_np(signal)
    rows, lead = _rows(signal, e, torch.float64, "gpu_envelope_detection")
    r = e.envelope_rows(rows, smooth_window=int(smooth_window), min_delay=0)
    env, q, pk = r["envelope"], r["quality"], r["peaks"].cpu().numpy()
    peaks = [row[row >= 0].astype(np.int64) for row in pk]
    env = _ret(env.reshape(lead + (env.shape[-1],)), like_np)
    if not lead:
        return env, peaks[0], float(q[0])
    return env, peaks, _ret(q.reshape(lead), like_np)


# --------------------------------------------------------------------------- confidence / fusion / Kalman
def calculate_band_confidence(snr, amplitude, band_idx):
    """airSonar_optimized.py:222-243; scalars (the reference's call) or arrays of equal shape."""
    e = get_engine()
    scalar = np.ndim(snr) == 0 and not isinstance(snr, torch.Tensor)
    like_np = _is_np(snr)
    s = e._dev(np.asarray(snr, dtype=np.float64) if like_np else snr, torch.float64)
    shape = tuple(s.shape)
    a = e._dev(np.broadcast_to(np.asarray(amplitude, dtype=np.float64), shape).copy() if _is_np(amplitude) else amplitude,
               torch.float64)
    j = e._dev(np.broadcast_to(np.asarray(band_idx, dtype=np.int32), shape).copy() if _is_np(band_idx) else band_idx,
               torch.int32)
    out = e.band_confidence(s, a, j).reshape(shape)
    if scalar:
        return float(out)
    return _ret(out, like_np)


def normalize_confidences(confidences):
    """airSonar_optimized.py:245-262: the confidences of the valid bands of one frame (list / 1-D), or rows of
    them [..., n]; each row is scaled to sum to 100."""
    e = get_engine()
    like_np = _is_np(confidences)
    rows, lead = _rows(np.asarray(confidences, dtype=np.float64) if like_np else confidences, e, torch.float64,
                       "normalize_confidences")
    out = e.normalize_confidences(rows)
    return _ret(out.reshape(lead + (out.shape[-1],)), like_np)


class ScalarKalman:
    """airSonar_optimized.py:265-283 with the same scalar interface and defaults, backed by a one-stream
    BatchedKalman (state in device memory)."""

    def __init__(self, q=0.005, r=0.1):
        self.q, self.r = q, r
        self._k = BatchedKalman(get_engine(), 1, q=q, r=r)

    @property
    def x(self):
        return float(self._k.x[0]) if int(self._k.init[0]) else None

    @property
    def p(self):
        return float(self._k.p[0])

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
