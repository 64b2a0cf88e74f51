"""Batched echo-ranging engine: the host-side mirror of the reference's DSP function surface.

One ``EchoEngine`` owns one library handle on one GPU.  Its methods carry the reference's function names
with a leading batch dimension and take/return torch tensors resident in HBM (torch is used only for
device memory and streams).  The module-level functions at the bottom are the drop-in spellings
(``bandpass``, ``gpu_correlate``, ``first_strong_peak`` ...) that also accept host NumPy arrays the way
the reference's CuPy wrappers do (upload, compute, download) -- airSonar_optimized.py:185-368.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from .config import SonarConfig, optimized_config
from .design import SonarTables, build_tables

RESULT_DTYPE = np.dtype([
    ("dist", np.float32, (3,)), ("conf", np.float32, (3,)), ("snr_db", np.float32, (3,)),
    ("peak_idx", np.int32, (3,)), ("fused_dist", np.float32), ("avg_conf", np.float32),
    ("kalman", np.float32), ("valid_mask", np.uint32),
])
assert RESULT_DTYPE.itemsize == 64


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class EchoEngine:
    def __init__(self, cfg: Optional[SonarConfig] = None, device: Optional[torch.device] = None,
                 tables: Optional[SonarTables] = None):
        """``tables`` overrides the filters / chirps designed from ``cfg`` (e.g. a caller-supplied (b, a, taps), or the
        identity filter design_filters falls back to, airSonar_optimized.py:178-182); sizes are taken from it."""
        self.cfg = cfg or optimized_config()
        self.tables: SonarTables = tables if tables is not None else build_tables(self.cfg)
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.EchoError("no GPU visible: the echo-ranging kernels need a gfx950 device")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        c, t = self.cfg, self.tables
        self._keep = [np.ascontiguousarray(x, dtype=np.float64) for x in (t.b, t.a, t.zi, t.taps, t.chirps)]
        nb = t.b.shape[0]
        if nb != c.n_bands or t.chirps.shape != (nb, c.n_chirp) or t.a.shape != t.b.shape or t.taps.shape[0] != nb:
            raise _lib.EchoError("tables do not match the configuration (bands, chirp length)")
        cc = _lib.EchoConfigC(
            fs=c.FS, n_rx=c.n_rx, n_chirp=c.n_chirp, n_bands=c.n_bands, iir_ncoef=t.b.shape[1],
            n_taps=t.taps.shape[1], min_delay=c.min_delay, noise_n=c.noise_n, detector=c.DETECTOR,
            confidence=c.CONFIDENCE, fusion=c.FUSION, range_offset=c.range_offset, snr_gate_db=c.SNR_GATE_DB,
            sound_speed=c.sound_speed, kalman_q=c.KALMAN_Q, kalman_r=c.KALMAN_R)
        for j in range(3):
            cc.freq_weights[j] = c.FREQ_WEIGHTS[j]
        cc.b, cc.a, cc.zi, cc.taps, cc.chirps = (x.ctypes.data for x in self._keep)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.echo_create(C.byref(cc), C.byref(h)), "echo_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            with torch.cuda.device(self.device):
                self.lib.echo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _call(self, name, *args):
        """One library call with the engine's device current (the handle belongs to it)."""
        with torch.cuda.device(self.device):
            _lib.check(getattr(self.lib, name)(*args), name)

    def _dev(self, x, dtype):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x))
        if not isinstance(x, torch.Tensor):
            x = torch.as_tensor(x)
        return x.to(device=self.device, dtype=dtype).contiguous()

    def _in(self, x, dtype, shape, name):
        """Input tensor on the device, converted if needed, with its shape checked: -1 entries of ``shape`` are free
        (the batch).  The kernels index by the configured sizes, so a wrong trailing dimension must never reach them."""
        t = self._dev(x, dtype)
        self._shape(t, shape, name)
        return t

    @staticmethod
    def _shape(t, shape, name):
        if t.dim() != len(shape) or any(w >= 0 and int(g) != w for g, w in zip(t.shape, shape)):
            want = "[" + ", ".join("B" if w < 0 else str(w) for w in shape) + "]"
            raise _lib.EchoError(f"{name} must have shape {want}, got {list(t.shape)}")

    def _out(self, t, dtype, shape, name):
        """Caller-supplied output tensor: written in place, so it must already be exactly what the kernel writes."""
        if (not isinstance(t, torch.Tensor) or t.device != self.device or t.dtype != dtype or not t.is_contiguous()
                or tuple(t.shape) != tuple(shape)):
            raise _lib.EchoError(f"{name} must be a contiguous {dtype} tensor of shape {list(shape)} on {self.device}")
        return t

    def _empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def reserve(self, max_frames: int):
        self._call("echo_reserve", self._h, int(max_frames))

    # ------------------------------------------------------------------ stages (device tensors)
    def fir_bank(self, rx):
        """FIR only (the reference's CuPy gpu_bandpass, airSonar_optimized.py:305-313). [B,N] f32 -> [B,nb,N] f64."""
        rx = self._in(rx, torch.float32, (-1, self.cfg.n_rx), "rx")
        B = rx.shape[0]
        out = self._empty((B, self.cfg.n_bands, self.cfg.n_rx), torch.float64)
        self._call("echo_fir_bank", self._h, _ptr(rx), _ptr(out), B, self._stream())
        return out

    def bandpass_bank(self, rx):
        """bandpass for every band (airSonar_optimized.py:185-197). [B,N] f32 -> [B,nb,N] f64."""
        rx = self._in(rx, torch.float32, (-1, self.cfg.n_rx), "rx")
        B = rx.shape[0]
        out = self._empty((B, self.cfg.n_bands, self.cfg.n_rx), torch.float64)
        self._call("echo_bandpass_bank", self._h, _ptr(rx), _ptr(out), B, self._stream())
        return out

    def spectrum(self, band_sig, db=True):
        """|rfft| or its dB (airSonar_optimized.py:710, :332-343). [B,nb,N] f64 -> [B,nb,N/2+1] f32."""
        band_sig = self._in(band_sig, torch.float64, (-1, self.cfg.n_bands, self.cfg.n_rx), "band_sig")
        B = band_sig.shape[0]
        out = self._empty((B, self.cfg.n_bands, self.cfg.n_bins), torch.float32)
        self._call("echo_spectrum", self._h, _ptr(band_sig), _ptr(out), 1 if db else 0, B, self._stream())
        return out

    def spectrum_rows(self, x, db=False):
        """gpu_spectrum on individual signals (airSonar_optimized.py:332-343). [rows,N] f64 -> [rows,N/2+1] f32."""
        x = self._in(x, torch.float64, (-1, self.cfg.n_rx), "signal")
        out = self._empty((x.shape[0], self.cfg.n_bins), torch.float32)
        self._call("echo_spectrum_rows", self._h, _ptr(x), _ptr(out), 1 if db else 0, x.shape[0], self._stream())
        return out

    def mag2db(self, x):
        x = self._dev(x, torch.float32)
        out = torch.empty_like(x)
        self._call("echo_mag2db", _ptr(x), _ptr(out), x.numel(), self._stream())
        return out

    def correlate(self, band_sig):
        """gpu_correlate against each band's chirp (airSonar_optimized.py:290-303). -> [B,nb,L] f64."""
        band_sig = self._in(band_sig, torch.float64, (-1, self.cfg.n_bands, self.cfg.n_rx), "band_sig")
        B = band_sig.shape[0]
        out = self._empty((B, self.cfg.n_bands, self.cfg.n_corr), torch.float64)
        self._call("echo_correlate", self._h, _ptr(band_sig), _ptr(out), B, self._stream())
        return out

    def xcorr(self, a, b):
        """gpu_correlate(a, b) for arbitrary operands (airSonar_optimized.py:290-303): a [rows,n], b [m] ->
        scipy.signal.correlate(a[row], b, 'full') [rows, n+m-1], f64."""
        a = self._dev(a, torch.float64); b = self._dev(b, torch.float64)
        if a.dim() != 2 or b.dim() != 1 or a.shape[1] < 1 or b.shape[0] < 1:
            raise _lib.EchoError(f"xcorr takes a [rows, n] and b [m], got {list(a.shape)} and {list(b.shape)}")
        rows, n = a.shape
        m = b.shape[0]
        out = self._empty((rows, n + m - 1), torch.float64)
        self._call("echo_xcorr", _ptr(a), _ptr(b), _ptr(out), rows, n, m, self._stream())
        return out

    def first_strong_peak(self, corr):
        """first_strong_peak on [B,nb,L] f64 (mutated in place). -> (peak_idx i32 [B,nb], snr_db f64, amplitude f64)."""
        if not isinstance(corr, torch.Tensor) or corr.dtype != torch.float64 or not corr.is_contiguous() \
                or corr.device != self.device:
            raise _lib.EchoError("first_strong_peak mutates corr in place: it must be a contiguous float64 tensor on "
                                 f"{self.device}")
        self._shape(corr, (-1, self.cfg.n_bands, self.cfg.n_corr), "corr")
        B = corr.shape[0]
        nb = self.cfg.n_bands
        peak = self._empty((B, nb), torch.int32)
        snr = self._empty((B, nb), torch.float64)
        amp = self._empty((B, nb), torch.float64)
        self._call("echo_first_strong_peak", self._h, _ptr(corr), _ptr(peak), _ptr(snr), _ptr(amp), B, self._stream())
        return peak, snr, amp

    def first_strong_peak_rows(self, corr, min_delay: int, noise_n: int, snr_gate_db: float = 6.0):
        """first_strong_peak(corr, fs, min_delay_samples) with the caller's own arguments (airSonar_optimized.py:200-219)
        on rows of any length: corr [rows, len] f64, mutated in place (blind zone zeroed), ``noise_n`` = int(SNR_NOISE_MS
        * fs).  -> (peak_idx i32 [rows] with -1 for None, snr_db f64 [rows])."""
        if not isinstance(corr, torch.Tensor) or corr.dtype != torch.float64 or not corr.is_contiguous() \
                or corr.device != self.device or corr.dim() != 2 or corr.shape[1] < 1:
            raise _lib.EchoError("first_strong_peak_rows mutates corr in place: it must be a contiguous float64 tensor "
                                 f"[rows, len] on {self.device}")
        if min_delay < 0 or noise_n < 0:
            raise _lib.EchoError("first_strong_peak_rows: min_delay and noise_n must not be negative")
        rows, L = corr.shape
        peak = self._empty((rows,), torch.int32)
        snr = self._empty((rows,), torch.float64)
        self._call("echo_first_strong_peak_rows", _ptr(corr), rows, L, int(min_delay), int(noise_n), float(snr_gate_db),
                   _ptr(peak), _ptr(snr), self._stream())
        return peak, snr

    def spectrum_any(self, x, db=False):
        """gpu_spectrum / mag2db(np.fft.rfft(x)) for rows of any length (airSonar_optimized.py:332-343, :710).
        x [rows, n] f64 -> [rows, n // 2 + 1] f32."""
        x = self._dev(x, torch.float64)
        if x.dim() != 2 or x.shape[1] < 1:
            raise _lib.EchoError(f"spectrum_any takes [rows, n >= 1], got {list(x.shape)}")
        out = self._empty((x.shape[0], x.shape[1] // 2 + 1), torch.float32)
        self._call("echo_spectrum_any", _ptr(x), _ptr(out), 1 if db else 0, x.shape[0], x.shape[1], self._stream())
        return out

    def argmax(self, x):
        x = self._dev(x, torch.float64)
        if x.dim() < 1 or x.shape[-1] < 1:
            raise _lib.EchoError("argmax needs a non-empty last axis")
        rows = int(np.prod(x.shape[:-1])) if x.dim() > 1 else 1
        out = self._empty(x.shape[:-1] if x.dim() > 1 else (1,), torch.int32)
        self._call("echo_argmax", _ptr(x), _ptr(out), rows, x.shape[-1], self._stream())
        return out

    def envelope_detection(self, corr, want_envelope: bool = True):
        """gpu_envelope_detection(corr, 'hilbert', smooth_window=5) + envelope-path peak choice
        (airSonar_optimized.py:431-475, :642-654).  corr [B,nb,L] f64, not modified.
        -> dict(envelope [B,nb,L] f64 or None, peak_idx int32 (-1: fall back to first_strong_peak), snr_db,
        quality (corr_quality), n_peaks, peaks int32 [B,nb,64] ascending, -1 padded)."""
        corr = self._in(corr, torch.float64, (-1, self.cfg.n_bands, self.cfg.n_corr), "corr")
        B, nb, L = corr.shape
        env = self._empty((B, nb, L), torch.float64) if want_envelope else None
        peak = self._empty((B, nb), torch.int32); npk = self._empty((B, nb), torch.int32)
        snr = self._empty((B, nb), torch.float64); q = self._empty((B, nb), torch.float64)
        peaks = self._empty((B, nb, 64), torch.int32)
        self._call("echo_envelope_detection", self._h, _ptr(corr), _ptr(env), _ptr(peak), _ptr(snr), _ptr(q),
                   _ptr(npk), _ptr(peaks), B, self._stream())
        return dict(envelope=env, peak_idx=peak, snr_db=snr, quality=q, n_peaks=npk, peaks=peaks)

    def envelope_rows(self, x, smooth_window: int = 5, min_delay: int = 0, want_envelope: bool = True):
        """gpu_envelope_detection(x, 'hilbert', smooth_window) on rows of any even length the kernel holds
        (airSonar_optimized.py:431-475; the band-signal call uses smooth_window = 3, :636).  x [rows, len] f64.
        -> dict as ``envelope_detection`` with a leading ``rows`` axis only."""
        x = self._dev(x, torch.float64)
        if x.dim() != 2 or x.shape[1] < 1:
            raise _lib.EchoError(f"envelope_rows takes [rows, len], got {list(x.shape)}")
        rows, L = x.shape
        env = self._empty((rows, L), torch.float64) if want_envelope else None
        peak = self._empty((rows,), torch.int32); npk = self._empty((rows,), torch.int32)
        snr = self._empty((rows,), torch.float64); q = self._empty((rows,), torch.float64)
        peaks = self._empty((rows, 64), torch.int32)
        self._call("echo_envelope_rows", self._h, _ptr(x), L, int(smooth_window), int(min_delay), _ptr(env), _ptr(peak),
                   _ptr(snr), _ptr(q), _ptr(npk), _ptr(peaks), rows, self._stream())
        return dict(envelope=env, peak_idx=peak, snr_db=snr, quality=q, n_peaks=npk, peaks=peaks)

    def band_confidence(self, snr_db, amplitude, band_idx):
        """calculate_band_confidence elementwise (airSonar_optimized.py:222-243). 1-D inputs of equal length -> f64."""
        snr_db = self._dev(snr_db, torch.float64).reshape(-1); amplitude = self._dev(amplitude, torch.float64).reshape(-1)
        band_idx = self._dev(band_idx, torch.int32).reshape(-1)
        n = snr_db.numel()
        if amplitude.numel() != n or band_idx.numel() != n:
            raise _lib.EchoError("band_confidence: snr_db, amplitude and band_idx must have the same number of elements")
        if n and (int(band_idx.min()) < 0 or int(band_idx.max()) >= 3):
            raise _lib.EchoError("band_confidence: band_idx must be 0, 1 or 2")
        out = self._empty((n,), torch.float64)
        self._call("echo_band_confidence", self._h, _ptr(snr_db), _ptr(amplitude), _ptr(band_idx), _ptr(out), n,
                   self._stream())
        return out

    def normalize_confidences(self, conf):
        """normalize_confidences over the last axis (airSonar_optimized.py:245-262). [rows, n] f64 -> same shape."""
        conf = self._dev(conf, torch.float64)
        if conf.dim() != 2 or conf.shape[1] < 1:
            raise _lib.EchoError(f"normalize_confidences takes [rows, n >= 1], got {list(conf.shape)}")
        out = torch.empty_like(conf)
        self._call("echo_normalize_confidences", _ptr(conf), _ptr(out), conf.shape[0], conf.shape[1], self._stream())
        return out

    def confidence_fuse(self, peak_idx, snr_db, amplitude):
        """Range conversion + band confidence + fusion. -> uint8 tensor [B,64] viewable as RESULT_DTYPE."""
        nb = self.cfg.n_bands
        peak_idx = self._in(peak_idx, torch.int32, (-1, nb), "peak_idx")
        B = peak_idx.shape[0]
        snr_db = self._in(snr_db, torch.float64, (B, nb), "snr_db")
        amplitude = self._in(amplitude, torch.float64, (B, nb), "amplitude")
        out = self._empty((B, 64), torch.uint8)
        self._call("echo_confidence_fuse", self._h, _ptr(peak_idx), _ptr(snr_db), _ptr(amplitude), _ptr(out), B,
                   self._stream())
        return out

    def chain(self, rx, n_streams: Optional[int] = None, kalman: Optional["BatchedKalman"] = None,
              want_spectra: bool = True, out=None, spec_out=None, mode: str = "named", quality_out=None):
        """The fused per-frame chain. rx [B,N] f32 (time-major over n_streams streams).
        mode="named": first_strong_peak path; mode="full": _process_band_gpu with its envelope branch
        (airSonar_optimized.py:629-676); quality_out (full mode): f64 [B,nb,2] tensor receiving
        (signal_quality, corr_quality).
        Returns (records uint8 [B,64], spec_db f32 [B,nb,bins] or None)."""
        rx = self._in(rx, torch.float32, (-1, self.cfg.n_rx), "rx")
        B = rx.shape[0]
        nb = self.cfg.n_bands
        out = self._empty((B, 64), torch.uint8) if out is None else self._out(out, torch.uint8, (B, 64), "out")
        spec = None
        if want_spectra:
            spec = (self._empty((B, nb, self.cfg.n_bins), torch.float32) if spec_out is None
                    else self._out(spec_out, torch.float32, (B, nb, self.cfg.n_bins), "spec_out"))
        if quality_out is not None:
            self._out(quality_out, torch.float64, (B, nb, 2), "quality_out")
        S = n_streams if n_streams is not None else (kalman.S if kalman is not None else max(B, 1))
        kx = kp = ki = None
        if kalman is not None:
            if kalman.S != S or kalman.engine.device != self.device:
                raise _lib.EchoError("kalman state must hold n_streams streams on the engine's device")
            kx, kp, ki = kalman.x, kalman.p, kalman.init
        if mode == "full":
            self._call("echo_chain_full", self._h, _ptr(rx), _ptr(out), _ptr(spec), _ptr(quality_out), _ptr(kx),
                       _ptr(kp), _ptr(ki), B, S, self._stream())
        elif mode == "named":
            self._call("echo_chain", self._h, _ptr(rx), _ptr(out), _ptr(spec), _ptr(kx), _ptr(kp), _ptr(ki),
                       B, S, self._stream())
        else:
            raise _lib.EchoError(f"unknown chain mode {mode!r}")
        return out, spec

    def chain_wave(self, rx, n_streams: Optional[int] = None, kalman: Optional["BatchedKalman"] = None):
        """One pass of SonarWorker.run()'s loop body with everything it hands on (airSonar_optimized.py:703-757): the
        full chain's records plus, for every frame, the dB spectra and the raw correlations of the waveSig payload.
        Returns (records uint8 [B,64], spec_db f32 [B,nb,bins], corr f64 [B,nb,L])."""
        rx = self._in(rx, torch.float32, (-1, self.cfg.n_rx), "rx")
        B, nb = rx.shape[0], self.cfg.n_bands
        out = self._empty((B, 64), torch.uint8)
        spec = self._empty((B, nb, self.cfg.n_bins), torch.float32)
        corr = self._empty((B, nb, self.cfg.n_corr), torch.float64)
        S = n_streams if n_streams is not None else (kalman.S if kalman is not None else max(B, 1))
        kx = kp = ki = None
        if kalman is not None:
            if kalman.S != S or kalman.engine.device != self.device:
                raise _lib.EchoError("kalman state must hold n_streams streams on the engine's device")
            kx, kp, ki = kalman.x, kalman.p, kalman.init
        self._call("echo_chain_wave", self._h, _ptr(rx), _ptr(out), _ptr(spec), _ptr(corr), None, _ptr(kx), _ptr(kp),
                   _ptr(ki), B, S, self._stream())
        return out, spec, corr

    def chain_noalloc(self, rx: torch.Tensor, out: torch.Tensor, spec_out: Optional[torch.Tensor] = None,
                      n_streams: Optional[int] = None, kalman: Optional["BatchedKalman"] = None):
        """``chain`` (named mode) through the entry point that allocates nothing: every buffer is the caller's
        (``rx`` [B,N] f32, ``out`` uint8 [B,64], ``spec_out`` f32 [B,nb,bins] or None), the pass's scratch is the
        library's fixed per-GPU array.  B <= ``noalloc_max_frames()``; larger batches raise EchoError."""
        if not isinstance(rx, torch.Tensor) or rx.dim() != 2:
            raise _lib.EchoError("chain_noalloc: rx must be a [B, n_rx] tensor")
        B = rx.shape[0]
        nb, N, nbins = self.cfg.n_bands, self.cfg.n_rx, self.cfg.n_bins
        self._out(rx, torch.float32, (B, N), "rx")
        self._out(out, torch.uint8, (B, 64), "out")
        if spec_out is not None:
            self._out(spec_out, torch.float32, (B, nb, nbins), "spec_out")
        S = n_streams if n_streams is not None else (kalman.S if kalman is not None else max(B, 1))
        kx = kp = ki = None
        if kalman is not None:
            if kalman.S != S:
                raise _lib.EchoError("chain_noalloc: kalman.S differs from n_streams")
            kx, kp, ki = kalman.x, kalman.p, kalman.init
        self._call("echo_chain_noalloc", self._h, _ptr(rx), _ptr(out), _ptr(spec_out), _ptr(kx), _ptr(kp), _ptr(ki),
                   B, S, self._stream())
        return out, spec_out

    def noalloc_max_frames(self) -> int:
        n = self.lib.echo_noalloc_max_frames(self._h)
        if n < 0:
            _lib.check(n, "echo_noalloc_max_frames")
        return n

    def synth_frames(self, count: int, first_frame: int = 0, seed: int = 1234, out=None) -> torch.Tensor:
        """``count`` synthetic rx frames generated in device memory (frame g = a function of (seed, g)); float32
        [count, n_rx] on the int16 grid.  Same recipe as dsp_amd.synth with its own random stream."""
        if getattr(self, "_tx_mix_dev", None) is None:
            # (two threads arriving here together both build the same array; the later assignment wins, either is valid)
            self._tx_mix_dev = torch.from_numpy((self.tables.tx_pcm.astype(np.float64) / 32768.0).astype(np.float32)
                                                ).to(self.device)
        rx = (self._empty((count, self.cfg.n_rx), torch.float32) if out is None
              else self._out(out, torch.float32, (count, self.cfg.n_rx), "out"))
        self._call("echo_synth_frames", self._h, _ptr(self._tx_mix_dev), _ptr(rx), count, first_frame, seed,
                   self._stream())
        return rx

    def chain_prefetch(self, rx: torch.Tensor):
        """Start the IIR of the batch the next ``chain`` call will get (same tensor), on the engine's own stream, so it
        overlaps the current batch's kernels.  ``rx`` must be a complete, contiguous float32 device tensor [B, N]
        that stays unchanged until that call."""
        if not isinstance(rx, torch.Tensor) or rx.dim() != 2:
            raise _lib.EchoError("chain_prefetch: rx must be a [B, n_rx] device tensor")
        self._out(rx, torch.float32, (rx.shape[0], self.cfg.n_rx), "rx")
        self._call("echo_chain_prefetch", self._h, _ptr(rx), rx.shape[0])

    def prefetch_ms(self) -> float:
        v = C.c_float()
        self._call("echo_prefetch_ms", self._h, C.byref(v))
        return v.value

    def capture_chain(self, rx: torch.Tensor, kalman: "BatchedKalman", want_spectra: bool = False) -> "ChainGraph":
        """Streaming mode: records one chain call over the device buffer ``rx`` [S, N] (one frame per stream, i.e.
        one tick of realtime_sonar.py's loop) into a HIP graph.  Per tick, copy the new frames into ``rx`` and call
        ``graph.launch()``; records (and spectra) land in ``graph.out`` / ``graph.spec``."""
        S = kalman.S
        self._out(rx, torch.float32, (S, self.cfg.n_rx), "rx")
        out = self._empty((S, 64), torch.uint8)
        spec = self._empty((S, self.cfg.n_bands, self.cfg.n_bins), torch.float32) if want_spectra else None
        side = torch.cuda.Stream(device=self.device)          # capture needs a stream other than the null stream
        side.wait_stream(torch.cuda.current_stream(self.device))
        g = C.c_void_p()
        self._call("echo_graph_create", self._h, _ptr(rx), _ptr(out), _ptr(spec), _ptr(kalman.x), _ptr(kalman.p),
                   _ptr(kalman.init), S, S, C.c_void_p(side.cuda_stream), C.byref(g))
        return ChainGraph(self, g, rx, out, spec, kalman)

    def set_profiling(self, slots: int, every: int = 1):
        """Arm HIP-event timing for ``slots`` of the next chain calls, every ``every``-th one (0 slots disarms)."""
        self._call("echo_set_profiling_every", self._h, int(slots), int(every))

    def last_kernel_ms(self):
        """Average device time (HIP events on the call's stream) of the stages of the armed chain calls: 'band' is the
        whole band stage; 'pair' and 'exact' its two kernels where it runs as two (0 otherwise)."""
        ms = (C.c_float * 6)()
        self._call("echo_kernel_times", self._h, ms, 6)
        return dict(zip(("iir", "band", "spec", "fuse", "pair", "exact"), (float(x) for x in ms)))

    def confined_iir_passes(self) -> int:
        """Passes of this engine whose IIR ran on the device's shared stream confined to a share of the compute units
        (what the library does by itself for long frames and the full chain while other engines' batches are in
        flight).  A scheduling diagnostic: results do not depend on it."""
        return int(self.lib.echo_confined_iir_passes(self._h))


class ChainGraph:
    """A recorded chain call (HIP graph) over fixed device buffers; see EchoEngine.capture_chain."""

    def __init__(self, engine, handle, rx, out, spec, kalman):
        self.engine, self._g, self.rx, self.out, self.spec, self.kalman = engine, handle, rx, out, spec, kalman

    def launch(self):
        self.engine._call("echo_graph_launch", self._g, self.engine._stream())
        return self.out

    def close(self):
        if self._g:
            self.engine.lib.echo_graph_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def records_to_numpy(rec: torch.Tensor) -> np.ndarray:
    """uint8 [B,64] device tensor -> structured NumPy array of EchoResult rows."""
    return rec.cpu().numpy().view(RESULT_DTYPE).reshape(-1)


class BatchedKalman:
    """ScalarKalman for S independent streams (airSonar_optimized.py:265-283); state lives in HBM."""

    def __init__(self, engine: EchoEngine, n_streams: int, q: Optional[float] = None, r: Optional[float] = None):
        """q, r: process / measurement noise of ``ScalarKalman(q, r)``; default: the engine's configuration.  (The fused
        chain always filters with the configuration's values.)"""
        self.engine = engine
        self.S = int(n_streams)
        if self.S < 1:
            raise _lib.EchoError("BatchedKalman needs at least one stream")
        self.q = float(engine.cfg.KALMAN_Q if q is None else q)
        self.r = float(engine.cfg.KALMAN_R if r is None else r)
        d = engine.device
        self.x = torch.zeros(self.S, dtype=torch.float64, device=d)
        self.p = torch.ones(self.S, dtype=torch.float64, device=d)
        self.init = torch.zeros(self.S, dtype=torch.uint8, device=d)

    def update(self, z, valid=None):
        """z [T,S] or [S] float64; valid same shape (uint8/bool). Returns smoothed values, same shape."""
        e = self.engine
        z = e._dev(z, torch.float64)
        shape = z.shape
        if z.numel() % self.S != 0 or (z.dim() > 0 and z.shape[-1] != self.S):
            raise _lib.EchoError(f"z must be [T, {self.S}] or [{self.S}], got {list(shape)}")
        z2 = z.reshape(-1, self.S)
        if valid is None:
            v = torch.ones_like(z2, dtype=torch.uint8)
        else:
            v = e._dev(valid, torch.uint8)
            if v.numel() != z2.numel():
                raise _lib.EchoError("valid must have the shape of z")
            v = v.reshape(-1, self.S)
        out = torch.empty_like(z2)
        e._call("echo_kalman_update_qr", _ptr(self.x), _ptr(self.p), _ptr(self.init), _ptr(z2), _ptr(v), _ptr(out),
                self.S, z2.shape[0], self.q, self.r, e._stream())
        return out.reshape(shape)
