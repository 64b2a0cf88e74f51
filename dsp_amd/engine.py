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
    def __init__(self, cfg: Optional[SonarConfig] = None, device: Optional[torch.device] = None):
        self.cfg = cfg or optimized_config()
        self.tables: SonarTables = build_tables(self.cfg)
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.EchoError("no GPU visible: the echo-ranging kernels need a gfx950 device")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        c, t = self.cfg, self.tables
        self._keep = [np.ascontiguousarray(x, dtype=np.float64) for x in (t.b, t.a, t.zi, t.taps, t.chirps)]
        cc = _lib.EchoConfigC(
            fs=c.FS, n_rx=c.n_rx, n_chirp=c.n_chirp, n_bands=c.n_bands, iir_ncoef=2 * c.IIR_ORDER + 1,
            n_taps=c.FIR_TAPS, min_delay=c.min_delay, noise_n=c.noise_n, detector=c.DETECTOR,
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

    def _dev(self, x, dtype):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x))
        return x.to(device=self.device, dtype=dtype).contiguous()

    def _empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def reserve(self, max_frames: int):
        _lib.check(self.lib.echo_reserve(self._h, int(max_frames)), "echo_reserve")

    # ------------------------------------------------------------------ stages (device tensors)
    def fir_bank(self, rx):
        """FIR only (the reference's CuPy gpu_bandpass, airSonar_optimized.py:305-313). [B,N] f32 -> [B,nb,N] f64."""
        rx = self._dev(rx, torch.float32)
        B = rx.shape[0]
        out = self._empty((B, self.cfg.n_bands, self.cfg.n_rx), torch.float64)
        _lib.check(self.lib.echo_fir_bank(self._h, _ptr(rx), _ptr(out), B, self._stream()), "echo_fir_bank")
        return out

    def bandpass_bank(self, rx):
        """bandpass for every band (airSonar_optimized.py:185-197). [B,N] f32 -> [B,nb,N] f64."""
        rx = self._dev(rx, torch.float32)
        assert rx.dim() == 2 and rx.shape[1] == self.cfg.n_rx, "rx must be [B, n_rx]"
        B = rx.shape[0]
        out = self._empty((B, self.cfg.n_bands, self.cfg.n_rx), torch.float64)
        _lib.check(self.lib.echo_bandpass_bank(self._h, _ptr(rx), _ptr(out), B, self._stream()), "echo_bandpass_bank")
        return out

    def spectrum(self, band_sig, db=True):
        """|rfft| or its dB (airSonar_optimized.py:710, :332-343). [B,nb,N] f64 -> [B,nb,N/2+1] f32."""
        band_sig = self._dev(band_sig, torch.float64)
        B = band_sig.shape[0]
        out = self._empty((B, self.cfg.n_bands, self.cfg.n_bins), torch.float32)
        _lib.check(self.lib.echo_spectrum(self._h, _ptr(band_sig), _ptr(out), 1 if db else 0, B, self._stream()),
                   "echo_spectrum")
        return out

    def mag2db(self, x):
        x = self._dev(x, torch.float32)
        out = torch.empty_like(x)
        _lib.check(self.lib.echo_mag2db(_ptr(x), _ptr(out), x.numel(), self._stream()), "echo_mag2db")
        return out

    def correlate(self, band_sig):
        """gpu_correlate against each band's chirp (airSonar_optimized.py:290-303). -> [B,nb,L] f64."""
        band_sig = self._dev(band_sig, torch.float64)
        B = band_sig.shape[0]
        out = self._empty((B, self.cfg.n_bands, self.cfg.n_corr), torch.float64)
        _lib.check(self.lib.echo_correlate(self._h, _ptr(band_sig), _ptr(out), B, self._stream()), "echo_correlate")
        return out

    def first_strong_peak(self, corr):
        """first_strong_peak on [B,nb,L] f64 (mutated in place). -> (peak_idx i32 [B,nb], snr_db f64, amplitude f64)."""
        assert isinstance(corr, torch.Tensor) and corr.dtype == torch.float64 and corr.is_contiguous()
        B = corr.shape[0]
        nb = self.cfg.n_bands
        peak = self._empty((B, nb), torch.int32)
        snr = self._empty((B, nb), torch.float64)
        amp = self._empty((B, nb), torch.float64)
        _lib.check(self.lib.echo_first_strong_peak(self._h, _ptr(corr), _ptr(peak), _ptr(snr), _ptr(amp), B,
                                                   self._stream()), "echo_first_strong_peak")
        return peak, snr, amp

    def argmax(self, x):
        x = self._dev(x, torch.float64)
        rows = int(np.prod(x.shape[:-1])) if x.dim() > 1 else 1
        out = self._empty(x.shape[:-1] if x.dim() > 1 else (1,), torch.int32)
        _lib.check(self.lib.echo_argmax(_ptr(x), _ptr(out), rows, x.shape[-1], self._stream()), "echo_argmax")
        return out

    def envelope_detection(self, corr, want_envelope: bool = True):
        """gpu_envelope_detection(corr, 'hilbert', smooth_window=5) + envelope-path peak choice
        (airSonar_optimized.py:431-475, :642-654).  corr [B,nb,L] f64, not modified.
        -> dict(envelope [B,nb,L] f64 or None, peak_idx int32 (-1: fall back to first_strong_peak), snr_db,
        quality (corr_quality), n_peaks, peaks int32 [B,nb,64] ascending, -1 padded)."""
        corr = self._dev(corr, torch.float64)
        B, nb, L = corr.shape
        assert nb == self.cfg.n_bands and L == self.cfg.n_corr
        env = self._empty((B, nb, L), torch.float64) if want_envelope else None
        peak = self._empty((B, nb), torch.int32); npk = self._empty((B, nb), torch.int32)
        snr = self._empty((B, nb), torch.float64); q = self._empty((B, nb), torch.float64)
        peaks = self._empty((B, nb, 64), torch.int32)
        _lib.check(self.lib.echo_envelope_detection(self._h, _ptr(corr), _ptr(env), _ptr(peak), _ptr(snr), _ptr(q),
                                                    _ptr(npk), _ptr(peaks), B, self._stream()), "echo_envelope_detection")
        return dict(envelope=env, peak_idx=peak, snr_db=snr, quality=q, n_peaks=npk, peaks=peaks)

    def confidence_fuse(self, peak_idx, snr_db, amplitude):
        """Range conversion + band confidence + fusion. -> uint8 tensor [B,64] viewable as RESULT_DTYPE."""
        peak_idx = self._dev(peak_idx, torch.int32); snr_db = self._dev(snr_db, torch.float64)
        amplitude = self._dev(amplitude, torch.float64)
        B = peak_idx.shape[0]
        out = self._empty((B, 64), torch.uint8)
        _lib.check(self.lib.echo_confidence_fuse(self._h, _ptr(peak_idx), _ptr(snr_db), _ptr(amplitude), _ptr(out), B,
                                                 self._stream()), "echo_confidence_fuse")
        return out

    def chain(self, rx, n_streams: Optional[int] = None, kalman: Optional["BatchedKalman"] = None,
              want_spectra: bool = True, out=None, spec_out=None, mode: str = "named", quality_out=None):
        """The fused per-frame chain. rx [B,N] f32 (time-major over n_streams streams).
        mode="named": first_strong_peak path; mode="full": _process_band_gpu with its envelope branch
        (airSonar_optimized.py:629-676); quality_out (full mode): f64 [B,nb,2] tensor receiving
        (signal_quality, corr_quality).
        Returns (records uint8 [B,64], spec_db f32 [B,nb,bins] or None)."""
        rx = self._dev(rx, torch.float32)
        assert rx.dim() == 2 and rx.shape[1] == self.cfg.n_rx, "rx must be [B, n_rx]"
        B = rx.shape[0]
        if out is None:
            out = self._empty((B, 64), torch.uint8)
        spec = None
        if want_spectra:
            spec = spec_out if spec_out is not None else self._empty((B, self.cfg.n_bands, self.cfg.n_bins), torch.float32)
        S = n_streams if n_streams is not None else (kalman.S if kalman is not None else max(B, 1))
        kx = kp = ki = None
        if kalman is not None:
            assert kalman.S == S
            kx, kp, ki = kalman.x, kalman.p, kalman.init
        if mode == "full":
            _lib.check(self.lib.echo_chain_full(self._h, _ptr(rx), _ptr(out), _ptr(spec), _ptr(quality_out), _ptr(kx),
                                                _ptr(kp), _ptr(ki), B, S, self._stream()), "echo_chain_full")
        else:
            assert mode == "named"
            _lib.check(self.lib.echo_chain(self._h, _ptr(rx), _ptr(out), _ptr(spec), _ptr(kx), _ptr(kp), _ptr(ki),
                                           B, S, self._stream()), "echo_chain")
        return out, spec

    def chain_noalloc(self, rx: torch.Tensor, out: torch.Tensor, spec_out: Optional[torch.Tensor] = None,
                      n_streams: Optional[int] = None, kalman: Optional["BatchedKalman"] = None):
        """``chain`` (named mode) through the entry point that allocates nothing: every buffer is the caller's
        (``rx`` [B,N] f32, ``out`` uint8 [B,64], ``spec_out`` f32 [B,nb,bins] or None), the pass's scratch is the
        library's fixed per-GPU array.  B <= ``noalloc_max_frames()``; larger batches raise EchoError."""
        B = rx.shape[0]
        nb, N, nbins = self.cfg.n_bands, self.cfg.n_rx, self.cfg.n_bins
        for t, shape, dt, name in ((rx, (B, N), torch.float32, "rx"), (out, (B, 64), torch.uint8, "out"),
                                   (spec_out, (B, nb, nbins), torch.float32, "spec_out")):
            if t is None and name == "spec_out":
                continue
            if (not isinstance(t, torch.Tensor) or t.device != self.device or t.dtype != dt or tuple(t.shape) != shape
                    or not t.is_contiguous()):
                raise _lib.EchoError(f"chain_noalloc: {name} must be a contiguous {dt} tensor of shape {shape} on {self.device}")
        S = n_streams if n_streams is not None else (kalman.S if kalman is not None else max(B, 1))
        kx = kp = ki = None
        if kalman is not None:
            if kalman.S != S:
                raise _lib.EchoError("chain_noalloc: kalman.S differs from n_streams")
            kx, kp, ki = kalman.x, kalman.p, kalman.init
        with torch.cuda.device(self.device):
            _lib.check(self.lib.echo_chain_noalloc(self._h, _ptr(rx), _ptr(out), _ptr(spec_out), _ptr(kx), _ptr(kp),
                                                   _ptr(ki), B, S, self._stream()), "echo_chain_noalloc")
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
            self._tx_mix_dev = torch.from_numpy((self.tables.tx_pcm.astype(np.float64) / 32768.0).astype(np.float32)
                                                ).to(self.device)
        rx = out if out is not None else self._empty((count, self.cfg.n_rx), torch.float32)
        _lib.check(self.lib.echo_synth_frames(self._h, _ptr(self._tx_mix_dev), _ptr(rx), count, first_frame, seed,
                                              self._stream()), "echo_synth_frames")
        return rx

    def chain_prefetch(self, rx: torch.Tensor):
        """Start the IIR of the batch the next ``chain`` call will get (same tensor), on the engine's own stream, so it
        overlaps the current batch's kernels.  ``rx`` must be a complete, contiguous float32 device tensor [B, N]
        that stays unchanged until that call."""
        assert rx.is_cuda and rx.dtype == torch.float32 and rx.is_contiguous() and rx.shape[1] == self.cfg.n_rx
        _lib.check(self.lib.echo_chain_prefetch(self._h, _ptr(rx), rx.shape[0]), "echo_chain_prefetch")

    def prefetch_ms(self) -> float:
        v = C.c_float()
        _lib.check(self.lib.echo_prefetch_ms(self._h, C.byref(v)), "echo_prefetch_ms")
        return v.value

    def capture_chain(self, rx: torch.Tensor, kalman: "BatchedKalman", want_spectra: bool = False) -> "ChainGraph":
        """Streaming mode: records one chain call over the device buffer ``rx`` [S, N] (one frame per stream, i.e.
        one tick of realtime_sonar.py's loop) into a HIP graph.  Per tick, copy the new frames into ``rx`` and call
        ``graph.launch()``; records (and spectra) land in ``graph.out`` / ``graph.spec``."""
        assert rx.is_cuda and rx.dtype == torch.float32 and rx.is_contiguous() and rx.shape == (kalman.S, self.cfg.n_rx)
        S = kalman.S
        out = self._empty((S, 64), torch.uint8)
        spec = self._empty((S, self.cfg.n_bands, self.cfg.n_bins), torch.float32) if want_spectra else None
        side = torch.cuda.Stream(device=self.device)          # capture needs a stream other than the null stream
        side.wait_stream(torch.cuda.current_stream(self.device))
        g = C.c_void_p()
        _lib.check(self.lib.echo_graph_create(self._h, _ptr(rx), _ptr(out), _ptr(spec), _ptr(kalman.x), _ptr(kalman.p),
                                              _ptr(kalman.init), S, S, C.c_void_p(side.cuda_stream), C.byref(g)),
                   "echo_graph_create")
        return ChainGraph(self, g, rx, out, spec, kalman)

    def set_profiling(self, slots: int):
        """Arm HIP-event timing for the next ``slots`` chain calls (0 disarms)."""
        _lib.check(self.lib.echo_set_profiling(self._h, int(slots)), "echo_set_profiling")

    def last_kernel_ms(self):
        a, b, c, d = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        _lib.check(self.lib.echo_last_kernel_ms(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)),
                   "echo_last_kernel_ms")
        return {"iir": a.value, "band": b.value, "spec": c.value, "fuse": d.value}


class ChainGraph:
    """A recorded chain call (HIP graph) over fixed device buffers; see EchoEngine.capture_chain."""

    def __init__(self, engine, handle, rx, out, spec, kalman):
        self.engine, self._g, self.rx, self.out, self.spec, self.kalman = engine, handle, rx, out, spec, kalman

    def launch(self):
        _lib.check(self.engine.lib.echo_graph_launch(self._g, self.engine._stream()), "echo_graph_launch")
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

    def __init__(self, engine: EchoEngine, n_streams: int):
        self.engine = engine
        self.S = int(n_streams)
        d = engine.device
        self.x = torch.zeros(self.S, dtype=torch.float64, device=d)
        self.p = torch.ones(self.S, dtype=torch.float64, device=d)
        self.init = torch.zeros(self.S, dtype=torch.uint8, device=d)

    def update(self, z, valid=None):
        """z [T,S] or [S] float64; valid same shape (uint8/bool). Returns smoothed values, same shape."""
        e = self.engine
        z = e._dev(z, torch.float64)
        shape = z.shape
        z2 = z.reshape(-1, self.S)
        v = torch.ones_like(z2, dtype=torch.uint8) if valid is None else e._dev(valid, torch.uint8).reshape(-1, self.S)
        out = torch.empty_like(z2)
        _lib.check(e.lib.echo_kalman_update(e._h, _ptr(self.x), _ptr(self.p), _ptr(self.init), _ptr(z2), _ptr(v),
                                            _ptr(out), self.S, z2.shape[0], e._stream()), "echo_kalman_update")
        return out.reshape(shape)
