"""Host-side, once-per-configuration set-up: chirps, filter design and the constant
tables that are uploaded to the GPU.

Mirrors ``generate_chirps`` (airSonar_optimized.py:157-165) and ``design_filters``
(airSonar_optimized.py:167-183).  Filter design is the reference's own SciPy calls
(``iirfilter``/``firwin``); it runs once on the host and is not part of the hot path.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
from scipy.signal import firwin, iirfilter, lfilter_zi

from .config import SonarConfig, optimized_config


def calculate_sound_speed(temperature_c):
    """airSonar_optimized.py:150-154."""
    return 331.3 + 0.606 * temperature_c


def linear_chirp(t: np.ndarray, f0: float, f1: float, t1: float) -> np.ndarray:
    """cos(2*pi*(f0*t + 0.5*beta*t^2)), beta = (f1-f0)/t1 -- the 'linear' method of
    scipy.signal.chirp as called at airSonar_optimized.py:161 (phi = 0)."""
    beta = (f1 - f0) / t1
    phase = 2 * np.pi * (f0 * t + 0.5 * beta * t * t)
    return np.cos(phase + 0.0)


def generate_chirps(cfg: Optional[SonarConfig] = None, fs=None, duration=None):
    """Returns (tx_pcm int16[M], [chirp float32[M]] per band). airSonar_optimized.py:157-165."""
    cfg = cfg or optimized_config()
    fs = fs if fs is not None else cfg.FS
    duration = duration if duration is not None else cfg.CHIRP_LEN
    t = np.linspace(0, duration, int(fs * duration), endpoint=False)
    chirps = [linear_chirp(t, l, h, duration).astype(np.float32) for l, h in cfg.BANDS]
    mix = np.sum(chirps, axis=0)
    mix *= 0.85 * (2 ** 15 - 1) / np.max(np.abs(mix))
    return mix.astype(np.int16), chirps


def design_filters(cfg: Optional[SonarConfig] = None, fs=None):
    """Per band (b, a, taps): Butterworth band-pass of order ``IIR_ORDER`` in transfer-function
    form plus a Hamming FIR. airSonar_optimized.py:167-183 (order 6 / 61 taps),
    airSonar.py:87-96 (order 8 / 101 taps)."""
    cfg = cfg or optimized_config()
    fs = fs if fs is not None else cfg.FS
    filters = []
    for low, high in cfg.BANDS:
        b, a = iirfilter(cfg.IIR_ORDER, [low / (0.5 * fs), high / (0.5 * fs)], btype='band', output='ba')
        taps = firwin(cfg.FIR_TAPS, [low, high], fs=fs, pass_zero=False, window='hamming')
        filters.append((b, a, taps))
    return filters


@dataclass
class SonarTables:
    """Everything the device needs, as contiguous host arrays (float64 unless noted)."""
    cfg: SonarConfig
    b: np.ndarray        # [nb, na]      numerator, a0-normalised
    a: np.ndarray        # [nb, na]      denominator, a[.,0] == 1
    zi: np.ndarray       # [nb, na-1]    lfilter_zi(b, a): steady-state DF2T state for unit step
    taps: np.ndarray     # [nb, ntap]
    chirps: np.ndarray   # [nb, M]       float32 values held in float64
    tx_pcm: np.ndarray   # [M] int16
    filters: list        # the reference-shaped [(b, a, taps)]
    chirps_f32: list     # the reference-shaped [float32[M]]


def build_tables(cfg: Optional[SonarConfig] = None) -> SonarTables:
    cfg = cfg or optimized_config()
    filters = design_filters(cfg)
    tx_pcm, chirps = generate_chirps(cfg)
    na = 2 * cfg.IIR_ORDER + 1
    nb = cfg.n_bands
    b = np.zeros((nb, na)); a = np.zeros((nb, na)); zi = np.zeros((nb, na - 1))
    taps = np.zeros((nb, cfg.FIR_TAPS))
    for i, (bi, ai, ti) in enumerate(filters):
        bi = np.asarray(bi, dtype=np.float64); ai = np.asarray(ai, dtype=np.float64)
        if ai[0] != 1.0:         # scipy.signal.lfilter normalises by a[0] before filtering
            bi = bi / ai[0]; ai = ai / ai[0]
        b[i] = bi; a[i] = ai
        zi[i] = lfilter_zi(bi, ai)
        taps[i] = ti
    ch = np.stack([c.astype(np.float64) for c in chirps])
    return SonarTables(cfg=cfg, b=b, a=a, zi=zi, taps=taps, chirps=ch, tx_pcm=tx_pcm,
                       filters=filters, chirps_f32=chirps)
