"""Synthetic rx frames: a delayed, scaled copy of the transmitted chirp mix in Gaussian noise,
quantised to the int16 grid exactly as ``AudioIO.record`` delivers it
(``np.frombuffer(raw, np.int16).astype(np.float32)/2**15``, airSonar_optimized.py:573).

Frame ``i`` is a pure function of ``seed + i`` so any rank can regenerate any shard.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .config import SonarConfig, optimized_config


def frame_params(i: int, cfg: SonarConfig, seed: int = 1234):
    """Draw order is fixed: range, amplitude, noise sigma, second-echo flag, second range."""
    rng = np.random.default_rng(seed + i)
    r1 = rng.uniform(1.5, 14.0)
    amp = rng.uniform(0.05, 0.4)
    sigma = rng.uniform(0.002, 0.02)
    second = rng.uniform() < 0.10
    r2 = rng.uniform(1.5, 14.0)
    return rng, r1, amp, sigma, second, r2


def quantise(x: np.ndarray) -> np.ndarray:
    """float -> int16 grid -> float32 in [-1, 1)."""
    q = np.clip(np.round(x * 32768.0), -32768, 32767).astype(np.int16)
    return q


def synth_frame_i16(i: int, cfg: SonarConfig, tx_mix: np.ndarray, seed: int = 1234) -> np.ndarray:
    n, m = cfg.n_rx, tx_mix.shape[0]
    rng, r1, amp, sigma, second, r2 = frame_params(i, cfg, seed)
    x = rng.normal(0.0, sigma, n)
    c = 343.42
    for r, a in ((r1, amp),) + (((r2, 0.5 * amp),) if second else ()):
        d = int(round(2.0 * r / c * cfg.FS))
        if d < n:
            k = min(m, n - d)
            x[d:d + k] += a * tx_mix[:k]
    return quantise(x)


def synth_frames(count: int, cfg: Optional[SonarConfig] = None, tx_pcm: Optional[np.ndarray] = None,
                 seed: int = 1234, start: int = 0) -> np.ndarray:
    """float32 [count, n_rx] frames ``start .. start+count-1``."""
    cfg = cfg or optimized_config()
    if tx_pcm is None:
        from .design import generate_chirps
        tx_pcm, _ = generate_chirps(cfg)
    tx_mix = tx_pcm.astype(np.float64) / 32768.0
    out = np.empty((count, cfg.n_rx), dtype=np.float32)
    for j in range(count):
        out[j] = synth_frame_i16(start + j, cfg, tx_mix, seed).astype(np.float32) / 2 ** 15
    return out


def tiled_frames(count: int, pool: np.ndarray, start: int = 0) -> np.ndarray:
    """Large batches: frame g is pool frame (g mod P) circularly shifted by 7*(g div P) samples.
    Cheap to produce for millions of frames, still distinct per frame."""
    p, n = pool.shape
    out = np.empty((count, n), dtype=np.float32)
    for j in range(count):
        g = start + j
        out[j] = np.roll(pool[g % p], 7 * (g // p))
    return out


# ---------------------------------------------------------------------------------------------------------------
# NumPy mirror of the parameter draw of the on-device generator (csrc/synth_kernel.hip): Philox4x32-10 with key
# (seed, seed >> 32) and counter (block, g, stream id, g >> 32); stream id 1 carries the per-frame parameters.
def _philox4x32_10(c0, c1, c2, c3, k0, k1):
    m32 = 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & m32, p1 & m32, ((p0 >> 32) ^ c3 ^ k1) & m32, p0 & m32
        k0, k1 = (k0 + 0x9E3779B9) & m32, (k1 + 0xBB67AE85) & m32
    return c0, c1, c2, c3


def _u01(x):
    return np.float32((np.float32(x >> 8) + np.float32(1.0)) * np.float32(1.0 / 16777216.0))


def device_frame_params(g: int, cfg: SonarConfig, seed: int = 1234):
    """(delay_1, amplitude, sigma, has_second, delay_2) of frame ``g`` of ``EchoEngine.synth_frames``."""
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    q = _philox4x32_10(0, g & 0xFFFFFFFF, 1, (g >> 32) & 0xFFFFFFFF, k0, k1)
    q2 = _philox4x32_10(1, g & 0xFFFFFFFF, 1, (g >> 32) & 0xFFFFFFFF, k0, k1)
    f32 = np.float32
    r1 = f32(1.5) + f32(12.5) * _u01(q[0])
    amp = f32(0.05) + f32(0.35) * _u01(q[1])
    sigma = f32(0.002) + f32(0.018) * _u01(q[2])
    second = bool(_u01(q[3]) < f32(0.10))
    r2 = f32(1.5) + f32(12.5) * _u01(q2[0])
    c = f32(343.42)
    d1 = int(np.rint(f32(2.0) * r1 / c * f32(cfg.FS)))
    d2 = int(np.rint(f32(2.0) * r2 / c * f32(cfg.FS)))
    return d1, float(amp), float(sigma), second, d2
