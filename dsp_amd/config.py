"""Sonar configuration: the reference's frozen ``Config`` dataclass restated with the
derived sizes the kernels need.

Mirrors ``airSonar_optimized.py:65-113`` (Config + derived properties).  The two
other parameter sets the reference ships are provided as presets:
``airSonar.py:50-60`` (FS 44100, IIR 8 / FIR 101, fixed 0.1 s chirp) and
``realtime_sonar.py:49-60`` (R_MIN 0.5, distinct bands).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, replace
from typing import Optional, Tuple

# detector kinds (first_strong_peak variants)
DET_MEDIAN = 0    # airSonar_optimized.py:200-219  median^2 floor, global argmax, 6 dB gate
DET_MAD = 1       # airSonar.py:104-117            first crossing of med + 6*MAD
DET_MEANPOW = 2   # realtime_sonar.py:127-142      mean-power floor over first 20 % of pos

DET_TWOPEAK = 3   # 666.py:153-200  strongest sample = direct path (time zero), second strongest = echo

# confidence kinds
CONF_CLAMPED = 0  # airSonar_optimized.py:222-243  per-term min(.,1)
CONF_RAW = 1      # realtime_sonar.py:393          only the final min(.,1)
CONF_NONE = 2     # airSonar.py has no confidence

# fusion kinds
FUSE_WEIGHTED_EPS = 0  # airSonar_optimized.py:733-738  weights = norm/100 + 1e-9
FUSE_WEIGHTED = 1      # realtime_sonar.py:337-339      weights = norm/100
FUSE_MEAN = 2          # airSonar.py:203-206            plain mean of valid distances
FUSE_SNRNORM = 3       # 666.py:372-385                 weights = snr / max(snr) * 100 + 1e-9

BANDS_TWOPEAK = ((3000, 7000), (10000, 14000), (17000, 20000))      # 666.py:68

BANDS_DEFAULT = ((3000, 5000), (3000, 5000), (3000, 5000))          # airSonar_optimized.py:75
BANDS_DISTINCT = ((9500, 11500), (13500, 15500), (17500, 19500))    # airSonar_optimized.py:74 (commented), realtime_sonar.py:60
BANDS_ORIGINAL = ((3000, 6000), (8000, 11000), (13000, 16000))      # airSonar.py:60


@dataclass(frozen=True)
class SonarConfig:
    FS: int = 48000
    BASE_TEMP: float = 28.0
    R_MIN: float = 1.0
    R_MAX: float = 15.0
    BANDS: Tuple[Tuple[float, float], ...] = BANDS_DEFAULT
    SNR_NOISE_MS: float = 0.005
    IIR_ORDER: int = 6
    FIR_TAPS: int = 61
    # fixed durations override the geometry-derived ones (airSonar.py:51-52)
    CHIRP_LEN_FIXED: Optional[float] = None
    LISTEN_LEN_FIXED: Optional[float] = None
    MIN_DELAY_FIXED: Optional[int] = None       # airSonar.py:104 uses 500
    DETECTOR: int = DET_MEDIAN
    CONFIDENCE: int = CONF_CLAMPED
    FUSION: int = FUSE_WEIGHTED_EPS
    SUBTRACT_CHIRP: bool = True                 # delay = peak - (M-1), airSonar_optimized.py:662
    SNR_GATE_DB: float = 6.0
    KALMAN_Q: float = 0.005                     # airSonar_optimized.py:266
    KALMAN_R: float = 0.1
    TEMPERATURE: float = 20.0                   # SonarWorker.temperature, airSonar_optimized.py:606
    FREQ_WEIGHTS: Tuple[float, float, float] = (0.8, 1.0, 0.9)  # airSonar_optimized.py:238

    # ---- derived, airSonar_optimized.py:93-103 ----
    @property
    def c_air(self) -> float:
        return 343.0 * math.sqrt(1 + (self.BASE_TEMP - 20) / 273.15)

    @property
    def CHIRP_LEN(self) -> float:
        if self.CHIRP_LEN_FIXED is not None:
            return self.CHIRP_LEN_FIXED
        return 2 * self.R_MIN / self.c_air

    @property
    def LISTEN_LEN(self) -> float:
        if self.LISTEN_LEN_FIXED is not None:
            return self.LISTEN_LEN_FIXED
        return 2 * self.R_MAX / self.c_air + 0.003

    # ---- sizes the kernels are launched with ----
    @property
    def n_bands(self) -> int:
        return len(self.BANDS)

    @property
    def n_rx(self) -> int:                       # samples per frame, airSonar_optimized.py:572
        return int(self.FS * self.LISTEN_LEN)

    @property
    def n_chirp(self) -> int:                    # airSonar_optimized.py:160
        return int(self.FS * self.CHIRP_LEN)

    @property
    def n_corr(self) -> int:
        return self.n_rx + self.n_chirp - 1

    @property
    def half(self) -> int:                       # airSonar_optimized.py:204
        return self.n_corr // 2

    @property
    def n_pos(self) -> int:
        return self.n_corr - self.half

    @property
    def min_delay(self) -> int:                  # airSonar_optimized.py:203, :645
        if self.MIN_DELAY_FIXED is not None:
            return self.MIN_DELAY_FIXED
        return int(self.FS * self.CHIRP_LEN * 1.2)

    @property
    def noise_n(self) -> int:
        if self.DETECTOR in (DET_MEANPOW, DET_TWOPEAK):   # realtime_sonar.py:135, 666.py:180
            return int(self.n_pos * 0.2)
        return int(self.SNR_NOISE_MS * self.FS)  # airSonar_optimized.py:209

    @property
    def n_bins(self) -> int:
        return self.n_rx // 2 + 1

    @property
    def padlen(self) -> int:                     # scipy filtfilt default 3*max(len(a), len(b))
        return 3 * (2 * self.IIR_ORDER + 1)

    @property
    def sound_speed(self) -> float:              # airSonar_optimized.py:150-154
        return 331.3 + 0.606 * self.TEMPERATURE

    @property
    def range_offset(self) -> int:
        return self.n_chirp - 1 if self.SUBTRACT_CHIRP else 0

    def with_bands(self, bands) -> "SonarConfig":
        return replace(self, BANDS=tuple(tuple(b) for b in bands))


def optimized_config(bands=BANDS_DEFAULT) -> SonarConfig:
    """airSonar_optimized.py parameters (the headline configuration)."""
    return SonarConfig(BANDS=tuple(bands))


def original_config() -> SonarConfig:
    """airSonar.py parameters: FS 44100, 0.1 s chirp, 0.2 s listen, IIR 8 / FIR 101,
    MAD first-crossing detector, mean fusion, Kalman q=.01 r=.2 (airSonar.py:50-60, :87-121, :203-229)."""
    return SonarConfig(FS=44100, BASE_TEMP=20.0, BANDS=BANDS_ORIGINAL, IIR_ORDER=8, FIR_TAPS=101,
                       CHIRP_LEN_FIXED=0.1, LISTEN_LEN_FIXED=0.2, MIN_DELAY_FIXED=500,
                       DETECTOR=DET_MAD, CONFIDENCE=CONF_NONE, FUSION=FUSE_MEAN,
                       SUBTRACT_CHIRP=False, KALMAN_Q=0.01, KALMAN_R=0.2)


def twopeak_config() -> SonarConfig:
    """666.py parameters: FS 48000, 0.05 s chirp (M = 2400), 0.15 s listen (N = 7200), IIR 6 / FIR 61, two-peak
    detector (delay = echo peak - direct-path peak), clamped confidence, SNR-normalised fusion
    (666.py:57-68, :114-128, :153-200, :372-385, :425-443)."""
    return SonarConfig(BANDS=BANDS_TWOPEAK, CHIRP_LEN_FIXED=0.05, LISTEN_LEN_FIXED=0.15, MIN_DELAY_FIXED=0,
                       DETECTOR=DET_TWOPEAK, CONFIDENCE=CONF_CLAMPED, FUSION=FUSE_SNRNORM, SUBTRACT_CHIRP=False)


def realtime_config() -> SonarConfig:
    """realtime_sonar.py parameters: R_MIN 0.5 (M = 137), distinct bands, mean-power noise floor,
    unclamped confidence terms, fusion weights without epsilon (realtime_sonar.py:49-60, :127-166, :334-397)."""
    return SonarConfig(R_MIN=0.5, BANDS=BANDS_DISTINCT, DETECTOR=DET_MEANPOW, CONFIDENCE=CONF_RAW,
                       FUSION=FUSE_WEIGHTED, SUBTRACT_CHIRP=False, KALMAN_Q=0.01, KALMAN_R=0.2)
