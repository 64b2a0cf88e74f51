"""The kernels' barrier-separated phases (dsp_amd/csrc/band_phases.h) compiled for the host and run thread by
thread, against the reference goldens.  This is how the index arithmetic of the band and spectrum kernels
(sliding dot products, swizzled FFT addressing, Bluestein tables, detector reductions) is checked without a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
from scipy.signal import filtfilt

from conftest import rx_from_i16
from dsp_amd import config as C
from dsp_amd.design import build_tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL = os.path.join(ROOT, "tests", "host_emul")


@pytest.fixture(scope="module")
def emul():
    subprocess.check_call(["make", "-C", EMUL], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(EMUL, "libemul.so"))
    P, I = ctypes.c_void_p, ctypes.c_int
    lib.emul_band.argtypes = [I] * 7 + [ctypes.c_double, P, P, I, P, P, P, I, P, P, I, I, P, P, P, I]
    return lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _run(lib, cfg, t, mode, rx=None, d_in=None, corr_io=None, want_spec=True, do_corr=1, corr_fast=0):
    B = len(rx) if rx is not None else len(d_in)
    nb, N = cfg.n_bands, cfg.n_rx
    band = np.zeros((B, nb, N)); spec = np.zeros((B, nb, cfg.n_bins), np.float32) if want_spec else None
    corr = corr_io if corr_io is not None else np.zeros((B, nb, cfg.n_corr))
    snr = np.zeros((B, nb)); amp = np.zeros((B, nb)); peak = np.zeros((B, nb), np.int32)
    taps = np.ascontiguousarray(t.taps); chirps = np.ascontiguousarray(t.chirps)
    rc = lib.emul_band(N, cfg.n_chirp, cfg.FIR_TAPS, nb, cfg.min_delay, cfg.noise_n, cfg.DETECTOR, cfg.SNR_GATE_DB,
                       _ptr(taps), _ptr(chirps), mode, _ptr(rx), _ptr(d_in), _ptr(corr), B, _ptr(band), _ptr(spec), 1,
                       do_corr, _ptr(snr), _ptr(amp), _ptr(peak), corr_fast)
    assert rc == 0
    return dict(band=band, corr=corr, spec=spec, snr=snr, amp=amp, peak=peak)


@pytest.mark.parametrize("which", ["default", "distinct"])
def test_emulated_band_and_spectrum_kernels(which, emul, gold_default, gold_distinct):
    g = gold_default if which == "default" else gold_distinct
    cfg = C.optimized_config(tuple(map(tuple, g["bands"].tolist())))
    t = build_tables(cfg)
    idx = g["stage_frames"]
    rx = rx_from_i16(g["rx_i16"])[idx]
    y = np.stack([[filtfilt(f[0], f[1], x) for f in t.filters] for x in rx])       # IIR stage input
    r = _run(emul, cfg, t, mode=1, d_in=np.ascontiguousarray(y))
    for got, ref in ((r["band"], g["stage_band_sig"]), (r["corr"], g["stage_corr"])):
        scale = np.abs(ref).max(axis=-1, keepdims=True); scale[scale == 0] = 1
        assert (np.abs(got - ref) / scale).max() <= 1e-13
    assert np.array_equal(r["peak"], g["peak_idx"][idx])
    np.testing.assert_allclose(r["snr"], g["snr_db"][idx], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(r["amp"], g["amplitude"][idx], rtol=1e-13)
    lin, ref = 10 ** (r["spec"].astype(np.float64) / 20), 10 ** (g["stage_spec_db"].astype(np.float64) / 20)
    live = ref.max(axis=-1) > 1e-9
    assert ((np.abs(lin - ref).max(axis=-1) / np.maximum(ref.max(axis=-1), 1e-300))[live] <= 1e-5).all()
    # the chain's split-precision correlation: float64 where the detector looks, float32 where only max|corr| is taken
    f = _run(emul, cfg, t, mode=1, d_in=np.ascontiguousarray(y), want_spec=False, corr_fast=1)
    assert np.array_equal(f["peak"], r["peak"]) and np.array_equal(f["snr"], r["snr"])       # detector inputs unchanged
    np.testing.assert_allclose(f["amp"], r["amp"], rtol=2e-6)


def test_emulated_first_strong_peak_mutates_like_reference(emul, gold_default):
    g = gold_default
    cfg = C.optimized_config()
    t = build_tables(cfg)
    corr = g["stage_corr"].copy()
    r = _run(emul, cfg, t, mode=3, d_in=corr, corr_io=corr, want_spec=False)
    assert np.array_equal(corr, g["stage_corr_after"])
    assert np.array_equal(r["peak"], g["peak_idx"][g["stage_frames"]])


def test_emulated_realtime_variant(emul, gold_realtime):
    g = gold_realtime
    cfg = C.realtime_config()
    t = build_tables(cfg)
    rx = rx_from_i16(g["rx_i16"])[:6]
    y = np.stack([[filtfilt(f[0], f[1], x) for f in t.filters] for x in rx])
    r = _run(emul, cfg, t, mode=1, d_in=np.ascontiguousarray(y), want_spec=False)
    assert np.array_equal(r["peak"], g["peak_idx"][:6])
    np.testing.assert_allclose(r["snr"], g["snr_db"][:6], rtol=1e-12)
    np.testing.assert_allclose(r["amp"], g["amplitude"][:6], rtol=1e-13)


def test_emulated_long_frame_path(emul, gold_original):
    """Original parameter set (airSonar.py: N = 8820, M = 4410, FIR 101, MAD first-crossing detector): tiled FIR,
    tiled matched filter and radix-select medians, thread by thread on the host, against the goldens."""
    g = gold_original
    cfg = C.original_config()
    t = build_tables(cfg)
    P, I = ctypes.c_void_p, ctypes.c_int
    emul.emul_long.argtypes = [I] * 6 + [P, P, P, P, I, I, P, P, P]
    rx = rx_from_i16(g["rx_i16"])
    B, nb, N, L = len(rx), cfg.n_bands, cfg.n_rx, cfg.n_corr
    y = np.ascontiguousarray(np.stack([[filtfilt(f[0], f[1], x) for f in t.filters] for x in rx]))
    taps = np.ascontiguousarray(t.taps); chirps = np.ascontiguousarray(t.chirps)
    args = (N, cfg.n_chirp, cfg.FIR_TAPS, nb, cfg.min_delay, cfg.DETECTOR, _ptr(taps), _ptr(chirps))
    # frame 0, every lag: band_sig and the full correlation against scipy's (FFT-method) values
    band = np.zeros((1, nb, N)); corr = np.zeros((1, nb, L)); peak = np.zeros((1, nb), np.int32)
    assert emul.emul_long(*args, _ptr(y[:1]), None, 1, 0, _ptr(band), _ptr(corr), _ptr(peak)) == 0
    # scipy.signal.correlate takes the FFT method at these sizes and transforms the float32 chirp in single
    # precision, so the reference correlation itself carries ~1e-8 of its maximum; the kernel sums in float64
    for got, ref, tol in ((band[0], g["stage_band_sig0"], 1e-12), (corr[0], g["stage_corr0"], 1e-7)):
        scale = np.abs(ref).max(axis=-1, keepdims=True)
        assert (np.abs(got - ref) / scale).max() <= tol
    assert np.array_equal(peak[0], g["peak_idx"][0])
    # all frames the way the chain runs them: only the lags the detector reads
    band = np.zeros((B, nb, N)); corr = np.zeros((B, nb, L)); peak = np.zeros((B, nb), np.int32)
    k_begin = cfg.half + cfg.min_delay
    assert emul.emul_long(*args, _ptr(y), None, B, k_begin, _ptr(band), _ptr(corr), _ptr(peak)) == 0
    assert np.array_equal(peak, g["peak_idx"])
    # direct first_strong_peak on a given correlation: same index, blind zone written back
    c0 = g["stage_corr0"].copy()[None]
    peak = np.zeros((1, nb), np.int32)
    assert emul.emul_long(*args, None, _ptr(c0), 1, 0, None, None, _ptr(peak)) == 0
    assert np.array_equal(peak[0], g["peak_idx"][0])
    want = g["stage_corr0"].copy(); want[:, cfg.half:cfg.half + cfg.min_delay] = 0
    assert np.array_equal(c0[0], want)
    # dB spectrum of the 8820-sample band signal (packed real FFT of length 4410 = 7*7*5*3*3*2, float32)
    emul.emul_long_spec.argtypes = [I, I, I, I, P, I, P]
    bs = np.ascontiguousarray(g["stage_band_sig0"][None])
    spec = np.zeros((1, nb, N // 2 + 1), np.float32)
    assert emul.emul_long_spec(N, cfg.n_chirp, cfg.FIR_TAPS, nb, _ptr(bs), 1, _ptr(spec)) == 0
    ref = np.abs(np.fft.rfft(bs[0], axis=-1))
    lin = 10 ** (spec[0].astype(np.float64) / 20)
    assert (np.abs(lin - ref).max(axis=-1) / ref.max(axis=-1)).max() <= 1e-5


def test_emulated_odd_length_spectrum(emul):
    """rfft of an odd number of samples (N = 4795): chirp-z convolution through the mixed-radix FFT, thread by thread
    on the host, against np.fft.rfft."""
    P, I = ctypes.c_void_p, ctypes.c_int
    emul.emul_long_spec.argtypes = [I, I, I, I, P, I, P]
    rng = np.random.default_rng(2)
    N, nb = 4795, 2
    t = np.arange(N)
    bs = rng.standard_normal((1, nb, N)) * 0.01 + 0.3 * np.sin(2 * np.pi * 0.0913 * t) + 0.1 * np.cos(2 * np.pi * 0.31 * t)
    bs = np.ascontiguousarray(bs)
    spec = np.zeros((1, nb, N // 2 + 1), np.float32)
    assert emul.emul_long_spec(N, 275, 61, nb, _ptr(bs), 1, _ptr(spec)) == 0
    ref = np.abs(np.fft.rfft(bs[0], axis=-1))
    lin = 10 ** (spec[0].astype(np.float64) / 20)
    assert (np.abs(lin - ref).max(axis=-1) / ref.max(axis=-1)).max() <= 1e-5


def test_emulated_odd_length_spectrum_in_place(emul):
    """Odd frame lengths beyond the two-buffer chirp-z kernel (N > ~6700): the same convolution with the transform in
    place in one LDS buffer (ospec_phases.h), thread by thread, against np.fft.rfft; lengths with other radix sets, the
    largest one near the LDS bound, and an even length whose half is prime."""
    P, I = ctypes.c_void_p, ctypes.c_int
    emul.emul_ospec.argtypes = [I, I, I, I, P, I, I, P]
    rng = np.random.default_rng(9)
    for N in (8821, 7001, 12001, 8818):                      # 8818 = 2 * 4409: even, its half a prime
        nb = 2
        bs = rng.standard_normal((1, nb, N)) * np.hanning(N)
        bs[0, 1] *= 1e-3
        bs = np.ascontiguousarray(bs)
        spec = np.zeros((1, nb, N // 2 + 1), np.float32)
        m = emul.emul_ospec(N, 275, 61, nb, _ptr(bs), 1, 0, _ptr(spec))
        assert m >= N + N // 2 and (m + m // 8 + 1) * 8 <= 160 * 1024, (N, m)
        ref = np.abs(np.fft.rfft(bs, axis=-1))
        assert (np.abs(spec - ref).max(axis=-1) / ref.max(axis=-1)).max() <= 2e-5, N
    # even lengths and short odd ones keep their own kernels
    bs = np.zeros((1, 1, 4795)); spec = np.zeros((1, 1, 2398), np.float32)
    assert emul.emul_ospec(4795, 275, 61, 1, _ptr(bs), 1, 0, _ptr(spec)) == -2
    assert emul.emul_ospec(8820, 275, 61, 1, _ptr(bs), 1, 0, _ptr(spec)) == -2


def test_emulated_long_frame_detectors_against_oracle(emul):
    """The four detectors of the tiled long-frame path, thread by thread, on synthetic correlations (noise, one or two
    peaks, silence, ties, a peak inside the blind zone) against the oracle's first_strong_peak functions: indices
    exact, SNR to 1e-12, and the in-place blind-zone zeroing."""
    from oracle import echo_oracle as O
    P, I, D = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    emul.emul_long_detect.argtypes = [I, I, I, I, I, I, I, D, P, I, P, P, P, P]
    rng = np.random.default_rng(3)
    N, M, nb, B = 5200, 400, 2, 6
    L = N + M - 1
    half = L // 2
    npos = L - half
    fs = 48000

    def make():
        c = rng.standard_normal((B, nb, L)) * 1e-3
        c[0, 0, half + 900] = 0.5                                  # one clear echo
        c[0, 1, half + 100] = 2.0; c[0, 1, half + 1500] = 0.3      # strongest sample inside the blind zone
        c[1, 0, half + 700] = 0.4; c[1, 0, half + 1900] = 0.4      # tie: first index wins
        c[1, 1] = 0.0                                              # silence
        c[2, 0, : half] *= 500.0                                   # amplitude dominated by the first half
        c[2, 1, half + npos - 1] = 1.0                             # peak at the last lag
        c[3, 0] += 2e-3                                            # median of pos away from zero
        c[3, 1, half + 1000:] = 0.0                                # more than half of pos is zero: MAD rank in the zeros
        c[4, 0] = np.where(c[4, 0] < 0, -1e-3, 1e-3); c[4, 0, half + 2000] = 0.1    # one magnitude shared by every lag
        c[4, 1] = np.round(c[4, 1] * 2e3) / 2e3                    # a few distinct magnitudes, many exact zeros
        return c

    for det, min_delay, noise_n in ((0, 331, 240), (0, 2300, 240), (2, 164, int(npos * 0.2)), (1, 500, 0), (3, 0, int(npos * 0.2))):
        c = make()
        ref_c = c.copy()
        peak = np.zeros((B, nb), np.int32); snr = np.zeros((B, nb)); amp = np.zeros((B, nb)); valid = np.zeros((B, nb), np.int32)
        assert emul.emul_long_detect(N, M, 101, nb, min_delay, det, noise_n, 6.0, _ptr(c), B, _ptr(peak), _ptr(snr),
                                     _ptr(amp), _ptr(valid)) == 0
        for b in range(B):
            for j in range(nb):
                r = ref_c[b, j]
                if det == 0:
                    class Cfg: SNR_NOISE_MS = noise_n / fs; CHIRP_LEN = 0.0
                    want, want_snr = O.first_strong_peak(r, fs, Cfg, min_delay)
                elif det == 2:
                    class Cfg: CHIRP_LEN = 0.0
                    want, want_snr = O.first_strong_peak_meanpow(r, fs, Cfg, min_delay)
                elif det == 1:
                    want, want_snr = O.first_strong_peak_mad(r, fs, min_delay), 0.0
                else:
                    want, want_snr = O.first_strong_peak_twopeak(r, fs)
                if det == 3:
                    assert bool(valid[b, j]) == (want is not None), (det, b, j)
                    if want is not None:
                        assert peak[b, j] == want, (det, b, j)
                else:
                    assert peak[b, j] == (-1 if want is None else want), (det, min_delay, b, j, peak[b, j], want)
                if det != 1:
                    np.testing.assert_allclose(snr[b, j], want_snr, rtol=1e-12, atol=1e-12)
                    np.testing.assert_allclose(amp[b, j], np.abs(r).max(), rtol=0, atol=0)
                if det != 3:
                    assert np.array_equal(c[b, j], r), (det, b, j)       # same in-place mutation as the reference's


def test_emulated_fft_matched_filter(emul, gold_original):
    """Matched filter of long frames by overlap-save FFT (cfft_phases.h), thread by thread: against the exact direct
    sum (np.correlate in float64) for the two long-frame parameter sets, every lag and the chain's lag range, and
    against the reference's own correlation of the golden frame."""
    P, I = ctypes.c_void_p, ctypes.c_int
    emul.emul_long_corr_fft.argtypes = [I, I, I, I, P, P, I, I, I, P]
    rng = np.random.default_rng(5)
    for N, M, k_begin, nt in ((8820, 4410, 0, 1024), (8820, 4410, 6614 + 500, 1024), (7200, 2400, 0, 256), (1000, 600, 37, 64)):
        nb, B, L = 2, 1, N + M - 1
        chirps = np.ascontiguousarray(rng.standard_normal((nb, M)))
        band = np.ascontiguousarray(rng.standard_normal((B, nb, N)))
        band[0, 1, N // 3:] *= 1e-3                                  # dynamic range inside one frame
        corr = np.full((B, nb, L), np.nan)
        assert emul.emul_long_corr_fft(N, M, 101, nb, _ptr(chirps), _ptr(band), B, k_begin, nt, _ptr(corr)) > 0
        for j in range(nb):
            ref = np.correlate(band[0, j], chirps[j], "full")
            assert np.isnan(corr[0, j, :k_begin]).all()             # lags below k_begin are not touched
            assert np.abs(corr[0, j, k_begin:] - ref[k_begin:]).max() <= 2e-14 * np.abs(ref).max() * np.sqrt(M)
    g = gold_original
    cfg = C.original_config()
    t = build_tables(cfg)
    nb, N, L = cfg.n_bands, cfg.n_rx, cfg.n_corr
    bs = np.ascontiguousarray(g["stage_band_sig0"][None])
    chirps = np.ascontiguousarray(t.chirps)
    corr = np.zeros((1, nb, L))
    assert emul.emul_long_corr_fft(N, cfg.n_chirp, cfg.FIR_TAPS, nb, _ptr(chirps), _ptr(bs), 1, 0, 1024, _ptr(corr)) > 0
    ref = g["stage_corr0"]
    assert (np.abs(corr[0] - ref) / np.abs(ref).max(axis=-1, keepdims=True)).max() <= 1e-7


def test_emulated_real_transform_matched_filter(emul):
    """Matched filter of long frames as one real cyclic correlation (rcf_phases.h), thread by thread: every lag from
    k_begin on against the exact direct sum (np.correlate in float64), lags below k_begin untouched; then the MAD and
    two-peak detectors run in the same workgroup on the packed layout against the oracle's detectors applied to those
    very correlation values (indices exact, SNR to 1e-12)."""
    from oracle import echo_oracle as O
    P, I, D = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    emul.emul_long_rcorr.argtypes = [I, I, I, I, P, P, I, I, I, P, I, I, I, D, P, P, P, P, I]
    rng = np.random.default_rng(11)
    fs = 48000
    # (N, M, detector, min_delay): airSonar.py's sizes (MAD detector, lags from half + min_delay), 666.py's (two-peak,
    # every lag), and two small ones with other radices
    for N, M, det, min_delay, nt in ((8820, 4410, 1, 500, 768), (7200, 2400, 3, 0, 768), (3000, 600, 1, 37, 512),
                                     (2000, 700, 3, 0, 512)):
        nb, B, L = 2, 2, N + M - 1
        half = L // 2
        k_begin = half + min_delay if det == 1 else 0
        chirps = np.ascontiguousarray(rng.standard_normal((nb, M)))
        band = np.ascontiguousarray(rng.standard_normal((B, nb, N)) * 1e-2)
        e0 = N // 2 + 300

        def echo(b, j, at, gain):
            n = min(M, N - at)
            band[b, j, at:at + n] += gain * chirps[j][:n]

        echo(0, 0, e0, 0.8)                                          # one echo
        echo(0, 1, e0, 0.8); echo(0, 1, e0 + N // 10, 0.5)          # two
        band[1, 1] = 0.0                                             # silence
        corr = np.full((B, nb, L), np.nan)
        peak = np.zeros((B, nb), np.int32); snr = np.zeros((B, nb)); amp = np.zeros((B, nb)); valid = np.zeros((B, nb), np.int32)
        H = emul.emul_long_rcorr(N, M, 101, nb, _ptr(chirps), _ptr(band), B, k_begin, nt, _ptr(corr), 0, min_delay, 0,
                                 6.0, None, None, None, None, 0)
        assert H > 0 and 2 * H >= L - k_begin + M - 1
        for b in range(B):
            for j in range(nb):
                ref = np.correlate(band[b, j], chirps[j], "full")
                assert np.isnan(corr[b, j, :k_begin]).all()
                assert np.abs(corr[b, j, k_begin:] - ref[k_begin:]).max() <= 4e-15 * max(np.abs(ref).max(), 1e-300)
        # the same lags in blocks, each a transform that leaves half of a compute unit's LDS to another workgroup
        if M >= 512 and N >= 7200:
            blocks = np.full((B, nb, L), np.nan)
            r = emul.emul_long_rcorr(N, M, 101, nb, _ptr(chirps), _ptr(band), B, k_begin, 512, _ptr(blocks), 0, min_delay,
                                     0, 6.0, None, None, None, None, 1)
            Hs, nblk = r // 16, r % 16
            assert r > 0 and nblk >= 2 and (Hs + Hs // 8 + 1) * 16 <= 80 * 1024 and nblk * (2 * Hs - (M - 1)) >= L - k_begin
            assert np.isnan(blocks[:, :, :k_begin]).all()
            scale = np.abs(corr[:, :, k_begin:]).max(axis=-1, keepdims=True)
            assert (np.abs(blocks[:, :, k_begin:] - corr[:, :, k_begin:]) <= 4e-15 * np.maximum(scale, 1e-300)).all()
        noise_n = int((L - half) * 0.2)
        assert emul.emul_long_rcorr(N, M, 101, nb, _ptr(chirps), _ptr(band), B, k_begin, nt, None, det, min_delay,
                                    noise_n, 6.0, _ptr(peak), _ptr(snr), _ptr(amp), _ptr(valid), 0) == H
        for b in range(B):
            for j in range(nb):
                c = np.where(np.isnan(corr[b, j]), 0.0, corr[b, j])
                if det == 1:
                    want = O.first_strong_peak_mad(c, fs, min_delay)
                    assert peak[b, j] == (-1 if want is None else want), (N, b, j, peak[b, j], want)
                else:
                    want, want_snr = O.first_strong_peak_twopeak(c, fs)
                    assert bool(valid[b, j]) == (want is not None), (N, b, j, peak[b, j], want)
                    if want is not None:
                        assert peak[b, j] == want
                    np.testing.assert_allclose(snr[b, j], want_snr, rtol=1e-12, atol=1e-12)
                    np.testing.assert_allclose(amp[b, j], np.abs(c).max(), rtol=0, atol=0)


def test_emulated_generic_fft():
    subprocess.check_call(["make", "-C", EMUL], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(EMUL, "libemul.so"))
    P, I = ctypes.c_void_p, ctypes.c_int
    lib.emul_gfft.argtypes = [I, I, I, P, P]
    rng = np.random.default_rng(0)
    for n in (2278, 4410, 2205, 64, 17, 6, 4, 3, 2, 1):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        xin = np.ascontiguousarray(np.stack([x.real, x.imag], -1))
        for inv in (0, 1):
            ref = np.fft.ifft(x) * n if inv else np.fft.fft(x)
            for use_float, tol in ((0, 5e-15), (1, 1e-6)):
                out = np.zeros((n, 2))
                assert lib.emul_gfft(n, inv, use_float, _ptr(xin), _ptr(out)) > 0
                assert np.abs(out[:, 0] + 1j * out[:, 1] - ref).max() <= tol * np.abs(ref).max()


def test_emulated_fft_with_small_radix_butterflies():
    """The same transform with the small prime factors merged into radices of up to 10 and every such pass run as
    in-register butterflies (the plan of the long-frame spectrum kernels): every radix 2..10 alone and in products, the
    lengths the kernels use (4410 = 10*9*7*7, 3600, the chirp-z lengths of odd frames), lengths that keep a large prime
    factor as an r-term pass beside butterfly passes (2278 = 67*17*2), forward and inverse, float32 and float64."""
    subprocess.check_call(["make", "-C", EMUL], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(EMUL, "libemul.so"))
    P, I = ctypes.c_void_p, ctypes.c_int
    lib.emul_gfft_bfly.argtypes = [I, I, I, P, P]
    rng = np.random.default_rng(1)
    for n in (2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 14, 35, 49, 81, 100, 210, 4410, 2205, 3600, 7200, 7203, 2278, 11 * 30, 1):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        xin = np.ascontiguousarray(np.stack([x.real, x.imag], -1))
        for inv in (0, 1):
            ref = np.fft.ifft(x) * n if inv else np.fft.fft(x)
            for use_float, tol in ((0, 5e-15), (1, 1e-6)):
                out = np.zeros((n, 2))
                assert lib.emul_gfft_bfly(n, inv, use_float, _ptr(xin), _ptr(out)) > 0, n
                err = np.abs(out[:, 0] + 1j * out[:, 1] - ref).max() / np.abs(ref).max()
                assert err <= tol, (n, inv, use_float, err)


def test_emulated_envelope_branch(emul, gold_default):
    """Correlation envelope branch (float64 Hilbert transform, uniform_filter1d(5), find_peaks with height, distance and
    prominence, peak choice, SNR, corr_quality) and the band-signal quality that rides on the spectrum kernel, thread by
    thread on the host, against SciPy (through the oracle) and the goldens.  Both forms of the Hilbert transform: one
    real cyclic convolution of length 9216 through the in-place 4608-point transform (what signals of 3072 samples and
    more take), and the two half-length transforms (mixed-radix sums for L / 2 = 2278, Bluestein for 2141)."""
    from oracle import echo_oracle as O
    g = gold_default
    cfg = C.optimized_config()
    t = build_tables(cfg)
    P, I = ctypes.c_void_p, ctypes.c_int
    emul.emul_env_opt.argtypes = [I, I, I, I, P, I, P, P, P, P, P, P, I]
    corr = np.ascontiguousarray(g["stage_corr"][:6])
    B, nb, L = corr.shape
    ref = g["stage_corr_envelope"][:6].astype(np.float64)
    bs = np.ascontiguousarray(g["stage_band_sig"][:3])
    Bb, _, N = bs.shape
    for allow_lin in (1, 0):
        env = np.zeros_like(corr); peak = np.zeros((B, nb), np.int32); npk = np.zeros((B, nb), np.int32)
        snr = np.zeros((B, nb)); amp = np.zeros((B, nb)); q = np.zeros((B, nb))
        assert emul.emul_env_opt(L, nb, cfg.min_delay, 5, _ptr(corr), B, _ptr(env), _ptr(peak), _ptr(snr), _ptr(amp),
                                 _ptr(q), _ptr(npk), allow_lin) == (2 if allow_lin else 1)
        assert np.abs(env - ref).max() <= 2e-7 * np.abs(ref).max()
        for f in range(B):
            for b in range(nb):
                e, pk, qual = O.gpu_envelope_detection(corr[f, b], 5)
                assert np.abs(env[f, b] - e).max() <= 1e-12 * max(e.max(), 1e-300)
                vp = pk[pk >= cfg.min_delay]
                want = int(vp[np.argmax(e[vp])]) if len(vp) else -1
                assert (want, len(pk)) == (peak[f, b], npk[f, b])
                assert abs(qual - q[f, b]) <= 1e-9 and amp[f, b] == np.abs(corr[f, b]).max()
        # the band-signal call gpu_envelope_detection(band_sig, 'hilbert', 3): N = 4282 = 2 * 2141 (prime), so the
        # half-length transforms, where they are taken, run as Bluestein convolutions; envelope, peaks and quality
        # against SciPy
        envb = np.zeros_like(bs); pk = np.zeros((Bb, nb), np.int32); npkb = np.zeros((Bb, nb), np.int32)
        snrb = np.zeros((Bb, nb)); ampb = np.zeros((Bb, nb)); qb = np.zeros((Bb, nb))
        assert emul.emul_env_opt(N, nb, 0, 3, _ptr(bs), Bb, _ptr(envb), _ptr(pk), _ptr(snrb), _ptr(ampb), _ptr(qb),
                                 _ptr(npkb), allow_lin) == (2 if allow_lin else 1)
        for f in range(Bb):
            for b in range(nb):
                e, pks, qual = O.gpu_envelope_detection(bs[f, b], 3)
                assert np.abs(envb[f, b] - e).max() <= 1e-12 * max(e.max(), 1e-300)
                assert len(pks) == npkb[f, b] and abs(qual - qb[f, b]) <= 1e-9
    # a signal below the 3072 samples from which the long convolution pays keeps the half-length transforms; the
    # longest the kernel takes (4608) and an odd-half length go through the convolution
    rng = np.random.default_rng(5)
    for Lx, want_rc in ((3000, 1), (3072, 2), (4608, 2), (4562, 2)):
        x = np.ascontiguousarray(rng.standard_normal((1, 1, Lx)) * np.hanning(Lx))
        envx = np.zeros_like(x); pk1 = np.zeros((1, 1), np.int32); n1 = np.zeros((1, 1), np.int32)
        s1 = np.zeros((1, 1)); a1 = np.zeros((1, 1)); q1 = np.zeros((1, 1))
        assert emul.emul_env_opt(Lx, 1, 0, 5, _ptr(x), 1, _ptr(envx), _ptr(pk1), _ptr(s1), _ptr(a1), _ptr(q1), _ptr(n1),
                                 1) == want_rc
        e, pks, qual = O.gpu_envelope_detection(x[0, 0], 5)
        assert np.abs(envx[0, 0] - e).max() <= 1e-12 * e.max()
        assert len(pks) == n1[0, 0] and abs(qual - q1[0, 0]) <= 1e-9
    # signal_quality from the float32 spectrum kernel with the second Bluestein round
    emul.emul_spec_hilbert.argtypes = [I, I, I, I, P, P, P, I, P, P, I]
    band = np.ascontiguousarray(g["stage_band_sig"][:4].astype(np.float32))
    taps = np.ascontiguousarray(t.taps); chirps = np.ascontiguousarray(t.chirps)
    for f9 in (0, 1):                    # 8192-point and 4608-point cyclic convolutions
        spec = np.zeros((4, nb, cfg.n_bins), np.float32); sq = np.zeros((4, nb))
        assert emul.emul_spec_hilbert(cfg.n_rx, cfg.n_chirp, cfg.FIR_TAPS, nb, _ptr(taps), _ptr(chirps), _ptr(band), 4,
                                      _ptr(spec), _ptr(sq), f9) == 0
        np.testing.assert_allclose(sq, g["signal_quality"][g["stage_frames"][:4]], rtol=0, atol=2e-6)
        lin, rl = 10 ** (spec.astype(np.float64) / 20), 10 ** (g["stage_spec_db"][:4].astype(np.float64) / 20)
        live = rl.max(axis=-1) > 1e-9
        assert ((np.abs(lin - rl).max(axis=-1) / np.maximum(rl.max(axis=-1), 1e-300))[live] <= 1e-5).all()
    # plain spectrum through the 4608-point variant
    emul.emul_spec9.argtypes = [I, I, I, I, P, P, P, I, P]
    bd = np.ascontiguousarray(g["stage_band_sig"][:4])
    spec = np.zeros((4, nb, cfg.n_bins), np.float32)
    assert emul.emul_spec9(cfg.n_rx, cfg.n_chirp, cfg.FIR_TAPS, nb, _ptr(taps), _ptr(chirps), _ptr(bd), 4, _ptr(spec)) == 0
    lin = 10 ** (spec.astype(np.float64) / 20)
    assert ((np.abs(lin - rl).max(axis=-1) / np.maximum(rl.max(axis=-1), 1e-300))[live] <= 1e-5).all()


def test_emulated_fused_full_chain_kernel(emul, gold_default):
    """FIR + correlation as one real cyclic convolution + envelope branch in one workgroup (fce_phases.h), thread by
    thread on the host: band signal and correlation against the reference's stages in the goldens, envelope peaks,
    SNR and quality against SciPy on the reference's correlation."""
    from oracle import echo_oracle as O
    g = gold_default
    cfg = C.optimized_config()
    t = build_tables(cfg)
    P, I = ctypes.c_void_p, ctypes.c_int
    emul.emul_fce.argtypes = [I, I, I, I, I, P, P, P, I, P, P, P, P, P, P, P]
    B, nb = 4, cfg.n_bands
    rx = rx_from_i16(g["rx_i16"])[g["stage_frames"][:B]]
    y = np.ascontiguousarray(np.stack([[filtfilt(f[0], f[1], x) for f in t.filters] for x in rx]))
    taps = np.ascontiguousarray(t.taps); chirps = np.ascontiguousarray(t.chirps)
    corr = np.zeros((B, nb, cfg.n_corr)); bandf = np.zeros((B, nb, cfg.n_rx), np.float32)
    peak = np.zeros((B, nb), np.int32); npk = np.zeros((B, nb), np.int32)
    snr = np.zeros((B, nb)); amp = np.zeros((B, nb)); q = np.zeros((B, nb))
    assert emul.emul_fce(cfg.n_rx, cfg.n_chirp, cfg.FIR_TAPS, nb, cfg.min_delay, _ptr(taps), _ptr(chirps), _ptr(y), B,
                         _ptr(corr), _ptr(bandf), _ptr(peak), _ptr(snr), _ptr(amp), _ptr(q), _ptr(npk)) == 0
    ref_band = g["stage_band_sig"][:B].astype(np.float64)
    ref_corr = g["stage_corr"][:B].astype(np.float64)
    assert np.abs(bandf - ref_band).max() <= 1e-6 * np.abs(ref_band).max()
    for f in range(B):
        for b in range(nb):
            scale = np.abs(ref_corr[f, b]).max()
            assert np.abs(corr[f, b] - ref_corr[f, b]).max() <= 1e-13 * scale
            e, pk, qual = O.gpu_envelope_detection(ref_corr[f, b], 5)
            vp = pk[pk >= cfg.min_delay]
            want = int(vp[np.argmax(e[vp])]) if len(vp) else -1
            assert (want, len(pk)) == (peak[f, b], npk[f, b])
            assert abs(qual - q[f, b]) <= 1e-9 and abs(amp[f, b] - scale) <= 1e-13 * scale
    # other sizes: the shortest correlation the fused kernel takes upward, the longest (4608 lags exactly), other taps
    from dataclasses import replace
    from dsp_amd.synth import synth_frames
    for n_rx, n_chirp, ntaps in ((3000, 275, 61), (4282, 327, 61), (4100, 275, 101)):
        cfg2 = replace(C.optimized_config(C.BANDS_DISTINCT), LISTEN_LEN_FIXED=(n_rx + 0.5) / 48000,
                       CHIRP_LEN_FIXED=(n_chirp + 0.5) / 48000, FIR_TAPS=ntaps)
        t2 = build_tables(cfg2)
        rx2 = synth_frames(2, cfg2, t2.tx_pcm, start=3)
        y2 = np.ascontiguousarray(np.stack([[filtfilt(f[0], f[1], x) for f in t2.filters] for x in rx2]))
        taps2 = np.ascontiguousarray(t2.taps); chirps2 = np.ascontiguousarray(t2.chirps)
        corr2 = np.zeros((2, nb, cfg2.n_corr)); bandf2 = np.zeros((2, nb, cfg2.n_rx), np.float32)
        peak2 = np.zeros((2, nb), np.int32); npk2 = np.zeros((2, nb), np.int32)
        snr2 = np.zeros((2, nb)); amp2 = np.zeros((2, nb)); q2 = np.zeros((2, nb))
        assert emul.emul_fce(cfg2.n_rx, cfg2.n_chirp, ntaps, nb, cfg2.min_delay, _ptr(taps2), _ptr(chirps2), _ptr(y2), 2,
                             _ptr(corr2), _ptr(bandf2), _ptr(peak2), _ptr(snr2), _ptr(amp2), _ptr(q2), _ptr(npk2)) == 0
        for f in range(2):
            for b in range(nb):
                band = O.bandpass(rx2[f], t2.filters[b])
                rc = O.gpu_correlate(band, t2.chirps_f32[b])
                assert np.abs(bandf2[f, b] - band).max() <= 1e-6 * np.abs(band).max()
                assert np.abs(corr2[f, b] - rc).max() <= 1e-13 * np.abs(rc).max()
                e, pk, qual = O.gpu_envelope_detection(rc, 5)
                vp = pk[pk >= cfg2.min_delay]
                want = int(vp[np.argmax(e[vp])]) if len(vp) else -1
                assert (want, len(pk)) == (peak2[f, b], npk2[f, b]) and abs(qual - q2[f, b]) <= 1e-9
