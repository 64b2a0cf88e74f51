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
