"""GPU: the captured loop of the reference (tests/golden/run_loop_*.npz: SonarWorker.run() itself on the golden and
the adversarial frames) against the HIP chains, and the two detector paths of the named chain against each other."""
import numpy as np
import pytest

from conftest import rx_from_i16
from test_run_loop_golden import check_wire_against_captured_loop

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from dsp_amd import config as C, wire                 # noqa: E402
from dsp_amd.synth import synth_frames                 # noqa: E402
from oracle import echo_oracle as O                    # noqa: E402

RTOL = 1e-4


def _cfg(which):
    return C.optimized_config(C.BANDS_DEFAULT if which == "default" else C.BANDS_DISTINCT)


def _rec(t):
    from dsp_amd.engine import records_to_numpy
    return records_to_numpy(t)


def _valid(rec, nb=3):
    return ((rec["valid_mask"][:, None] >> np.arange(nb)) & 1).astype(bool)


ILL_POSED = "echo_only_in_first_half"
NOISE_BOUND = 1e-14          # of max|corr|: what the reference's FFT-based FIR leaves where the true correlation is ~0


def _ill_posed_index(g):
    """'echo_only_in_first_half' (no noise, the echo before corr[half]): the true values of pos[min_delay:] are ~1e-18
    while max|corr| is 9.3, and what the reference finds there (4e-16) is the rounding noise of fftconvolve, so its
    argmax follows pocketfft's rounding (tools/make_goldens.py, adversarial_frames)."""
    return int(g["n_golden"]) + list(g["adversarial_names"]).index(ILL_POSED)


def _exact_arithmetic_detector(rx, cfg, tables, band):
    """The reference's first_strong_peak on the correlation computed without FFT rounding: filtfilt (bit-identical in
    the kernel), then the FIR and the matched filter as float64 direct sums.
    -> (peak_idx or -1, snr_db, raw corr, max|corr| after the detector's in-place zeroing)."""
    b, a, taps = tables.filters[band]
    band_sig = O.fir_same_direct(O.filtfilt_df2t(b, a, rx), taps)
    corr = np.correlate(band_sig, np.asarray(tables.chirps_f32[band], dtype=np.float64), "full")
    raw = corr.copy()
    idx, snr = O.first_strong_peak(corr, cfg.FS, cfg, cfg.min_delay)
    return (-1 if idx is None else int(idx)), float(snr), raw, float(np.abs(corr).max())


@pytest.mark.parametrize("which", ["default", "distinct"])
def test_named_chain_on_adversarial_frames(which, gold_loop_default, gold_loop_distinct):
    """first_strong_peak path on frames built to stress the argmax (equal and nearly equal echoes, tones with a flat
    correlation envelope, silence next to a loud frame, blind-zone edge): peak indices and validity exactly as the
    reference, SNR and the confidence (through it max|corr|) within 1e-4, in batch order and reversed (the frames then
    meet other partners in the pair kernel).  The one frame whose reference index is decided by the reference's own
    FFT rounding noise is held to the exact-arithmetic index instead (next test)."""
    from dsp_amd.engine import EchoEngine
    g = gold_loop_default if which == "default" else gold_loop_distinct
    cfg = _cfg(which)
    e = EchoEngine(cfg)
    rx = rx_from_i16(g["rx_i16"])
    want_conf = np.zeros_like(g["snr_db"])
    for i, j in np.argwhere(g["valid"]):
        want_conf[i, j] = O.calculate_band_confidence(g["snr_db"][i, j], g["amplitude"][i, j], j)
    ill = _ill_posed_index(g)
    want_idx, want_valid, want_snr = g["peak_idx"].copy(), g["valid"].copy(), g["snr_db"].copy()
    for j in range(cfg.n_bands):                               # the ill-posed frame: exact-arithmetic expectations
        idx, snr, _, amp = _exact_arithmetic_detector(rx[ill], cfg, e.tables, j)
        want_idx[ill, j], want_valid[ill, j], want_snr[ill, j] = idx, idx >= 0, snr
        if idx >= 0:
            want_conf[ill, j] = O.calculate_band_confidence(snr, amp, j)
    for order in (np.arange(len(rx)), np.arange(len(rx))[::-1], np.r_[np.arange(1, len(rx)), 0]):
        rec, _ = e.chain(np.ascontiguousarray(rx[order]), want_spectra=False)
        r = _rec(rec)
        v = want_valid[order]
        assert np.array_equal(_valid(r), v)
        assert np.array_equal(r["peak_idx"], want_idx[order])
        np.testing.assert_allclose(r["snr_db"][v], want_snr[order][v], rtol=RTOL)
        np.testing.assert_allclose(r["conf"][v], want_conf[order][v], rtol=RTOL)
    e.close()


@pytest.mark.parametrize("which", ["default", "distinct"])
def test_ill_posed_frame_takes_the_exact_arithmetic_index(which, gold_loop_default, gold_loop_distinct):
    """'echo_only_in_first_half': the kernels' index is the one the reference's detector gives on the exact float64
    direct-sum correlation, alone and inside a batch; where that differs from the captured reference index, the
    reference's own decision hung on less than NOISE_BOUND of max|corr| -- its correlation (fftconvolve + correlate,
    recomputed here through the oracle) differs from the exact one by rounding noise of that size, and neither holds
    anything larger in pos[min_delay:]."""
    from dsp_amd.engine import EchoEngine
    g = gold_loop_default if which == "default" else gold_loop_distinct
    cfg = _cfg(which)
    e = EchoEngine(cfg)
    rx_all = rx_from_i16(g["rx_i16"])
    ill = _ill_posed_index(g)
    rx = rx_all[ill]
    alone = _rec(e.chain(rx[None], want_spectra=False)[0])
    batch = _rec(e.chain(rx_all, want_spectra=False)[0])
    differing = 0
    for j in range(cfg.n_bands):
        idx, snr, c_exact, _ = _exact_arithmetic_detector(rx, cfg, e.tables, j)
        for r in (alone[0], batch[ill]):
            assert int(r["peak_idx"][j]) == idx, (j, int(r["peak_idx"][j]), idx)
            assert bool((r["valid_mask"] >> j) & 1) == (idx >= 0)
            if idx >= 0:
                assert r["snr_db"][j] == pytest.approx(snr, rel=RTOL)
        # the reference's arithmetic on the same frame reproduces the captured index ...
        c_ref = O.gpu_correlate(O.bandpass(rx, e.tables.filters[j]), e.tables.chirps_f32[j])
        scale = np.abs(c_ref).max()
        ref_idx, _ = O.first_strong_peak(c_ref.copy(), cfg.FS, cfg, cfg.min_delay)
        assert (-1 if ref_idx is None else int(ref_idx)) == int(g["peak_idx"][ill, j])
        # ... and everything either detector could choose from lies below the bound: the reference's values there are
        # the rounding noise of its FFTs (~1e-16 of the maximum), the exact ones the filters' decaying tails
        pos_ref, pos_exact = c_ref[cfg.half + cfg.min_delay:], c_exact[cfg.half + cfg.min_delay:]
        assert np.abs(pos_ref).max() <= NOISE_BOUND * scale and np.abs(pos_exact).max() <= NOISE_BOUND * scale
        assert np.abs(c_ref - c_exact).max() <= NOISE_BOUND * scale
        differing += int(g["peak_idx"][ill, j]) != idx
    assert differing > 0, "the frame no longer separates the reference's rounding from exact arithmetic"
    e.close()


@pytest.mark.parametrize("which", ["default", "distinct"])
def test_chain_wave_reproduces_the_captured_loop(which, gold_loop_default, gold_loop_distinct):
    """echo_chain_wave = the loop body of run() with everything it hands on: the (distance, confidence, snr) triples
    of _process_band_gpu, the log lines / distanceSig arguments / CSV file built from the records by dsp_amd.wire, and
    the waveSig arrays (dB spectra, raw correlations) -- against what the reference's own loop produced."""
    from dsp_amd.engine import BatchedKalman, EchoEngine
    g = gold_loop_default if which == "default" else gold_loop_distinct
    cfg = _cfg(which)
    e = EchoEngine(cfg)
    rx = rx_from_i16(g["rx_i16"])
    rec, spec, corr = e.chain_wave(rx, n_streams=1, kalman=BatchedKalman(e, 1))
    r = _rec(rec)
    v = g["full_valid"]
    assert np.array_equal(_valid(r), v)
    want_idx = np.rint(g["full_distance"] * 2 * cfg.FS / cfg.sound_speed).astype(np.int64) + cfg.range_offset
    assert np.array_equal(r["peak_idx"][v], want_idx[v])
    np.testing.assert_allclose(r["dist"][v], g["full_distance"][v], rtol=RTOL)
    np.testing.assert_allclose(r["conf"][v], g["full_confidence"][v], rtol=RTOL)
    np.testing.assert_allclose(r["snr_db"][v], g["full_snr"][v], rtol=RTOL, atol=1e-6)
    check_wire_against_captured_loop(r, g)
    k = len(g["wave_spectra"])
    waves = wire.wave_sigs(rx[:k], spec[:k].cpu().numpy(), corr[:k].cpu().numpy())
    for i, w in enumerate(waves):
        assert sorted(w.keys()) == list(g["wave_keys"]) and w["rx_id"] == g["wave_rx_id"][i]
        assert w["band_lengths"] == list(g["wave_band_lengths"][i])
        c = np.stack(w["correlations"]); ref = g["wave_correlations"][i]
        assert np.abs(c - ref).max() <= 1e-12 * np.abs(ref).max()
        lin, rl = 10 ** (np.stack(w["band_spectra"]).astype(np.float64) / 20), 10 ** (g["wave_spectra"][i].astype(np.float64) / 20)
        assert (np.abs(lin - rl).max(axis=-1) <= 1e-4 * rl.max(axis=-1)).all()
    # the same call leaves the same records as echo_chain_full
    rec2, spec2 = e.chain(rx, n_streams=1, kalman=BatchedKalman(e, 1), mode="full")
    assert torch.equal(rec, rec2) and torch.equal(spec, spec2)
    e.close()


@pytest.mark.parametrize("which", ["default", "distinct"])
def test_pair_kernel_path_equals_single_frame_kernel_path(which, monkeypatch, gold_loop_default, gold_loop_distinct):
    """The named chain's fast path (float32 FFT matched filter over frame pairs + exact float64 re-evaluation) against
    the single-frame kernel that sums every lag of pos[min_delay:] in float64 (ECHO_BAND2=0): same peak indices and
    validity, SNR to rounding, on 4096 synthetic frames and the adversarial set, at an odd batch size too."""
    from dsp_amd.engine import EchoEngine
    g = gold_loop_default if which == "default" else gold_loop_distinct
    cfg = _cfg(which)
    fast = EchoEngine(cfg)
    monkeypatch.setenv("ECHO_BAND2", "0")
    slow = EchoEngine(cfg)
    monkeypatch.delenv("ECHO_BAND2")
    rx = np.concatenate([synth_frames(4096, cfg, fast.tables.tx_pcm, start=9000), rx_from_i16(g["rx_i16"])])[:4141]
    a, sa = fast.chain(rx)
    b, sb = slow.chain(rx)
    ra, rb = _rec(a), _rec(b)
    assert np.array_equal(ra["peak_idx"], rb["peak_idx"]) and np.array_equal(ra["valid_mask"], rb["valid_mask"])
    v = _valid(ra)
    np.testing.assert_allclose(ra["snr_db"][v], rb["snr_db"][v], rtol=1e-6)
    np.testing.assert_allclose(ra["conf"][v], rb["conf"][v], rtol=RTOL)       # max|corr|: exact here, float32 sums there
    assert np.array_equal(ra["dist"], rb["dist"])
    lin, rl = 10 ** (sa.cpu().numpy().astype(np.float64) / 20), 10 ** (sb.cpu().numpy().astype(np.float64) / 20)
    pk = rl.max(axis=-1, keepdims=True)
    assert (np.abs(lin - rl) <= 1e-5 * np.maximum(pk, 1e-30)).all()
    fast.close(); slow.close()
