"""The drop-in boundary on the GPU: the reference's own call shapes through ``dsp_amd.dropin`` (one band's 1-D signal,
a chirp, a (b, a, taps) tuple, scalars), and argument validation -- a mis-shaped tensor must raise EchoError before it
reaches a kernel.  Expected values: the goldens produced by importing the reference, and SciPy through the oracle."""
import numpy as np
import pytest

from conftest import rx_from_i16

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from dsp_amd import config as C                      # noqa: E402
from dsp_amd._lib import EchoError                    # noqa: E402
from oracle import echo_oracle as O                   # noqa: E402


@pytest.fixture(scope="module")
def engines():
    from dsp_amd.engine import EchoEngine
    e = {"default": EchoEngine(C.optimized_config(C.BANDS_DEFAULT)), "distinct": EchoEngine(C.optimized_config(C.BANDS_DISTINCT))}
    yield e
    for v in e.values():
        v.close()


class _Worker:
    """``SonarWorker._process_band_gpu`` of airSonar_optimized.py:629-676, line for line, with the module-level names
    bound to dsp_amd.dropin (what a maintainer gets by replacing the reference's gpu_* functions)."""

    def __init__(self, D, cfg):
        self.D, self.cfg, self.temperature = D, cfg, 20.0

    def _process_band_gpu(self, rx, chirp_sig, filt, band_idx):
        D, cfg = self.D, self.cfg
        band_sig = D.bandpass(rx, filt)
        envelope, peaks, signal_quality = D.gpu_envelope_detection(band_sig, method='hilbert', smooth_window=3)
        corr = D.gpu_correlate(band_sig, chirp_sig)
        corr_envelope, corr_peaks, corr_quality = D.gpu_envelope_detection(corr, method='hilbert', smooth_window=5)
        min_delay_samples = int(cfg.FS * cfg.CHIRP_LEN * 1.2)
        valid_peaks = corr_peaks[corr_peaks >= min_delay_samples]
        if len(valid_peaks) > 0:
            peak_idx = valid_peaks[np.argmax(corr_envelope[valid_peaks])]
            signal_power = corr_envelope[peak_idx] ** 2
            noise_power = np.mean(corr_envelope[:min_delay_samples] ** 2) + 1e-10
            snr = 10 * np.log10(signal_power / noise_power)
        else:
            peak_idx, snr = D.first_strong_peak(corr, cfg.FS, min_delay_samples)
            if peak_idx is None:
                return None, 0.0, 0.0
        delay = peak_idx - (len(chirp_sig) - 1)
        distance = delay / cfg.FS * D.calculate_sound_speed(self.temperature) / 2
        amplitude = np.max(np.abs(corr))
        base_confidence = D.calculate_band_confidence(snr, amplitude, band_idx)
        quality_factor = (signal_quality + corr_quality) / 2.0
        enhanced_confidence = base_confidence * (0.5 + 0.5 * quality_factor)
        return distance, enhanced_confidence, snr


@pytest.mark.parametrize("which", ["default", "distinct"])
def test_process_band_gpu_transliterated_with_dropin_names(which, engines, gold_default, gold_distinct):
    """The reference's per-band function, unchanged but for the module the names come from, reproduces the
    (distance, enhanced_confidence, snr) triples the reference itself produced, frame by frame and band by band; so do
    the fusion lines of run() (:733-739) with dropin.normalize_confidences and dropin.ScalarKalman."""
    from dsp_amd import dropin as D
    g = gold_default if which == "default" else gold_distinct
    D.use_engine(engines[which])
    cfg = engines[which].cfg
    w = _Worker(D, cfg)
    tx_pcm, chirps = D.generate_chirps(cfg.FS, cfg.CHIRP_LEN)
    filters = D.design_filters(cfg.FS)
    assert np.array_equal(tx_pcm, g["tx_pcm"])
    kf = D.ScalarKalman()
    rx_all = rx_from_i16(g["rx_i16"])
    for i, rx in enumerate(rx_all):
        results = []
        for j, (chirp, filt) in enumerate(zip(chirps, filters)):
            distance, confidence, snr = w._process_band_gpu(rx, chirp, filt, j)
            assert (distance is not None) == bool(g["full_valid"][i, j]), (i, j)
            if distance is None:
                continue
            assert distance == pytest.approx(g["full_distance"][i, j], rel=1e-12, abs=1e-15), (i, j)   # same index
            assert snr == pytest.approx(g["full_snr"][i, j], rel=1e-9, abs=1e-9), (i, j)
            assert confidence == pytest.approx(g["full_confidence"][i, j], rel=1e-9), (i, j)
            results.append((distance, confidence, snr))
        if results:
            distances, confidences, snrs = zip(*results)
            confidences_norm = D.normalize_confidences(confidences)
            weights = confidences_norm / 100.0 + 1e-9
            weighted_dist = np.average(distances, weights=weights)
            avg_confidence = np.mean(confidences_norm)
            dist_kf = kf.update(weighted_dist)
            assert weighted_dist == pytest.approx(g["full_fused_dist"][i], rel=1e-9)
            assert avg_confidence == pytest.approx(g["full_avg_conf"][i], rel=1e-9)
            assert dist_kf == pytest.approx(g["full_kalman"][i], rel=1e-9)
        else:
            assert not g["full_valid_any"][i]


def test_process_band_with_output_call_shapes(engines, gold_distinct):
    """process_band_with_output (:703-715): bandpass(rx[N], filt), the band-signal envelope call, mag2db(rfft) as
    spectrum_db / gpu_spectrum on one band, gpu_correlate(band_sig[N], chirp[M]) -- against the per-stage goldens."""
    from dsp_amd import dropin as D
    g = gold_distinct
    D.use_engine(engines["distinct"])
    filters = D.design_filters()
    _, chirps = D.generate_chirps()
    for k, i in enumerate(g["stage_frames"][:4]):
        rx = rx_from_i16(g["rx_i16"][i])
        for j in range(3):
            band_sig = D.bandpass(rx, filters[j])
            ref = g["stage_band_sig"][k, j]
            assert band_sig.shape == ref.shape and band_sig.dtype == np.float64
            assert np.abs(band_sig - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-300)
            env, peaks, q = D.gpu_envelope_detection(band_sig, method='hilbert', smooth_window=3)
            e_ref, p_ref, q_ref = O.gpu_envelope_detection(ref, 3)
            assert np.abs(env - e_ref).max() <= 1e-11 * max(e_ref.max(), 1e-300)
            assert np.array_equal(peaks, p_ref) and abs(q - q_ref) <= 1e-9
            assert abs(q - g["signal_quality"][i, j]) <= 1e-9
            spec = D.spectrum_db(band_sig)
            lin, rl = 10 ** (spec.astype(np.float64) / 20), 10 ** (g["stage_spec_db"][k, j].astype(np.float64) / 20)
            assert spec.shape == (2142,) and np.abs(lin - rl).max() <= 1e-4 * max(rl.max(), 1e-30)
            mag = D.gpu_spectrum(band_sig)
            assert np.abs(mag - np.abs(np.fft.rfft(ref))).max() <= 1e-4 * max(rl.max(), 1e-30)
            corr = D.gpu_correlate(band_sig, chirps[j])
            refc = g["stage_corr"][k, j]
            assert corr.shape == refc.shape and np.abs(corr - refc).max() <= 1e-12 * max(np.abs(refc).max(), 1e-300)


def test_gpu_correlate_uses_b(engines):
    """gpu_correlate(a, b) correlates with the b it is given -- any length, not the engine's chirp -- and takes rows."""
    from dsp_amd import dropin as D
    from scipy.signal import correlate
    D.use_engine(engines["default"])
    rng = np.random.default_rng(21)
    for n, m in ((4282, 275), (1000, 33), (50, 200), (9, 9), (7200, 2400)):
        a = rng.standard_normal(n); b = rng.standard_normal(m).astype(np.float32)
        got = D.gpu_correlate(a, b)
        ref = correlate(a, b.astype(np.float64), 'full', method='direct')
        assert got.shape == (n + m - 1,) and np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max(), (n, m)
    a = rng.standard_normal((3, 5, 640)); b = rng.standard_normal(77)
    got = D.gpu_correlate(a, b)
    assert got.shape == (3, 5, 716)
    assert np.abs(got[2, 4] - correlate(a[2, 4], b, 'full', method='direct')).max() <= 1e-12 * 100
    t = torch.from_numpy(a[0]).cuda()
    assert torch.is_tensor(D.gpu_correlate(t, b)) and D.gpu_correlate(t, b).is_cuda
    with pytest.raises(EchoError):
        D.gpu_correlate(a, np.zeros((2, 5)))
    with pytest.raises(EchoError):
        D.gpu_correlate(np.zeros(100), np.zeros(20000))            # b beyond one workgroup's LDS


def test_confidence_helpers_against_goldens(engines, gold_distinct):
    """calculate_band_confidence / normalize_confidences run on the device (echo_band_confidence,
    echo_normalize_confidences): golden (snr, amplitude) -> golden confidence, scalars and arrays; normalisation
    against the oracle, zero-sum rows included."""
    from dsp_amd import dropin as D
    g = gold_distinct
    D.use_engine(engines["distinct"])
    v = g["valid"]
    bands = np.broadcast_to(np.arange(3), v.shape)
    got = D.calculate_band_confidence(g["snr_db"], g["amplitude"], bands)
    assert got.shape == v.shape and np.array_equal(got[v], g["confidence"][v])          # float64, same operations
    i, j = np.argwhere(v)[0]
    c = D.calculate_band_confidence(float(g["snr_db"][i, j]), float(g["amplitude"][i, j]), int(j))
    assert isinstance(c, float) and c == g["confidence"][i, j]
    assert D.calculate_band_confidence(-5.0, 0.01, 2) == O.calculate_band_confidence(-5.0, 0.01, 2)   # no lower clamp
    for confs in ([0.3, 0.9, 0.5], [0.7], (0.2, 0.2), [0.0, 0.0, 0.0]):
        got = D.normalize_confidences(confs)
        assert np.array_equal(got, O.normalize_confidences(confs)), confs
    rows = np.array([[0.3, 0.9, 0.5], [0.0, 0.0, 0.0], [1.0, 0.25, 0.125]])
    got = D.normalize_confidences(rows)
    assert np.array_equal(got, np.stack([O.normalize_confidences(r) for r in rows]))
    with pytest.raises(EchoError):
        D.calculate_band_confidence(1.0, 1.0, 3)


def test_scalar_kalman_with_arbitrary_noise_parameters(engines):
    from dsp_amd import dropin as D
    D.use_engine(engines["default"])
    rng = np.random.default_rng(8)
    for q, r in ((0.005, 0.1), (0.01, 0.2), (0.37, 1.9)):
        kf, ref = D.ScalarKalman(q, r), O.ScalarKalman(q, r)
        assert kf.x is None
        for z in rng.uniform(1, 14, size=12):
            assert kf.update(z) == ref.update(z)                                          # float64, same operations
        assert kf.x == ref.x and kf.p == ref.p
    kf = D.ScalarKalman()
    assert (kf.q, kf.r) == (0.005, 0.1)


def test_bandpass_with_other_filters(engines, gold_default):
    """bandpass(sig, filt) with a (b, a, taps) that is not one of the engine's bands: another design (order 6 / 61 taps,
    order 8 / 101 taps) and the identity filter design_filters falls back to (airSonar_optimized.py:178-182)."""
    from dsp_amd import dropin as D
    from scipy.signal import firwin, iirfilter
    D.use_engine(engines["default"])
    rx = rx_from_i16(gold_default["rx_i16"][:3])
    for order, ntap, lo, hi in ((6, 61, 7000.0, 9000.0), (8, 101, 5000.0, 8000.0)):
        b, a = iirfilter(order, [lo / 24000.0, hi / 24000.0], btype='band', output='ba')
        taps = firwin(ntap, [lo, hi], fs=48000, pass_zero=False, window='hamming')
        for x in rx:
            got = D.bandpass(x, (b, a, taps))
            ref = O.bandpass(x, (b, a, taps))
            assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    ident = ([1.0], [1.0], [1.0])
    got = D.bandpass(rx[0], ident)
    # SciPy's filtfilt raises for a one-coefficient filter (lfilter_zi), so the reference's bandpass lands in its
    # except branch and returns sig unchanged (:195-197): the same values
    assert got.dtype == np.float64 and np.array_equal(got, rx[0].astype(np.float64))
    got = D.bandpass(rx, ident)
    assert got.shape == rx.shape and np.array_equal(got, rx.astype(np.float64))
    with pytest.raises(EchoError):
        D.bandpass(rx[0], ([1.0, 0.5, 0.2], [1.0, -0.3, 0.1], [1.0]))           # an order the kernels do not hold
    f441 = D.design_filters(44100)
    ref441 = [iirfilter(6, [lo / 22050.0, hi / 22050.0], btype='band', output='ba') for lo, hi in engines["default"].cfg.BANDS]
    assert all(np.array_equal(f[0], r[0]) and np.array_equal(f[1], r[1]) for f, r in zip(f441, ref441))


def test_envelope_detection_other_lengths_and_windows(engines):
    """gpu_envelope_detection on lengths and windows beyond the two the chain uses: mixed-radix and Bluestein
    half-length transforms, windows 1..9, rows; envelope, peaks and quality against SciPy."""
    from dsp_amd import dropin as D
    D.use_engine(engines["default"])
    rng = np.random.default_rng(4)
    for L, w in ((4282, 3), (4556, 5), (4000, 1), (3998, 7), (2048, 9), (4698, 5)):
        t = np.arange(L)
        x = rng.standard_normal(L) * 0.05 + np.exp(-((t - L * 0.6) / 40.0) ** 2) * np.sin(0.5 * t) \
            + 0.5 * np.exp(-((t - L * 0.25) / 25.0) ** 2) * np.sin(0.9 * t)
        env, peaks, q = D.gpu_envelope_detection(x, 'hilbert', w)
        e_ref, p_ref, q_ref = O.gpu_envelope_detection(x, w)
        assert np.abs(env - e_ref).max() <= 1e-11 * e_ref.max(), (L, w)
        assert np.array_equal(peaks, p_ref) and abs(q - q_ref) <= 1e-9, (L, w)
    x = rng.standard_normal((2, 3, 4282))
    env, peaks, q = D.gpu_envelope_detection(x, 'hilbert', 3)
    assert env.shape == x.shape and q.shape == (2, 3) and len(peaks) == 6
    e_ref, p_ref, q_ref = O.gpu_envelope_detection(x[1, 2], 3)
    assert np.array_equal(peaks[5], p_ref) and abs(q[1, 2] - q_ref) <= 1e-9
    for bad in (np.zeros(4283), np.zeros(6), np.zeros(20000)):
        with pytest.raises(EchoError):
            D.gpu_envelope_detection(bad, 'hilbert', 5)
    with pytest.raises(EchoError):
        D.gpu_envelope_detection(np.zeros(4282), 'hilbert', 4)
    with pytest.raises(EchoError):
        D.gpu_envelope_detection(np.zeros(4282), 'rms', 5)


def test_mis_shaped_arguments_raise_instead_of_reaching_a_kernel(engines):
    """Every EchoEngine method that indexes by the configured sizes rejects a 1-D input and a wrong trailing dimension
    (the kernels would read past the buffer), and caller-supplied outputs of the wrong shape, dtype or device."""
    from dsp_amd.engine import BatchedKalman
    e = engines["default"]
    cfg = e.cfg
    N, L, nb, bins = cfg.n_rx, cfg.n_corr, cfg.n_bands, cfg.n_bins
    f32, f64 = np.float32, np.float64
    cases = [
        (e.fir_bank, [np.zeros(N, f32), np.zeros((2, N - 1), f32), np.zeros((2, nb, N), f32)]),
        (e.bandpass_bank, [np.zeros(N, f32), np.zeros((2, N + 2), f32)]),
        (e.spectrum, [np.zeros(N, f64), np.zeros((nb, N), f64), np.zeros((2, nb, N - 2), f64), np.zeros((2, nb + 1, N), f64)]),
        (e.spectrum_rows, [np.zeros(N, f64), np.zeros((2, N - 2), f64)]),
        (e.correlate, [np.zeros(N, f64), np.zeros((nb, N), f64), np.zeros((2, nb, N + 1), f64), np.zeros((2, 1, N), f64)]),
        (e.envelope_detection, [np.zeros(L, f64), np.zeros((2, nb, L - 2), f64)]),
        (e.chain, [np.zeros(N, f32), np.zeros((4, N - 1), f32), np.zeros((4, nb, N), f32)]),
    ]
    for fn, bads in cases:
        for bad in bads:
            with pytest.raises(EchoError):
                fn(bad)
    for bad in (torch.zeros(L, dtype=torch.float64, device=e.device),
                torch.zeros((2, nb, L - 1), dtype=torch.float64, device=e.device),
                torch.zeros((2, nb, L), dtype=torch.float32, device=e.device),
                torch.zeros((2, nb, L), dtype=torch.float64),                         # host tensor
                torch.zeros((2, nb, 2 * L), dtype=torch.float64, device=e.device)[:, :, ::2]):   # not contiguous
        with pytest.raises(EchoError):
            e.first_strong_peak(bad)
    with pytest.raises(EchoError):
        e.confidence_fuse(np.zeros((4, nb), np.int32), np.zeros((4, nb + 1)), np.zeros((4, nb)))
    with pytest.raises(EchoError):
        e.confidence_fuse(np.zeros(nb, np.int32), np.zeros(nb), np.zeros(nb))
    rx = torch.zeros((4, N), dtype=torch.float32, device=e.device)
    for kw in (dict(out=torch.zeros((3, 64), dtype=torch.uint8, device=e.device)),
               dict(out=torch.zeros((4, 64), dtype=torch.uint8)),
               dict(spec_out=torch.zeros((4, nb, bins - 1), dtype=torch.float32, device=e.device)),
               dict(spec_out=torch.zeros((4, nb, bins), dtype=torch.float64, device=e.device)),
               dict(mode="full", quality_out=torch.zeros((4, nb), dtype=torch.float64, device=e.device)),
               dict(kalman=BatchedKalman(e, 2), n_streams=4),
               dict(mode="other")):
        with pytest.raises(EchoError):
            e.chain(rx, **kw)
    with pytest.raises(EchoError):
        e.chain_prefetch(torch.zeros((4, N - 1), dtype=torch.float32, device=e.device))
    with pytest.raises(EchoError):
        BatchedKalman(e, 4).update(np.zeros((3, 5)))
    # the dropin spellings: 1-D calls of the wrong length
    from dsp_amd import dropin as D
    D.use_engine(e)
    with pytest.raises(EchoError):
        D.first_strong_peak(np.zeros(L), cfg.FS, -1)
    with pytest.raises(EchoError):
        D.first_strong_peak([0.0] * L, cfg.FS)                     # the blind zone is written into corr: no lists
    with pytest.raises(EchoError):
        D.gpu_spectrum(np.zeros((2, 0)))
    with pytest.raises(EchoError):
        D.bandpass(np.zeros(N + 1, f32), D.design_filters()[0])
    # and the engine still works afterwards
    rec, spec = e.chain(rx)
    assert rec.shape == (4, 64) and spec.shape == (4, nb, bins)


def test_first_strong_peak_with_the_callers_own_arguments(engines, gold_distinct):
    """first_strong_peak(corr, fs, min_delay_samples) (airSonar_optimized.py:200-219) for any correlation length, blind
    zone and sample rate -- the noise window is int(SNR_NOISE_MS * fs) samples, the default blind zone
    int(fs * CHIRP_LEN * 1.2): index, SNR and the in-place zeroing against the oracle's restatement; even and odd
    counts in the noise window, a window cut short by the end of pos, ties, silence, pos no longer than the blind zone."""
    from dsp_amd import dropin as D
    e = engines["distinct"]
    D.use_engine(e)
    cfg = e.cfg
    rng = np.random.default_rng(12)
    g = gold_distinct
    cases = [(4556, 48000, None), (4556, 48000, 331), (4556, 48000, 0), (4556, 48000, 1000), (4556, 44100, 500),
             (4555, 48000, 331), (13229, 44100, 500), (901, 48000, 100), (700, 48000, 300), (700, 9800, 349),
             (640, 48000, 320), (640, 48000, 5000), (2, 48000, 0), (1, 48000, 0), (30001, 96000, 2000)]
    for L, fs, md in cases:
        t = np.arange(L)
        corr = rng.standard_normal(L) * 0.1 + 3.0 * np.exp(-((t - int(L * 0.8)) / 6.0) ** 2) * np.cos(0.7 * t)
        want_c = corr.copy()
        want = O.first_strong_peak(want_c, fs, cfg, md)
        got_c = corr.copy()
        got = D.first_strong_peak(got_c, fs, md)
        assert (got[0] is None) == (want[0] is None), (L, fs, md, got, want)
        if want[0] is not None:
            assert got[0] == int(want[0]) and got[1] == pytest.approx(float(want[1]), rel=1e-12), (L, fs, md)
        else:
            assert got == (None, 0.0)
        assert np.array_equal(got_c, want_c), (L, fs, md)
    # the reference's own correlations, batched [frames, bands, L], another blind zone
    corr = g["stage_corr"][:4].copy()
    peak, snr = D.first_strong_peak(corr, cfg.FS, 700)
    assert peak.shape == (4, 3) and peak.dtype == np.int32
    for k in range(4):
        for j in range(3):
            c = g["stage_corr"][k, j].copy()
            p, sn = O.first_strong_peak(c, cfg.FS, cfg, 700)
            assert peak[k, j] == (-1 if p is None else p) and snr[k, j] == pytest.approx(sn, rel=1e-12)
            assert np.array_equal(corr[k, j], c)
    # ties take the first index; silence and a constant are gated
    tie = np.zeros(1200); tie[600:] = 0.001 * np.sin(np.arange(600)); tie[[900, 1000]] = 5.0
    want = O.first_strong_peak(tie.copy(), 48000, cfg, 100)
    assert D.first_strong_peak(tie.copy(), 48000, 100)[0] == want[0] == 300
    assert D.first_strong_peak(np.zeros(4556), 48000) == (None, 0.0)
    assert D.first_strong_peak(np.full(4556, 0.25), 48000) == O.first_strong_peak(np.full(4556, 0.25), 48000, cfg)
    # a device tensor is mutated where it lies
    c = torch.from_numpy(g["stage_corr"][0, 0].copy()).to(e.device)
    p, _ = D.first_strong_peak(c, cfg.FS)
    assert p == g["peak_idx"][int(g["stage_frames"][0]), 0] and bool((c[cfg.half:cfg.half + cfg.min_delay] == 0).all())


def test_spectrum_of_any_length(engines):
    """gpu_spectrum / mag2db(np.fft.rfft(x)) (airSonar_optimized.py:332-343, :710) for lengths other than the engine's
    frame: lengths the FFT kernels hold (a plan per length), lengths only the direct sums take (a large prime, an odd
    length beyond a workgroup's LDS, tiny ones), rows; |dX| / max|X| <= 1e-4 against NumPy."""
    from dsp_amd import dropin as D
    D.use_engine(engines["default"])
    rng = np.random.default_rng(3)
    for n in (4282, 4800, 8820, 4795, 1000, 64, 4283, 9973, 19997, 8821, 8818, 63, 7, 2, 1, 6000, 12000):
        t = np.arange(n)
        x = rng.standard_normal(n) * 0.1 + np.sin(2 * np.pi * 0.11 * t) + 0.3 * np.cos(2 * np.pi * 0.31 * t)
        ref = np.abs(np.fft.rfft(x))
        mag = D.gpu_spectrum(x)
        assert mag.shape == ref.shape and mag.dtype == np.float32, n
        assert np.abs(mag - ref).max() <= 1e-4 * max(ref.max(), 1e-30), (n, np.abs(mag - ref).max() / ref.max())
        db = D.spectrum_db(x)
        lin = 10 ** (db.astype(np.float64) / 20)
        assert np.abs(lin - np.maximum(ref, 1e-12)).max() <= 1e-4 * max(ref.max(), 1e-30), n
    x = rng.standard_normal((2, 3, 1500))
    mag = D.gpu_spectrum(x)
    assert mag.shape == (2, 3, 751)
    assert np.abs(mag[1, 2] - np.abs(np.fft.rfft(x[1, 2]))).max() <= 1e-4 * np.abs(np.fft.rfft(x[1, 2])).max()
    tt = torch.from_numpy(x[0]).cuda()
    assert torch.is_tensor(D.gpu_spectrum(tt)) and D.gpu_spectrum(tt).shape == (3, 751)
