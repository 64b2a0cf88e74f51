"""The drop-in boundary under the reference's own caller: ``SonarWorker.run()`` submits ``process_band_with_output``
for the three bands to ``ThreadPoolExecutor(max_workers=3)`` (airSonar_optimized.py:703-730), so ``bandpass``,
``gpu_envelope_detection``, ``gpu_correlate``, ``first_strong_peak`` ... run from three threads at once on one engine --
the first time with a cold handle whose scratch, envelope plans and filter engines do not exist yet.  Expected values:
the goldens produced by importing the reference."""
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import rx_from_i16
from test_gpu_dropin import _Worker

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from dsp_amd import config as C                      # noqa: E402


def _run_frame(D, worker, rx, chirps, filters, start: threading.Barrier):
    """Lines 698-730 of airSonar_optimized.py with the module-level names bound to dsp_amd.dropin; ``start`` lines the
    three pool threads up so that they enter the library together."""
    n = len(filters)
    band_spectra, correlations, band_lengths = [None] * n, [None] * n, [0] * n

    def process_band_with_output(args):
        i, chirp, filt = args
        start.wait(timeout=60)
        band_sig = D.bandpass(rx, filt)
        envelope, peaks, signal_quality = D.gpu_envelope_detection(band_sig, method='hilbert', smooth_window=3)
        band_spec = D.spectrum_db(band_sig)                       # mag2db(np.fft.rfft(band_sig))
        corr = D.gpu_correlate(band_sig, chirp)
        distance, confidence, snr = worker._process_band_gpu(rx, chirp, filt, i)
        return i, band_spec, corr, len(band_sig), (distance, confidence, snr)

    triples = [None] * n
    with ThreadPoolExecutor(max_workers=n) as pool:
        futs = [pool.submit(process_band_with_output, (i, chirp, filt))
                for i, (chirp, filt) in enumerate(zip(chirps, filters))]
        for fut in futs:
            i, band_spec, corr, band_len, result = fut.result(timeout=120)
            band_spectra[i], correlations[i], band_lengths[i], triples[i] = band_spec, corr, band_len, result
    return band_spectra, correlations, band_lengths, triples


@pytest.mark.parametrize("which", ["default", "distinct"])
def test_three_pool_threads_on_a_cold_engine(which, gold_default, gold_distinct):
    """Every frame of the golden set through the thread pool, the very first call of each thread on a handle that has
    allocated nothing yet: triples, spectra and correlations as the reference produced them."""
    from dsp_amd import dropin as D
    from dsp_amd.engine import EchoEngine
    g = gold_default if which == "default" else gold_distinct
    cfg = C.optimized_config(C.BANDS_DEFAULT if which == "default" else C.BANDS_DISTINCT)
    eng = EchoEngine(cfg)                                     # cold: no scratch, no envelope plan
    D.use_engine(eng)
    w = _Worker(D, cfg)
    _, chirps = D.generate_chirps(cfg.FS, cfg.CHIRP_LEN)
    filters = D.design_filters(cfg.FS)
    rx_all = rx_from_i16(g["rx_i16"])
    stage = {int(i): k for k, i in enumerate(g["stage_frames"])}
    start = threading.Barrier(3)
    for i, rx in enumerate(rx_all):
        spectra, corrs, lengths, triples = _run_frame(D, w, rx, chirps, filters, start)
        assert lengths == [cfg.n_rx] * 3
        for j, (distance, confidence, snr) in enumerate(triples):
            assert (distance is not None) == bool(g["full_valid"][i, j]), (i, j)
            if distance is None:
                assert (confidence, snr) == (0.0, 0.0)
                continue
            assert distance == pytest.approx(g["full_distance"][i, j], rel=1e-12, abs=1e-15), (i, j)
            assert snr == pytest.approx(g["full_snr"][i, j], rel=1e-9, abs=1e-9), (i, j)
            assert confidence == pytest.approx(g["full_confidence"][i, j], rel=1e-9), (i, j)
        if i in stage:
            k = stage[i]
            for j in range(3):
                refc = g["stage_corr"][k, j]
                assert np.abs(corrs[j] - refc).max() <= 1e-12 * max(np.abs(refc).max(), 1e-300), (i, j)
                lin = 10 ** (spectra[j].astype(np.float64) / 20)
                rl = 10 ** (g["stage_spec_db"][k, j].astype(np.float64) / 20)
                assert np.abs(lin - rl).max() <= 1e-4 * max(rl.max(), 1e-30), (i, j)
    eng.close()


def test_scratch_growth_under_contention(gold_distinct):
    """Three threads call the batched drop-in functions on one engine with batch sizes that keep forcing the handle's
    scratch to be freed and re-allocated while the other threads are between their calls; a caller-supplied filter
    adds a cold filter engine.  Every result equals the one the same call gives alone."""
    from dsp_amd import dropin as D
    from dsp_amd.engine import EchoEngine
    from scipy.signal import firwin, iirfilter
    g = gold_distinct
    cfg = C.optimized_config(C.BANDS_DISTINCT)
    rx = np.tile(rx_from_i16(g["rx_i16"]), (12, 1))           # 372 frames
    filters = None
    b, a = iirfilter(6, [7000.0 / 24000.0, 9000.0 / 24000.0], btype='band', output='ba')
    other = (b, a, firwin(61, [7000.0, 9000.0], fs=48000, pass_zero=False, window='hamming'))
    sizes = [(1, 200, 17), (64, 3, 372), (300, 31, 128)]       # per thread: rows of its successive calls

    def work(D, filters, tid, start):
        out = []
        if start is not None:
            start.wait(timeout=60)
        for r in sizes[tid]:
            band = D.bandpass(rx[:r], filters[tid])
            corr = D.gpu_correlate(band, D.generate_chirps()[1][tid])
            peak, snr = D.first_strong_peak(np.repeat(corr[:, None, :], 3, axis=1).copy())
            oth = D.bandpass(rx[:min(r, 8)], other)
            out.append((band, corr, peak, snr, oth))
        return out

    ref_eng = EchoEngine(cfg)
    D.use_engine(ref_eng)
    filters = D.design_filters()
    want = [work(D, filters, t, None) for t in range(3)]
    ref_eng.close()

    eng = EchoEngine(cfg)                                      # cold again
    D.use_engine(eng)
    start = threading.Barrier(3)
    with ThreadPoolExecutor(max_workers=3) as pool:
        futs = [pool.submit(work, D, filters, t, start) for t in range(3)]
        got = [f.result(timeout=300) for f in futs]
    for t in range(3):
        for call, (gt, wt) in enumerate(zip(got[t], want[t])):
            for name, x, y in zip(("band", "corr", "peak", "snr", "other"), gt, wt):
                assert np.array_equal(x, y), (t, call, name)
    eng.close()


def test_calls_on_one_handle_from_different_streams_are_ordered(gold_default):
    """Three threads, each on a stream of its own, share one engine: the handle's scratch then serves calls on different
    streams back to back, and the library orders them on the device (the later stream waits for the earlier one).
    The chain's records and spectra equal those of the same batches run one after the other on one stream."""
    from dsp_amd.engine import EchoEngine
    g = gold_default
    cfg = C.optimized_config(C.BANDS_DEFAULT)
    eng = EchoEngine(cfg)
    rx = torch.from_numpy(np.tile(rx_from_i16(g["rx_i16"]), (40, 1))).to(eng.device)   # 1240 frames
    batches = [rx[:1240], rx[100:612], rx[7:1031]]
    want = [eng.chain(b) for b in batches]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=eng.device) for _ in range(3)]
    start = threading.Barrier(3)

    def work(t):
        res = []
        with torch.cuda.stream(streams[t]):
            start.wait(timeout=60)
            for _ in range(6):
                res.append(eng.chain(batches[t]))
        streams[t].synchronize()
        return res

    with ThreadPoolExecutor(max_workers=3) as pool:
        got = [f.result(timeout=300) for f in [pool.submit(work, t) for t in range(3)]]
    for t in range(3):
        for rec, spec in got[t]:
            assert torch.equal(rec, want[t][0]) and torch.equal(spec, want[t][1]), t
    eng.close()
