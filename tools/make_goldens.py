#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own functions (imported from
/root/reference with audio/GUI modules stubbed) on synthetic frames.

Run in the build container only:  python tools/make_goldens.py
The reference never travels; only the resulting data files are committed.
Each .npz records numpy/scipy versions -- the reference pins none, so the oracle is defined as
"the reference's code on the numpy/scipy of this image".
"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import numpy as np
import scipy

import ref_stubs

ref_stubs.install()
_scratch = tempfile.mkdtemp(prefix="refimport_")   # importing opens sonar.log in the CWD
os.chdir(_scratch)
sys.path.insert(0, "/root/reference")

from dsp_amd import config as C          # noqa: E402
from dsp_amd.synth import synth_frame_i16, quantise   # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


class _Self:
    temperature = 20.0


def edge_frames(cfg, tx_mix, rng):
    """int16 frames for the edge cases: silence, noise only, echo in the blind zone, echo in the
    detectable half, two equal echoes, clipped input."""
    n, m = cfg.n_rx, len(tx_mix)
    frames, names = [], []

    def put(name, x):
        frames.append(quantise(x)); names.append(name)

    put("silence", np.zeros(n))
    put("noise_only", rng.normal(0, 0.01, n))
    lag_blind = cfg.half - (m - 1) + cfg.min_delay // 2      # peak lands inside pos[:min_delay]
    x = rng.normal(0, 0.003, n); x[lag_blind:lag_blind + m] += 0.3 * tx_mix[: max(0, min(m, n - lag_blind))]
    put("echo_in_blind_zone", x)
    lag_far = cfg.half - (m - 1) + cfg.min_delay + 700
    x = rng.normal(0, 0.003, n); k = min(m, n - lag_far); x[lag_far:lag_far + k] += 0.3 * tx_mix[:k]
    put("echo_far", x)
    x = rng.normal(0, 0.003, n)
    for lag in (lag_far - 300, lag_far + 500):
        k = min(m, n - lag); x[lag:lag + k] += 0.25 * tx_mix[:k]
    put("two_equal_echoes", x)
    x = rng.normal(0, 0.4, n); k = min(m, n - lag_far); x[lag_far:lag_far + k] += 3.0 * tx_mix[:k]
    put("clipped", x)
    lag_end = n - m // 2                                      # echo truncated by the frame end
    x = rng.normal(0, 0.003, n); k = n - lag_end; x[lag_end:] += 0.3 * tx_mix[:k]
    put("echo_truncated", x)
    return np.stack(frames), names


def run_optimized(tag, bands, n_random=24, n_full=3):
    """airSonar_optimized.py on both band sets: per-stage arrays for a few frames, scalars for all."""
    import importlib
    import airSonar_optimized as ref
    # the module-level cfg is frozen; rebuild it with the wanted bands
    ref.cfg = ref.Config(BANDS=tuple(bands))
    cfg = C.optimized_config(bands)
    assert not ref.GPU_AVAILABLE
    tx_pcm, chirps = ref.generate_chirps(ref.cfg.FS, ref.cfg.CHIRP_LEN)
    filters = ref.design_filters(ref.cfg.FS)
    assert len(tx_pcm) == cfg.n_chirp and int(ref.cfg.FS * ref.cfg.LISTEN_LEN) == cfg.n_rx
    tx_mix = tx_pcm.astype(np.float64) / 32768.0
    rnd = np.stack([synth_frame_i16(i, cfg, tx_mix) for i in range(n_random)])
    edge, edge_names = edge_frames(cfg, tx_mix, np.random.default_rng(99))
    rx_i16 = np.concatenate([rnd, edge])
    B, nb = len(rx_i16), len(bands)
    rx = rx_i16.astype(np.float32) / 2 ** 15
    g = dict(rx_i16=rx_i16, edge_names=np.array(edge_names), n_random=n_random,
             tx_pcm=tx_pcm, chirps=np.stack(chirps),
             filt_b=np.stack([f[0] for f in filters]), filt_a=np.stack([f[1] for f in filters]),
             filt_taps=np.stack([f[2] for f in filters]),
             numpy_version=np.__version__, scipy_version=scipy.__version__, bands=np.array(bands))
    peak = np.full((B, nb), -1, np.int32); snr = np.zeros((B, nb)); amp = np.zeros((B, nb))
    dist = np.zeros((B, nb)); conf = np.zeros((B, nb)); valid = np.zeros((B, nb), bool)
    full_dist = np.zeros((B, nb)); full_conf = np.zeros((B, nb)); full_snr = np.zeros((B, nb))
    full_valid = np.zeros((B, nb), bool)
    fused = np.zeros(B); avgc = np.zeros(B); vany = np.zeros(B, bool); kal = np.zeros(B)
    ffused = np.zeros(B); favgc = np.zeros(B); fvany = np.zeros(B, bool); fkal = np.zeros(B)
    full_idx = list(range(n_full)) + list(range(n_random, B))      # frames with per-stage arrays
    st_band = np.zeros((len(full_idx), nb, cfg.n_rx)); st_spec = np.zeros((len(full_idx), nb, cfg.n_bins))
    st_corr = np.zeros((len(full_idx), nb, cfg.n_corr)); st_corr_after = np.zeros_like(st_corr)
    st_env = np.zeros_like(st_corr)
    sigq = np.zeros((B, nb)); corrq = np.zeros((B, nb))
    kf = ref.ScalarKalman(); kf_full = ref.ScalarKalman()
    me = _Self()
    for i in range(B):
        res_named, res_full = [], []
        for j in range(nb):
            # ---- named chain: the first_strong_peak path of _process_band_gpu (lines 633, 639, 657-667)
            band_sig = ref.bandpass(rx[i], filters[j])
            spec = ref.mag2db(np.fft.rfft(band_sig))
            corr = ref.gpu_correlate(band_sig, chirps[j])
            corr_raw = corr.copy()
            p, s = ref.first_strong_peak(corr, ref.cfg.FS, cfg.min_delay)
            a = float(np.max(np.abs(corr)))
            amp[i, j] = a
            if p is not None:
                peak[i, j] = p; snr[i, j] = s; valid[i, j] = True
                dist[i, j] = (p - (len(chirps[j]) - 1)) / ref.cfg.FS * ref.calculate_sound_speed(20.0) / 2
                conf[i, j] = ref.calculate_band_confidence(s, a, j)
                res_named.append((dist[i, j], conf[i, j], s))
            # ---- full chain: the reference method itself
            d, c, s2 = ref.SonarWorker._process_band_gpu(me, rx[i], chirps[j], filters[j], j)
            if d is not None:
                full_dist[i, j] = d; full_conf[i, j] = c; full_snr[i, j] = s2; full_valid[i, j] = True
                res_full.append((d, c, s2))
            _, _, sigq[i, j] = ref.gpu_envelope_detection(band_sig, method='hilbert', smooth_window=3)
            env, _, corrq[i, j] = ref.gpu_envelope_detection(corr_raw, method='hilbert', smooth_window=5)
            if i in full_idx:
                k = full_idx.index(i)
                st_band[k, j] = band_sig; st_spec[k, j] = spec; st_corr[k, j] = corr_raw
                st_corr_after[k, j] = corr; st_env[k, j] = env
        for results, (fa, aa, va, ka, kfi) in ((res_named, (fused, avgc, vany, kal, kf)),
                                                (res_full, (ffused, favgc, fvany, fkal, kf_full))):
            if results:   # airSonar_optimized.py:733-739
                distances, confidences, snrs = zip(*results)
                cn = ref.normalize_confidences(confidences)
                w = cn / 100.0 + 1e-9
                fa[i] = np.average(distances, weights=w); aa[i] = np.mean(cn); va[i] = True
                ka[i] = kfi.update(fa[i])
            else:
                ka[i] = kfi.x if kfi.x is not None else 0.0
    g.update(peak_idx=peak, snr_db=snr, amplitude=amp, distance=dist, confidence=conf, valid=valid,
             fused_dist=fused, avg_conf=avgc, valid_any=vany, kalman=kal,
             full_distance=full_dist, full_confidence=full_conf, full_snr=full_snr, full_valid=full_valid,
             full_fused_dist=ffused, full_avg_conf=favgc, full_valid_any=fvany, full_kalman=fkal,
             signal_quality=sigq, corr_quality=corrq,
             stage_frames=np.array(full_idx), stage_band_sig=st_band, stage_spec_db=st_spec.astype(np.float32),
             stage_corr=st_corr, stage_corr_after=st_corr_after, stage_corr_envelope=st_env.astype(np.float32))
    path = os.path.join(GOLD, f"optimized_{tag}.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB",
          "valid named", int(valid.sum()), "of", valid.size, "full", int(full_valid.sum()))


def run_original(n_random=6):
    """airSonar.py (config 4): IIR 8 / FIR 101, N = 8820, M = 4410, MAD first-crossing detector."""
    import airSonar as ref
    cfg = C.original_config()
    tx_pcm, chirps = ref.generate_chirps()
    filters = ref.design_filters()
    assert len(tx_pcm) == cfg.n_chirp and int(ref.FS * ref.LISTEN_LEN) == cfg.n_rx
    tx_mix = tx_pcm.astype(np.float64) / 32768.0
    rx_i16 = np.stack([synth_frame_i16(i, cfg, tx_mix) for i in range(n_random)]
                      + [quantise(np.zeros(cfg.n_rx))])
    rx = rx_i16.astype(np.float32) / 2 ** 15
    B, nb = len(rx), len(filters)
    peak = np.full((B, nb), -1, np.int32); dist = np.zeros((B, nb)); valid = np.zeros((B, nb), bool)
    fused = np.zeros(B); vany = np.zeros(B, bool); kal = np.zeros(B)
    band0 = np.zeros((nb, cfg.n_rx)); corr0 = np.zeros((nb, cfg.n_corr))
    kf = ref.ScalarKalman(); me = _Self()
    for i in range(B):
        ds = []
        for j in range(nb):
            y = ref.bandpass(rx[i], filters[j])
            corr = ref.correlate(y, chirps[j], 'full')
            if i == 0:
                band0[j] = y; corr0[j] = corr
            p = ref.first_strong_peak(corr, ref.FS)
            d = ref.SonarWorker._process_band(me, rx[i], chirps[j], filters[j])
            if p is not None:
                peak[i, j] = p; valid[i, j] = True; dist[i, j] = d; ds.append(d)
            else:
                assert d is None
        if ds:
            fused[i] = float(np.mean(ds)); vany[i] = True; kal[i] = kf.update(fused[i])
        else:
            kal[i] = kf.x if kf.x is not None else 0.0
    path = os.path.join(GOLD, "original.npz")
    np.savez_compressed(path, rx_i16=rx_i16, tx_pcm=tx_pcm, chirps=np.stack(chirps),
                        filt_b=np.stack([f[0] for f in filters]), filt_a=np.stack([f[1] for f in filters]),
                        filt_taps=np.stack([f[2] for f in filters]),
                        peak_idx=peak, distance=dist, valid=valid, fused_dist=fused, valid_any=vany, kalman=kal,
                        stage_band_sig0=band0, stage_corr0=corr0,
                        numpy_version=np.__version__, scipy_version=scipy.__version__)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", "valid", int(valid.sum()), "of", valid.size)


def run_realtime(n_streams=4, n_steps=6):
    """realtime_sonar.py (config 5): S streams x T frames, its detector / confidence / fusion / Kalman."""
    import realtime_sonar as ref
    cfg = C.realtime_config()
    tx_pcm, chirps = ref.generate_chirps()
    filters = ref.design_filters()
    assert len(tx_pcm) == cfg.n_chirp and int(ref.cfg.FS * ref.cfg.LISTEN_LEN) == cfg.n_rx
    tx_mix = tx_pcm.astype(np.float64) / 32768.0
    B = n_streams * n_steps
    rx_i16 = np.stack([synth_frame_i16(i, cfg, tx_mix) for i in range(B)])
    rx = rx_i16.astype(np.float32) / 2 ** 15
    nb = len(filters)
    peak = np.full((B, nb), -1, np.int32); snr = np.zeros((B, nb)); dist = np.zeros((B, nb))
    conf = np.zeros((B, nb)); valid = np.zeros((B, nb), bool); amp = np.zeros((B, nb))
    fused = np.zeros(B); avgc = np.zeros(B); vany = np.zeros(B, bool); kal = np.zeros(B)
    kfs = [ref.ScalarKalman() for _ in range(n_streams)]
    me = _Self()
    for i in range(B):            # time-major: stream = i % S
        res = []
        for j in range(nb):
            y = ref.gpu_bandpass(rx[i], filters[j])
            corr = ref.gpu_correlate(y, chirps[j])
            p, s = ref.first_strong_peak(corr, ref.cfg.FS)
            amp[i, j] = np.max(np.abs(corr))
            d, c, s2 = ref.SonarWorker._process_band(me, rx[i], chirps[j], filters[j], j)
            if p is not None:
                peak[i, j] = p; snr[i, j] = s; valid[i, j] = True; dist[i, j] = d; conf[i, j] = c
                res.append((d, c, s2))
            else:
                assert d is None
        kf = kfs[i % n_streams]
        if res:
            distances, confs, snrs = zip(*res)
            nc = ref.normalize_confidences(confs)
            w = nc / 100.0
            fused[i] = np.average(distances, weights=w); avgc[i] = np.mean(nc); vany[i] = True
            kal[i] = kf.update(fused[i])
        else:
            kal[i] = kf.x if kf.x is not None else 0.0
    path = os.path.join(GOLD, "realtime.npz")
    np.savez_compressed(path, rx_i16=rx_i16, n_streams=n_streams, tx_pcm=tx_pcm, chirps=np.stack(chirps),
                        filt_b=np.stack([f[0] for f in filters]), filt_a=np.stack([f[1] for f in filters]),
                        filt_taps=np.stack([f[2] for f in filters]),
                        peak_idx=peak, snr_db=snr, amplitude=amp, distance=dist, confidence=conf, valid=valid,
                        fused_dist=fused, avg_conf=avgc, valid_any=vany, kalman=kal,
                        numpy_version=np.__version__, scipy_version=scipy.__version__)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", "valid", int(valid.sum()), "of", valid.size)


def run_twopeak(n_random=6):
    """666.py: two-peak detector (direct path = time zero), SNR-normalised fusion; N = 7200, M = 2400."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref666", "/root/reference/666.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    cfg = C.twopeak_config()
    tx_pcm, chirps = ref.generate_chirps()
    filters = ref.design_filters()
    assert len(tx_pcm) == cfg.n_chirp and int(ref.FS * ref.LISTEN_LEN) == cfg.n_rx
    tx_mix = tx_pcm.astype(np.float64) / 32768.0
    rng = np.random.default_rng(666)
    frames = []
    for i in range(n_random):
        # direct-path pulse near the start (what the detector calls time zero) plus the synthetic echo frame
        q = synth_frame_i16(i, cfg, tx_mix).astype(np.float64) / 32768.0
        d0 = int(rng.integers(0, 200))
        q[d0:d0 + cfg.n_chirp] += 0.5 * tx_mix[:cfg.n_rx - d0][:cfg.n_chirp]
        frames.append(quantise(q))
    frames.append(quantise(np.zeros(cfg.n_rx)))
    frames.append(synth_frame_i16(99, cfg, tx_mix))            # echo only, no direct path
    rx_i16 = np.stack(frames)
    rx = rx_i16.astype(np.float32) / 2 ** 15
    B, nb = len(rx), len(filters)
    delay = np.full((B, nb), -1, np.int32); snr = np.zeros((B, nb)); dist = np.zeros((B, nb))
    conf = np.zeros((B, nb)); valid = np.zeros((B, nb), bool); amp = np.zeros((B, nb))
    fused = np.zeros(B); avgc = np.zeros(B); vany = np.zeros(B, bool); kal = np.zeros(B)
    band0 = np.zeros((nb, cfg.n_rx)); corr0 = np.zeros((nb, cfg.n_corr))
    kf = ref.ScalarKalman(); me = _Self()
    for i in range(B):
        res = []
        for j in range(nb):
            y = ref.bandpass(rx[i], filters[j])
            corr = ref.correlate(y, chirps[j], 'full')
            if i == 0:
                band0[j] = y; corr0[j] = corr
            amp[i, j] = np.max(np.abs(corr))
            dl, s = ref.first_strong_peak(corr, ref.FS)
            d, c, s2 = ref.SonarWorker._process_band(me, rx[i], chirps[j], filters[j], j)
            if dl is not None:
                delay[i, j] = dl; snr[i, j] = s; valid[i, j] = True; dist[i, j] = d; conf[i, j] = c
                res.append((d, c, s2))
            else:
                assert d is None
        if res:
            distances, confs, snrs = zip(*res)
            snr_array = np.array(snrs)                                     # 666.py:372-385
            normalized = snr_array / np.max(snr_array) * 100 if np.max(snr_array) > 0 else np.zeros_like(snr_array)
            weights = normalized + 1e-9
            fused[i] = np.average(distances, weights=weights); avgc[i] = np.mean(normalized); vany[i] = True
            kal[i] = kf.update(fused[i])
        else:
            kal[i] = kf.x if kf.x is not None else 0.0
    path = os.path.join(GOLD, "twopeak.npz")
    np.savez_compressed(path, rx_i16=rx_i16, tx_pcm=tx_pcm, chirps=np.stack(chirps),
                        filt_b=np.stack([f[0] for f in filters]), filt_a=np.stack([f[1] for f in filters]),
                        filt_taps=np.stack([f[2] for f in filters]),
                        peak_idx=delay, snr_db=snr, amplitude=amp, distance=dist, confidence=conf, valid=valid,
                        fused_dist=fused, avg_conf=avgc, valid_any=vany, kalman=kal,
                        stage_band_sig0=band0, stage_corr0=corr0,
                        numpy_version=np.__version__, scipy_version=scipy.__version__)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", "valid", int(valid.sum()), "of", valid.size)


def adversarial_frames(cfg, tx_mix, rng):
    """Frames built to stress the argmax: echoes of (nearly) equal strength, echoes a few samples apart, a noise-free
    periodic input whose correlation repeats, a tone at the band centre (flat correlation envelope), a loud frame
    next to silence, an echo straddling the blind-zone edge, impulses."""
    n, m = cfg.n_rx, len(tx_mix)
    base = cfg.half - (m - 1) + cfg.min_delay
    frames, names = [], []

    def put(name, x):
        frames.append(quantise(x)); names.append(name)

    def echo(x, lag, a):
        k = max(0, min(m, n - lag))
        x[lag:lag + k] += a * tx_mix[:k]

    x = np.zeros(n); echo(x, base + 300, 0.25); echo(x, base + 1100, 0.25)
    put("two_identical_echoes_no_noise", x)
    x = np.zeros(n)
    for q in range(5):
        echo(x, base + 200 + 320 * q, 0.2)
    put("five_identical_echoes_no_noise", x)
    x = rng.normal(0, 0.0005, n); echo(x, base + 300, 0.25); echo(x, base + 1100, 0.2501)
    put("two_echoes_1e-4_apart", x)
    x = rng.normal(0, 0.002, n); echo(x, base + 500, 0.2); echo(x, base + 503, 0.2)
    put("two_echoes_three_samples_apart", x)
    t = np.arange(n)
    put("tone_band0_centre", 0.3 * np.sin(2 * np.pi * (cfg.BANDS[0][0] + cfg.BANDS[0][1]) / 2 / cfg.FS * t))
    put("tone_band2_centre", 0.3 * np.sin(2 * np.pi * (cfg.BANDS[2][0] + cfg.BANDS[2][1]) / 2 / cfg.FS * t))
    x = rng.normal(0, 0.3, n); echo(x, base + 900, 0.9)
    put("loud", x)
    put("silence_after_loud", np.zeros(n))
    put("faint_noise", rng.normal(0, 0.0002, n))
    x = rng.normal(0, 0.002, n); echo(x, cfg.half - (m - 1) + cfg.min_delay - 1, 0.3)
    put("echo_peak_last_blind_sample", x)
    x = rng.normal(0, 0.002, n); echo(x, cfg.half - (m - 1) + cfg.min_delay, 0.3)
    put("echo_peak_first_open_sample", x)
    x = np.zeros(n); x[base + 400] = 0.9; x[base + 1500] = -0.9
    put("impulses", x)
    x = rng.normal(0, 0.004, n); echo(x, n - m - 2, 0.3)
    put("echo_at_frame_end", x)
    x = np.zeros(n); echo(x, 900, 0.3)
    put("echo_only_in_first_half", x)
    x = np.full(n, 0.5)
    put("dc", x)
    return np.stack(frames), names


def run_loop_capture(tag="default", bands=C.BANDS_DEFAULT):
    """SonarWorker.run() of airSonar_optimized.py itself (:678-768) on the golden frames followed by adversarial ones,
    with the sound card replaced by the frames and the sinks captured: the INFO lines (:741-743), the distanceSig
    arguments (:745), the CSV file (:746-747) and the waveSig payload of the first frames (:749-757).  This pins the
    result / wire formats and the whole loop body (envelope branch, fusion, Kalman) on the reference's own code."""
    import csv
    import logging
    import types
    import airSonar_optimized as ref
    ref.cfg = ref.Config(BANDS=tuple(bands))
    cfg = C.optimized_config(bands)
    gold = np.load(os.path.join(GOLD, f"optimized_{tag}.npz"))
    tx_mix = gold["tx_pcm"].astype(np.float64) / 32768.0
    adv, adv_names = adversarial_frames(cfg, tx_mix, np.random.default_rng(4242))
    rx_i16 = np.concatenate([gold["rx_i16"], adv])
    frames = [q.astype(np.float32) / 2 ** 15 for q in rx_i16]

    class FakeAudio:
        def __init__(self):
            self.i = 0
        def play(self, pcm):
            pass
        def record(self):
            x = frames[self.i]; self.i += 1
            if self.i >= len(frames):
                worker.stop_event.set()
            return x
        def close(self):
            pass

    class Capture:
        def __init__(self):
            self.calls = []
        def emit(self, *a):
            self.calls.append(a)

    lines = []
    class Handler(logging.Handler):
        def emit(self, record):
            lines.append(record.getMessage())
    h = Handler(); ref.logger.addHandler(h)
    if os.path.exists("distances.csv"):
        os.remove("distances.csv")
    real_time = ref.time
    ref.time = types.SimpleNamespace(perf_counter=real_time.perf_counter, sleep=lambda s: None, time=lambda: 1234.5)
    ref.AudioIO = FakeAudio
    worker = ref.SonarWorker()
    worker.distanceSig = Capture(); worker.waveSig = Capture(); worker.errorSig = Capture()
    try:
        worker.run()
    finally:
        ref.time = real_time
        ref.logger.removeHandler(h)
    assert not worker.errorSig.calls, worker.errorSig.calls
    waves = worker.waveSig.calls
    assert len(waves) == len(frames)
    with open("distances.csv", newline="") as f:
        csv_text = f.read()
    log = [l for l in lines if l.startswith("distance = ")]
    dsig = worker.distanceSig.calls
    assert len(log) == len(dsig)
    # which frames produced a result: recovered from the waveSig ids and the order of the emissions
    nb = len(bands)
    B = len(frames)
    keep = 3                                                   # waveSig arrays of the first frames only (size)
    out = dict(
        rx_i16=rx_i16, n_golden=len(gold["rx_i16"]), adversarial_names=np.array(adv_names), bands=np.array(bands),
        log_lines=np.array(log), csv_text=np.array(csv_text),
        dsig_dist=np.array([d[0] for d in dsig]), dsig_avg_conf=np.array([d[2] for d in dsig]),
        dsig_nsnr=np.array([len(d[1]) for d in dsig]), dsig_snrs=np.array([s for d in dsig for s in d[1]]),
        wave_keys=np.array(sorted(waves[0][0].keys())),
        wave_rx_id=np.array([w[0]["rx_id"] for w in waves]),
        wave_band_lengths=np.array([w[0]["band_lengths"] for w in waves]),
        wave_spectra=np.stack([np.stack(w[0]["band_spectra"]) for w in waves[:keep]]).astype(np.float32),
        wave_correlations=np.stack([np.stack(w[0]["correlations"]) for w in waves[:keep]]),
        numpy_version=np.__version__, scipy_version=scipy.__version__)
    # per frame and band, the reference's own functions once more for the scalars the parity tests compare
    filters = worker.filters; chirps = worker.chirps
    me = _Self()
    peak = np.full((B, nb), -1, np.int32); snr = np.zeros((B, nb)); amp = np.zeros((B, nb)); valid = np.zeros((B, nb), bool)
    fd = np.zeros((B, nb)); fc = np.zeros((B, nb)); fs = np.zeros((B, nb)); fv = np.zeros((B, nb), bool)
    for i in range(B):
        for j in range(nb):
            band_sig = ref.bandpass(frames[i], filters[j])
            corr = ref.gpu_correlate(band_sig, chirps[j])
            p, s = ref.first_strong_peak(corr, ref.cfg.FS, cfg.min_delay)
            amp[i, j] = float(np.max(np.abs(corr)))
            if p is not None:
                peak[i, j] = p; snr[i, j] = s; valid[i, j] = True
            d, c, s2 = ref.SonarWorker._process_band_gpu(me, frames[i], chirps[j], filters[j], j)
            if d is not None:
                fd[i, j] = d; fc[i, j] = c; fs[i, j] = s2; fv[i, j] = True
    out.update(peak_idx=peak, snr_db=snr, amplitude=amp, valid=valid,
               full_distance=fd, full_confidence=fc, full_snr=fs, full_valid=fv)
    path = os.path.join(GOLD, f"run_loop_{tag}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", "frames", B, "results", len(dsig),
          "named valid", int(valid.sum()), "full valid", int(fv.sum()))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "run_loop":     # only the captured run() loops (needs the optimized goldens)
        run_loop_capture("default", C.BANDS_DEFAULT)
        run_loop_capture("distinct", C.BANDS_DISTINCT)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "twopeak":      # only the 666.py variant
        run_twopeak()
        sys.exit(0)
    run_optimized("default", C.BANDS_DEFAULT)
    run_optimized("distinct", C.BANDS_DISTINCT)
    run_original()
    run_realtime()
    run_twopeak()
    run_loop_capture("default", C.BANDS_DEFAULT)
    run_loop_capture("distinct", C.BANDS_DISTINCT)
