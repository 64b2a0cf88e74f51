"""The CPU oracle against golden vectors produced by the reference itself (tools/make_goldens.py).
Also pins the host-side set-up (chirps, filter design) the product uses."""
import numpy as np
import pytest

from conftest import rx_from_i16
from dsp_amd import config as C
from dsp_amd.design import build_tables, design_filters, generate_chirps
from dsp_amd.synth import synth_frame_i16
from oracle import echo_oracle as O


def _cfg_for(gold):
    return C.optimized_config(tuple(map(tuple, gold["bands"].tolist())))


@pytest.mark.parametrize("which", ["default", "distinct"])
def test_setup_matches_reference(which, gold_default, gold_distinct):
    g = gold_default if which == "default" else gold_distinct
    cfg = _cfg_for(g)
    tx_pcm, chirps = generate_chirps(cfg)
    assert np.array_equal(tx_pcm, g["tx_pcm"])
    assert np.array_equal(np.stack(chirps), g["chirps"])           # bit-exact float32
    filters = design_filters(cfg)
    for j, (b, a, taps) in enumerate(filters):
        assert np.array_equal(b, g["filt_b"][j]) and np.array_equal(a, g["filt_a"][j])
        assert np.array_equal(taps, g["filt_taps"][j])
    assert (cfg.n_rx, cfg.n_chirp, cfg.n_corr, cfg.half, cfg.min_delay, cfg.noise_n, cfg.n_bins) == \
        (4282, 275, 4556, 2278, 331, 240, 2142)


def test_synthetic_frames_reproducible(gold_default):
    cfg = _cfg_for(gold_default)
    tx_mix = gold_default["tx_pcm"].astype(np.float64) / 32768.0
    for i in (0, 5, 17):
        assert np.array_equal(synth_frame_i16(i, cfg, tx_mix), gold_default["rx_i16"][i])


@pytest.mark.parametrize("which", ["default", "distinct"])
def test_oracle_stages(which, gold_default, gold_distinct):
    g = gold_default if which == "default" else gold_distinct
    cfg = _cfg_for(g)
    t = build_tables(cfg)
    rx = rx_from_i16(g["rx_i16"])
    for k, i in enumerate(g["stage_frames"]):
        for j in range(cfg.n_bands):
            out = O.process_band(rx[i], t.chirps_f32[j], t.filters[j], j, cfg, mode="named")
            assert np.array_equal(out["band_sig"], g["stage_band_sig"][k, j])
            assert np.array_equal(out["corr_raw"], g["stage_corr"][k, j])
            assert np.array_equal(out["corr_after"], g["stage_corr_after"][k, j])
            np.testing.assert_allclose(out["spec_db"], g["stage_spec_db"][k, j], rtol=1e-6, atol=1e-4)


@pytest.mark.parametrize("which", ["default", "distinct"])
@pytest.mark.parametrize("mode", ["named", "full"])
def test_oracle_chain(which, mode, gold_default, gold_distinct):
    g = gold_default if which == "default" else gold_distinct
    cfg = _cfg_for(g)
    t = build_tables(cfg)
    rx = rx_from_i16(g["rx_i16"])
    r = O.chain(rx, t.chirps_f32, t.filters, cfg, n_streams=1, mode=mode)
    p = "" if mode == "named" else "full_"
    if mode == "named":
        assert np.array_equal(r["peak_idx"], g["peak_idx"])
        assert np.array_equal(r["snr_db"], g["snr_db"])
        assert np.array_equal(r["amplitude"], g["amplitude"])
    else:
        assert np.array_equal(r["snr_db"][g["full_valid"]], g["full_snr"][g["full_valid"]])
    assert np.array_equal(r["valid"], g[p + "valid"])
    assert np.array_equal(r["distance"], g[p + "distance"])
    assert np.array_equal(r["confidence"], g[p + "confidence"])
    assert np.array_equal(r["valid_any"], g[p + "valid_any"])
    assert np.array_equal(r["fused_dist"], g[p + "fused_dist"])
    assert np.array_equal(r["avg_conf"], g[p + "avg_conf"])
    assert np.array_equal(r["kalman"], g[p + "kalman"])


def test_filtfilt_restatement_is_bit_exact(gold_default):
    """The written-out DF2T filtfilt (the algorithm of the IIR kernel) equals scipy's bit for bit,
    and direct FIR 'same' equals fftconvolve to FFT rounding."""
    g = gold_default
    cfg = _cfg_for(g)
    t = build_tables(cfg)
    from scipy.signal import filtfilt
    rx = rx_from_i16(g["rx_i16"])
    for i in (0, int(g["n_random"]) + 5):      # a random frame and the clipped edge frame
        b, a, taps = t.filters[0]
        y_ref = filtfilt(b, a, rx[i])
        y = O.filtfilt_df2t(b, a, rx[i])
        assert np.array_equal(y, y_ref)
        bs = O.fir_same_direct(y, taps)
        k = list(g["stage_frames"]).index(i)
        ref = g["stage_band_sig"][k, 0]
        assert np.max(np.abs(bs - ref)) <= 1e-14 * max(1e-300, np.max(np.abs(ref)))


def test_oracle_original(gold_original):
    g = gold_original
    cfg = C.original_config()
    t = build_tables(cfg)
    assert np.array_equal(t.tx_pcm, g["tx_pcm"]) and np.array_equal(np.stack(t.chirps_f32), g["chirps"])
    assert all(np.array_equal(t.filters[j][0], g["filt_b"][j]) for j in range(3))
    assert (cfg.n_rx, cfg.n_chirp, cfg.n_corr, cfg.half) == (8820, 4410, 13229, 6614)
    rx = rx_from_i16(g["rx_i16"])
    r = O.chain(rx, t.chirps_f32, t.filters, cfg, n_streams=1, mode="named")
    assert np.array_equal(r["peak_idx"], g["peak_idx"])
    assert np.array_equal(r["distance"], g["distance"])
    assert np.array_equal(r["fused_dist"], g["fused_dist"])
    assert np.array_equal(r["kalman"], g["kalman"])


def test_oracle_realtime(gold_realtime):
    g = gold_realtime
    cfg = C.realtime_config()
    t = build_tables(cfg)
    assert np.array_equal(t.tx_pcm, g["tx_pcm"]) and cfg.n_chirp == 137 and cfg.n_corr == 4418
    rx = rx_from_i16(g["rx_i16"])
    r = O.chain(rx, t.chirps_f32, t.filters, cfg, n_streams=int(g["n_streams"]), mode="named")
    for k in ("peak_idx", "snr_db", "amplitude", "distance", "confidence", "valid", "fused_dist",
              "avg_conf", "valid_any", "kalman"):
        assert np.array_equal(r[k], g[k]), k


def test_oracle_twopeak(gold_twopeak):
    """666.py: two-peak detector (delay = echo peak - direct-path peak, may be negative), SNR-normalised fusion."""
    g = gold_twopeak
    cfg = C.twopeak_config()
    t = build_tables(cfg)
    assert np.array_equal(t.tx_pcm, g["tx_pcm"]) and np.array_equal(np.stack(t.chirps_f32), g["chirps"])
    assert (cfg.n_rx, cfg.n_chirp, cfg.n_corr, cfg.half, cfg.noise_n) == (7200, 2400, 9599, 4799, 960)
    rx = rx_from_i16(g["rx_i16"])
    r = O.chain(rx, t.chirps_f32, t.filters, cfg, n_streams=1, mode="named")
    for k in ("peak_idx", "snr_db", "amplitude", "distance", "confidence", "valid", "fused_dist",
              "avg_conf", "valid_any", "kalman"):
        assert np.array_equal(r[k], g[k]), k
    assert (g["peak_idx"][g["valid"]] < 0).any()          # the fixture does contain valid negative delays
