"""The captured loop of the reference: tools/make_goldens.py ran SonarWorker.run() of airSonar_optimized.py itself on
the golden frames plus adversarial ones (equal echoes, echoes three samples apart, tones, a loud frame next to
silence, echoes at the blind-zone edge ...) with the sound card replaced by the frames, and kept what the loop logged,
emitted and wrote.  CPU part: the oracle reproduces the per-band results bit for bit, and dsp_amd.wire formats oracle
records into the reference's own log lines, signal arguments and CSV file."""
import csv
import io

import numpy as np
import pytest

from conftest import rx_from_i16
from dsp_amd import config as C, wire
from dsp_amd.design import build_tables
from dsp_amd.engine import RESULT_DTYPE
from oracle import echo_oracle as O


def _cfg(which):
    return C.optimized_config(C.BANDS_DEFAULT if which == "default" else C.BANDS_DISTINCT)


def rows_from_chain(res):
    """Oracle chain dict -> EchoResult rows (float32 fields, as the device writes them)."""
    B = len(res["kalman"])
    r = np.zeros(B, RESULT_DTYPE)
    r["dist"] = res["distance"]; r["conf"] = res["confidence"]; r["snr_db"] = res["snr_db"]
    r["peak_idx"] = res["peak_idx"]; r["fused_dist"] = res["fused_dist"]; r["avg_conf"] = res["avg_conf"]
    r["kalman"] = res["kalman"]
    r["valid_mask"] = (res["valid"] << np.arange(res["valid"].shape[1])).sum(axis=1)
    return r


def check_wire_against_captured_loop(rows, g):
    """Log lines, distanceSig arguments and the CSV file made from EchoResult rows against what the reference's loop
    produced.  The records are float32, so numbers are compared at 1e-4 and the formatted text exactly except where a
    value sits within 1e-4 of a rounding boundary of its format."""
    lines = [l for l in wire.log_lines(rows) if l is not None]
    sigs = [s for s in wire.distance_sig_args(rows) if s is not None]
    assert len(lines) == len(g["log_lines"]) == len(sigs) == len(g["dsig_dist"])
    np.testing.assert_allclose([s[0] for s in sigs], g["dsig_dist"], rtol=1e-4)
    np.testing.assert_allclose([s[2] for s in sigs], g["dsig_avg_conf"], rtol=1e-4)
    assert [len(s[1]) for s in sigs] == list(g["dsig_nsnr"])
    np.testing.assert_allclose([x for s in sigs for x in s[1]], g["dsig_snrs"], rtol=1e-4, atol=1e-6)

    def near_boundary(x, digits):
        f = abs(x) * 10 ** digits
        return abs(f - np.floor(f) - 0.5) < 1e-4 * max(1.0, f)
    off = 0
    for k, (mine, ref) in enumerate(zip(lines, g["log_lines"])):
        n = int(g["dsig_nsnr"][k])
        vals = [(g["dsig_dist"][k], 2), (g["dsig_avg_conf"][k], 1)] + [(x, 1) for x in g["dsig_snrs"][off:off + n]]
        off += n
        if not any(near_boundary(x, d) for x, d in vals):
            assert mine == str(ref), k
        assert mine.startswith("distance = ") and mine.endswith("%)")
    text = wire.csv_text(rows, timestamps=[1234.5] * len(rows))
    mine = list(csv.reader(io.StringIO(text))); ref = list(csv.reader(io.StringIO(str(g["csv_text"]))))
    assert mine[0] == ref[0] == ["timestamp", "distance", "confidence", "band_snrs"] and len(mine) == len(ref)
    for a, b in zip(mine[1:], ref[1:]):
        assert a[0] == b[0] and len(a) == len(b) == 4
        np.testing.assert_allclose(float(a[1]), float(b[1]), rtol=1e-4)
        np.testing.assert_allclose(float(a[2]), float(b[2]), rtol=1e-4)
        np.testing.assert_allclose(eval(a[3]), eval(b[3]), rtol=1e-4, atol=1e-6)     # "[12.3, 4.5]": a list literal


@pytest.mark.parametrize("which", ["default", "distinct"])
def test_oracle_reproduces_the_captured_loop(which, gold_loop_default, gold_loop_distinct):
    g = gold_loop_default if which == "default" else gold_loop_distinct
    cfg = _cfg(which)
    t = build_tables(cfg)
    rx = rx_from_i16(g["rx_i16"])
    named = O.chain(rx, t.chirps_f32, t.filters, cfg, n_streams=1, mode="named")
    assert np.array_equal(named["peak_idx"], g["peak_idx"]) and np.array_equal(named["valid"], g["valid"])
    assert np.array_equal(named["snr_db"], g["snr_db"]) and np.array_equal(named["amplitude"], g["amplitude"])
    full = O.chain(rx, t.chirps_f32, t.filters, cfg, n_streams=1, mode="full", keep=("spec_db", "corr_raw"))
    assert np.array_equal(full["valid"], g["full_valid"])
    v = g["full_valid"]
    assert np.array_equal(full["distance"][v], g["full_distance"][v])
    assert np.array_equal(full["confidence"][v], g["full_confidence"][v])
    assert np.array_equal(full["snr_db"][v], g["full_snr"][v])
    # what the loop itself emitted: Kalman-smoothed distance, SNRs of the valid bands, mean normalised confidence
    va = full["valid_any"]
    assert va.sum() == len(g["dsig_dist"])
    assert np.array_equal(full["kalman"][va], g["dsig_dist"]) and np.array_equal(full["avg_conf"][va], g["dsig_avg_conf"])
    assert np.array_equal(full["snr_db"][v], g["dsig_snrs"])
    k = len(g["wave_spectra"])
    assert np.array_equal(full["corr_raw"][:k], g["wave_correlations"])
    assert np.array_equal(full["spec_db"][:k].astype(np.float32), g["wave_spectra"])
    assert list(g["wave_rx_id"]) == list(range(1, len(rx) + 1)) and (g["wave_band_lengths"] == cfg.n_rx).all()
    assert list(g["wave_keys"]) == sorted(wire.wave_sig(rx[0], [], None, 1).keys())
    check_wire_against_captured_loop(rows_from_chain(full), g)
