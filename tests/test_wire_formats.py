"""Result/wire shapes (dsp_amd/wire.py) against the reference's own formatting of the same numbers."""
import numpy as np

from dsp_amd import wire
from dsp_amd.engine import RESULT_DTYPE


def _rows():
    r = np.zeros(3, RESULT_DTYPE)
    r["valid_mask"] = [0b101, 0, 0b111]
    r["snr_db"] = [[12.34, 0, 7.81], [0, 0, 0], [20.0, 21.5, 19.25]]
    r["kalman"] = [3.14159, 0.0, 7.5]
    r["avg_conf"] = [50.0, 0.0, 33.3333]
    return r


def test_csv_rows_and_distance_sig():
    rows = _rows()
    sig = wire.distance_sig_args(rows)
    assert sig[1] is None and len(sig[0][1]) == 2 and len(sig[2][1]) == 3
    csv = wire.csv_rows(rows, timestamps=[10.0, 11.0, 12.0])
    assert [c[0] for c in csv] == [10.0, 12.0]                       # the frame without a valid band writes no row
    assert csv[0][1] == np.float32(3.14159) and csv[0][2] == 50.0
    np.testing.assert_allclose(csv[0][3], [12.34, 7.81], rtol=1e-6)


def test_log_line_matches_reference_format():
    rows = _rows()
    dist_kf, snrs, avg = float(rows["kalman"][2]), [float(s) for s in rows["snr_db"][2]], float(rows["avg_conf"][2])
    want = (f"distance = {dist_kf:.2f} m  "                          # airSonar_optimized.py:741-743, verbatim format
            f"(band SNRs: {['%.1f'%s for s in snrs]}, "
            f"confidence: {avg:.1f}%)")
    lines = wire.log_lines(rows)
    assert lines[2] == want and lines[1] is None


def test_wave_sig_payload():
    rx = np.zeros(4282, np.float32)
    spec = [np.zeros(2142, np.float32)] * 3
    w = wire.wave_sig(rx, spec, None, rx_id=5)
    assert set(w) == {"rx", "band_spectra", "correlations", "band_lengths", "rx_id", "update_spectrum"}
    assert w["band_lengths"] == [4282] * 3 and w["rx_id"] == 5 and w["update_spectrum"] is True
    assert all(c.size == 0 for c in w["correlations"])
