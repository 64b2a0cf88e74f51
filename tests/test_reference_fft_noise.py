"""The five frame-bands (of 1.2 M) where the kernels and the reference disagree on an index for the long-chirp parameter
sets (tools/bulk_parity.py, 200,000 frames each): scipy.signal.correlate takes its FFT method at these chirp lengths and
transforms the float32 chirp in single precision (airSonar.py:224, 666.py), so the reference's correlation carries
~1e-8 of its maximum in rounding noise, and in each of these frames the detector's decision hangs on less.
tests/golden/disputed_long_chirp.npz holds the frames with the indices the reference itself gives on its own
correlation and on the exact float64 direct sum (tools/make_disputed_fixture.py).  CPU: the oracle reproduces both.
GPU (test_gpu_parity.py::test_disputed_long_chirp_indices): the kernels give the exact-arithmetic one."""
import numpy as np
import pytest

from conftest import load_golden, rx_from_i16
from dsp_amd import config as C
from dsp_amd.design import build_tables
from dsp_amd.synth import synth_frames
from oracle import echo_oracle as O

MARGIN_BOUND = 1e-6      # of max|pos|: every disputed decision of the reference is closer than this


def disputed_cases():
    g = load_golden("disputed_long_chirp")
    for k in range(len(g["frame"])):
        yield (str(g["variant"][k]), int(g["frame"][k]), int(g["band"][k]), g[f"rx_i16_{k}"], int(g["ref_index"][k]),
               int(g["exact_index"][k]), float(g["margin_over_max"][k]), float(g["corr_noise_over_max"][k]))


def variant_config(variant):
    return C.original_config() if variant == "original" else C.twopeak_config()


def detect(cfg, c):
    c = c.copy()
    if cfg.DETECTOR == O.DET_TWOPEAK:
        idx, _ = O.first_strong_peak_twopeak(c, cfg.FS)
    else:
        idx = O.first_strong_peak_mad(c, cfg.FS, cfg.min_delay)
    return -1 if idx is None else int(idx)


@pytest.mark.parametrize("case", list(disputed_cases()), ids=lambda c: f"{c[0]}-{c[1]}-{c[2]}")
def test_disputed_index_is_the_exact_arithmetic_one(case):
    variant, frame, band, rx_i16, ref_index, exact_index, margin, noise = case
    cfg = variant_config(variant)
    t = build_tables(cfg)
    rx = rx_from_i16(rx_i16)
    assert np.array_equal(rx, synth_frames(1, cfg, t.tx_pcm, start=int(load_golden("disputed_long_chirp")["first_frame_seed_index"]) + frame)[0])
    band_sig = O.bandpass(rx, t.filters[band])
    c_ref = O.gpu_correlate(band_sig, t.chirps_f32[band])
    c_exact = np.correlate(band_sig.astype(np.float64), np.asarray(t.chirps_f32[band], dtype=np.float64), "full")
    assert 1e-10 < np.abs(c_ref - c_exact).max() / np.abs(c_exact).max() < 1e-7     # the reference's own rounding noise
    assert detect(cfg, c_ref) == ref_index            # what the reference itself returned when the fixture was made
    assert detect(cfg, c_exact) == exact_index
    assert ref_index != exact_index and margin < MARGIN_BOUND and noise < 1e-7
