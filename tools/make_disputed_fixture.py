#!/usr/bin/env python3
"""Writes tests/golden/disputed_long_chirp.npz: the five frame-bands (of 1.2 M checked) where the kernels and the
reference disagree on an index for the long-chirp parameter sets (airSonar.py, 666.py).

At these chirp lengths scipy.signal.correlate takes its FFT method and transforms the float32 chirp in single
precision, so the reference's correlation carries ~1e-8 of its maximum in rounding noise; in each of these frames the
detector's decision hangs on less.  For every case this script regenerates the frame, runs the REFERENCE's own
functions on it (imported from /root/reference with audio/GUI modules stubbed, as tools/make_goldens.py does) and the
reference's own detector on the exact float64 direct-sum correlation, and records both indices and how close the
reference's decision was.  Run in the build container only; the reference never travels, the .npz does.
"""
import importlib.util
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
os.chdir(tempfile.mkdtemp(prefix="refimport_"))           # importing the reference opens sonar.log in the CWD
sys.path.insert(0, "/root/reference")

from dsp_amd import config as C                            # noqa: E402
from dsp_amd.design import build_tables                    # noqa: E402
from dsp_amd.synth import synth_frames                     # noqa: E402

START = 100000          # first frame of the bulk parity runs (tools/bulk_parity.py)
CASES = [               # variant, frame of the bulk run, band
    ("original", 62133, 1), ("original", 69412, 2), ("original", 99730, 1),
    ("twopeak", 83562, 0), ("twopeak", 126947, 0),
]


def _load(variant):
    if variant == "original":
        import airSonar as ref
        return ref
    spec = importlib.util.spec_from_file_location("ref666", "/root/reference/666.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref


def _detect(ref, variant, corr):
    """The reference's first_strong_peak on a copy -> (index or -1; for 666.py the delay echo - direct path)."""
    r = ref.first_strong_peak(corr.copy(), ref.FS)
    if variant == "twopeak":
        r = r[0]
    return -1 if r is None else int(r)


def _margin(variant, cfg, c, i_ref, i_exact):
    """How close the decision between the two indices is in correlation ``c``, relative to max|pos|."""
    pos = c[cfg.half:].copy()
    scale = np.abs(pos).max()
    if variant == "original":                              # airSonar.py:104-117: first crossing of med + 6 mad
        pos[:cfg.min_delay] = 0
        med = np.median(pos); mad = np.median(np.abs(pos - med)) + 1e-9
        thr = med + 6 * mad
        first = min(i for i in (i_ref, i_exact) if i >= 0)   # the earlier candidate: is it above the threshold?
        return float(abs(pos[first] - thr) / scale)
    ref0 = int(np.argmax(pos))                             # 666.py:153-200: the two highest echo candidates
    p2 = pos.copy(); p2[ref0] = 0
    a, b = np.sort(p2)[-1], np.sort(p2)[-2]
    return float((a - b) / scale)


def main():
    recs = []
    for variant, frame, band in CASES:
        ref = _load(variant)
        cfg = C.original_config() if variant == "original" else C.twopeak_config()
        t = build_tables(cfg)
        tx_pcm, chirps = ref.generate_chirps()
        filters = ref.design_filters()
        assert np.array_equal(tx_pcm, t.tx_pcm)
        rx = synth_frames(1, cfg, t.tx_pcm, start=START + frame)[0]
        q = np.rint(rx.astype(np.float64) * 32768).astype(np.int16)
        assert np.array_equal(q.astype(np.float32) / 2 ** 15, rx)
        band_sig = ref.bandpass(rx, filters[band])
        c_ref = ref.correlate(band_sig, chirps[band], 'full')
        c_exact = np.correlate(np.asarray(band_sig, dtype=np.float64), np.asarray(chirps[band], dtype=np.float64), "full")
        i_ref, i_exact = _detect(ref, variant, c_ref), _detect(ref, variant, c_exact)
        rec = dict(variant=variant, frame=frame, band=band, rx_i16=q, ref_index=i_ref, exact_index=i_exact,
                   corr_noise=float(np.abs(c_ref - c_exact).max() / np.abs(c_exact).max()),
                   margin=_margin(variant, cfg, c_ref, i_ref, i_exact))
        print({k: v for k, v in rec.items() if k != "rx_i16"})
        assert i_ref != i_exact
        recs.append(rec)
    out = {"variant": np.array([r["variant"] for r in recs]), "frame": np.array([r["frame"] for r in recs]),
           "band": np.array([r["band"] for r in recs]), "ref_index": np.array([r["ref_index"] for r in recs]),
           "exact_index": np.array([r["exact_index"] for r in recs]),
           "corr_noise_over_max": np.array([r["corr_noise"] for r in recs]),
           "margin_over_max": np.array([r["margin"] for r in recs]),
           "first_frame_seed_index": START, "numpy_version": np.__version__, "scipy_version": scipy.__version__}
    for k, r in enumerate(recs):
        out[f"rx_i16_{k}"] = r["rx_i16"]
    path = os.path.join(ROOT, "tests", "golden", "disputed_long_chirp.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
