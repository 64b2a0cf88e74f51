#!/usr/bin/env python3
"""Looks into the index mismatches a bulk parity run lists (tools/bulk_parity.py).  CPU only.

For the long-chirp parameter sets scipy.signal.correlate takes its FFT method and transforms the float32 chirp in
single precision, so the oracle's correlation (the reference's own arithmetic) carries ~1e-8 of its maximum in
rounding noise; the kernels sum in float64.  A detector decision that hangs on less than that noise -- a sample within
1e-8 of the MAD threshold, two samples of nearly equal height -- can fall either way.  For every listed mismatch this
tool regenerates the frame, evaluates the same detector on (a) the oracle's correlation and (b) the exact float64
direct sum np.correlate(band_sig, chirp), and reports which index each gives and how close the decision was.
Usage: index_margin_probe.py bulk_parity_<variant>.json [...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    from bulk_parity import START, _config
    from dsp_amd.design import build_tables
    from dsp_amd.synth import synth_frames
    from oracle import echo_oracle as O
    out = []
    for path in sys.argv[1:]:
        summ = json.load(open(path))
        for leg in summ["legs"]:
            cfg = _config(leg["bands"])
            t = build_tables(cfg)
            half, md = cfg.half, cfg.min_delay
            for frame, band, got, want in leg["index_mismatch_list_frame_band_got_want"]:
                rx = synth_frames(1, cfg, t.tx_pcm, start=START + frame)[0]
                band_sig = O.bandpass(rx, t.filters[band])
                c_ref = O.gpu_correlate(band_sig, t.chirps_f32[band])
                c_exact = np.correlate(band_sig.astype(np.float64), np.asarray(t.chirps_f32[band], dtype=np.float64), "full")
                rec = {"file": os.path.basename(path), "variant": leg["bands"], "frame": frame, "band": band,
                       "gpu_index": got, "oracle_index": want,
                       "oracle_corr_vs_exact_max_abs_over_max": float(np.abs(c_ref - c_exact).max() / np.abs(c_exact).max())}
                for name, c in (("oracle_corr", c_ref), ("exact_corr", c_exact)):
                    c = c.copy()
                    if cfg.DETECTOR == O.DET_TWOPEAK:
                        idx, _ = O.first_strong_peak_twopeak(c, cfg.FS)
                        pos = c[half:].copy()
                        ref = int(np.argmax(pos)); pos2 = pos.copy(); pos2[ref] = 0
                        order = np.argsort(pos2)[::-1][:2]
                        rec[name] = {"index": -1 if idx is None else int(idx), "direct_path_index": ref,
                                     "two_highest_echo_candidates": [int(i) for i in order],
                                     "their_relative_difference": float((pos2[order[0]] - pos2[order[1]]) / abs(pos2[order[0]])),
                                     "two_highest_of_pos": [int(i) for i in np.argsort(pos)[::-1][:2]],
                                     "their_relative_difference_pos": float((np.sort(pos)[-1] - np.sort(pos)[-2]) / abs(np.sort(pos)[-1]))}
                    else:
                        idx = O.first_strong_peak_mad(c, cfg.FS, md)
                        pos = c[half:].copy(); pos[:md] = 0
                        med = np.median(pos); mad = np.median(np.abs(pos - med)) + 1e-9
                        thr = med + 6 * mad
                        scale = np.abs(pos).max()
                        rec[name] = {"index": -1 if idx is None else int(idx), "threshold": float(thr),
                                     "margin_at_gpu_index_over_max": float((pos[got] - thr) / scale) if got >= 0 else None,
                                     "margin_at_oracle_index_over_max": float((pos[want] - thr) / scale) if want >= 0 else None}
                rec["gpu_agrees_with_exact_arithmetic"] = rec["exact_corr"]["index"] == got
                print(json.dumps(rec), flush=True)
                out.append(rec)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "index_margin_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
