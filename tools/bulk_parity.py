#!/usr/bin/env python3
"""One-off larger parity run: the GPU chain against the oracle on many synthetic frames (named and full mode, both
band sets).  Prints mismatch counts; exits non-zero on any index mismatch.  Usage: bulk_parity.py [frames]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd import config as C
from dsp_amd.engine import EchoEngine, BatchedKalman, records_to_numpy
from dsp_amd.synth import synth_frames
from oracle import echo_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
bad = 0
for bands in (C.BANDS_DEFAULT, C.BANDS_DISTINCT):
    cfg = C.optimized_config(bands)
    eng = EchoEngine(cfg)
    rx = synth_frames(n, cfg, eng.tables.tx_pcm, start=100000)
    for mode in ("named", "full"):
        t0 = time.time()
        o = O.chain(rx, eng.tables.chirps_f32, eng.tables.filters, cfg, n_streams=64, mode=mode)
        rec, _ = eng.chain(rx, n_streams=64, kalman=BatchedKalman(eng, 64), mode=mode, want_spectra=False)
        r = records_to_numpy(rec)
        mism = int((r["peak_idx"][:, :3] != o["peak_idx"]).sum())
        v = o["valid"]
        dconf = np.abs(r["conf"][:, :3][v] - o["confidence"][v]) / np.maximum(np.abs(o["confidence"][v]), 1e-12)
        dsnr = np.abs(r["snr_db"][:, :3][v] - o["snr_db"][v]) / np.maximum(np.abs(o["snr_db"][v]), 1e-12)
        dk = np.abs(r["kalman"] - o["kalman"]).max()
        print(f"bands={bands[0]}.. mode={mode}: {n} frames, index mismatches {mism}, max rel conf err {dconf.max():.2e}, "
              f"max rel snr err {dsnr.max():.2e}, max kalman abs err {dk:.2e}  ({time.time() - t0:.0f} s)", flush=True)
        bad += mism
    eng.close()
sys.exit(1 if bad else 0)
