#!/usr/bin/env python3
"""Development aid: time of the correlation-envelope kernel when it leaves after stage ECHO_ENV_STOP (1 forward FFT,
2 inverse FFT, 3 smoothing + statistics, 4 maxima + distance rule, 5 prominence, 0 everything)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd.config import optimized_config
from dsp_amd.engine import EchoEngine
from dsp_amd.synth import synth_frames

cfg = optimized_config()
eng = EchoEngine(cfg)
rx = synth_frames(4096, cfg, eng.tables.tx_pcm)
corr = eng.correlate(eng.bandpass_bank(rx))
for _ in range(2):
    eng.envelope_detection(corr, want_envelope=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    eng.envelope_detection(corr, want_envelope=False)
torch.cuda.synchronize()
print("stop", os.environ.get("ECHO_ENV_STOP", "0"), "ms per 4096 frames: %.3f" % ((time.perf_counter() - t0) / 5 * 1e3))
