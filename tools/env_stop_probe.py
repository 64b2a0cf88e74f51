#!/usr/bin/env python3
"""Development aid: cumulative time of the envelope-branch kernel's stages (ECHO_ENV_STOP leaves the sequence early;
results are then meaningless, only the time counts).  Prints ms per 4096-frame full chain for each stop point."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd import config as C
from dsp_amd.engine import EchoEngine
from dsp_amd.synth import synth_frames

cfg = C.optimized_config(C.BANDS_DEFAULT)
names = {6: "load", 1: "+forward transform", 2: "+mask, inverse transform", 10: "+abs, smooth", 11: "+mean/std",
         3: "+second moments", 12: "+candidates", 13: "+maxima", 4: "+distance rounds", 5: "+prominence", 0: "everything"}
rx = None
for stop in (6, 1, 2, 10, 11, 3, 12, 13, 4, 5, 0):
    os.environ["ECHO_ENV_STOP"] = str(stop)
    e = EchoEngine(cfg)
    if rx is None:
        rx = torch.from_numpy(synth_frames(4096, cfg, e.tables.tx_pcm)).to(e.device)
    for _ in range(3):
        e.chain(rx, want_spectra=True, mode="full")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        e.chain(rx, want_spectra=True, mode="full")
    torch.cuda.synchronize()
    print(f"stop {stop:2d} {names[stop]:28s} {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms", flush=True)
    e.close()
