"""Time of the spectrum entry point for long frames of even and odd length (4096 frames x 3 bands)."""
import sys, time
from dataclasses import replace
import torch
sys.path.insert(0, ".")
from dsp_amd import config as C
from dsp_amd.engine import EchoEngine

for n in (8820, 8821, 9999, 12001):
    cfg = replace(C.original_config(), LISTEN_LEN_FIXED=n / 44100 + 1e-7)
    assert cfg.n_rx == n
    e = EchoEngine(cfg)
    B = 4096
    band = torch.randn((B, cfg.n_bands, n), dtype=torch.float64, device="cuda")
    e.spectrum(band[:64]); torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(3):
        e.spectrum(band)
    t1.record(); torch.cuda.synchronize()
    print(f"N={n}: {t0.elapsed_time(t1) / 3:.3f} ms per {B} frames x {cfg.n_bands} bands")
    e.close()
