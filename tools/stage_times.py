#!/usr/bin/env python3
"""Times each C-ABI stage on one GPU (torch CUDA events on the current stream). Development aid."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dsp_amd.config import optimized_config
from dsp_amd.engine import EchoEngine, BatchedKalman
from dsp_amd.synth import synth_frames

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = optimized_config(); e = EchoEngine(cfg)
rx = torch.from_numpy(synth_frames(256, cfg, e.tables.tx_pcm)).cuda().repeat(B // 256, 1).contiguous()
e.reserve(B)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
band = e.bandpass_bank(rx); corr = e.correlate(band)
rec = torch.empty((B, 64), dtype=torch.uint8, device='cuda'); spec = torch.empty((B, 3, cfg.n_bins), dtype=torch.float32, device='cuda')
res = {
 "fir_bank": t(lambda: e.fir_bank(rx)),
 "bandpass_bank(iir+fir+store)": t(lambda: e.bandpass_bank(rx)),
 "correlate(load+corr+store)": t(lambda: e.correlate(band)),
 "spectrum": t(lambda: e.spectrum(band)),
 "first_strong_peak": t(lambda: e.first_strong_peak(corr)),
 "chain_no_spectra": t(lambda: e.chain(rx, want_spectra=False, out=rec)),
 "chain": t(lambda: e.chain(rx, out=rec, spec_out=spec)),
}
print(json.dumps({"B": B, "ms": res}, indent=1))
