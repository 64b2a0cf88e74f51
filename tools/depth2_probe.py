#!/usr/bin/env python3
"""Development probe: two handles taking alternate batches, each prefetching its own next batch, i.e. every IIR is
issued two steps ahead and two IIRs are in flight.  Prints frames/s for the IIR mapping selected by ECHO_IIR_MODE."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd.config import optimized_config            # noqa: E402
from dsp_amd.engine import BatchedKalman, EchoEngine   # noqa: E402
from dsp_amd.synth import synth_frames                 # noqa: E402


def main():
    B, steps = 4096, 60
    cfg = optimized_config()
    engs = [EchoEngine(cfg), EchoEngine(cfg)]
    rx = torch.from_numpy(synth_frames(B, cfg, engs[0].tables.tx_pcm)).to(engs[0].device)
    rxs = [rx, rx.clone()]
    kal = [BatchedKalman(e, 64) for e in engs]
    outs = [torch.empty((B, 64), dtype=torch.uint8, device=rx.device) for _ in range(2)]
    spec = torch.empty((B, cfg.n_bands, cfg.n_bins), dtype=torch.float32, device=rx.device)
    for e in engs:
        e.reserve(B)

    def step(k):
        i = k & 1
        engs[i].chain(rxs[i], n_streams=64, kalman=kal[i], want_spectra=True, out=outs[i], spec_out=spec)
        engs[i].chain_prefetch(rxs[i])

    for k in range(6):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"IIR mode {os.environ.get('ECHO_IIR_MODE', '0')}: {B * steps / dt:.0f} frames/s, {dt / steps * 1e3:.3f} ms/step, "
          f"prefetched IIR {engs[0].prefetch_ms():.3f} ms")


if __name__ == "__main__":
    main()
