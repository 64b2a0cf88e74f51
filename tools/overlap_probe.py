#!/usr/bin/env python3
"""Development probe: aggregate throughput of two engines running the chain concurrently on two streams (the
latency-bound IIR of one batch overlapping the VALU-bound kernels of the other) against one engine alone."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd.config import optimized_config
from dsp_amd.engine import EchoEngine, BatchedKalman
from dsp_amd.synth import synth_frames

cfg = optimized_config()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
engs = [EchoEngine(cfg), EchoEngine(cfg)]
rx = torch.from_numpy(synth_frames(B, cfg, engs[0].tables.tx_pcm)).cuda()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
kal = [BatchedKalman(e, 64) for e in engs]
outs = [(torch.empty((B, 64), dtype=torch.uint8, device="cuda"), torch.empty((B, 3, cfg.n_bins), dtype=torch.float32, device="cuda")) for _ in engs]
for e in engs:
    e.reserve(B)

def run(n_eng, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        i = k % n_eng
        with torch.cuda.stream(streams[i]):
            engs[i].chain(rx, n_streams=64, kalman=kal[i], out=outs[i][0], spec_out=outs[i][1])
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)

run(2, 6)
print("one engine, one stream : %.0f frames/s" % run(1, 40))
print("two engines, two streams: %.0f frames/s" % run(2, 40))
