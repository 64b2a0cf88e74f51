#!/usr/bin/env python3
"""Development probe: aggregate throughput of n engines running the chain concurrently on n streams (the kernels of
one batch overlapping those of the others) against one engine alone.  Usage: overlap_probe.py [frames] [max engines]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd.config import optimized_config
from dsp_amd.engine import EchoEngine, BatchedKalman
from dsp_amd.synth import synth_frames

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NE = int(sys.argv[2]) if len(sys.argv) > 2 else 4
MODE = sys.argv[3] if len(sys.argv) > 3 else "named"
WL = sys.argv[4] if len(sys.argv) > 4 else "optimized"
from dsp_amd import config as _C
cfg = getattr(_C, WL + "_config")()
engs = [EchoEngine(cfg) for _ in range(NE)]
rx = torch.from_numpy(synth_frames(B, cfg, engs[0].tables.tx_pcm)).cuda()
streams = [torch.cuda.Stream() for _ in engs]
kal = [BatchedKalman(e, 64) for e in engs]
outs = [(torch.empty((B, 64), dtype=torch.uint8, device="cuda"), torch.empty((B, 3, cfg.n_bins), dtype=torch.float32, device="cuda")) for _ in engs]
for e in engs:
    e.reserve(B)

def run(n_eng, steps, prefetch=False):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        i = k % n_eng
        with torch.cuda.stream(streams[i]):
            engs[i].chain(rx, n_streams=64, kalman=kal[i], out=outs[i][0], spec_out=outs[i][1], mode=MODE)
            if prefetch:
                engs[i].chain_prefetch(rx)
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)

run(NE, 2 * NE)
for n in range(1, NE + 1):
    print("%d engine(s) on %d stream(s): %.0f frames/s" % (n, n, run(n, 48)), flush=True)
run(NE, 2 * NE, True)
for n in range(1, min(NE, 3) + 1):
    print("%d engine(s), each prefetching its next IIR: %.0f frames/s" % (n, run(n, 48, True)), flush=True)
