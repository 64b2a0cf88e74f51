"""Development probe: throughput of single echo_chain calls of growing size (synchronised after each call), with the
library's own three-stream split for large calls on or off (ECHO_LANES=0)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd.config import optimized_config
from dsp_amd.engine import EchoEngine, BatchedKalman
cfg = optimized_config(); e = EchoEngine(cfg)
for B in (8192, 12288, 16384, 32768, 65536):
    rx = e.synth_frames(B)
    out = torch.empty((B,64),dtype=torch.uint8,device="cuda"); spec = torch.empty((B,3,cfg.n_bins),dtype=torch.float32,device="cuda")
    kal = BatchedKalman(e, 64)
    for _ in range(3): e.chain(rx, n_streams=64, kalman=kal, out=out, spec_out=spec)
    torch.cuda.synchronize(); t0=time.perf_counter(); n = max(3, 200000 // B)
    for _ in range(n):
        e.chain(rx, n_streams=64, kalman=kal, out=out, spec_out=spec); torch.cuda.synchronize()
    print(B, "frames per call, synchronised after each: %.0f frames/s" % (n*B/(time.perf_counter()-t0)), flush=True)
