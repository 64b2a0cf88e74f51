import sys, os, torch, json
sys.path.insert(0, os.getcwd())
from dsp_amd.config import realtime_config
from dsp_amd.engine import BatchedKalman, EchoEngine
from dsp_amd.synth import synth_frames
cfg = realtime_config(); eng = EchoEngine(cfg)
pool = torch.from_numpy(synth_frames(256, cfg, eng.tables.tx_pcm)).to(eng.device)
for S in (1, 256):
    rx = pool[:S].contiguous(); kal = BatchedKalman(eng, S)
    out = torch.empty((S, 64), dtype=torch.uint8, device=eng.device)
    for _ in range(10): eng.chain(rx, n_streams=S, kalman=kal, want_spectra=False, out=out)
    torch.cuda.synchronize(); eng.set_profiling(32)
    for _ in range(32): eng.chain(rx, n_streams=S, kalman=kal, want_spectra=False, out=out)
    torch.cuda.synchronize(); print(S, eng.last_kernel_ms()); eng.set_profiling(0)
