#!/usr/bin/env python3
"""Development aid: duration of the long-frame detector kernel when only its first stages / selection passes run
(ECHO_DET_STAGES, ECHO_DET_PASSES are read by echo_create).  Results are wrong by construction; only the time counts."""
import json
import os
import subprocess
import sys

CHILD = r'''
import torch, numpy as np, json, sys
from dsp_amd import config as C
from dsp_amd.engine import EchoEngine
cfg = C.original_config()
e = EchoEngine(cfg)
B = 2048
g = torch.Generator(device="cpu").manual_seed(1)
corr = torch.randn((B, cfg.n_bands, cfg.n_corr), generator=g, dtype=torch.float64).to(e.device)
for _ in range(2): e.first_strong_peak(corr.clone())
torch.cuda.synchronize()
c = corr.clone()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(5): e.first_strong_peak(c)
t1.record(); torch.cuda.synchronize()
print(json.dumps({"ms": t0.elapsed_time(t1) / 5}))
'''

def run(stages, passes):
    env = dict(os.environ, ECHO_DET_STAGES=str(stages), ECHO_DET_PASSES=str(passes))
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=120)
    if out.returncode != 0:
        return out.stderr[-300:]
    return json.loads(out.stdout.strip().splitlines()[-1])["ms"]

if __name__ == "__main__":
    for stages, passes in ((1, 0), (2, 1), (2, 2), (2, 3), (2, 0), (0, 1), (0, 0)):
        print(f"stages={stages} passes={passes}: {run(stages, passes)}", flush=True)
