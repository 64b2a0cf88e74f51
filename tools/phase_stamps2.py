#!/usr/bin/env python3
"""Per-phase clock shares of the pair kernel from the diagnostic build (make -C dsp_amd/csrc diag). Development aid."""
import ctypes as C, os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from dsp_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "dsp_amd", "libechohip_diag.so")
from dsp_amd.config import optimized_config
from dsp_amd.engine import EchoEngine
from dsp_amd.synth import synth_frames
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = optimized_config(); e = EchoEngine(cfg)
rx = torch.from_numpy(synth_frames(B, cfg, e.tables.tx_pcm)).cuda()
for _ in range(3): e.chain(rx)
torch.cuda.synchronize()
lib = e.lib
nb, ns = C.c_int(), C.c_int(); lib.echo_diag_stamp_shape(C.byref(nb), C.byref(ns))
buf = np.zeros(nb.value * ns.value, dtype=np.uint64)
lib.echo_diag_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.echo_diag_read_stamps(buf.ctypes.data, buf.size) == 0
st = buf.reshape(nb.value, ns.value)[: min(nb.value, (B + 1) // 2 * 3)].astype(np.int64)
d = np.diff(st[:, :6], axis=1)
names = ["tables+y load", "FIR x2", "x write+reduce", "FFT round", "detect+candidates"]
print(json.dumps({"B": B, "total_median": float(np.median(st[:, 5] - st[:, 0])),
                  "phase_median": dict(zip(names, map(float, np.median(d, axis=0)))),
                  "phase_mean": dict(zip(names, map(float, d.mean(axis=0))))}, indent=1))
