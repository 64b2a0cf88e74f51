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
ex = buf.reshape(nb.value, ns.value)[: min(nb.value, B * 3), 12:17].astype(np.int64)
de = np.diff(ex, axis=1)
allst = buf.reshape(nb.value, ns.value).astype(np.int64)
def span(a, b, rows):
    x = allst[:rows, a]; y = allst[:rows, b]
    return int(y.max() - x[x > 0].min()), int((x == 0).sum())
print("kernel spans in ticks (and unstamped rows): pair", span(0, 5, len(st)), " exact", span(12, 16, min(nb.value, B * 3)))
sg = buf.reshape(nb.value, ns.value)[: min(nb.value, B * 3)].astype(np.int64)
print("noise segment: stage, FIR, corr, fold:", np.median(sg[:, 17] - sg[:, 13]), np.median(sg[:, 18] - sg[:, 17]), np.median(sg[:, 19] - sg[:, 18]), np.median(sg[:, 14] - sg[:, 19]))
print(json.dumps({"exact_total_median": float(np.median(ex[:, 4] - ex[:, 0])), "exact_total_max": float((ex[:, 4] - ex[:, 0]).max()),
                  "exact_phase_median": dict(zip(["setup", "noise segment", "candidate segments", "median"], map(float, np.median(de, axis=0)))),
                  "exact_phase_mean": dict(zip(["setup", "noise segment", "candidate segments", "median"], map(float, de.mean(axis=0))))}))
print(json.dumps({"B": B, "total_median": float(np.median(st[:, 5] - st[:, 0])),
                  "phase_median": dict(zip(names, map(float, np.median(d, axis=0)))),
                  "phase_mean": dict(zip(names, map(float, d.mean(axis=0))))}, indent=1))
