#!/usr/bin/env python3
"""Per-phase cycle shares of the band kernel from the diagnostic build (make -C dsp_amd/csrc diag). Development aid."""
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
rx = torch.from_numpy(synth_frames(256, cfg, e.tables.tx_pcm)).cuda().repeat(B // 256, 1).contiguous()
for _ in range(3): e.chain(rx)
torch.cuda.synchronize()
lib = e.lib
nb, ns = C.c_int(), C.c_int(); lib.echo_diag_stamp_shape(C.byref(nb), C.byref(ns))
buf = np.zeros(nb.value * ns.value, dtype=np.uint64)
lib.echo_diag_read_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.echo_diag_read_stamps(buf.ctypes.data, buf.size) == 0
st = buf.reshape(nb.value, ns.value)[: min(nb.value, B * 3), :17].astype(np.int64)
d = np.diff(st, axis=1)
d[:, 6] = 0   # stamp 6 (band kernel end) and stamp 7 (spectrum kernel start) belong to different launches
names = ["load", "fir", "corr", "reduce+rank_count", "rank_pick", "finish", "(kernel boundary)", "spec_load", "r8 q1024", "r8 q128",
         "r8 q16", "mid", "ir8 q16", "ir8 q128", "ir8 q1024", "spec_store"]
med = np.median(d, axis=0); tot = np.median((st[:, 6] - st[:, 0]) + (st[:, 16] - st[:, 7]))
print(json.dumps({"B": B, "total_cycles_median": float(tot), "phase_cycles_median": dict(zip(names, map(float, med)))}, indent=1))
