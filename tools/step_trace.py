"""Completion times of the batches of one fenced run of K steps (three lanes), as bench.py times it: where the
fixed cost of a short run goes (filling and draining the lanes, the fence)."""
import sys, time
import torch
sys.path.insert(0, ".")
from dsp_amd import config as C
from dsp_amd.engine import BatchedKalman, EchoEngine
from dsp_amd.lanes import ChainLanes

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = C.optimized_config()
eng = EchoEngine(cfg)
B, S = 4096, 64
rx = eng.synth_frames(B, first_frame=0)
lanes = ChainLanes(eng, 3); lanes.reserve(B)
kal = BatchedKalman(eng, S)
recs = [torch.empty((B, 64), dtype=torch.uint8, device="cuda") for _ in range(3)]
specs = [torch.empty((B, cfg.n_bands, cfg.n_bins), dtype=torch.float32, device="cuda") for _ in range(3)]

STAGGER_US = int(sys.argv[3]) if len(sys.argv) > 3 else 0


def run(n, events):
    evs = []
    for i in range(n):
        if STAGGER_US and events and 0 < i < 3:
            t = time.perf_counter()
            while (time.perf_counter() - t) * 1e6 < STAGGER_US:
                pass
        evs.append(lanes.submit(rx, n_streams=S, kalman=kal, out=recs[i % 3], spec_out=specs[i % 3], inputs_ready=True,
                                want_event=events))
    return evs

def fence():
    lanes.join(); torch.cuda.synchronize()

run(W, False); fence()
for trial in range(3):
    start = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    start.record(torch.cuda.current_stream())
    evs = run(K, True)
    t_issue = time.perf_counter() - t0
    fence()
    dt = time.perf_counter() - t0
    ts = [start.elapsed_time(e) for e in evs]
    d = [ts[0]] + [ts[i] - ts[i - 1] for i in range(1, K)]
    print(f"trial {trial}: wall {dt*1e3:.3f} ms ({B*K/dt/1e6:.3f} M/s), host issue {t_issue*1e3:.3f} ms, last batch done at {ts[-1]:.3f} ms")
    print("  completion deltas (ms):", " ".join(f"{x:.2f}" for x in d))
