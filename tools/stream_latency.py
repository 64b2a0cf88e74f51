#!/usr/bin/env python3
"""Per-tick latency of the streaming mode (realtime_sonar.py parameters): S streams, one new frame per stream per
tick, Kalman state carried across ticks.  Compares a direct echo_chain call with a replay of the recorded HIP graph.
Latency = host time from the launch call to the completion of the tick's last kernel (stream synchronise), input
already in HBM.  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsp_amd.config import realtime_config           # noqa: E402
from dsp_amd.engine import BatchedKalman, EchoEngine  # noqa: E402
from dsp_amd.synth import synth_frames                # noqa: E402


def main():
    cfg = realtime_config()
    eng = EchoEngine(cfg)
    pool = torch.from_numpy(synth_frames(1024, cfg, eng.tables.tx_pcm)).to(eng.device)
    eng.reserve(1024)                     # recorded graphs pin the scratch: size it once for the largest tick
    rows = []
    for S in (1, 16, 64, 256, 1024):
        ticks = 200
        rx = pool[:S].contiguous()
        res = {"streams": S}
        for mode in ("direct", "graph"):
            kal = BatchedKalman(eng, S)
            if mode == "graph":
                g = eng.capture_chain(rx, kal, want_spectra=False)
                fire = g.launch
            else:
                out = torch.empty((S, 64), dtype=torch.uint8, device=eng.device)
                fire = lambda: eng.chain(rx, n_streams=S, kalman=kal, want_spectra=False, out=out)   # noqa: E731
            for _ in range(20):
                fire()
            torch.cuda.synchronize()
            lat = []
            for _ in range(ticks):
                t0 = time.perf_counter()
                fire()
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t0)
            lat = np.array(lat) * 1e6
            res[mode + "_us_median"] = float(np.median(lat))
            res[mode + "_us_p99"] = float(np.percentile(lat, 99))
            if mode == "graph":
                g.close()
        res["frames_per_s_graph"] = S / (res["graph_us_median"] * 1e-6)
        rows.append(res)
    print(json.dumps({"workload": "realtime_sonar.py parameters, N=4282, M=137, 3 bands, results only (no spectra)",
                      "latency": rows}))


if __name__ == "__main__":
    main()
