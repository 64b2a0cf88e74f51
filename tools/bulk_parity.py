#!/usr/bin/env python3
"""Larger parity run: the GPU chain against the oracle on many synthetic frames (named and full chain, both band
sets).  The oracle runs in worker processes on frame shards (they never touch the GPU); the parent runs the GPU chain,
replays the Kalman filter over the oracle's fused ranges and writes a JSON summary.
Usage: bulk_parity.py [frames] [workers] [out.json] [lanes] [variant]; exits non-zero on any index or validity
mismatch.  variant: optimized (default: both band sets, named and full chain), or original / twopeak / realtime (the
other parameter sets, named chain).
BULK_START=<index> moves the run to another range of synthetic frames (default 100000).
With lanes > 1 the GPU side runs as batches of 4096 frames in flight on that many streams (ChainLanes), which also
checks that concurrent handles do not disturb each other."""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
START = int(os.environ.get("BULK_START", "100000"))   # index of the first synthetic frame (frame i is a function of seed + i)
CHUNK = 500


def _config(which):
    from dsp_amd import config as C
    return {"default": lambda: C.optimized_config(C.BANDS_DEFAULT), "distinct": lambda: C.optimized_config(C.BANDS_DISTINCT),
            "original": C.original_config, "twopeak": C.twopeak_config, "realtime": C.realtime_config}[which]()


def _oracle_shard(job):
    which, mode, lo, n = job
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
    from dsp_amd import config as C
    from dsp_amd.design import build_tables
    from dsp_amd.synth import synth_frames
    from oracle import echo_oracle as O
    cfg = _config(which)
    t = build_tables(cfg)
    rx = synth_frames(n, cfg, t.tx_pcm, start=START + lo)
    o = O.chain(rx, t.chirps_f32, t.filters, cfg, n_streams=n, mode=mode)
    return lo, {k: o[k] for k in ("peak_idx", "valid", "snr_db", "confidence", "distance", "fused_dist", "avg_conf", "valid_any")}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else min(16, os.cpu_count() or 1)
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    n_lanes = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    variant = sys.argv[5] if len(sys.argv) > 5 else "optimized"
    legs = [("default", ("named", "full")), ("distinct", ("named", "full"))] if variant == "optimized" else [(variant, ("named",))]
    if os.environ.get("BULK_MODES"):                        # e.g. BULK_MODES=full: only those chain modes
        keep = os.environ["BULK_MODES"].split(",")
        legs = [(w, tuple(m for m in modes if m in keep)) for w, modes in legs]
    from dsp_amd import config as C
    from dsp_amd.engine import BatchedKalman, EchoEngine, records_to_numpy
    from dsp_amd.synth import synth_frames
    from oracle import echo_oracle as O
    S = 64
    n -= n % S
    summary = {"variant": variant, "lanes": n_lanes, "frames_per_leg": n, "first_frame_seed_index": START, "n_streams": S, "oracle_workers": workers, "legs": []}
    bad = 0
    ctx = mp.get_context("spawn")
    with ctx.Pool(workers) as pool:
        for which, modes in legs:
            cfg = _config(which)
            eng = EchoEngine(cfg)
            rx = np.empty((n, cfg.n_rx), dtype=np.float32)
            for lo in range(0, n, 50000):                       # in pieces, with a sign of life for long runs
                hi = min(n, lo + 50000)
                rx[lo:hi] = synth_frames(hi - lo, cfg, eng.tables.tx_pcm, start=START + lo)
                print(f"  {which}: frames {hi}/{n} generated", flush=True)
            for mode in modes:
                t0 = time.time()
                jobs = [(which, mode, lo, min(CHUNK, n - lo)) for lo in range(0, n, CHUNK)]
                o = {}
                done = 0
                for lo, part in pool.imap_unordered(_oracle_shard, jobs):
                    for k, a in part.items():
                        o.setdefault(k, np.zeros((n,) + a.shape[1:], a.dtype))[lo:lo + len(a)] = a
                    done += 1
                    if done % 20 == 0:
                        print(f"  {which}/{mode}: oracle {done * CHUNK}/{n} frames, {time.time() - t0:.0f} s", flush=True)
                kfs = [O.ScalarKalman(cfg.KALMAN_Q, cfg.KALMAN_R) for _ in range(S)]
                kal = np.zeros(n)
                for i in range(n):
                    kf = kfs[i % S]
                    kal[i] = kf.update(o["fused_dist"][i]) if o["valid_any"][i] else (kf.x if kf.x is not None else 0.0)
                if n_lanes > 1:
                    import torch
                    from dsp_amd.lanes import ChainLanes
                    ln = ChainLanes(eng, n_lanes)
                    rxd = torch.from_numpy(rx).to(eng.device)
                    rec = torch.empty((n, 64), dtype=torch.uint8, device=eng.device)
                    kal_l = BatchedKalman(eng, S)
                    for lo in range(0, n, 4096):
                        hi = min(n, lo + 4096)
                        ln.submit(rxd[lo:hi], n_streams=S, kalman=kal_l, want_spectra=False, out=rec[lo:hi], mode=mode)
                    ln.join(); torch.cuda.synchronize(); ln.close()
                else:
                    rec, _ = eng.chain(rx, n_streams=S, kalman=BatchedKalman(eng, S), mode=mode, want_spectra=False)
                r = records_to_numpy(rec)
                valid = ((r["valid_mask"][:, None] >> np.arange(3)) & 1).astype(bool)
                v = o["valid"]
                if mode == "full":      # the oracle's full-chain index: recovered from the distance (an index into corr)
                    want_idx = np.rint(o["distance"] * 2 * cfg.FS / cfg.sound_speed).astype(np.int64) + cfg.range_offset
                    diff = (r["peak_idx"][:, :3] != want_idx) & v
                else:
                    want_idx = o["peak_idx"]
                    diff = r["peak_idx"][:, :3] != want_idx
                mism = int(diff.sum())
                where = [[int(i), int(j), int(r["peak_idx"][i, j]), int(want_idx[i, j])] for i, j in zip(*np.nonzero(diff))][:50]
                vmism = int((valid != v).sum())
                rel = lambda a, b: float((np.abs(a - b) / np.maximum(np.abs(b), 1e-12)).max()) if len(b) else 0.0
                leg = {"bands": which, "mode": mode, "frames": n, "frame_bands": int(n * 3), "valid_frame_bands": int(v.sum()),
                       "index_mismatches": mism, "validity_mismatches": vmism,
                       "index_mismatch_list_frame_band_got_want": where,
                       "max_rel_err_snr": rel(r["snr_db"][:, :3][v], o["snr_db"][v]),
                       "max_rel_err_confidence": rel(r["conf"][:, :3][v], o["confidence"][v]),
                       "max_rel_err_distance": rel(r["dist"][:, :3][v], o["distance"][v]),
                       "max_rel_err_fused": rel(r["fused_dist"][o["valid_any"]], o["fused_dist"][o["valid_any"]]),
                       "max_abs_err_fused": float(np.abs(r["fused_dist"][o["valid_any"]] - o["fused_dist"][o["valid_any"]]).max()) if o["valid_any"].any() else 0.0,
                       "max_abs_err_kalman": float(np.abs(r["kalman"] - kal).max()), "oracle_seconds": round(time.time() - t0, 1)}
                print(json.dumps(leg), flush=True)
                summary["legs"].append(leg)
                bad += mism + vmism
            eng.close()
    summary["total_index_or_validity_mismatches"] = bad
    if out_path:
        with open(out_path, "w") as f:
            json.dump(summary, f, indent=1)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
