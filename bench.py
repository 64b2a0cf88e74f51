#!/usr/bin/env python3
"""Headline benchmark: echo frames/sec through the full 3-band chain on MI355X.

  python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path (echo_chain: zero-phase IIR + FIR bank, dB spectra, matched-filter
correlation, first_strong_peak, confidence fusion, Kalman) over one batch of 4096 synthetic rx frames per GPU,
inputs resident in HBM before the timed region.  With N > 1 (one process per GPU, launched by
torch.distributed.run) every rank processes its own 4096-frame shard of the global batch and the 64-byte
result records are all-gathered over RCCL each step -- the only exchange on this path.
Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAMES_PER_STEP = 4096          # BASELINE.json configs[1]
N_STREAMS = 64                  # 64 streams x 64 time steps per batch for the Kalman stage
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_STEP, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (one-core leg)")
    ap.add_argument("--cpu-workers", type=int, default=16, help="processes of the all-cores CPU-baseline leg")
    ap.add_argument("--cpu-worker", nargs=2, type=int, metavar=("START", "COUNT"),
                    help="internal: run the CPU oracle on one frame shard and print the seconds taken")
    ap.add_argument("--no-spectra", action="store_true", help="results only (17,188 B/frame variant)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the collectives even with one rank (exercises the "
                         "multi-GPU code path on a one-GPU box)")
    ap.add_argument("--device-synth", action="store_true",
                    help="generate the batch in device memory (every frame distinct, no upload); the CPU baseline "
                         "then runs on frames downloaded from the device")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do not start the next step's IIR on the engine's own stream during the current step")
    ap.add_argument("--mode", choices=("named", "full"), default="named",
                    help="'named': the chain the headline metric names (filter + rfft + first_strong_peak + fusion); "
                         "'full': with the reference's Hilbert-envelope branch (secondary measurement)")
    ap.add_argument("--workload", choices=("optimized", "original", "realtime", "twopeak"), default="optimized",
                    help="parameter set; the headline line is 'optimized' (airSonar_optimized.py), the others are "
                         "secondary measurements of the variants (airSonar.py, realtime_sonar.py)")
    return ap.parse_args()


def _cpu_worker(job):
    """One host process of the all-cores baseline: rebuilds the tables and runs the oracle on its own frame shard."""
    start, count, n_streams = job
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
    from dsp_amd.config import optimized_config
    from dsp_amd.design import build_tables
    from dsp_amd.synth import synth_frames
    from oracle import echo_oracle as O
    cfg = optimized_config()
    tables = build_tables(cfg)
    rx = synth_frames(count, cfg, tables.tx_pcm, start=start)
    t0 = time.perf_counter()
    O.chain(rx, tables.chirps_f32, tables.filters, cfg, n_streams=n_streams, mode="named", keep=("spec_db",))
    return time.perf_counter() - t0


def cpu_baseline(cfg, tables, rx_host, budget_s, workers):
    """The oracle (NumPy/SciPy restatement of the reference chain) on the host: one core on a bounded sample of
    the GPU batch, then the same chain on `workers` processes, each with its own shard of frames."""
    from oracle import echo_oracle as O
    t0 = time.perf_counter()
    O.chain(rx_host[:32], tables.chirps_f32, tables.filters, cfg, n_streams=32, mode="named")
    per = (time.perf_counter() - t0) / 32
    n = int(max(64, budget_s / per))
    n -= n % 64
    t0 = time.perf_counter()
    for f0 in range(0, n, len(rx_host)):                   # beyond one batch the sample goes round the batch again
        m = min(len(rx_host), n - f0)
        O.chain(rx_host[:m], tables.chirps_f32, tables.filters, cfg, n_streams=min(m, N_STREAMS), mode="named",
                keep=("spec_db",))
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{n} frames drawn from the {len(rx_host)} frames of the GPU batch, oracle/echo_oracle.py chain "
                     f"(filtfilt, fftconvolve, rfft+mag2db, correlate, first_strong_peak, fusion, Kalman), "
                     f"{dt:.1f} s on 1 thread"}
    if workers > 1:
        per_worker = max(64, int(n / 2) - int(n / 2) % 64)
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker",
                                   str(w * per_worker), str(per_worker)], stdout=subprocess.PIPE, text=True)
                 for w in range(workers)]
        inner = []
        for p in procs:
            text, _ = p.communicate(timeout=600)
            if p.returncode != 0:
                raise RuntimeError("CPU-baseline worker failed")
            inner.append(float(text.strip().splitlines()[-1]))
        wall = max(inner)
        out["all_cores"] = {"value": workers * per_worker / wall, "unit": "frames/s", "cores": workers,
                            "sample": f"{workers} processes x {per_worker} frames, slowest process {wall:.1f} s "
                                      f"(wall {time.perf_counter() - t0:.1f} s incl. interpreter start-up)"}
    return out


def main():
    args = parse()
    if args.cpu_worker:
        print(_cpu_worker((args.cpu_worker[0], args.cpu_worker[1], 64)))
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = world > 1 or args.force_dist
    if args.gpus > 1 and "RANK" not in os.environ:
        # convenience: re-launch under torch.distributed.run as a child process (nothing has touched the GPU yet)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch
    import torch.distributed as dist

    from dsp_amd.config import optimized_config, original_config, realtime_config, twopeak_config
    from dsp_amd.engine import BatchedKalman, EchoEngine
    from dsp_amd.synth import synth_frames

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    cfg = {"optimized": optimized_config, "original": original_config, "realtime": realtime_config,
           "twopeak": twopeak_config}[args.workload]()
    eng = EchoEngine(cfg, dev)
    B = args.frames
    # batches beyond 4096 frames repeat a 4096-frame pool (the sharded 1M-frame configuration)
    if args.device_synth:
        rx = eng.synth_frames(B, first_frame=rank * B)
        rx_host = rx[:min(B, 4096)].cpu().numpy()
    else:
        rx_host = synth_frames(min(B, 4096), cfg, eng.tables.tx_pcm, start=rank * min(B, 4096))
        rx = torch.from_numpy(rx_host).to(dev)
        if B > 4096:
            rx = rx.repeat((B + 4095) // 4096, 1)[:B].contiguous()
    S = N_STREAMS if B % N_STREAMS == 0 else 1
    kal = BatchedKalman(eng, S)
    eng.reserve(B)
    want_spec = not args.no_spectra
    # two record buffers: with several ranks the all-gather of step k is asynchronous and only has to be complete
    # before step k+2 writes the same buffer again, so the collective never stalls the compute stream
    recs = [torch.empty((B, 64), dtype=torch.uint8, device=dev) for _ in range(2)]
    rec = recs[0]
    spec = torch.empty((B, cfg.n_bands, cfg.n_bins), dtype=torch.float32, device=dev) if want_spec else None
    gathered = [torch.empty((world * B, 64), dtype=torch.uint8, device=dev) for _ in range(2)] if use_dist else None
    works = [None, None]
    counter = [0]

    # Steady-state pipelining: the input of step k+1 is resident, so its zero-phase IIR (a sequential recurrence that
    # leaves the arithmetic units idle) is started on the engine's own stream as soon as step k has been issued.
    # Each step consumes one prefetched IIR and issues one, so K timed steps do the work of K full chains.
    pipeline = not args.no_pipeline and B <= (16384 if args.workload in ("optimized", "realtime") else 4096)

    def step():
        buf = counter[0] & 1
        counter[0] += 1
        if works[buf] is not None:
            works[buf].wait()
        eng.chain(rx, n_streams=S, kalman=kal, want_spectra=want_spec, out=recs[buf], spec_out=spec, mode=args.mode)
        if pipeline:
            eng.chain_prefetch(rx)
        if use_dist:
            works[buf] = dist.all_gather_into_tensor(gathered[buf], recs[buf], async_op=True)

    def fence():
        for w in works:
            if w is not None:
                w.wait()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    eng.set_profiling(min(args.steps, 64))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    kms = eng.last_kernel_ms()
    eng.set_profiling(0)
    if pipeline:
        kms["iir_wait"] = kms["iir"]           # on the caller's stream only the wait for the prefetched IIR remains
        kms["iir"] = eng.prefetch_ms()         # device time of the IIR kernel on the engine's stream
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        frames = world * B * args.steps
        bytes_per_frame = cfg.n_rx * 4 + 60 + (cfg.n_bands * cfg.n_bins * 4 if want_spec else 0)   # 42,892 for the headline
        band_s = kms["band"] / 1e3
        # echo_chain runs batches beyond 16384 frames in passes of that size, and calls of 65536 frames or more in
        # self-pipelined passes of 8192 (long-frame parameter sets: passes of 4096)
        if args.workload in ("original", "twopeak"):
            launch_frames = min(B, 4096)
        else:
            launch_frames = 8192 if B >= 65536 and os.environ.get("ECHO_INNER_PIPELINE", "1") != "0" else min(B, 16384)
        achieved = bytes_per_frame * launch_frames / band_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_band_kernel.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch") if args.workload == "optimized" and B == FRAMES_PER_STEP else None
            except Exception:
                traffic = None
        out = {
            "metric": "echo frames/sec (3-band filter+rfft+peak)",
            "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (generated on the device)" if args.device_synth else "synthetic",
            "config": {"workload": (f"configs[1]: " if args.workload == "optimized" else f"variant '{args.workload}': ")
                                   + f"{B} synthetic rx frames per GPU per step, {cfg.n_bands} bands, N={cfg.n_rx}, "
                                   f"M={cfg.n_chirp}, IIR order {cfg.IIR_ORDER} (filtfilt) + FIR {cfg.FIR_TAPS}, "
                                   f"rfft({cfg.n_rx}) dB spectra "
                                   + ("stored" if want_spec else "not stored")
                                   + ", correlate, first_strong_peak, confidence fusion, Kalman over "
                                   f"{S} streams" + (", RCCL all-gather of 64-B result records" if world > 1 else "")
                                   + (", IIR of the next step prefetched on a second stream" if pipeline else ""),
                       "frames_per_gpu_per_step": B, "n_streams": S, "spectra": want_spec, "chain_mode": args.mode,
                       "pipelined_iir": pipeline},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "kernel": "band_kernel", "kernel_ms": kms["band"],
                         "algorithmic_bytes_per_launch": bytes_per_frame * launch_frames,
                         "note": "VALU/LDS-bound chain (~13 MFLOP/frame f64+f32): HBM fraction is low by construction"},
            "kernel_ms": kms,
        }
        if args.workload != "original" and args.mode == "named":
            # what bounds this kernel: multiply-adds of the FIR and the matched filter against the vector peaks
            # (float64 78.6 TFLOP/s, float32 157.3 TFLOP/s; the chain splits each by what the detector reads)
            nexact = cfg.half + cfg.min_delay - (cfg.n_chirp - 1)
            f64 = cfg.FIR_TAPS * (cfg.n_rx - nexact) + cfg.n_chirp * (cfg.n_corr - cfg.half - cfg.min_delay)
            f32 = cfg.FIR_TAPS * nexact + cfg.n_chirp * cfg.half
            per_launch = launch_frames * cfg.n_bands
            t_peak = per_launch * 2.0 * (f64 / 78.6e12 + f32 / 157.3e12)
            out["valu"] = {"kernel": "band_kernel", "f64_fma_per_frame_band": f64, "f32_fma_per_frame_band": f32,
                           "seconds_at_vector_peak": t_peak, "frac_of_vector_peak": t_peak / band_s,
                           "note": "supplementary: the band kernel is bound by vector instruction issue, not by HBM"
                                   + ("; its duration here includes the co-running prefetched IIR" if pipeline else "")}
        if world == 1 and not args.no_cpu_baseline and args.workload == "optimized":
            out["cpu_baseline"] = cpu_baseline(cfg, eng.tables, rx_host, args.cpu_seconds, args.cpu_workers)
        print(json.dumps(out), flush=True)
    if use_dist:
        last = (counter[0] - 1) & 1
        assert torch.equal(gathered[last][rank * B:(rank + 1) * B], recs[last])   # the gathered records hold this rank's shard
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
