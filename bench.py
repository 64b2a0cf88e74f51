#!/usr/bin/env python3
"""Headline benchmark: echo frames/sec through the full 3-band chain on MI355X.

  python bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path (echo_chain: zero-phase IIR + FIR bank, dB spectra, matched-filter
correlation, first_strong_peak, confidence fusion, Kalman) over one batch of 4096 synthetic rx frames per GPU,
inputs resident in HBM before the timed region.  With N > 1 (one process per GPU, launched by
torch.distributed.run) every rank processes its own 4096-frame shard of the global batch and the 64-byte
result records are all-gathered over RCCL each step -- the only exchange on this path.
With --total-frames T the job is the sharded configuration instead (strong scaling): T distinct frames, generated on
the devices, are split into contiguous shards over the ranks, a step is one pass over all T frames plus one gather of
the records.
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
PREWARM_STEPS = 40              # untimed steps ahead of the warm-up steps (GPU clock ramp), see main()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_STEP, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--total-frames", type=int, default=0,
                    help="strong-scaling job: this many distinct device-synthesised frames split over the ranks "
                         "(BASELINE configs[2] is 1000000); a step is one pass over all of them")
    ap.add_argument("--cpu-frames", type=int, default=2048, help="CPU baseline: frames per timed run")
    ap.add_argument("--cpu-runs", type=int, default=5, help="CPU baseline: timed runs of the one-core leg (median)")
    ap.add_argument("--cpu-workers", type=int, default=0,
                    help="processes of the all-cores CPU-baseline leg (0: this process's CPU share -- "
                         "OMP_NUM_THREADS when set, else the CPUs it may run on)")
    ap.add_argument("--cpu-worker", nargs=4, type=int, metavar=("START", "COUNT", "RUNS", "PIN"),
                    help="internal: time the CPU oracle RUNS times on one frame shard, pinned to core PIN (-1: not "
                         "pinned); prints the seconds of every run")
    ap.add_argument("--no-spectra", action="store_true", help="results only (17,188 B/frame variant)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the collectives even with one rank (exercises the "
                         "multi-GPU code path on a one-GPU box)")
    ap.add_argument("--device-synth", action="store_true",
                    help="generate the batch in device memory (every frame distinct, no upload)")
    ap.add_argument("--lanes", type=int, default=3,
                    help="batches in flight: consecutive steps run on this many streams (one library handle each), "
                         "only their Kalman stages ordered; 1 = one stream, the next step's IIR prefetched beside it")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="one stream, nothing overlapped: every step runs its kernels one after the other")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="development aid: no event timing of the kernels at all (the roofline object is then "
                         "meaningless); shows what the sampled timing costs")
    ap.add_argument("--mode", choices=("named", "full"), default="named",
                    help="'named': the chain the headline metric names (filter + rfft + first_strong_peak + fusion); "
                         "'full': with the reference's Hilbert-envelope branch (secondary measurement)")
    ap.add_argument("--workload", choices=("optimized", "original", "realtime", "twopeak"), default="optimized",
                    help="parameter set; the headline line is 'optimized' (airSonar_optimized.py), the others are "
                         "secondary measurements of the variants (airSonar.py, realtime_sonar.py)")
    return ap.parse_args()


def _cpu_worker(start, count, runs, pin):
    """One host process of the CPU baseline: rebuilds the tables, synthesises its own frame shard and times the oracle
    chain `runs` times over it, single-threaded, pinned to one core when asked."""
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
    if pin >= 0:
        try:
            os.sched_setaffinity(0, {pin})
        except OSError:
            pass
    from dsp_amd.config import optimized_config
    from dsp_amd.design import build_tables
    from dsp_amd.synth import synth_frames
    from oracle import echo_oracle as O
    cfg = optimized_config()
    tables = build_tables(cfg)
    rx = synth_frames(count, cfg, tables.tx_pcm, start=start)
    ns = N_STREAMS if count % N_STREAMS == 0 else 1
    O.chain(rx[:ns], tables.chirps_f32, tables.filters, cfg, n_streams=ns, mode="named", keep=("spec_db",))   # warm-up
    out = []
    for _ in range(runs):
        t0 = time.perf_counter()
        O.chain(rx, tables.chirps_f32, tables.filters, cfg, n_streams=ns, mode="named", keep=("spec_db",))
        out.append(time.perf_counter() - t0)
    return out


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def cpu_baseline(frames, runs, workers):
    """The oracle (NumPy/SciPy restatement of the reference chain) on the host, on the same synthetic frames as the GPU
    batch (frame g is a function of g): (i) one process pinned to one core, single-threaded BLAS/FFT, median of `runs`
    runs over `frames` frames; (ii) the same chain in `workers` processes, each with its own shard of frames and its own
    core, median of three rounds of the slowest worker."""
    allowed = sorted(os.sched_getaffinity(0))
    if workers <= 0:
        share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
        workers = min(len(allowed), share) if share > 0 else len(allowed)

    def spawn(start, count, nruns, pin):
        return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(start), str(count),
                                 str(nruns), str(pin)], stdout=subprocess.PIPE, text=True)

    def collect(p):
        text, _ = p.communicate(timeout=900)
        if p.returncode != 0:
            raise RuntimeError("CPU-baseline worker failed")
        return [float(x) for x in text.strip().splitlines()[-1].split()]

    one = collect(spawn(0, frames, runs, allowed[0]))
    out = {"value": frames / _median(one), "unit": "frames/s", "cores": 1, "kind": "port",
           "cpu_model": _cpu_model(), "host_logical_cpus": os.cpu_count(), "cpus_this_process_may_use": len(allowed),
           "sample": f"frames 0..{frames - 1} of the GPU batch, oracle/echo_oracle.py chain (filtfilt, fftconvolve, "
                     f"rfft+mag2db, correlate, first_strong_peak, fusion, Kalman), one process pinned to core "
                     f"{allowed[0]}, 1 thread, median of {runs} runs ({', '.join(f'{t:.2f}' for t in one)} s)"}
    if workers > 1:
        rounds = 3
        t0 = time.perf_counter()
        procs = [spawn(w * frames, frames, rounds, allowed[w % len(allowed)]) for w in range(workers)]
        times = [collect(p) for p in procs]
        slowest = [max(t[r] for t in times) for r in range(rounds)]
        out["all_cores"] = {"value": workers * frames / _median(slowest), "unit": "frames/s", "cores": workers,
                            "sample": f"{workers} pinned processes x {frames} frames each, median of {rounds} rounds of "
                                      f"the slowest process ({', '.join(f'{t:.2f}' for t in slowest)} s; wall "
                                      f"{time.perf_counter() - t0:.1f} s incl. interpreter start-up)"}
    return out


def _largest_divisor(n, cap):
    return max(d for d in range(1, cap + 1) if n % d == 0)


def main():
    args = parse()
    if args.cpu_worker:
        print(" ".join(f"{t:.6f}" for t in _cpu_worker(*args.cpu_worker)))
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    use_dist = world > 1 or args.force_dist
    if args.gpus > 1 and "RANK" not in os.environ:
        # convenience: re-launch under torch.distributed.run as a child process (nothing has touched the GPU yet)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    # the JSON line must be the only thing on stdout: libraries loaded below (the collective library prints a banner
    # there) get stderr as their file descriptor 1, and the line goes to the saved descriptor
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist

    from dsp_amd.config import optimized_config, original_config, realtime_config, twopeak_config
    from dsp_amd.dist import GatherRing, shard_bounds
    from dsp_amd.engine import BatchedKalman, EchoEngine
    from dsp_amd.lanes import ChainLanes
    from dsp_amd.synth import synth_frames

    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = {"optimized": optimized_config, "original": original_config, "realtime": realtime_config,
           "twopeak": twopeak_config}[args.workload]()
    eng = EchoEngine(cfg, dev)
    strong = args.total_frames > 0
    if strong:
        # the sharded configuration: contiguous shards of one global batch of distinct frames (frame g depends on g only)
        bounds = shard_bounds(args.total_frames, world)
        counts = [hi - lo for lo, hi in bounds]
        lo, hi = bounds[rank]
        B, Bmax = hi - lo, max(counts)
        rx = eng.synth_frames(B, first_frame=lo)
    else:
        B = Bmax = args.frames
        counts = [B] * world
        # batches beyond 4096 frames repeat a 4096-frame pool
        if args.device_synth:
            rx = eng.synth_frames(B, first_frame=rank * B)
        else:
            rx_host = synth_frames(min(B, 4096), cfg, eng.tables.tx_pcm, start=rank * min(B, 4096))
            rx = torch.from_numpy(rx_host).to(dev)
            if B > 4096:
                rx = rx.repeat((B + 4095) // 4096, 1)[:B].contiguous()
    # Batches in flight.  Up to the fusion a batch depends on nothing but its frames, and every kernel of the chain
    # keeps one unit of the GPU busy at a time, so consecutive steps run on `lanes` streams (ChainLanes: one library
    # handle each) with only their Kalman stages ordered.
    # With one lane the step runs on one stream and only the next step's IIR is started beside it (chain_prefetch).
    # The sharded job is one chain call per step over the whole shard: the library splits a call of that size into
    # chunks on three streams of its own.
    n_lanes = 1 if (args.no_pipeline or strong) else max(1, args.lanes)
    CHUNK = 4096                                            # the library's chunk for large calls
    S = _largest_divisor(B, N_STREAMS)                      # 64 whenever the batch is a multiple of it
    kal = BatchedKalman(eng, S)
    lanes = ChainLanes(eng, n_lanes) if n_lanes > 1 else None
    if lanes is not None:
        lanes.reserve(B)
    else:
        eng.reserve(min(B, 16384))
    if lanes is not None:
        # the lanes' streams take their hardware queues first (one untimed batch each), the collective library's
        # streams after them: created the other way round, lanes ended up sharing a queue and ran one after the other
        warm = torch.empty((min(B, 64), 64), dtype=torch.uint8, device=dev)
        for _ in range(n_lanes):
            lanes.submit(rx[:warm.shape[0]], want_spectra=False, out=warm, mode=args.mode)
        lanes.join()
        torch.cuda.synchronize(dev)
    want_spec = not args.no_spectra
    if lanes is None and use_dist and not args.no_pipeline and not strong and B <= 4096:
        # one stream with the next step's IIR prefetched: the engine's prefetch stream likewise takes its hardware queue
        # (one small prefetched batch) before the collective library creates its streams
        warm_rx = rx[:min(B, 64)].contiguous()
        eng.chain_prefetch(warm_rx)
        eng.chain(warm_rx, want_spectra=False)
        torch.cuda.synchronize(dev)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
    # Output buffers, one set per lane (at least two), and the exchange of the records (dsp_amd.dist.GatherRing): with
    # several ranks the all-gather of step k is asynchronous and only has to be complete before the step that writes the
    # same buffer again; with lanes it is issued from the batch's own lane, n_lanes - 1 steps later.  Uneven shards are
    # padded to the largest one inside the buffer the collective sends.
    nbuf = max(2, n_lanes) if not strong else 2
    ring = GatherRing(counts, rank, dev, nbuf=nbuf, delay=n_lanes if lanes is not None else 1, collective=use_dist)
    recs = ring.recs
    nspec = 1 if (strong or n_lanes == 1) else nbuf         # the sharded job's spectra are one array per rank
    specs = [torch.empty((B, cfg.n_bands, cfg.n_bins), dtype=torch.float32, device=dev) if want_spec else None
             for _ in range(nspec)]
    pipeline = (lanes is None and not args.no_pipeline and not strong
                and B <= (16384 if args.workload in ("optimized", "realtime") else 4096))

    def step(want_event=False):
        buf = ring.acquire()                                # this buffer's previous gather is waited for (stream order)
        spec = specs[buf % nspec]
        if lanes is not None:
            # rx is resident and never rewritten; the buffers of this slot are free once their gather has been waited
            # for, which the lane has to see only when there is a collective
            ev = lanes.submit(rx, n_streams=S, kalman=kal, want_spectra=want_spec, out=recs[buf], spec_out=spec,
                              mode=args.mode, inputs_ready=not use_dist, want_event=want_event)
            ring.commit(buf, lanes.last_stream)
            return ev
        eng.chain(rx, n_streams=S, kalman=kal, want_spectra=want_spec, out=recs[buf], spec_out=spec, mode=args.mode)
        if pipeline:
            eng.chain_prefetch(rx)
        ring.commit(buf)
        return None

    def fence():
        ring.flush()
        if lanes is not None:
            lanes.join()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # The GPU's clocks ramp for tens of milliseconds after the first kernels of a process; W warm-up steps of about a
    # millisecond each do not cover that (measured: 3.40-3.48 M frames/s over 20 timed steps behind 5 warm-up steps,
    # 3.57 M behind 50).  The untimed steps below keep the GPU busy before the W warm-up steps start; they are part of
    # neither W nor K and are reported as `gpu_prewarm_steps`.
    prewarm = 0 if strong else PREWARM_STEPS
    for _ in range(prewarm):
        step()
    if prewarm:
        fence()
    kms_serial = None
    prof_on = not args.no_kernel_timing
    if lanes is not None and prof_on and not strong:
        # outside the timed region: the same kernels one after the other on one stream, every call timed.  With several
        # batches in flight a kernel's duration above is stretched by the other lanes' kernels running beside it;
        # these are the durations each kernel has with the GPU to itself.
        eng.set_profiling(6)
        rec_x = torch.empty((B, 64), dtype=torch.uint8, device=dev)
        for _ in range(6):
            eng.chain(rx, n_streams=S, kalman=BatchedKalman(eng, S), want_spectra=want_spec, out=rec_x, spec_out=specs[0],
                      mode=args.mode)
        torch.cuda.synchronize(dev)
        kms_serial = eng.last_kernel_ms()
        eng.set_profiling(0)
    for _ in range(args.warmup):
        step()
    fence()
    # kernel durations by HIP events on the calls' own streams, inside the timed region.  Event records keep a stream's
    # kernels from running back to back, so only a sample of the calls is timed: every eighth call of the first lane
    # (every call on a single stream).
    if prof_on:
        eng.set_profiling(min(args.steps, 64), every=8 if lanes is not None else 1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_issue = time.perf_counter() - t0                      # host time to queue the steps (the GPU runs behind)
    fence()
    dt = time.perf_counter() - t0
    kms = eng.last_kernel_ms() if prof_on else {k: 1e-9 for k in ("iir", "band", "spec", "fuse", "pair", "exact")}
    eng.set_profiling(0)
    if pipeline:
        kms["iir_wait"] = kms["iir"]           # on the caller's stream only the wait for the prefetched IIR remains
        kms["iir"] = eng.prefetch_ms()         # device time of the IIR kernel on the engine's stream
    steady = None
    if lanes is not None and not strong and not use_dist:
        # Beside the fenced wall-clock value, never instead of it: the rate between the completion of two batches of
        # one continuous run, so that neither the filling nor the draining of the lanes is inside the interval.
        n_ss = max(args.steps, 2 * n_lanes)
        marks = (n_lanes - 1, n_lanes - 1 + n_ss)          # only these two batches carry a (timing) event
        evs = [step(want_event=i in marks) for i in range(n_ss + 2 * n_lanes)]
        fence()
        ms_ss = evs[marks[0]].elapsed_time(evs[marks[1]])
        steady = {"value": B * n_ss / (ms_ss / 1e3), "unit": "frames/s", "steps": n_ss, "ms_per_step": ms_ss / n_ss,
                  "note": f"GPU events at the completion of batch {n_lanes} and batch {n_lanes + n_ss} of one run of "
                          f"{n_ss + 2 * n_lanes} batches after the timed region, lanes neither filling nor draining "
                          "inside the interval; `value` above is the fenced wall-clock figure"}
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        frames = sum(counts) * args.steps
        bytes_per_frame = cfg.n_rx * 4 + 60 + (cfg.n_bands * cfg.n_bins * 4 if want_spec else 0)   # 42,892 for the headline
        # echo_chain runs batches beyond 16384 frames in passes of that size, and calls of 65536 frames or more in
        # self-pipelined passes of 8192 (long-frame parameter sets: passes of 4096); the event times are those of
        # the first pass of each call
        if args.workload in ("original", "twopeak"):
            launch_frames = min(B, 4096)
        elif B >= 3 * CHUNK and os.environ.get("ECHO_LANES", "1") != "0":
            launch_frames = CHUNK                           # large calls: chunks on the library's own streams
        else:
            launch_frames = 8192 if B >= 65536 and os.environ.get("ECHO_INNER_PIPELINE", "1") != "0" else min(B, 16384)
        # the dominant kernel: the longest of the chain's kernels by its event-measured duration in this run (the
        # band stage is one kernel or, on the fast path of the named chain, the pair kernel + the exact kernel)
        split = kms.get("pair", 0.0) > 0.0
        cand = {"iir": kms["iir"], "spec": kms["spec"]}
        if split:
            cand.update({"pair": kms["pair"], "exact": kms["exact"]})
        else:
            cand["band"] = kms["band"]
        dom = max(cand, key=cand.get)
        names = {"iir": "iir_filtfilt kernel", "band": "band_kernel", "pair": "band2_kernel", "exact": "exact_kernel",
                 "spec": "spectrum kernel"}
        if args.workload in ("original", "twopeak"):
            names.update({"band": "long-frame band stage (FIR + matched filter + detector kernels)"})
        dom_s = cand[dom] / 1e3
        tinfo = {}
        tpath = os.path.join(ROOT, "profiles", "traffic_kernels.json")
        if os.path.exists(tpath) and args.workload == "optimized" and args.mode == "named" and launch_frames == FRAMES_PER_STEP:
            try:
                tinfo = json.load(open(tpath))
            except Exception:
                tinfo = {}
        per_kernel_traffic = tinfo.get("hbm_bytes_per_launch", {})
        wl = (f"configs[2]: {args.total_frames} distinct synthetic rx frames (generated on the devices) sharded over "
              f"{world} GPU(s), one pass per step, " if strong else
              (f"configs[1]: " if args.workload == "optimized" else f"variant '{args.workload}': ")
              + f"{B} synthetic rx frames per GPU per step, ")
        # Top-level roofline: the algorithmic bytes the job moved per second of the timed region (value x bytes per
        # frame) -- the figure tied to the same clock as `value`.  The per-kernel prices of one launch sit beside it:
        # `dominant_kernel_alone` (the longest kernel with the GPU to itself) and `as_benched` (the longest kernel as it
        # ran in the timed region, stretched by whatever ran beside it).
        job_gbps = bytes_per_frame * frames / dt / 1e9 / world      # per GPU
        as_benched = {"kernel": names[dom], "kernel_ms": cand[dom],
                      "achieved": bytes_per_frame * launch_frames / dom_s / 1e9,
                      "frac": bytes_per_frame * launch_frames / dom_s / 1e9 / HBM_PEAK_GBPS,
                      "traffic": per_kernel_traffic.get(dom),
                      "note": "algorithmic bytes of one launch over this kernel's event-measured duration inside the "
                              "timed region"
                              + ("; the IIR runs on the engine's stream beside the other kernels" if pipeline else "")
                              + (f"; {n_lanes} batches are in flight, so the duration (first lane's calls, every eighth "
                                 "timed) includes the time the kernel shares the GPU with the other lanes' kernels and can "
                                 "exceed ms_per_step" if lanes is not None else "")}
        out = {
            "metric": "echo frames/sec (3-band filter+rfft+peak)",
            "value": frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64 (IIR, detector, fusion, Kalman) + f32 (FIR, FFT matched filter, spectrum)",
            "data": "synthetic (generated on the device)" if (args.device_synth or strong) else "synthetic",
            "config": {"workload": wl + f"{cfg.n_bands} bands, N={cfg.n_rx}, "
                                   f"M={cfg.n_chirp}, IIR order {cfg.IIR_ORDER} (filtfilt) + FIR {cfg.FIR_TAPS}, "
                                   f"rfft({cfg.n_rx}) dB spectra "
                                   + ("stored" if want_spec else "not stored")
                                   + ", correlate, first_strong_peak, confidence fusion, Kalman over "
                                   f"{S} streams" + (", RCCL all-gather of 64-B result records" if use_dist else "")
                                   + (", IIR of the next step prefetched on a second stream" if pipeline else "")
                                   + (f", {n_lanes} batches in flight on {n_lanes} streams (Kalman stages in order)"
                                      if lanes is not None else "")
                                   + (", one chain call per step (the library runs its chunks of 4096 frames on three "
                                      "streams of its own)" if strong else ""),
                       "frames_per_gpu_per_step": B, "n_streams": S, "spectra": want_spec, "chain_mode": args.mode,
                       "pipelined_iir": pipeline, "lanes": n_lanes},
            "roofline": {"bound": "hbm", "achieved": job_gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": job_gbps / HBM_PEAK_GBPS,
                         "traffic": tinfo.get("total_bytes_per_4096_frames"),
                         "basis": "whole job, per GPU: value x algorithmic bytes per frame / n_gpus",
                         "algorithmic_bytes_per_frame": bytes_per_frame, "frames_per_launch": launch_frames,
                         "algorithmic_bytes_per_launch": bytes_per_frame * launch_frames,
                         "note": "algorithmic bytes (rx in, spectra + record out) per second of the timed region; "
                                 "`traffic` = HBM bytes all kernels of one 4096-frame step move (PMC counters). The "
                                 "chain is bound by vector instruction issue, LDS and the IIR's latency "
                                 "(~13 MFLOP/frame), so the HBM fraction is low by construction",
                         "as_benched": as_benched},
            "kernel_ms": kms,
            "host_issue_ms_per_step": t_issue / args.steps * 1e3,
            "gpu_prewarm_steps": prewarm,
        }
        if kms_serial is not None:
            cs = {"iir": kms_serial["iir"], "spec": kms_serial["spec"]}
            if kms_serial.get("pair", 0.0) > 0.0:
                cs.update({"pair": kms_serial["pair"], "exact": kms_serial["exact"]})
            else:
                cs["band"] = kms_serial["band"]
            ds = max(cs, key=cs.get)
            a2 = bytes_per_frame * launch_frames / (cs[ds] / 1e3) / 1e9
            out["roofline"]["dominant_kernel_alone"] = {
                "kernel": names[ds], "kernel_ms": cs[ds], "achieved": a2, "frac": a2 / HBM_PEAK_GBPS,
                "traffic": per_kernel_traffic.get(ds), "kernel_ms_all": kms_serial,
                "note": "measured before the warm-up steps: 6 calls on one stream, nothing overlapped"}
        if steady is not None:
            out["steady_state"] = steady
        if strong:
            out["config"]["total_frames"] = args.total_frames
            out["config"]["frames_per_rank"] = counts
        if args.workload in ("optimized", "realtime") and args.mode == "named" and not split:
            # what bounds the single-frame band kernel: multiply-adds of the FIR and the matched filter against the
            # vector peaks (float64 78.6 TFLOP/s, float32 157.3 TFLOP/s; the chain splits each by what the detector reads)
            nexact = cfg.half + cfg.min_delay - (cfg.n_chirp - 1)
            f64 = cfg.FIR_TAPS * (cfg.n_rx - nexact) + cfg.n_chirp * (cfg.n_corr - cfg.half - cfg.min_delay)
            f32 = cfg.FIR_TAPS * nexact + cfg.n_chirp * cfg.half
            per_launch = launch_frames * cfg.n_bands
            t_peak = per_launch * 2.0 * (f64 / 78.6e12 + f32 / 157.3e12)
            out["valu"] = {"kernel": "band_kernel", "f64_fma_per_frame_band": f64, "f32_fma_per_frame_band": f32,
                           "seconds_at_vector_peak": t_peak, "frac_of_vector_peak": t_peak / (kms["band"] / 1e3),
                           "note": "supplementary: the single-frame band kernel is bound by vector instruction issue"
                                   + ("; its duration here includes the co-running prefetched IIR" if pipeline else "")}
        if world == 1 and not strong and not args.no_cpu_baseline and args.workload == "optimized":
            out["cpu_baseline"] = cpu_baseline(args.cpu_frames, args.cpu_runs, args.cpu_workers)
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if use_dist:
        last = ring.step_of.index(max(ring.step_of))
        assert torch.equal(ring.own_rows(last), recs[last])           # the gathered records hold this rank's shard
        assert ring.result(last).shape[0] == sum(counts)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
