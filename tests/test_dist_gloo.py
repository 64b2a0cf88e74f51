"""The sharding and gather logic of the multi-GPU driver, rehearsed with two gloo ranks on CPU.
The engine is replaced by a stand-in whose records are a pure function of the frame contents, so the test
checks exactly what the distributed layer adds: shard bounds, stream-wise splitting and result ordering."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dsp_amd.dist import (GatherRing, gather_records, gather_stream_records, local_stream_frames,  # noqa: E402
                          process_sharded, shard_bounds, shard_range)


class FakeEngine:
    """chain(): record row = [frame checksum bytes, running per-stream count] -- sequential in time per stream."""

    def chain(self, rx, n_streams=None, kalman=None, want_spectra=False, out=None, spec_out=None):
        b = rx.shape[0]
        s = n_streams or b
        rec = torch.zeros((b, 64), dtype=torch.uint8)
        chk = (rx.double().sum(dim=1) * 1000).round().long()
        state = torch.zeros(s, dtype=torch.long)
        for i in range(b):
            state[i % s] += chk[i]                      # "Kalman": depends on the stream's history
            rec[i, :8] = torch.frombuffer(bytearray(int(chk[i]).to_bytes(8, "little", signed=True)), dtype=torch.uint8)
            rec[i, 8:16] = torch.frombuffer(bytearray(int(state[i % s]).to_bytes(8, "little", signed=True)),
                                            dtype=torch.uint8)
        return rec, None


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        rx = torch.randn(7 * 5 + 0, 16, generator=g)           # 35 frames: uneven split over 2 ranks
        eng = FakeEngine()
        full, _ = eng.chain(rx)
        got, _ = process_sharded(eng, rx)
        ok1 = torch.equal(got, full)
        # streams: T = 7 steps x S = 5 streams, time-major; split 3 + 2 streams
        full_s, _ = eng.chain(rx, n_streams=5)
        got_s, _ = process_sharded(eng, rx, n_streams=5)
        ok2 = torch.equal(got_s, full_s)
        # equal-count fast path
        rec = torch.full((4, 64), rank, dtype=torch.uint8)
        allr = gather_records(rec)
        ok3 = allr.shape == (8, 64) and bool((allr[:4] == 0).all()) and bool((allr[4:] == 1).all())
        # fewer streams / frames than ranks: one rank contributes nothing
        rx1 = torch.randn(3, 16, generator=g)                   # T = 3 steps of S = 1 stream
        full1, _ = eng.chain(rx1, n_streams=1)
        got1, _ = process_sharded(eng, rx1, n_streams=1)
        rxf = torch.randn(1, 16, generator=g)                   # a single frame
        fullf, _ = eng.chain(rxf)
        gotf, _ = process_sharded(eng, rxf)
        ok4 = torch.equal(got1, full1) and torch.equal(gotf, fullf)
        q.put((rank, ok1, ok2, ok3, ok4))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok1, ok2, ok3, ok4 in res:
        assert ok1, f"rank {rank}: frame-sharded records differ from the single-process run"
        assert ok2, f"rank {rank}: stream-sharded records differ from the single-process run"
        assert ok3, f"rank {rank}: equal-count gather is mis-ordered"
        assert ok4, f"rank {rank}: a shard without streams or frames breaks the gather"


def test_shard_bounds_cover_and_balance():
    for total in (0, 1, 7, 4096, 1_000_000):
        for world in (1, 2, 3, 8):
            b = shard_bounds(total, world)
            assert b[0][0] == 0 and b[-1][1] == total
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
            assert shard_range(total, world - 1, world) == b[-1]


def test_stream_split_keeps_time_order():
    t, s, n = 4, 6, 3
    rx = torch.arange(t * s * n, dtype=torch.float32).reshape(t * s, n)
    loc, (s0, s1) = local_stream_frames(rx, s, 1, 2)
    assert (s0, s1) == (3, 6) and loc.shape == (t * 3, n)
    assert torch.equal(loc[0], rx[3]) and torch.equal(loc[3], rx[s + 3])      # step 1 follows step 0
    assert gather_stream_records(loc, s) is loc                                # no process group: identity


# ---------------------------------------------------------------------------------------------------------------------
# The steady-state loop of bench.py with several ranks: one chain call per step into a ring of record buffers, the
# all-gather of a step asynchronous and issued `delay - 1` steps late, uneven shards padded inside the send buffer.
def _ring_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        eng = FakeEngine()
        problems = []
        for total, delay, nbuf, steps in ((35, 1, 2, 7), (35, 3, 3, 9), (34, 3, 4, 8), (5, 2, 2, 7), (1, 1, 2, 3)):
            bounds = shard_bounds(total, world)
            counts = [hi - lo for lo, hi in bounds]
            lo, hi = bounds[rank]
            g = torch.Generator().manual_seed(100 + total + delay)
            batches = [torch.randn(total, 16, generator=g) for _ in range(steps)]      # the same on every rank
            want = [eng.chain(b)[0] for b in batches]                                   # single-process result per step
            ring = GatherRing(counts, rank, torch.device("cpu"), nbuf=nbuf, delay=delay)
            checked = []
            held = [-1] * nbuf                          # which step's records each buffer holds

            def check(buf):
                k = held[buf]
                if k < 0:
                    return
                got = ring.result(buf)
                if got.shape != want[k].shape or not torch.equal(got, want[k]):
                    problems.append(("records", total, delay, k))
                if not torch.equal(ring.own_rows(buf), want[k][lo:hi]):
                    problems.append(("own rows", total, delay, k))
                checked.append(k)

            for k in range(steps):
                buf = ring.acquire()                    # waits for the gather of the step that used this buffer before
                if buf != k % nbuf or ring.step_of[buf] != k:
                    problems.append(("buffer order", total, delay, k))
                check(buf)                              # ... whose gathered records are complete now, not yet overwritten
                rec, _ = eng.chain(batches[k][lo:hi]) if hi > lo else (torch.zeros((0, 64), dtype=torch.uint8), None)
                ring.recs[buf].copy_(rec)
                held[buf] = k
                ring.commit(buf)
                owed = len(ring._owed)
                if owed != min(k + 1, delay - 1):       # gathers are issued delay - 1 steps late, never later
                    problems.append(("owed", total, delay, k, owed))
            ring.flush()
            for buf in range(nbuf):
                check(buf)
            if sorted(checked) != list(range(steps)):
                problems.append(("steps checked", total, delay, sorted(checked)))
        # a ring without a collective only rotates its buffers
        solo = GatherRing([4], 0, torch.device("cpu"), nbuf=2, delay=1, collective=False)
        b0 = solo.acquire(); solo.commit(b0); b1 = solo.acquire(); solo.commit(b1); solo.flush()
        if (b0, b1) != (0, 1) or solo.result(0) is not solo.recs[0]:
            problems.append(("solo",))
        q.put((rank, problems))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gather_ring():
    """GatherRing (what bench.py's multi-rank loop runs on) with two gloo ranks and the stand-in engine: >= 7 steps,
    uneven shards (18 + 17 rows; 3 + 2; 1 + 0), gathers issued at once and two steps late (lanes in {1, 3}), two to
    four buffers; every step's gathered records equal the single-process result, and a rank's own rows sit where the
    end-of-run check of bench.py looks for them."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 200)
    procs = [ctx.Process(target=_ring_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, problems in res:
        assert not problems, f"rank {rank}: {problems}"


def test_gather_ring_rejects_a_ring_shorter_than_its_delay():
    with pytest.raises(ValueError):
        GatherRing([4, 4], 0, torch.device("cpu"), nbuf=2, delay=3, collective=False)
