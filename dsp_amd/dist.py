"""Multi-GPU driver: one process per GPU, frames (or whole streams) sharded across ranks, per-frame result
records gathered with one collective (RCCL on GPUs, gloo on CPU tensors in the unit tests).

The reference has no distributed path; the only exchange this hot path needs is the gather of the 64-byte
result rows (range, confidence, SNR per band, fused range, Kalman output) -- spectra stay on the rank that
produced them.  Sharding rules:
  * independent frames: contiguous blocks, rank r takes frames [r*B/W, (r+1)*B/W) (remainder to the low ranks);
  * Kalman streams: a stream is sequential in time, so streams -- never time steps -- are split across ranks.
"""
from __future__ import annotations

from contextlib import nullcontext
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(total: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) per rank; the first total % world ranks get one extra item."""
    base, rem = divmod(total, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    return shard_bounds(total, world)[rank]


def local_stream_frames(rx_time_major: torch.Tensor, n_streams: int, rank: int, world: int):
    """rx [T*S, N] time-major over S streams -> this rank's [T*S_r, N] (still time-major) and its stream range."""
    ts, n = rx_time_major.shape
    t = ts // n_streams
    s0, s1 = shard_range(n_streams, rank, world)
    local = rx_time_major.reshape(t, n_streams, n)[:, s0:s1].reshape(t * (s1 - s0), n)
    return local.contiguous(), (s0, s1)


def gather_records(rec_local: torch.Tensor, counts: Optional[List[int]] = None, group=None) -> torch.Tensor:
    """All-gather of per-frame record rows [B_r, 64] uint8 -> [sum B_r, 64] in rank order, on every rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return rec_local
    world = dist.get_world_size(group)
    if counts is None or len(set(counts)) == 1:
        out = torch.empty((world * rec_local.shape[0],) + tuple(rec_local.shape[1:]), dtype=rec_local.dtype,
                          device=rec_local.device)
        dist.all_gather_into_tensor(out, rec_local.contiguous(), group=group)
        return out
    # uneven shards: pad every contribution to the largest count, gather once, drop the padding
    cmax = max(counts)
    padded = torch.zeros((cmax,) + tuple(rec_local.shape[1:]), dtype=rec_local.dtype, device=rec_local.device)
    padded[: rec_local.shape[0]] = rec_local
    out = torch.empty((world * cmax,) + tuple(rec_local.shape[1:]), dtype=rec_local.dtype, device=rec_local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * cmax: r * cmax + c] for r, c in enumerate(counts)], dim=0)


def gather_stream_records(rec_local: torch.Tensor, n_streams: int, group=None,
                          n_steps: Optional[int] = None) -> torch.Tensor:
    """Records of stream-sharded, time-major batches -> global time-major order [T*S, 64].
    ``n_steps`` (T) is needed when a rank may hold no stream at all (fewer streams than ranks): its contribution is
    empty, so T cannot be read off its rows."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return rec_local
    world = dist.get_world_size(group)
    bounds = shard_bounds(n_streams, world)
    s_r = [hi - lo for lo, hi in bounds]
    mine = s_r[dist.get_rank(group)]
    if n_steps is None:
        if mine == 0:
            raise ValueError("gather_stream_records: this rank holds no stream; pass n_steps")
        n_steps = rec_local.shape[0] // mine
    t = int(n_steps)
    if rec_local.shape[0] != t * mine:
        raise ValueError(f"gather_stream_records: expected {t * mine} local rows, got {rec_local.shape[0]}")
    flat = gather_records(rec_local, [t * s for s in s_r], group)
    parts, off = [], 0
    width = flat.shape[1]
    for s in s_r:
        if s:
            parts.append(flat[off:off + t * s].reshape(t, s, width))
        off += t * s
    return torch.cat(parts, dim=1).reshape(t * n_streams, width)


def process_sharded(engine, rx_global: torch.Tensor, n_streams: Optional[int] = None, kalman=None,
                    want_spectra: bool = False, group=None):
    """Run the chain on this rank's shard of rx_global and gather the records of all ranks.
    With n_streams the batch is time-major over streams and is split by stream; otherwise by frame.
    Returns (records of all frames in global order, this rank's spectra or None)."""
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if n_streams:
        if rx_global.shape[0] % n_streams:
            raise ValueError("process_sharded: the batch must hold whole time steps of n_streams frames")
        t = rx_global.shape[0] // n_streams
        local, (s0, s1) = local_stream_frames(rx_global, n_streams, rank, world)
        if s1 > s0:
            rec, spec = engine.chain(local, n_streams=s1 - s0, kalman=kalman, want_spectra=want_spectra)
        else:                                  # more ranks than streams: this rank contributes no rows
            rec, spec = torch.empty((0, 64), dtype=torch.uint8, device=rx_global.device), None
        return gather_stream_records(rec, n_streams, group, n_steps=t), spec
    lo, hi = shard_range(rx_global.shape[0], rank, world)
    if hi > lo:
        rec, spec = engine.chain(rx_global[lo:hi], want_spectra=want_spectra)
    else:                                      # more ranks than frames
        rec, spec = torch.empty((0, 64), dtype=torch.uint8, device=rx_global.device), None
    counts = [b - a for a, b in shard_bounds(rx_global.shape[0], world)]
    return gather_records(rec, counts, group), spec


class GatherRing:
    """The record exchange of a steady-state multi-rank loop: step k writes this rank's records into buffer
    k % nbuf, and the all-gather of a step is asynchronous -- it only has to be complete before the step that writes
    the same buffer again, so the collective never stalls the compute streams.

    * Uneven shards (``counts`` = rows per rank) are padded to the largest one inside the buffer the collective sends;
      ``result`` drops the padding again.
    * ``delay``: the gather of a step is issued when ``delay - 1`` later steps have been committed (the rest at
      ``flush``), from the stream the step itself was queued on.  With several batches in flight on several streams
      (``ChainLanes``) ``delay`` = the number of lanes: by then every lane's newer work sits behind work that cannot start
      before this batch is done anyway, so the collective's wait for the batch -- its stream shares one of the few
      hardware queues with some lane -- holds nothing back.  ``nbuf >= delay`` keeps a buffer from coming round before
      its gather has been issued.
    * Without a process group (one rank, no collective) the ring only rotates the buffers.

    Per step: ``buf = ring.acquire()``; queue the work that writes ``ring.recs[buf]`` (ordered after the caller's
    current stream, which ``acquire`` made wait for the buffer's previous gather); ``ring.commit(buf, stream)``.
    """

    def __init__(self, counts: List[int], rank: int, device, nbuf: int = 2, delay: int = 1, group=None,
                 width: int = 64, dtype=torch.uint8, collective: Optional[bool] = None):
        if nbuf < 1 or delay < 1 or nbuf < delay:
            raise ValueError("GatherRing: need nbuf >= delay >= 1")
        self.counts = list(counts)
        self.world, self.rank, self.group = len(self.counts), rank, group
        self.rows, self.rows_max = self.counts[rank], max(self.counts)
        self.nbuf, self.delay = nbuf, delay
        self.collective = (dist.is_initialized() if collective is None else collective)
        self.send = [torch.zeros((self.rows_max, width), dtype=dtype, device=device) for _ in range(nbuf)]
        self.recs = [b[: self.rows] for b in self.send]           # what a step writes: this rank's rows
        self.gathered = ([torch.empty((self.world * self.rows_max, width), dtype=dtype, device=device)
                          for _ in range(nbuf)] if self.collective else None)
        self._works = [None] * nbuf
        self._owed: List[Tuple[int, object]] = []                  # committed, gather not issued yet: (buffer, stream)
        self._steps = 0
        self.step_of = [-1] * nbuf                                 # which step's records each buffer holds

    def acquire(self) -> int:
        """The buffer of the next step.  Its previous gather must be over before it is written again: the wait is
        queued on the caller's current stream (RCCL) or blocks the host (gloo)."""
        buf = self._steps % self.nbuf
        if any(b == buf for b, _ in self._owed):                   # cannot happen with nbuf >= delay
            raise RuntimeError("GatherRing: buffer re-acquired before its gather was issued")
        w = self._works[buf]
        if w is not None:
            w.wait()
            self._works[buf] = None
        self.step_of[buf] = self._steps
        self._steps += 1
        return buf

    def commit(self, buf: int, stream=None):
        """The work that fills ``recs[buf]`` has been queued on ``stream`` (None: the current stream, or host code that
        has finished)."""
        if not self.collective:
            return
        self._owed.append((buf, stream))
        while len(self._owed) >= self.delay:
            self._issue()

    def _issue(self):
        buf, st = self._owed.pop(0)
        with (torch.cuda.stream(st) if st is not None else nullcontext()):
            self._works[buf] = dist.all_gather_into_tensor(self.gathered[buf], self.send[buf], group=self.group,
                                                           async_op=True)

    def flush(self):
        """Issue what is owed and wait for every gather in flight."""
        while self._owed:
            self._issue()
        for i, w in enumerate(self._works):
            if w is not None:
                w.wait()
                self._works[i] = None

    def result(self, buf: int) -> torch.Tensor:
        """Records of all ranks for the step in ``buf``, global order, padding dropped: [sum(counts), width].  Valid
        once that step's gather has been waited for (``flush``, or the ``acquire`` that hands the buffer out again --
        before the new step's work is queued)."""
        if not self.collective:
            return self.recs[buf]
        g = self.gathered[buf]
        if len(set(self.counts)) == 1:
            return g
        return torch.cat([g[r * self.rows_max: r * self.rows_max + c] for r, c in enumerate(self.counts)], dim=0)

    def own_rows(self, buf: int) -> torch.Tensor:
        """This rank's rows inside the gathered buffer (they must equal ``recs[buf]``)."""
        off = self.rank * self.rows_max
        return self.gathered[buf][off: off + self.rows] if self.collective else self.recs[buf]
