"""Several batches in flight: the chain on a few streams at once.

Every kernel of the chain saturates one unit of the GPU at a time (memory while a workgroup stages its input, the
vector ALUs in the FIR, LDS in the transforms, instruction latency in the IIR recurrence), so one batch after the
other leaves most of the chip waiting most of the time.  Batches are independent up to the fusion -- only the Kalman
stage continues the state of the batch before (ScalarKalman, airSonar_optimized.py:265-283, :739) -- so ``ChainLanes``
runs consecutive batches on a few lanes (one library handle and one stream each) and orders the Kalman stages alone
(``echo_chain_ordered``).  Results are byte-identical to ``EchoEngine.chain`` called batch after batch.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from . import _lib
from .engine import BatchedKalman, EchoEngine, _ptr


class ChainLanes:
    """``submit`` is ``EchoEngine.chain`` that returns at once; ``join`` makes the caller's stream wait for everything
    submitted.  ``rx``, ``out`` and ``spec_out`` of a submitted batch must stay untouched until ``join`` (or until the
    event ``submit`` returns has completed)."""

    def __init__(self, engine: EchoEngine, lanes: int = 3):
        if lanes < 1:
            raise _lib.EchoError("ChainLanes needs at least one lane")
        self.engine = engine
        self.device = engine.device
        # lane 0 is the caller's engine; the others share its tables and configuration
        self.engines: List[EchoEngine] = [engine] + [EchoEngine(engine.cfg, engine.device, tables=engine.tables)
                                                     for _ in range(lanes - 1)]
        with torch.cuda.device(self.device):
            # default priority on purpose: with high-priority lanes a collective queued at normal priority behind
            # them never got to run on this driver (the lanes' wait packets kept the high-priority queues active)
            self.streams = [torch.cuda.Stream(device=self.device) for _ in range(lanes)]
            # one event per lane for the order of the Kalman stages, recorded once here so that it exists: a lane's
            # event is recorded again by every ordered call on that lane, and the next batch's wait is queued before
            # the lane's following call records it once more
            self._kal_ev = [torch.cuda.Event() for _ in range(lanes)]
            for e, st in zip(self._kal_ev, self.streams):
                e.record(st)
        self._next = 0
        self.last_stream: Optional[torch.cuda.Stream] = None    # the lane of the batch submitted last
        self._kal_done: Optional[torch.cuda.Event] = None      # Kalman stage of the batch submitted last

    def reserve(self, max_frames: int):
        for e in self.engines:
            e.reserve(max_frames)

    def submit(self, rx: torch.Tensor, n_streams: Optional[int] = None, kalman: Optional[BatchedKalman] = None,
               want_spectra: bool = True, out: Optional[torch.Tensor] = None, spec_out: Optional[torch.Tensor] = None,
               mode: str = "named", inputs_ready: bool = False, want_event: bool = False) -> Optional[torch.cuda.Event]:
        """Queue one batch on the next lane.  Arguments as ``EchoEngine.chain``; ``out`` (uint8 [B,64]) is required, and
        ``spec_out`` when ``want_spectra``.  ``inputs_ready``: ``rx`` is complete already (nothing pending on the
        caller's stream produces it), so the lane need not wait for that stream.  Returns the event that completes with
        the batch when ``want_event``; ``last_stream`` is the lane the batch went to."""
        lane = self._next
        eng = self.engines[lane]
        stream = self.streams[lane]
        self.last_stream = stream
        self._next = (self._next + 1) % len(self.engines)
        cfg = eng.cfg
        if not isinstance(rx, torch.Tensor) or rx.dim() != 2:
            raise _lib.EchoError("submit: rx must be a [B, n_rx] device tensor")
        B, nb = rx.shape[0], cfg.n_bands
        eng._out(rx, torch.float32, (B, cfg.n_rx), "rx")
        eng._out(out, torch.uint8, (B, 64), "out")
        if want_spectra:
            eng._out(spec_out, torch.float32, (B, nb, cfg.n_bins), "spec_out")
        if mode not in ("named", "full"):
            raise _lib.EchoError(f"unknown chain mode {mode!r}")
        S = n_streams if n_streams is not None else (kalman.S if kalman is not None else max(B, 1))
        if kalman is not None and (kalman.S != S or kalman.engine.device != self.device):
            raise _lib.EchoError("kalman state must hold n_streams streams on the engine's device")
        if not inputs_ready:
            stream.wait_stream(torch.cuda.current_stream(self.device))   # the inputs are ready once that stream gets here
        done = None
        with torch.cuda.device(self.device), torch.cuda.stream(stream):
            sp = C.c_void_p(stream.cuda_stream)
            spec = spec_out if want_spectra else None
            if kalman is None:
                name = "echo_chain_full" if mode == "full" else "echo_chain"
                args = (eng._h, _ptr(rx), _ptr(out), _ptr(spec)) + ((None,) if mode == "full" else ()) + \
                       (None, None, None, B, S, sp)
                _lib.check(getattr(eng.lib, name)(*args), name)
            else:
                kdone = self._kal_ev[lane]
                wait = self._kal_done
                _lib.check(eng.lib.echo_chain_ordered(
                    eng._h, _ptr(rx), _ptr(out), _ptr(spec), _ptr(kalman.x), _ptr(kalman.p), _ptr(kalman.init), B, S,
                    1 if mode == "full" else 0, sp, C.c_void_p(wait.cuda_event) if wait is not None else None,
                    C.c_void_p(kdone.cuda_event)), "echo_chain_ordered")
                self._kal_done = kdone
            if want_event:
                done = torch.cuda.Event(enable_timing=True)
                done.record(stream)
        return done

    def join(self):
        """The caller's stream waits for every submitted batch."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)

    def close(self):
        for e in self.engines[1:]:
            e.close()
        self.engines = self.engines[:1]
