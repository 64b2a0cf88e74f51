"""The entry point that allocates nothing (echo_chain_noalloc: scratch in a fixed per-GPU device array) against
echo_chain, through the C ABI: records, spectra and Kalman states byte-identical, on two calls in a row, the second
with a smaller input, and a batch beyond the fixed scratch refused with a clear error."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


@pytest.mark.parametrize("which", ["optimized", "realtime", "original"])
def test_noalloc_chain_matches_chain_two_calls_in_a_row(which):
    from dsp_amd import _lib, config as cfgs
    from dsp_amd.engine import BatchedKalman, EchoEngine
    from dsp_amd.synth import synth_frames
    cfg = {"optimized": lambda: cfgs.optimized_config(cfgs.BANDS_DISTINCT), "realtime": cfgs.realtime_config,
           "original": cfgs.original_config}[which]()
    e = EchoEngine(cfg)
    lib, h = e.lib, e._h
    nmax = lib.echo_noalloc_max_frames(h)
    assert nmax >= 64, nmax
    if which == "optimized":
        assert nmax >= 256
    S = 8
    sizes = (min(nmax, 192) // S * S, 40)                 # second call smaller: stale scratch beyond it must not matter
    pool = torch.from_numpy(synth_frames(sizes[0] + sizes[1], cfg, e.tables.tx_pcm, start=31)).to(e.device)
    k_ref, k_new = BatchedKalman(e, S), BatchedKalman(e, S)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    off = 0
    for B in sizes:
        rx = pool[off:off + B].contiguous(); off += B
        want = torch.empty((B, 64), dtype=torch.uint8, device=e.device)
        want_spec = torch.empty((B, cfg.n_bands, cfg.n_bins), dtype=torch.float32, device=e.device)
        got = torch.full_like(want, 0xAB); got_spec = torch.full_like(want_spec, float("nan"))
        rc = lib.echo_chain(h, _p(rx), _p(want), _p(want_spec), _p(k_ref.x), _p(k_ref.p), _p(k_ref.init), B, S, stream)
        assert rc == 0
        torch.cuda.synchronize()
        rc = lib.echo_chain_noalloc(h, _p(rx), _p(got), _p(got_spec), _p(k_new.x), _p(k_new.p), _p(k_new.init), B, S, stream)
        assert rc == 0, lib.echo_strerror(rc)
        torch.cuda.synchronize()
        assert torch.equal(got, want), B
        assert torch.equal(got_spec, want_spec), B
        assert torch.equal(k_ref.x, k_new.x) and torch.equal(k_ref.p, k_new.p) and torch.equal(k_ref.init, k_new.init)
    # without spectra and without a Kalman state
    rx = pool[:24].contiguous()
    want = torch.empty((24, 64), dtype=torch.uint8, device=e.device); got = torch.zeros_like(want)
    assert lib.echo_chain(h, _p(rx), _p(want), None, None, None, None, 24, 24, stream) == 0
    assert lib.echo_chain_noalloc(h, _p(rx), _p(got), None, None, None, None, 24, 24, stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    # beyond the fixed scratch: refused, nothing written
    B = nmax + 1
    big = pool[:1].expand(B, cfg.n_rx).contiguous()
    out = torch.full((B, 64), 0x5A, dtype=torch.uint8, device=e.device)
    rc = lib.echo_chain_noalloc(h, _p(big), _p(out), None, None, None, None, B, B, stream)
    assert rc == -6 and b"fixed scratch" in lib.echo_strerror(rc)
    torch.cuda.synchronize()
    assert bool((out == 0x5A).all())
    with pytest.raises(_lib.EchoError):
        e.chain_noalloc(big, out)
    assert lib.echo_chain_noalloc(h, _p(big), _p(out), None, None, None, None, 0, 1, stream) == 0     # empty batch
    e.close()


def test_noalloc_chain_inside_a_stream_capture():
    """echo_chain_noalloc between hipStreamBeginCapture and hipStreamEndCapture (torch's graph capture in its 'global'
    error mode, where a hipMalloc or an un-captured event operation anywhere in the process fails the capture):
    nothing executes while capturing, every replay leaves the bytes echo_chain leaves -- Kalman state carried from
    replay to replay -- and a prefetch that was pending on the handle is still honoured afterwards."""
    from dsp_amd import config as cfgs
    from dsp_amd.engine import BatchedKalman, EchoEngine
    from dsp_amd.synth import synth_frames
    cfg = cfgs.optimized_config(cfgs.BANDS_DISTINCT)
    e = EchoEngine(cfg)
    lib, h = e.lib, e._h
    B, S = 64, 8
    pool = torch.from_numpy(synth_frames(3 * B, cfg, e.tables.tx_pcm, start=77)).to(e.device)
    ticks = [pool[:B].contiguous(), pool[B:2 * B].contiguous()]
    other = pool[2 * B:].contiguous()
    k_ref, k_new = BatchedKalman(e, S), BatchedKalman(e, S)
    want = []
    for rx in ticks:
        want.append(e.chain(rx, n_streams=S, kalman=k_ref))
    want_other = e.chain(other)
    torch.cuda.synchronize()
    e.chain_prefetch(other)                                   # stays pending across the captured call
    rx = torch.empty_like(ticks[0])
    out = torch.full((B, 64), 0xAB, dtype=torch.uint8, device=e.device)
    spec = torch.full((B, cfg.n_bands, cfg.n_bins), float("nan"), dtype=torch.float32, device=e.device)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=e.device)
    with torch.cuda.graph(g, stream=side):
        rc = lib.echo_chain_noalloc(h, _p(rx), _p(out), _p(spec), _p(k_new.x), _p(k_new.p), _p(k_new.init), B, S,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.echo_strerror(rc)
    torch.cuda.synchronize()
    assert bool((out == 0xAB).all())                          # recorded, not run
    for t, frames in enumerate(ticks):
        rx.copy_(frames)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want[t][0]) and torch.equal(spec, want[t][1]), t
    assert torch.equal(k_ref.x, k_new.x) and torch.equal(k_ref.p, k_new.p) and torch.equal(k_ref.init, k_new.init)
    got_other = e.chain(other)
    torch.cuda.synchronize()
    assert torch.equal(got_other[0], want_other[0]) and torch.equal(got_other[1], want_other[1])
    del g
    e.close()
