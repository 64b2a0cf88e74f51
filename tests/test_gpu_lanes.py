"""Several batches in flight (ChainLanes / echo_chain_ordered): records, spectra and the Kalman column must be the
bytes of the same batches run one after the other on one stream, although only the Kalman stages are ordered."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from dsp_amd import config as C                          # noqa: E402
from dsp_amd.synth import synth_frames                    # noqa: E402


@pytest.mark.parametrize("mode,lanes", [("named", 3), ("named", 2), ("full", 3)])
def test_lanes_match_sequential_calls(mode, lanes):
    from dsp_amd.engine import BatchedKalman, EchoEngine
    from dsp_amd.lanes import ChainLanes
    cfg = C.optimized_config(C.BANDS_DISTINCT)
    eng = EchoEngine(cfg)
    S, nbatch = 16, 7
    sizes = [S * t for t in (6, 1, 9, 4, 12, 3, 5)]                       # ragged batches, whole time steps each
    rx = torch.from_numpy(synth_frames(sum(sizes), cfg, eng.tables.tx_pcm, start=31000)).cuda()
    offs = np.concatenate([[0], np.cumsum(sizes)])
    # one stream, one call after the other
    kal = BatchedKalman(eng, S)
    want = [eng.chain(rx[offs[i]:offs[i + 1]], n_streams=S, kalman=kal, mode=mode) for i in range(nbatch)]
    want = [(r.clone(), s.clone()) for r, s in want]
    kx = kal.x.clone()
    # the same batches on lanes
    ln = ChainLanes(eng, lanes)
    kal2 = BatchedKalman(eng, S)
    outs = [torch.empty((n, 64), dtype=torch.uint8, device="cuda") for n in sizes]
    specs = [torch.empty((n, cfg.n_bands, cfg.n_bins), dtype=torch.float32, device="cuda") for n in sizes]
    for i in range(nbatch):
        ln.submit(rx[offs[i]:offs[i + 1]], n_streams=S, kalman=kal2, out=outs[i], spec_out=specs[i], mode=mode)
    ln.join()
    torch.cuda.synchronize()
    for i in range(nbatch):
        assert torch.equal(outs[i], want[i][0]), f"batch {i}: records differ"
        assert torch.equal(specs[i], want[i][1]), f"batch {i}: spectra differ"
    assert torch.equal(kal2.x, kx)
    # without a Kalman state the lanes are plain concurrent chain calls
    plain, _ = eng.chain(rx[:sizes[0]], want_spectra=False, mode=mode)
    o = torch.empty((sizes[0], 64), dtype=torch.uint8, device="cuda")
    ln.submit(rx[:sizes[0]], want_spectra=False, out=o, mode=mode)
    ln.join(); torch.cuda.synchronize()
    assert torch.equal(o, plain)
    ln.close(); eng.close()


def test_ordered_call_validates_and_passes_empty_batches_on():
    import ctypes as Ct
    from dsp_amd import _lib
    from dsp_amd.engine import BatchedKalman, EchoEngine, _ptr
    cfg = C.optimized_config()
    eng = EchoEngine(cfg)
    kal = BatchedKalman(eng, 4)
    s = torch.cuda.current_stream()
    sp = Ct.c_void_p(s.cuda_stream)
    rx = torch.zeros((4, cfg.n_rx), dtype=torch.float32, device="cuda")
    out = torch.empty((4, 64), dtype=torch.uint8, device="cuda")
    # no Kalman pointers: nothing to order
    rc = eng.lib.echo_chain_ordered(eng._h, _ptr(rx), _ptr(out), None, None, None, None, 4, 4, 0, sp, None, None)
    assert rc != 0
    # an empty batch still records its event, so the chain of batches is not broken
    ev = torch.cuda.Event(); ev.record(s)
    rc = eng.lib.echo_chain_ordered(eng._h, _ptr(rx), _ptr(out), None, _ptr(kal.x), _ptr(kal.p), _ptr(kal.init), 0, 4, 0,
                                    sp, None, Ct.c_void_p(ev.cuda_event))
    assert rc == 0
    ev.synchronize()
    eng.close()


def test_lanes_with_batches_of_several_internal_passes():
    """Batches beyond one internal pass (16384 frames): the Kalman stage of every pass follows the previous batch's."""
    from dsp_amd.engine import BatchedKalman, EchoEngine
    from dsp_amd.lanes import ChainLanes
    cfg = C.optimized_config()
    eng = EchoEngine(cfg)
    S = 64
    sizes = [16384 + 640, 4096, 16384 + 64]
    rx = eng.synth_frames(sum(sizes), first_frame=5000)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    kal = BatchedKalman(eng, S)
    want = [eng.chain(rx[offs[i]:offs[i + 1]], n_streams=S, kalman=kal, want_spectra=False)[0].clone() for i in range(3)]
    ln = ChainLanes(eng, 2)
    kal2 = BatchedKalman(eng, S)
    outs = [torch.empty((n, 64), dtype=torch.uint8, device="cuda") for n in sizes]
    for i in range(3):
        ln.submit(rx[offs[i]:offs[i + 1]], n_streams=S, kalman=kal2, want_spectra=False, out=outs[i])
    ln.join(); torch.cuda.synchronize()
    for i in range(3):
        assert torch.equal(outs[i], want[i]), f"batch {i}: records differ"
    assert torch.equal(kal2.x, kal.x)
    ln.close(); eng.close()


@pytest.mark.parametrize("env", [{"ECHO_IIR_SIDE": "1"}, {"ECHO_IIR_SIDE": "1", "ECHO_IIR_CUS": "192"},
                                 {"ECHO_IIR_PRIO": "3"}])
def test_scheduling_switches_leave_the_bytes_alone(env, monkeypatch):
    """The switches behind the scheduling measurements -- every handle's IIR on one stream shared by the device
    (ECHO_IIR_SIDE), that stream confined to a subset of the compute units (ECHO_IIR_CUS), the IIR's waves at a raised
    priority (ECHO_IIR_PRIO) -- change where and when the kernels run, not what they compute: three batches in flight
    give the bytes of the plain engine."""
    from dsp_amd.engine import BatchedKalman, EchoEngine
    from dsp_amd.lanes import ChainLanes
    cfg = C.optimized_config(C.BANDS_DISTINCT)
    plain = EchoEngine(cfg)
    S, sizes = 16, [16 * 40, 16 * 7, 16 * 64, 16 * 3, 16 * 33]
    rx = torch.from_numpy(synth_frames(sum(sizes), cfg, plain.tables.tx_pcm, start=777)).cuda()
    offs = np.concatenate([[0], np.cumsum(sizes)])
    kal = BatchedKalman(plain, S)
    want = [tuple(t.clone() for t in plain.chain(rx[offs[i]:offs[i + 1]], n_streams=S, kalman=kal)) for i in range(len(sizes))]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    eng = EchoEngine(cfg)
    ln = ChainLanes(eng, 3)
    for k in env:
        monkeypatch.delenv(k)
    kal2 = BatchedKalman(eng, S)
    outs = [torch.empty((n, 64), dtype=torch.uint8, device="cuda") for n in sizes]
    specs = [torch.empty((n, cfg.n_bands, cfg.n_bins), dtype=torch.float32, device="cuda") for n in sizes]
    for i in range(len(sizes)):
        ln.submit(rx[offs[i]:offs[i + 1]], n_streams=S, kalman=kal2, out=outs[i], spec_out=specs[i])
    ln.join(); torch.cuda.synchronize()
    for i in range(len(sizes)):
        assert torch.equal(outs[i], want[i][0]) and torch.equal(specs[i], want[i][1]), i
    assert torch.equal(kal2.x, kal.x)
    ln.close(); eng.close(); plain.close()


@pytest.mark.parametrize("which,mode", [("original", "named"), ("twopeak", "named"), ("optimized", "full")])
def test_lanes_with_the_confined_iir_stream(which, mode, monkeypatch):
    """Long frames (airSonar.py's and 666.py's sizes) and the full chain of the short ones in its two-kernel form
    (ECHO_FCE=0: band kernel + envelope kernel; the fused kernel keeps its IIR in line): with batches of 2048 frames and
    more in flight on several handles the library moves every IIR to one stream confined to a share of the compute
    units.  Where the IIR runs changes nothing in what is computed: the lanes' bytes equal those of an engine with that
    switched off (ECHO_IIR_SIDE=0) called batch after batch."""
    from dsp_amd.engine import BatchedKalman, EchoEngine
    from dsp_amd.lanes import ChainLanes
    cfg = {"original": C.original_config, "twopeak": C.twopeak_config, "optimized": C.optimized_config}[which]()
    if mode == "full":
        monkeypatch.setenv("ECHO_FCE", "0")
    monkeypatch.setenv("ECHO_IIR_SIDE", "0")
    plain = EchoEngine(cfg)
    monkeypatch.delenv("ECHO_IIR_SIDE")
    S, sizes = 64, [2048, 2560, 2048, 3072, 2048, 1024]
    rx = plain.synth_frames(sum(sizes), first_frame=900)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    kal = BatchedKalman(plain, S)
    want = [tuple(t.clone() for t in plain.chain(rx[offs[i]:offs[i + 1]], n_streams=S, kalman=kal, mode=mode))
            for i in range(len(sizes))]
    eng = EchoEngine(cfg)
    ln = ChainLanes(eng, 3)
    if mode == "full":
        monkeypatch.delenv("ECHO_FCE")
    kal2 = BatchedKalman(eng, S)
    outs = [torch.empty((n, 64), dtype=torch.uint8, device="cuda") for n in sizes]
    specs = [torch.empty((n, cfg.n_bands, cfg.n_bins), dtype=torch.float32, device="cuda") for n in sizes]
    for rounds in range(2):                                  # the second round starts with batches already in flight
        for i in range(len(sizes)):
            ln.submit(rx[offs[i]:offs[i + 1]], n_streams=S, kalman=kal2 if rounds == 0 else None, out=outs[i],
                      spec_out=specs[i], mode=mode)
        ln.join(); torch.cuda.synchronize()
        for i in range(len(sizes)):
            if rounds == 0:
                assert torch.equal(outs[i], want[i][0]), i
            assert torch.equal(specs[i], want[i][1]), i
    assert torch.equal(kal2.x, kal.x)
    # the route was taken (by the lanes' engines, for batches of 2048 frames and more) and only there
    assert sum(e.confined_iir_passes() for e in ln.engines) >= 4
    assert plain.confined_iir_passes() == 0
    ln.close(); eng.close(); plain.close()
