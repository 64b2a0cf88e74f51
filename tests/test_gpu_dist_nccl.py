"""The multi-GPU driver with the real engine on a real process group: one rank, backend nccl (RCCL).
`process_sharded` -- frame-sharded and stream-sharded, with the Kalman state of the local streams -- must return
the bytes of the unsharded chain call, and the gather must run through the collective (not the one-rank shortcut)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import torch.distributed as dist                         # noqa: E402

from dsp_amd import config as C                          # noqa: E402
from dsp_amd import dist as D                             # noqa: E402
from dsp_amd.synth import synth_frames                    # noqa: E402


@pytest.fixture(scope="module")
def group():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    yield dev
    dist.destroy_process_group()


def test_process_sharded_real_engine_over_rccl(group):
    from dsp_amd.engine import BatchedKalman, EchoEngine
    cfg = C.optimized_config(C.BANDS_DISTINCT)
    eng = EchoEngine(cfg, group)
    S, T = 6, 5
    rx = torch.from_numpy(synth_frames(S * T, cfg, eng.tables.tx_pcm, start=4000)).to(group)
    # frame-sharded
    want, _ = eng.chain(rx, want_spectra=False)
    got, _ = D.process_sharded(eng, rx)
    assert torch.equal(got, want)
    # stream-sharded with Kalman state
    want_s, _ = eng.chain(rx, n_streams=S, kalman=BatchedKalman(eng, S), want_spectra=False)
    got_s, spec = D.process_sharded(eng, rx, n_streams=S, kalman=BatchedKalman(eng, S), want_spectra=True)
    assert torch.equal(got_s, want_s) and spec.shape == (S * T, cfg.n_bands, cfg.n_bins)
    # the collective itself (the helpers short-cut a one-rank group): equal and padded forms
    rec = want.contiguous()
    out = torch.empty_like(rec)
    dist.all_gather_into_tensor(out, rec)
    torch.cuda.synchronize()
    assert torch.equal(out, rec)
    lst = [torch.empty_like(rec)]
    dist.all_gather(lst, rec)
    torch.cuda.synchronize()
    assert torch.equal(lst[0], rec)
    eng.close()


def test_bench_total_frames_split_covers_every_frame_once():
    """The strong-scaling split of bench.py --total-frames: shard_range over ranks covers [0, total) exactly once and
    device-synthesised shards are slices of the whole (frame g depends on g only)."""
    from dsp_amd.engine import EchoEngine
    cfg = C.optimized_config()
    eng = EchoEngine(cfg)
    total, world = 1000, 3
    whole = eng.synth_frames(total, first_frame=0)
    for r in range(world):
        lo, hi = D.shard_range(total, r, world)
        part = eng.synth_frames(hi - lo, first_frame=lo)
        assert torch.equal(part, whole[lo:hi])
    eng.close()
