"""The wavefront-level primitives of the kernels, alone, on the GPU: DPP folds and lane reads (`wave_fold`,
`wave_max_int`), the argmax butterfly of the detector partials, the combined LDS atomics of the radix-selection
histogram (`wave_hist_add`, the per-wavefront population count) and the shuffle scan of the bin search.  The host
emulation of the phases has no wavefronts, so these are compared here with their definitions evaluated in NumPy:
the fold bit for bit in its documented order, the integer primitives exactly."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def unit():
    path = os.path.join(HERE, "gpu_unit", "libwaveunit.so")
    assert os.path.exists(path), "tests/gpu_unit/libwaveunit.so has not been built (python __graft_entry__.py)"
    lib = ctypes.CDLL(path)
    P, I = ctypes.c_void_p, ctypes.c_int
    lib.wave_unit_fold.argtypes = [P, P, I, I, P]
    lib.wave_unit_max_int.argtypes = [P, P, I, P]
    lib.wave_unit_argmax.argtypes = [P, P, P, P, I, P]
    lib.wave_unit_hist.argtypes = [P, P, P, I, I, P, P, P]
    lib.wave_unit_scan.argtypes = [P, P, I, P]
    for f in (lib.wave_unit_fold, lib.wave_unit_max_int, lib.wave_unit_argmax, lib.wave_unit_hist, lib.wave_unit_scan):
        f.restype = I
    return lib


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _fold_definition(x, op):
    """wave_fold as documented: neighbours, pairs, mirrored half row, mirrored row, then (r0 + r1) + (r2 + r3)."""
    f = {0: np.add, 1: np.maximum, 2: np.minimum}[op]
    a = x.reshape(-1, 64).copy()
    lane = np.arange(64)
    for perm in (lane ^ 1, lane ^ 2, (lane & ~7) | (7 - (lane & 7)), (lane & ~15) | (15 - (lane & 15))):
        a = f(a, a[:, perm])
    tot = f(f(a[:, 0], a[:, 16]), f(a[:, 32], a[:, 48]))
    return np.repeat(tot, 64)


@pytest.mark.parametrize("op", [0, 1, 2])
def test_wave_fold_bits(unit, op):
    rng = np.random.default_rng(100 + op)
    n = 256 * 64
    x = rng.standard_normal(n) * np.exp(rng.uniform(-30, 30, n))       # sums that round differently in any other order
    x[:64] = 0.0
    x[64:128] = -1.5
    x[128:192] = np.where(np.arange(64) == 37, 7.0, -7.0)              # a single extreme lane
    x[192:256] = np.arange(64, dtype=np.float64)[::-1]
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    dx = _dev(x)
    assert unit.wave_unit_fold(dx.data_ptr(), out.data_ptr(), n, op, _stream()) == 0
    got = out.cpu().numpy()
    want = _fold_definition(x, op)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    if op:                                                            # max and min do not depend on the order at all
        ref = (x.reshape(-1, 64).max(axis=1) if op == 1 else x.reshape(-1, 64).min(axis=1))
        assert np.array_equal(got[::64], ref)
    else:
        exact = np.array([float(np.sum(r.astype(np.longdouble))) for r in x.reshape(-1, 64)])
        scale = np.abs(x.reshape(-1, 64)).sum(axis=1) + 1e-300
        assert (np.abs(got[::64] - exact) <= 64 * 2.3e-16 * scale).all()


def test_wave_max_int(unit):
    rng = np.random.default_rng(7)
    n = 256 * 32
    x = rng.integers(-2**31, 2**31 - 1, n, dtype=np.int64).astype(np.int32)
    x[:64] = -5
    x[64:128] = np.where(np.arange(64) == 63, 11, 3)
    x[128:192] = np.where(np.arange(64) == 0, 2**31 - 1, -2**31)
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    dx = _dev(x)
    assert unit.wave_unit_max_int(dx.data_ptr(), out.data_ptr(), n, _stream()) == 0
    want = np.repeat(x.reshape(-1, 64).max(axis=1), 64)
    assert np.array_equal(out.cpu().numpy(), want)


def test_argmax_butterfly_keeps_the_first_index_among_equals(unit):
    rng = np.random.default_rng(8)
    n = 256 * 32
    val = rng.standard_normal(n)
    v2 = val.reshape(-1, 64)
    for w in range(0, v2.shape[0], 2):                                 # every other wavefront: its maximum in several lanes
        lanes = rng.choice(64, size=rng.integers(2, 9), replace=False)
        v2[w, lanes] = v2[w].max() + 1.0
    v2[1, :] = 0.25                                                    # all lanes equal
    idx = rng.permutation(n).astype(np.int32)                          # indices in no lane order
    best = torch.empty(n, dtype=torch.float64, device="cuda")
    bidx = torch.empty(n, dtype=torch.int32, device="cuda")
    dval, didx = _dev(val), _dev(idx)
    assert unit.wave_unit_argmax(dval.data_ptr(), didx.data_ptr(), best.data_ptr(), bidx.data_ptr(), n, _stream()) == 0
    i2 = idx.reshape(-1, 64)
    wmax = v2.max(axis=1)
    widx = np.array([i2[w][v2[w] == wmax[w]].min() for w in range(v2.shape[0])], dtype=np.int32)
    assert np.array_equal(best.cpu().numpy(), np.repeat(wmax, 64))
    assert np.array_equal(bidx.cpu().numpy(), np.repeat(widx, 64))


@pytest.mark.parametrize("kind", ["clustered", "spread", "one_digit", "mixed"])
def test_histogram_and_population_count(unit, kind):
    rng = np.random.default_rng({"clustered": 1, "spread": 2, "one_digit": 3, "mixed": 4}[kind])
    blocks, rounds = 24, 19
    n = blocks * rounds * 256
    if kind == "clustered":                                            # a few digits hold nearly everything
        digit = rng.choice([3, 3, 3, 3, 130, 130, 131, 255, 0], size=n)
    elif kind == "spread":
        digit = rng.integers(0, 256, n)
    elif kind == "one_digit":
        digit = np.full(n, 77)
    else:                                                              # more than eight shared digits per wavefront
        digit = (rng.integers(0, 12, n) * 21) % 256
        digit[::5] = rng.integers(0, 256, len(digit[::5]))
    active = (rng.random(n) < 0.8).astype(np.int32)
    flag = (rng.random(n) < 0.3).astype(np.int32)
    if kind == "one_digit":
        active[:256] = 0                                               # a round with no active lane at all
        flag[:256] = 0
    digit = digit.astype(np.int32)
    hist = torch.empty((blocks, 256), dtype=torch.int32, device="cuda")
    count = torch.empty(blocks, dtype=torch.int32, device="cuda")
    dd, da, df = _dev(digit), _dev(active), _dev(flag)
    assert unit.wave_unit_hist(dd.data_ptr(), da.data_ptr(), df.data_ptr(), blocks, rounds, hist.data_ptr(),
                               count.data_ptr(), _stream()) == 0
    d2, a2 = digit.reshape(blocks, -1), active.reshape(blocks, -1)
    want = np.stack([np.bincount(d2[b][a2[b] != 0], minlength=256) for b in range(blocks)]).astype(np.int32)
    assert np.array_equal(hist.cpu().numpy(), want)
    assert np.array_equal(count.cpu().numpy(), flag.reshape(blocks, -1).sum(axis=1).astype(np.int32))


def test_shuffle_scan(unit):
    rng = np.random.default_rng(9)
    n = 256 * 16
    x = rng.integers(0, 5000, n).astype(np.int32)
    x[:64] = 0
    out = torch.empty(n, dtype=torch.int32, device="cuda")
    dx = _dev(x)
    assert unit.wave_unit_scan(dx.data_ptr(), out.data_ptr(), n, _stream()) == 0
    want = np.cumsum(x.reshape(-1, 64), axis=1).reshape(-1).astype(np.int32)
    assert np.array_equal(out.cpu().numpy(), want)
