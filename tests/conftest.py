import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped, not failed, on a host without a GPU or without the built library."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    have_lib = os.path.exists(os.path.join(ROOT, "dsp_amd", "libechohip.so"))
    if have_gpu and have_lib:
        return
    why = "no GPU visible" if not have_gpu else "dsp_amd/libechohip.so has not been built"
    skip = pytest.mark.skip(reason=why)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def gold_default():
    return load_golden("optimized_default")


@pytest.fixture(scope="session")
def gold_distinct():
    return load_golden("optimized_distinct")


@pytest.fixture(scope="session")
def gold_original():
    return load_golden("original")


@pytest.fixture(scope="session")
def gold_twopeak():
    return load_golden("twopeak")


@pytest.fixture(scope="session")
def gold_realtime():
    return load_golden("realtime")


@pytest.fixture(scope="session")
def gold_loop_default():
    return load_golden("run_loop_default")


@pytest.fixture(scope="session")
def gold_loop_distinct():
    return load_golden("run_loop_distinct")


def rx_from_i16(q):
    """AudioIO.record quantisation (airSonar_optimized.py:573)."""
    return q.astype(np.float32) / 2 ** 15
