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


def rx_from_i16(q):
    """AudioIO.record quantisation (airSonar_optimized.py:573)."""
    return q.astype(np.float32) / 2 ** 15
