import lzma
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def corpus():
    """alice29 || asyoulik (tests/golden/make_golden.py item 5)."""
    with lzma.open(os.path.join(GOLDEN, "corpus_alice29_asyoulik.xz")) as f:
        return np.frombuffer(f.read(), dtype=np.uint8).copy()


@pytest.fixture(scope="session")
def random_then_unicode():
    with lzma.open(os.path.join(GOLDEN, "random_then_unicode.xz")) as f:
        return np.frombuffer(f.read(), dtype=np.uint8).copy()


@pytest.fixture(scope="session")
def shuffle384():
    return np.fromfile(os.path.join(GOLDEN, "shuffle384.bin"), dtype=np.uint8)
