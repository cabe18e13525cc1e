import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))


@pytest.fixture(scope="session")
def tiny_resnet():
    return np.load(os.path.join(ROOT, "tests", "golden", "tiny_resnet.npz"))


def synth_states(n: int, d: int, seed: int = 0) -> np.ndarray:
    """SURVEY §8d synthetic inputs: uniform random permutations of arange(d)."""
    rng = np.random.default_rng(seed)
    return rng.permuted(np.tile(np.arange(d, dtype=np.uint8), (n, 1)), axis=1)
