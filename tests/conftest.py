import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without a HIP device: tests marked gpu are skipped instead of failing at the first
    `_lib.require_gpu()`.  On a GPU box nothing is skipped — a missing libdca_hip.so must fail loudly there."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs an MI355X (no HIP device visible)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))


@pytest.fixture(scope="session")
def nets():
    """tests/golden/nets.npz: network forwards recorded from the reference (make_golden_nets.py)."""
    return np.load(os.path.join(ROOT, "tests", "golden", "nets.npz"))


@pytest.fixture(scope="session")
def tiny_resnet():
    return np.load(os.path.join(ROOT, "tests", "golden", "tiny_resnet.npz"))


def synth_states(n: int, d: int, seed: int = 0) -> np.ndarray:
    """SURVEY §8d synthetic inputs: uniform random permutations of arange(d)."""
    rng = np.random.default_rng(seed)
    return rng.permuted(np.tile(np.arange(d, dtype=np.uint8), (n, 1)), axis=1)
