import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "chainer-speech-recognition_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracles are loops of small torch operators (1000 time steps of a GRU on (32, 512) tensors): on a host that shows 256 logical
    # CPUs torch would fan every one of them out over 256 threads, and the full-size ragged GRU oracle took 140 s instead of 7.  The
    # GPU box gives a process 16 cores (the share bench.py's cpu_baseline reports as well).
    import torch
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
