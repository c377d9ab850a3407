import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        path = os.path.join(GOLDEN, name)
        if name.endswith(".json"):
            import json
            return json.load(open(path))
        return np.load(path, allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def oracle():
    from oracle import oov_oracle
    oov_oracle.build()
    return oov_oracle


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def bits_equal(a, b):
    """Bitwise equality of float32 arrays, signed zeros included.  NaNs must sit at the same
    places but their sign/payload is not compared: x86 produces 0xFFC00000 for 0/0, gfx950
    0x7FC00000, and IEEE 754 leaves that choice open."""
    a = np.ascontiguousarray(a, dtype=np.float32).copy()
    b = np.ascontiguousarray(b, dtype=np.float32).copy()
    if a.shape != b.shape:
        return False
    a[np.isnan(a)] = np.float32(np.nan)
    b[np.isnan(b)] = np.float32(np.nan)
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))
