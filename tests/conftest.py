"""pytest configuration: path setup + the ``gpu`` marker.

``-m "not gpu"``: oracle vs golden vectors, host logic, C-ABI export check (no GPU needed).
``-m gpu``: parity tests proper -- the HIP path through the C-ABI against the oracle, on a real MI355X.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "ipp-marl_amd"):
    path = os.path.join(ROOT, sub)
    if path not in sys.path:
        sys.path.insert(0, path)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        return cache[name]

    return load


def unpack_correctness(fx):
    """Episode fixtures store the reference's per-sensing correctness draws bit-packed, in call order."""
    out, off = [], 0
    for n in fx["corr_lens"]:
        nbytes = (int(n) + 7) // 8
        out.append(np.unpackbits(fx["corr_packed"][off:off + nbytes])[: int(n)])
        off += nbytes
    return out
