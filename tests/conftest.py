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


def assert_posteriors(actual, desired, strict, msg=""):
    """Posterior-map parity.

    strict=True  (``desired`` from the oracle in exact float64 mode): every cell within 1e-5 relative.
    strict=False (``desired`` produced by the reference itself / the oracle in reference mode): the reference
      re-quantises each map to float32 *probabilities* at every fusion (mappings.py:83), which perturbs 1-p by up
      to 2^-25 / 1e-4 = 3e-4 relative for cells that came close to the 0.9999 clip; a later contradicting
      observation turns that into a relative error of the same size in p.  That noise is the reference's, it is
      not reproducible by any other arithmetic, and it is bounded in log-odds (one such rounding per step while a
      cell sits near the clip; three of them ~ 1e-3).  So: >= 99.95 % of the cells within 1e-5 relative, and ALL
      cells within 1e-3 in log-odds (0.1 % in odds).
    """
    actual = np.asarray(actual, dtype=np.float64)
    desired = np.asarray(desired, dtype=np.float64)
    if strict:
        np.testing.assert_allclose(actual, desired, rtol=1e-5, atol=0, err_msg=msg)
        return
    rel = np.abs(actual - desired) / np.abs(desired)
    frac = float((rel <= 1e-5).mean())
    assert frac >= 0.9995, f"{msg}: only {frac:.6f} of the cells within 1e-5 relative"
    # beyond the clip bound the stored value only matters up to its clip (every consumer clips first), and float32
    # cannot resolve 1-p there at all; inside it the quantisation bound applies
    a, b = np.clip(actual, 1e-4, 0.9999), np.clip(desired, 1e-4, 0.9999)
    dl = np.abs(np.log(a / (1 - a)) - np.log(b / (1 - b)))
    assert float(dl.max()) <= 1e-3, f"{msg}: log-odds deviation {dl.max():.3e} exceeds the float32-quantisation bound"
