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


def local_subset(fx, local_maps):
    """The agents whose final local maps an episode fixture holds (all of them unless it says ``final_local_agents``)."""
    local_maps = np.asarray(local_maps)
    return local_maps[fx["final_local_agents"]] if "final_local_agents" in fx else local_maps


def unpack_correctness(fx):
    """Episode fixtures store the reference's per-sensing correctness draws bit-packed, in call order."""
    out, off = [], 0
    for n in fx["corr_lens"]:
        nbytes = (int(n) + 7) // 8
        out.append(np.unpackbits(fx["corr_packed"][off:off + nbytes])[: int(n)])
        off += nbytes
    return out


# Cells of the recorded reference episodes that carry the reference's OWN float32 re-quantisation noise beyond 1e-5: the reference
# re-quantises each map to float32 *probabilities* at every fusion (mappings.py:83), which perturbs 1-p by up to 2^-25 / 1e-4 =
# 3e-4 relative for cells that came close to the 0.9999 clip; a later contradicting observation turns that into a relative error
# of the same size in p.  No other arithmetic reproduces that noise.  Found by replaying every recorded episode through the
# oracle in exact-float64 mode (tests/test_oracle_golden.py::test_exact_mode_differs_only_by_reference_quantisation): one map
# cell, (6, 81) of episode_small5_e3, at 2.45e-5 -- in the global map and in the two local maps that received it; every other
# cell of every recording is within 1e-5.  The 493 x 493 recording (episode_default_e2, added in round 4) has two such cells, at
# 1.03e-5 .. 1.27e-5; the 8-UAV 512 x 512 recording (episode_c4_e2: twice the fusions per cell) five, the largest at 1.9e-4 --
# inside the 3e-4 the mechanism allows, which is the cap assert_posteriors holds the listed cells to.
REFERENCE_QUANTISATION_CELLS = {
    ("episode_c4_e2", "final_local"): [(0, 251, 99), (0, 256, 57), (0, 396, 75), (1, 178, 50), (1, 247, 142), (1, 251, 99), (1, 256, 57)],
    ("episode_c4_e2", "final_global"): [(251, 58), (251, 99), (424, 63)],
    ("episode_default_e2", "final_local"): [(1, 487, 373), (1, 490, 367)],
    ("episode_default_e2", "final_global"): [(487, 373)],
    ("episode_small5_e3", "final_local"): [(1, 6, 81), (3, 6, 81)],
    ("episode_small5_e3", "final_global"): [(6, 81)],
}


def assert_posteriors(actual, desired, strict, msg="", allow=None):
    """Posterior-map parity.

    strict=True  (``desired`` from the oracle in exact float64 mode): every cell within 1e-5 relative.
    strict=False, ``allow`` = list of cell indices (``desired`` recorded from the reference itself): every cell within 1e-5
      relative except the listed ones (REFERENCE_QUANTISATION_CELLS above), which must be within 3e-4 (= 2^-25 / 1e-4, what the
      reference's float32 re-quantisation of a probability next to the 0.9999 clip can do to it).
    """
    actual = np.asarray(actual, dtype=np.float64)
    desired = np.asarray(desired, dtype=np.float64)
    if strict:
        np.testing.assert_allclose(actual, desired, rtol=1e-5, atol=0, err_msg=msg)
        return
    with np.errstate(divide="ignore", invalid="ignore"):
        rel = np.where(actual == desired, 0.0, np.abs(actual - desired) / np.abs(desired))   # (cells at exactly 0: noise-free altitudes)
    assert float(rel.max()) <= 3e-4, f"{msg}: relative deviation {rel.max():.3e} exceeds the reference's own float32 re-quantisation noise"
    listed = np.zeros(rel.shape, dtype=bool)
    for idx in (allow or []):
        listed[tuple(idx)] = True
    worst = np.where(listed, 0.0, rel)
    assert float(worst.max()) <= 1e-5, (f"{msg}: cell {np.unravel_index(int(worst.argmax()), rel.shape)} deviates by {worst.max():.3e} "
                                        "and is not one of the listed re-quantisation cells of this recording")


TIE_ULPS = 32 * 2.0 ** -53   # float64 rounding of a weighted sum of <= 150 cells around 0.5


def assert_features_or_ties(got, want, deciders, rtol, atol, msg=""):
    """Feature-plane parity with the one exception the arithmetic forces.

    The class-weight planes (actor 3, 4; critic 8) are w(v) * H(v) with w switching at v = 0.499 / 0.501, v an 11 x 11 area
    average.  On small integer grids v can land ON a threshold in exact arithmetic (17 pixels per footprint:
    0.5 + 0.125 * 20 / 2500 = 0.501); which side it falls on is then decided by the last bit of whoever sums it (cv2 in the
    reference, float64 here in the oracle, integer counts and one float on the device).  An element may therefore differ --
    but ONLY if (1) its plane is a class-weight plane, (2) the oracle's deciding average is within 32 ulp of a threshold, and
    (3) the device's value is the same entropy under one of the other two class weights.  Returns the number of such ties.

    ``deciders``: {plane: float64 [11, 11] deciding average of that plane}."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    bad = ~np.isclose(got, want, rtol=rtol, atol=atol)
    ties = 0
    for x, y, p in zip(*np.nonzero(bad)):
        assert int(p) in deciders, f"{msg}: plane {p} differs at ({x},{y}): {got[x, y, p]} vs {want[x, y, p]} (not a class-weight plane)"
        v = float(deciders[int(p)][x, y])
        dist = min(abs(v - 0.499), abs(v - 0.501))
        assert dist <= TIE_ULPS, (f"{msg}: plane {p} differs at ({x},{y}): {got[x, y, p]} vs {want[x, y, p]}, and the deciding average "
                                 f"{v!r} is {dist:.3e} away from a class-weight threshold: not a tie")
        vc = min(max(v, 1e-4), 0.9999)
        h = -vc * np.log2(vc) - (1 - vc) * np.log2(1 - vc)
        assert min(abs(got[x, y, p] - c * h) for c in (0.0, 0.5, 1.0)) <= atol + rtol * h, \
            f"{msg}: plane {p} at ({x},{y}): {got[x, y, p]} is not the entropy {h} under any class weight"
        ties += 1
    return ties
