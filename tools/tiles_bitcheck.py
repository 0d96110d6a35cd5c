#!/usr/bin/env python
"""Bit-identity of tile storage against row-major maps on larger batches than tests/test_tile_storage.py flies: whole episodes under the uniform random policy,
every map compared after every step.  python tools/tiles_bitcheck.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
from configs import make_params  # noqa: E402
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM  # noqa: E402

CASES = [("c4", {}, 64, None), ("c5", {"experiment__missions__n_agents": 16}, 16, None),
         ("c5", {"experiment__missions__n_agents": 16}, 16, [2, 4, 8, 16] * 4),
         ("c2", {"experiment__missions__n_agents": 8, "experiment__uav__failure_rate": 0.3}, 256, None),
         ("c2", {"experiment__constraints__num_actions": 27}, 513, None)]
for name, over, n_envs, teams in CASES:
    params = make_params(name, **over)
    for track in (False, True):
        a = VecEnv(params, n_envs, map_layout="rows", track_area=track, terrain="random_field", team_sizes=teams)
        b = VecEnv(params, n_envs, map_layout="tiles", track_area=track, terrain="random_field", team_sizes=teams)
        for wave in range(2):
            eps = np.arange(1, n_envs + 1) * 7 + 5000 * wave
            a.reset(eps)
            b.reset(eps)
            for t in range(a.d.budget + 1):
                for env in (a, b):
                    if track:
                        env.build_observations(t, features=True)
                    env.steps(t, policy=POLICY_UNIFORM, features=track)
                assert torch.equal(b.rows_view(b.local), a.local) and torch.equal(b.rows_view(b.glob), a.glob), (name, track, wave, t)
                assert torch.equal(a.pos, b.pos) and torch.equal(a.code, b.code), (name, track, wave, t)
        assert int(a.fault.abs().sum()) == 0 and int(b.fault.abs().sum()) == 0
        print(name, over, n_envs, "teams" if teams else "", "tracked" if track else "env-only", ": identical over 2 x", a.d.budget + 1, "steps", flush=True)
        del a, b
        torch.cuda.empty_cache()
print("all cases bit-identical")
