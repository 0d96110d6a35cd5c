#!/usr/bin/env python
"""Env-only step time at a grid that is / is not a multiple of 4 cells wide (16-byte vs 4-byte lane accesses)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ipp-marl_amd"))
import torch
from ippmarl.params import grid256_params
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM
for n in (30, 29, 31):
    env = VecEnv(grid256_params(sensor__pixel__number_x=n, sensor__pixel__number_y=n), 1024, terrain="split", track_area=False)
    env.reset(torch.arange(1, 1025))
    T = env.d.budget + 1
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(T - 1):
            env.steps(t, policy=POLICY_UNIFORM, features=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (T - 1)
        env.reset(torch.arange(2000, 3024))
    print(f"pixels {n}: grid {env.d.grid_x}: {dt * 1e6:.1f} us per step (no resets)")
    del env
