#!/usr/bin/env python
"""Env-only step time at grids that are / are not a multiple of 4 cells wide (rows 16-byte vs only 4-byte aligned), and per cell at
the reference's default 493 x 493 against 496 x 496 (64 envs x 2 UAVs)."""
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

from ippmarl.params import default_params
# the reference's default parameters (50 m world, 57 pixels, 60 deg: 493 x 493 cells) against the same with a field of view of
# 59.72 deg, which makes the grid 496 x 496 (a multiple of 4 wide)
for angle in (60, 59.72):
    p = default_params(sensor__field_of_view__angle_x=angle, sensor__field_of_view__angle_y=angle, experiment__missions__n_agents=2)
    env = VecEnv(p, 256, terrain="split", track_area=False)
    env.reset(torch.arange(1, 257))
    T = env.d.budget + 1
    for rep in range(2):
        env.counters(reset=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(T - 1):
            env.steps(t, policy=POLICY_UNIFORM, features=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (T - 1)
        c = env.counters()
        env.reset(torch.arange(2000, 2256))
    cells = (c["sense_cells"] + c["fuse_local_cells"] + c["fuse_global_cells"]) / (T - 1)
    print(f"default params, fov {angle}: grid {env.d.grid_x} x {env.d.grid_y} (vec {env.d.vec}): {dt * 1e6:.1f} us per step of 256 envs x 2 UAVs, "
          f"{cells / 1e6:.2f} M cells per step -> {dt * 1e9 / cells * 1e3:.2f} ps per cell")
    del env
