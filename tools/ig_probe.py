#!/usr/bin/env python
"""Greedy information-gain policy (IG_baseline.py:56-325, batched: VecEnv.ig_actions) over 1024 envs x 4 UAVs x 256^2:
time per env step and per planner call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ipp-marl_amd"))
import torch
from ippmarl.params import grid256_params
from ippmarl.vec_env import VecEnv, POLICY_EXPLICIT
E = int(os.environ.get("ENVS", 1024))
# ENVS / AGENTS / ACTIONS / GRID (256, 512, 1024): e.g. BASELINE config 5's shape = ENVS=64 AGENTS=16 ACTIONS=27 GRID=1024
PIXELS = {128: 15, 256: 30, 512: 60, 1024: 120}
grid = int(os.environ.get("GRID", 256))
over = dict(experiment__missions__n_agents=int(os.environ.get("AGENTS", 4)), experiment__constraints__num_actions=int(os.environ.get("ACTIONS", 6)),
            sensor__pixel__number_x=PIXELS[grid], sensor__pixel__number_y=PIXELS[grid])
env = VecEnv(grid256_params(**over), E, terrain="random_field", track_area=False)
T = env.d.budget + 1
for rep in range(3):
    env.reset(torch.arange(1 + rep * E, 1 + (rep + 1) * E))
    torch.cuda.synchronize(); t0 = time.perf_counter(); plan_s = 0.0
    for t in range(T):
        env.build_observations(t, features=False)      # comm + local fusion: the planner reads the fused local maps
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); acts = env.ig_actions(); b.record()
        env.steps(t, policy=POLICY_EXPLICIT, actions=acts, features=False)
        torch.cuda.synchronize(); plan_s += a.elapsed_time(b) * 1e-3
    dt = time.perf_counter() - t0
print(f"IG policy ({E} envs x {env.d.n_agents} UAVs x {env.d.grid_x}^2, {env.d.n_actions} actions{', per-candidate K9' if os.environ.get('IPPM_IG_PER_CANDIDATE') else ''}): {dt / T * 1e3:.3f} ms per env step ({E * env.d.n_agents * T / dt / 1e6:.2f} M agent-env steps/s), planner {plan_s / T * 1e6:.0f} us per call")
