"""Follow-up of placement_probe.py: one VecEnv, candidate buffer sets drawn in different ways, one timed episode each."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
sys.path.insert(0, ROOT)
from bench import bench_params  # noqa: E402
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM  # noqa: E402


class A:
    envs, agents, grid, actions, terrain = 1024, 4, 256, None, "random_field"


def main():
    env = VecEnv(bench_params(A), 1024, philox_seed=3, terrain="random_field", track_area=False)
    T = env.d.budget + 1
    ids = list(range(1, 1025))
    hold = []

    def score():
        env._boxes_valid = False
        env.reset(ids)
        for t in range(T):
            env.steps(t, policy=POLICY_UNIFORM, features=False)
        env.reset(ids)
        env.profile = True
        for t in range(T):
            env.steps(t, policy=POLICY_UNIFORM, features=False)
        env.profile = False
        tm = env.event_times_us()
        return round(tm["sense"]["avg_us"], 1), round(tm["fuse"]["avg_us"], 1)

    def addrs():
        return {n: hex(getattr(env, n).data_ptr()) for n in ("local", "glob", "code", "truth")}

    print("initial", score(), score(), addrs())
    for mode in ("adjacent", "spaced", "adjacent", "spaced", "only-local", "only-glob", "only-code", "only-truth", "adjacent", "spaced"):
        names = {"only-local": ("local",), "only-glob": ("glob",), "only-code": ("code",), "only-truth": ("truth",)}.get(mode, ("local", "glob", "code", "truth"))
        for n in names:
            old = getattr(env, n)
            hold.append(old)
            if mode == "spaced":
                hold.append(torch.empty(301 * 1024 * 1024 + 12288, dtype=torch.uint8, device="cuda"))
            setattr(env, n, torch.empty_like(old))
        print(mode, score(), addrs())


main()
