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
    for rep in range(2):
        r = env.tune_placement(3)
        print("tune", r["map_kernels_us_per_step"], [x[1] for x in r["trace"]])

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

    print("probe2-style initial", score())
    for mode in ("all", "all", "only-local", "only-glob", "only-code", "only-truth", "all", "all"):
        names = {"only-local": ("local",), "only-glob": ("glob",), "only-code": ("code",), "only-truth": ("truth",)}.get(mode, ("local", "glob", "code", "truth"))
        for n in names:
            old = getattr(env, n)
            hold.append(old)
            setattr(env, n, torch.empty_like(old))
        print(mode, score())
    r = env.tune_placement(2)
    print("tune", r["map_kernels_us_per_step"], [x[1] for x in r["trace"]])


main()
