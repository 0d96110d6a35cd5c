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

    def carve(name, arena_bytes, offset=0):
        old = getattr(env, name)
        hold.append(old)
        nbytes = old.numel() * old.element_size()
        arena = torch.empty(arena_bytes, dtype=torch.uint8, device="cuda")
        hold.append(arena)
        new = arena[offset:offset + nbytes].view(old.dtype).view(old.shape)
        setattr(env, name, new)
        return hex(new.data_ptr())

    print("initial", score(), score())
    for rep in range(3):
        for name in ("glob", "code", "truth", "local"):
            for arena in ((1 << 30), (1 << 29), None, (1 << 31)):
                old = getattr(env, name)
                nbytes = old.numel() * old.element_size()
                if arena is None:
                    hold.append(old)
                    setattr(env, name, torch.empty_like(old))
                    where = hex(getattr(env, name).data_ptr())
                    tag = "own"
                elif arena < nbytes:
                    continue
                else:
                    where = carve(name, arena)
                    tag = f"{arena >> 20}MB-arena"
                print(rep, name, tag, where, score())


main()
