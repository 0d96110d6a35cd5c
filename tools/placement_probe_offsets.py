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

    print("initial", score(), score())
    name = sys.argv[1] if len(sys.argv) > 1 else "glob"
    old = getattr(env, name)
    nbytes = old.numel() * old.element_size()
    arena = torch.empty(nbytes + (1 << 30), dtype=torch.uint8, device="cuda")
    MB = 1 << 20
    offs = [0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 24, 28, 32, 48, 64, 96, 128, 192, 256, 384, 512, 640, 768, 1, 3, 0.5, 0.25, 0.125, 0.0625]
    for o in offs:
        off = int(o * MB)
        new = arena[off:off + nbytes].view(old.dtype).view(old.shape)
        setattr(env, name, new)
        print(name, "offset MB", o, hex(new.data_ptr()), score())


main()
