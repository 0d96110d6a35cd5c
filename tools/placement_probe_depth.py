"""Does the quality of an allocation depend on how deep into device memory it lies?  Ballast allocations are HELD, so that later
draws of the env's arena come from further down (KFD hands out device memory top-down)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
sys.path.insert(0, ROOT)
os.environ["IPPM_PLACEMENT_NO_EARLY"] = "1"
from bench import bench_params  # noqa: E402
from ippmarl.vec_env import VecEnv  # noqa: E402


class A:
    envs, agents, grid, actions, terrain = 1024, 4, 256, None, "random_field"


def main():
    env = VecEnv(bench_params(A), 1024, philox_seed=3, terrain="random_field", track_area=False)
    ballast, held = [], 0
    step = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    print("free/total GB", [round(x / 2 ** 30, 1) for x in torch.cuda.mem_get_info()])
    while True:
        r = env.tune_placement(4)
        print(f"ballast {held:4d} GB:", r["map_kernels_us_per_step"])
        free = torch.cuda.mem_get_info()[0] >> 30
        if free < step + 16:
            break
        ballast.append(torch.empty(step << 30, dtype=torch.uint8, device="cuda"))
        held += step


main()
