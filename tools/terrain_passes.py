"""us of each terrain pass (1024 fields of 256^2 by default), timed with events around repeated launches of ippm_terrain_field's two
kernels and ippm_terrain_pack: python tools/terrain_passes.py [envs]   (IPPMARL_LIB selects a variant library)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
sys.path.insert(0, ROOT)
from bench import bench_params  # noqa: E402
from ippmarl.vec_env import VecEnv  # noqa: E402


class A:
    envs, agents, grid, actions, terrain = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 4, 256, None, "random_field"


env = VecEnv(bench_params(A), A.envs, philox_seed=3, terrain="random_field", track_area=False)
env.reset(list(range(1, A.envs + 1)))
gen = env._terrain()
for _ in range(3):
    gen.generate(env.episode, env.truth, env.stream)
env.event_times_us()
env.profile = True
for _ in range(10):
    gen.generate(env.episode, env.truth, env.stream)
env.profile = False
import ctypes as C  # noqa: E402
from ippmarl import _ffi  # noqa: E402
# the class "terrain" lumps the three kernels: read the raw event pairs in launch order (x, y, pack, x, y, pack, ...)
ctx = env.ctx
n, tot, mn = C.c_int64(0), C.c_double(0.0), C.c_double(0.0)
tm = env.event_times_us(clear=False)["terrain"]
print(os.path.basename(os.environ.get("IPPMARL_LIB", "libippmarl.so")), "terrain class:", round(tm["avg_us"] * 3, 1), "us per synthesis (3 launches), min launch", round(tm["min_us"], 1))
