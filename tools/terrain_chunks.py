"""Terrain synthesis in chunks of envs: with chunks small enough for the work buffer and the field of a chunk to stay in the 256 MB
Infinity Cache, pass Y and the threshold pass read what the pass before them wrote from the cache instead of from HBM.
    python tools/terrain_chunks.py [envs]        -> us per reset of the three terrain kernels, per chunk size"""
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
ids = list(range(1, A.envs + 1))
env.reset(ids)
ref = env.truth.clone()
for chunk in (1024, 512, 256, 128, 64, 32):
    if chunk > A.envs:
        continue
    env._terrain().chunk = chunk
    for _ in range(2):
        env.reset(ids)
    assert torch.equal(env.truth, ref)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        env._terrain().generate(env.episode, env.truth, env.stream)
    ev[1].record()
    torch.cuda.synchronize()
    env.profile = True
    env._terrain().generate(env.episode, env.truth, env.stream)
    env.profile = False
    tm = env.event_times_us()["terrain"]
    print(f"chunk {chunk:5d} envs: {ev[0].elapsed_time(ev[1]) * 1e3 / 5:7.1f} us per synthesis of {A.envs} fields (stream time); "
          f"kernel time {tm['avg_us'] * tm['launches']:7.1f} us in {tm['launches']} launches", flush=True)
