"""TIMING-ONLY probe: how much would the step gain if the global map's fusion ran beside {local fusion -> K3} on a second stream?
(The two touch disjoint maps, but K3 overwrites the code tiles the global fusion reads: a real implementation double-buffers them.
Here the hazard is ignored -- results are NOT checked, only the loop's wall time.  Needs tools/probe/fuse_which.patch: the IPPM_FUSE_WHICH knob of
the tile fusion.)   python tools/overlap_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
sys.path.insert(0, ROOT)
from bench import bench_params  # noqa: E402
from ippmarl import _ffi  # noqa: E402
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM  # noqa: E402


class A:
    envs, agents, grid, actions, terrain, episode_comm_range = 1024, 4, 256, None, "random_field", False


env = VecEnv(bench_params(A), A.envs, philox_seed=3, terrain="random_field", track_area=False)
print("placement", env.tune_placement(12))
T = env.d.budget + 1
ids = list(range(1, A.envs + 1))
side = torch.cuda.Stream()
main = torch.cuda.current_stream()


def step_serial(t):
    env.steps(t, policy=POLICY_UNIFORM, features=False)


def step_overlap(t):
    env._plan_step(t, _ffi.STEP_COMM | _ffi.STEP_GLOBAL | _ffi.STEP_MOVE, None, POLICY_UNIFORM, None, None)
    planned = torch.cuda.Event()
    planned.record(main)
    os.environ["IPPM_FUSE_WHICH"] = "2"
    with torch.cuda.stream(side):
        side.wait_event(planned)
        env.ctx.call("ippm_fuse_step", env._p(env.local), env._p(env.glob), env._p(env.code), env._p(env.ws), env._p(env.sums), None,
                     env._p(env.work), env.E, side.cuda_stream)
        done = torch.cuda.Event()
        done.record(side)
    os.environ["IPPM_FUSE_WHICH"] = "1"
    env._fuse_step()
    os.environ["IPPM_FUSE_WHICH"] = "0"
    main.wait_event(done)          # (K3 completes the reward from the global fusion's sums)
    env._sense(stage=t + 1, close_step=True)
    env.t = t + 1


for name, step in (("serial", step_serial), ("overlap", step_overlap), ("serial", step_serial), ("overlap", step_overlap)):
    env.reset(ids)
    for t in range(T):
        step(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for rep in range(6):
        env.reset(ids)
        for t in range(T):
            step(t)
            n += 1
    torch.cuda.synchronize()
    print(f"{name}: {1e6 * (time.perf_counter() - t0) / n:.1f} us per step (resets included)", flush=True)
