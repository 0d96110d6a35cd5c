"""Round 5: does the map pitch decide the tile fusion's speed on config 5's shape?  At 1024 x 1024 a map is exactly 4 MiB, so cell
(r, c) of every map of a batch sits at the same offset modulo 4 MiB; a region a message covers is fused into up to 16 local maps of
one env at those same offsets.  This script steps a batch of config 5's shape with the maps `skew` floats apart:

    make -C ipp-marl_amd/csrc VARIANT=skew1088 EXTRA=-DIPPM_MAP_SKEW=1088
    IPPMARL_LIB=ipp-marl_amd/lib/libippmarl_skew1088.so python tools/c5_pitch_probe.py 1088 256 2,4,8,16

(the library's compile-time IPPM_MAP_SKEW and the first argument must agree, as in tools/alloc_skew_sample.py).  Prints K3, fusion
and k_reset_maps avg us per launch (the kernels' own dispatch-bound events) for a few episodes, whole batch per launch, one stream."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ipp-marl_amd"))
sys.path.insert(0, ROOT)
from bench import bench_params  # noqa: E402
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM  # noqa: E402


class A:
    envs, agents, grid, actions, terrain, episode_comm_range, comm_range = 64, 16, 1024, 27, "random_field", True, None


def score(env, ids, T):
    for timed in (False, True):
        env._boxes_valid = False
        env.reset(ids)
        env.profile = timed
        for t in range(T):
            env.steps(t, policy=POLICY_UNIFORM, features=False)
        env.reset(ids)
        env.profile = False
    tm = env.event_times_us()
    return tuple(round(tm[k]["avg_us"], 1) for k in ("sense", "fuse", "reset_maps", "plan"))


def main():
    skew = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    A.envs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    pattern = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 and sys.argv[3] != "-" else None
    if len(sys.argv) > 4:
        A.episode_comm_range, A.comm_range = False, float(sys.argv[4])
    teams = [pattern[e % len(pattern)] for e in range(A.envs)] if pattern else None
    env = VecEnv(bench_params(A), A.envs, philox_seed=3, terrain="random_field", track_area=False, team_sizes=teams)
    d, E, N = env.d, env.E, env.d.n_agents
    cells = d.grid_x * d.grid_y + skew
    if skew:
        env._hot_shapes = (("local", (E, N, cells), torch.float32), ("glob", (E, cells), torch.float32)) + tuple(env._hot_shapes[2:])
        env._place_hot()
    T = d.budget + 1
    for rep in range(4):
        ids = list(range(1 + rep * E, 1 + (rep + 1) * E))
        k3, fuse, rst, plan = score(env, ids, T)
        print(f"skew {skew:6d} floats, {E} envs, teams {pattern}, range {A.comm_range}, waves {os.environ.get('IPPM_TILE_WAVES', '-')}: "
              f"K3 {k3:7.1f}  fusion {fuse:7.1f}  reset_maps {rst:7.1f}  plan {plan:6.1f}", flush=True)


if __name__ == "__main__":
    main()
