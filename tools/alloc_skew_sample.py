"""Round 5: is a map pitch that is not a power of two a reliable way to the fast kind of allocation?

Round 4's layout sweep (profiles/r04/layout_skew.txt) had four draws per setting at a base rate of ~45 % fast; two settings (map
skew 16 floats = 64 B and 16448 floats = 65 792 B) came out 4/4 fast.  This script draws DRAWS fresh hipMalloc arenas in ONE
process for ONE setting and times an episode of config 2 on each with the kernels' own dispatch-bound events:

    make -C ipp-marl_amd/csrc VARIANT=skew16 EXTRA=-DIPPM_MAP_SKEW=16
    IPPMARL_LIB=ipp-marl_amd/lib/libippmarl_skew16.so python tools/alloc_skew_sample.py 16 14

(the library's compile-time IPPM_MAP_SKEW and the first argument must agree: K3's tile form, the tile fusion and k_reset_maps of
that library step maps IPPM_MAP_SKEW floats apart; nothing else of the library honours the pitch, so only the env-only step runs).
Prints one line per draw: (K3, fusion, k_reset_maps) avg us per launch, and the share of draws of the fast kind (K3 + fusion
below the midpoint of the two clusters of config 2, 118.5 us)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("ipp-marl_amd",):
    sys.path.insert(0, os.path.join(ROOT, sub))
sys.path.insert(0, ROOT)
from bench import bench_params  # noqa: E402
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM  # noqa: E402

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]


class Raw:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class A:
    envs, agents, grid, actions, terrain = 1024, 4, 256, None, "random_field"


def score(env, ids, T):
    for timed in (False, True):
        env._boxes_valid = False
        env.reset(ids)
        env.profile = timed
        for t in range(T):
            env.steps(t, policy=POLICY_UNIFORM, features=False)
        env.reset(ids)      # (the timed one: a box-restricted reset)
        env.profile = False
    tm = env.event_times_us()
    return tuple(round(tm[k]["avg_us"], 1) for k in ("sense", "fuse", "reset_maps"))


def main():
    skew = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    draws = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    ids = list(range(1, A.envs + 1))
    env = VecEnv(bench_params(A), A.envs, philox_seed=3, terrain="random_field", track_area=False)
    d, E, N = env.d, env.E, env.d.n_agents
    cells = d.grid_x * d.grid_y + skew
    # the maps `skew` floats apart (the variant library's pitch); code and truth planes as they are
    env._hot_shapes = (("local", (E, N, cells), torch.float32), ("glob", (E, cells), torch.float32)) + tuple(env._hot_shapes[2:])
    env._place_hot()
    T = d.budget + 1
    nbytes = env._arena.numel()
    score(env, ids, T)      # the process's first episodes run slow whatever the allocation
    out = [("torch", score(env, ids, T))]
    held = []
    for k in range(draws):
        ptr = C.c_void_p()
        if hip.hipMalloc(C.byref(ptr), nbytes + (k % 7) * (2 << 20)) != 0 or not ptr.value:
            print("hipMalloc failed at draw", k)
            break
        held.append(ptr)
        arena = torch.as_tensor(Raw(ptr.value, nbytes), device="cuda")
        arena.zero_()
        env._use_arena(arena)
        out.append((f"plain{k}", score(env, ids, T)))
        print(f"skew {skew:6d} floats ({skew * 4} B) draw {k:2d}: K3 {out[-1][1][0]:5.1f}  fusion {out[-1][1][1]:5.1f}  reset_maps {out[-1][1][2]:6.1f}", flush=True)
    plain = [v for n, v in out if n.startswith("plain")]
    fast = sum(1 for v in plain if v[0] + v[1] < 118.5)
    print(f"skew {skew:6d} floats ({skew * 4} B; map pitch {cells * 4} B): torch{out[0][1]}  fast {fast}/{len(plain)} plain draws "
          f"(K3 + fusion < 118.5 us); K3+fusion per draw: {[round(v[0] + v[1], 1) for v in plain]}", flush=True)
    torch.cuda.synchronize()
    del env
    for p in held:
        hip.hipFree(p)


main()
