"""The env's arena as physically contiguous device memory (hipExtMallocWithFlags + hipDeviceMallocContiguous) against plain
allocations: (K3, fusion) us of one timed episode per allocation.
    python tools/placement_contig.py [rounds]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
sys.path.insert(0, ROOT)
from bench import bench_params  # noqa: E402
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "tools", "probe", "libvmm_arena.so"))
lib.flagged_alloc.restype = C.c_void_p
lib.flagged_alloc.argtypes = [C.c_size_t, C.c_uint]
lib.flagged_free.argtypes = [C.c_void_p]


class Raw:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class A:
    envs, agents, grid, actions, terrain = 1024, 4, 256, None, "random_field"


def main():
    env = VecEnv(bench_params(A), 1024, philox_seed=3, terrain="random_field", track_area=False)
    T = env.d.budget + 1
    ids = list(range(1, 1025))
    nbytes = env._arena.numel()

    def score():
        env.reset(ids)
        for t in range(T):
            env.steps(t, policy=POLICY_UNIFORM, features=False)
        env.reset(ids)
        env.profile = True
        for t in range(T):
            env.steps(t, policy=POLICY_UNIFORM, features=False)
        env.profile = False
        tm = env.event_times_us()
        return round(tm["sense"]["avg_us"], 1), round(tm["fuse"]["avg_us"], 1), round(tm["reset_maps"]["avg_us"], 1) if "reset_maps" in tm else None

    score()
    print("torch allocation", score())
    held = []
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
        for flags, name, size in ((4, "contiguous", nbytes), (0, "plain", nbytes), (4, "contiguous 2 GB", 2 << 30), (4, "contiguous", nbytes)):
            ptr = lib.flagged_alloc(size, flags)
            if not ptr:
                print(name, "failed")
                continue
            env._use_arena(torch.as_tensor(Raw(ptr, nbytes), device="cuda"))
            print(f"rep {rep} {name:16s} at {ptr:#x}", score(), flush=True)
            held.append(ptr)
    torch.cuda.synchronize()


main()
