"""Can the good / bad kinds of allocation (VecEnv.tune_placement) be produced on purpose?  The env's arena is assembled from
physical chunks of a chosen size mapped in a chosen order (tools/probe/vmm_arena.cpp) and one episode is timed on it.
    python tools/placement_vmm.py"""
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
lib.vmm_arena_create.restype = C.c_void_p
lib.vmm_arena_create.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.c_uint, C.c_size_t, C.POINTER(C.c_void_p)]
lib.vmm_arena_destroy.argtypes = [C.c_void_p]
lib.vmm_arena_create_spread.restype = C.c_void_p
lib.vmm_arena_create_spread.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.c_uint, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]


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
        return round(tm["sense"]["avg_us"], 1), round(tm["fuse"]["avg_us"], 1)

    print("torch allocation", score())
    keep = []
    MB = 1 << 20
    for rep in range(3):
        for chunk, order, spread in ((2 * MB, 0, 1), (2 * MB, 2, 4), (2 * MB, 0, 4), (2 * MB, 2, 16), (8 * MB, 2, 4), (32 * MB, 2, 4), (2 * MB, 2, 1)):
            ptr = C.c_void_p()
            h = lib.vmm_arena_create_spread(nbytes, chunk, order, 17 + rep, 2 * MB, spread, C.byref(ptr))
            if not h:
                print("create failed", chunk, order, spread)
                continue
            arena = torch.as_tensor(Raw(ptr.value, nbytes), device="cuda")
            env._use_arena(arena)
            print(f"rep {rep} chunk {chunk / MB:6.1f} MB order {('created', 'reversed', 'shuffled')[order]:9s} spread {spread:2d}", score(), flush=True)
            keep.append(h)
    env._place_hot()
    print("torch allocation", score())
    torch.cuda.synchronize()


main()
