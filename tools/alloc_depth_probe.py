"""Round 5: does the kind of an allocation (fast / slow for the env's map kernels) depend on how much device memory
is already held?  tools/alloc_skew_sample.py saw, in three processes in a row, the first 9-10 fresh hipMalloc arenas slow and the next
ones fast.  Here: a ballast of B GB (one hipMalloc, or chunks) is taken first, then DRAWS arenas; (K3 + fusion) us per step of config 2.
    python tools/alloc_depth_probe.py "0,4,8,12,16,24,48" 4 [chunk_gb]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ipp-marl_amd"))
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


def malloc(n):
    p = C.c_void_p()
    if hip.hipMalloc(C.byref(p), n) != 0 or not p.value:
        raise SystemExit(f"hipMalloc({n}) failed")
    return p


def score(env, ids, T):
    for timed in (False, True):
        env._boxes_valid = False
        env.reset(ids)
        env.profile = timed
        for t in range(T):
            env.steps(t, policy=POLICY_UNIFORM, features=False)
        env.profile = False
    tm = env.event_times_us()
    return round(tm["sense"]["avg_us"] + tm["fuse"]["avg_us"], 1)


def main():
    ballasts = [float(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,4,8,12,16,24,48").split(",")]
    draws = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    chunk = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    ids = list(range(1, A.envs + 1))
    env = VecEnv(bench_params(A), A.envs, philox_seed=3, terrain="random_field", track_area=False)
    T = env.d.budget + 1
    nbytes = env._arena.numel()
    score(env, ids, T)
    print(f"torch's own arena: {score(env, ids, T)} us", flush=True)
    held_total = 0.0
    for gb in ballasts:
        ballast = []
        if gb > 0:
            if chunk > 0:
                k = 0
                while k * chunk < gb:
                    ballast.append(malloc(int(chunk * (1 << 30))))
                    k += 1
            else:
                ballast.append(malloc(int(gb * (1 << 30))))
        arenas, out = [], []
        for k in range(draws):
            p = malloc(nbytes + (k % 5) * (2 << 20))
            arenas.append(p)
            a = torch.as_tensor(Raw(p.value, nbytes), device="cuda")
            a.zero_()
            env._use_arena(a)
            out.append(score(env, ids, T))
        print(f"ballast {gb:5.1f} GB ({'one block' if chunk <= 0 else f'{chunk} GB chunks'}): {out}", flush=True)
        torch.cuda.synchronize()
        env._place_hot()     # back on a torch block before the probes are freed
        for p in arenas + ballast:
            hip.hipFree(p)


main()
