"""Layout experiment (round 4): does a non-power-of-two distance between maps remove the good / bad kinds of
allocation?  The maps of config 2 sit at exact power-of-two strides (256 KiB per map, 1 MiB per env); physically contiguous memory
(hipDeviceMallocContiguous) is deterministically the slow kind.  Here every map gets `skew` floats of padding behind it
(IPPM_MAP_SKEW, honoured by K3's tile form, the tile fusion and k_reset_maps only) and one episode is timed on a contiguous and on
plain allocations of the arena.
    python tools/layout_skew.py [skews in floats, comma separated] [plain draws]
Prints (K3, fusion, reset_maps) avg us per launch of one timed episode."""
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
    return tuple(round(tm[k]["avg_us"], 1) if k in tm else None for k in ("sense", "fuse", "reset_maps"))


def main():
    skews = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,64,1024,16448").split(",")]
    draws = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    ids = list(range(1, A.envs + 1))
    for skew in skews:
        os.environ["IPPM_MAP_SKEW"] = str(skew)
        env = VecEnv(bench_params(A), A.envs, philox_seed=3, terrain="random_field", track_area=False)
        T = env.d.budget + 1
        nbytes = env._arena.numel()
        score(env, ids, T)
        out = [("torch", score(env, ids, T))]
        held = []
        for flags, name in [(4, "contig"), (4, "contig")] + [(0, "plain")] * draws:
            ptr = lib.flagged_alloc(nbytes, flags)
            if not ptr:
                out.append((name, "failed"))
                continue
            held.append(ptr)
            env._use_arena(torch.as_tensor(Raw(ptr, nbytes), device="cuda"))
            out.append((name, score(env, ids, T)))
        print(f"skew {skew:6d} floats ({skew * 4} B; map pitch {(256 * 256 + skew) * 4} B):", " ".join(f"{n}{v}" for n, v in out), flush=True)
        torch.cuda.synchronize()
        del env
        for p in held:
            lib.flagged_free(p)
        torch.cuda.empty_cache()


main()
