"""Per-instance (XCD x L2-channel) memory-side counters of the map kernels on a good, a bad and a physically contiguous allocation
of the hot planes, in ONE process (round 4, VERDICT r03 item 1a: "read per-channel TCC_EA_RDREQ / WRREQ good vs bad").
Run under rocprofv3 with NON-summed counters (the rocpd database keeps one value per counter instance):
    rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ --kernel-trace -d out -o p -- python tools/channel_pmc.py
    python tools/channel_pmc.py --read out/**/p_results.db
The episodes are announced by k_stream_copy launches (1 before the best plain draw, 2 before the worst plain draw, 3 before the
contiguous arena) so that the dispatches can be told apart."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
sys.path.insert(0, ROOT)


def read(path):
    import sqlite3

    import numpy as np
    import pandas as pd
    db = sqlite3.connect(path)
    df = pd.read_sql("select name, dispatch_id, duration, counter_name, counter_value from pmc_events", db)
    d = df.drop_duplicates("dispatch_id").sort_values("dispatch_id")[["dispatch_id", "name"]].reset_index(drop=True)
    marks = d.index[d["name"].str.contains("k_stream_copy")].tolist()
    # ... copy | best | copy copy | worst | copy copy copy | contiguous
    runs, k = [], 0          # (copies in a row, index of the first, index of the last)
    while k < len(marks):
        j = k
        while j + 1 < len(marks) and marks[j + 1] == marks[j] + 1:
            j += 1
        runs.append((j - k + 1, marks[k], marks[j]))
        k = j + 1
    runs = runs[-3:]
    tags = {1: "best plain", 2: "worst plain", 3: "contiguous"}
    pd.set_option("display.width", 250)
    for i, (n, _, last) in enumerate(runs):
        end = runs[i + 1][1] if i + 1 < len(runs) else len(d)
        ids = set(d["dispatch_id"][last + 1:end])
        s = df[df["dispatch_id"].isin(ids) & df["name"].str.contains("k_fuse_tiles|k_sense_tiles|k_reset_maps")]
        s = s.assign(k=s["name"].str.extract(r"(k_\w+)")[0])
        print(f"--- {tags.get(n, n)}: launches {s.groupby('k')['dispatch_id'].nunique().to_dict()}")
        rows = []
        for (kname, cname), g in s.groupby(["k", "counter_name"]):
            per = g.groupby("dispatch_id")["counter_value"].apply(lambda v: np.asarray(v, dtype=np.float64))
            inst = min(len(v) for v in per)
            m = np.stack([v[:inst] for v in per])          # [launch, instance]
            tot = m.sum(1)
            share = m / np.maximum(tot[:, None], 1)
            rows.append(dict(kernel=kname, counter=cname, instances=inst, per_launch=tot.mean(),
                             max_over_mean=(m.max(1) / np.maximum(m.mean(1), 1e-9)).mean(),
                             min_over_mean=(m.min(1) / np.maximum(m.mean(1), 1e-9)).mean(),
                             cv=(m.std(1) / np.maximum(m.mean(1), 1e-9)).mean(),
                             avg_us=g.drop_duplicates("dispatch_id")["duration"].mean() / 1e3))
        print(pd.DataFrame(rows).round(3).to_string(index=False))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--read":
        return read(sys.argv[2])
    import torch
    from bench import bench_params
    from ippmarl import _ffi
    from ippmarl.vec_env import VecEnv, POLICY_UNIFORM

    lib = C.CDLL(os.path.join(ROOT, "tools", "probe", "libvmm_arena.so"))
    lib.flagged_alloc.restype = C.c_void_p
    lib.flagged_alloc.argtypes = [C.c_size_t, C.c_uint]

    class Raw:
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    class A:
        envs, agents, grid, actions, terrain = 1024, 4, 256, None, "random_field"

    env = VecEnv(bench_params(A), 1024, philox_seed=3, terrain="random_field", track_area=False)
    T = env.d.budget + 1
    ids = list(range(1, 1025))
    nbytes = env._arena.numel()

    def episode(timed=False):
        env._boxes_valid = False
        env.reset(ids)
        env.profile = timed
        for t in range(T):
            env.steps(t, policy=POLICY_UNIFORM, features=False)
        env.reset(ids)
        env.profile = False

    arenas, scores = [], []
    for k in range(int(os.environ.get("DRAWS", "8"))):
        if k:
            env._place_hot(slack_mb=66 * k)
        arenas.append(env._arena)
        episode()
        episode(True)
        tm = env.event_times_us()
        scores.append(tm["sense"]["avg_us"] + tm["fuse"]["avg_us"])
    print("scores", [round(x, 1) for x in scores], file=sys.stderr)
    ptr = lib.flagged_alloc(nbytes, 4)
    contig = torch.as_tensor(Raw(ptr, nbytes), device="cuda")
    scratch = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")

    def mark(n):
        for _ in range(n):
            env.ctx.call("ippm_stream_copy", _ffi.ptr(scratch), _ffi.ptr(scratch[32 << 20:]), 32 << 20, env.stream)

    for arena, n in ((arenas[scores.index(min(scores))], 1), (arenas[scores.index(max(scores))], 2), (contig, 3)):
        env._use_arena(arena)
        episode()
        torch.cuda.synchronize()
        mark(n)
        episode()
        torch.cuda.synchronize()


main()
