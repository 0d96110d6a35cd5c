"""Counters of the map kernels on a good and on a bad allocation of the env's hot planes, in one process (run under
rocprofv3 --pmc ... --kernel-trace): a search without early exit keeps every candidate, then one episode runs on the best and
one on the worst, each announced by k_stream_copy launches (1 before the best, 2 before the worst) so that the dispatches can
be told apart in the counter table.
    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d out -o p -- python tools/placement_pmc.py
    python tools/placement_pmc.py --read out/**/p_counter_collection.csv"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
sys.path.insert(0, ROOT)


def read(path):
    import pandas as pd
    c = pd.read_csv(path).sort_values("Dispatch_Id")
    d = c.drop_duplicates("Dispatch_Id")[["Dispatch_Id", "Kernel_Name"]].reset_index(drop=True)
    marks = d.index[d["Kernel_Name"].str.contains("k_stream_copy")].tolist()
    # ... copy, [best episode], copy, copy, [worst episode]
    if len(marks) < 3:
        print("markers not found", len(marks))
        return
    m1, m2, m3 = marks[-3], marks[-2], marks[-1]
    seg = {"best": set(d["Dispatch_Id"][m1 + 1:m2]), "worst": set(d["Dispatch_Id"][m3 + 1:])}
    for tag, ids in seg.items():
        s = c[c["Dispatch_Id"].isin(ids)]
        s = s[s["Kernel_Name"].str.contains("k_fuse_tiles|k_sense_tiles")]
        s = s.assign(k=s["Kernel_Name"].str.extract(r"(k_\w+)")[0])
        g = s.groupby(["k", "Counter_Name"])["Counter_Value"].sum() / s.groupby(["k", "Counter_Name"])["Dispatch_Id"].nunique()
        print(tag, "launches", s.groupby("k")["Dispatch_Id"].nunique().to_dict())
        print(g.unstack().round(0).to_string())


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--read":
        return read(sys.argv[2])
    import torch
    os.environ["IPPM_PLACEMENT_NO_EARLY"] = "1"
    from bench import bench_params
    from ippmarl import _ffi
    from ippmarl.vec_env import VecEnv, POLICY_UNIFORM

    class A:
        envs, agents, grid, actions, terrain = 1024, 4, 256, None, "random_field"

    env = VecEnv(bench_params(A), 1024, philox_seed=3, terrain="random_field", track_area=False)
    T = env.d.budget + 1
    ids = list(range(1, 1025))
    arenas, scores = [], []
    for k in range(8):
        if k:
            env._place_hot(slack_mb=66 * k)
        arenas.append(env._arena)
        env.reset(ids)
        for t in range(T):
            env.steps(t, policy=POLICY_UNIFORM, features=False)
        env.reset(ids)
        env.profile = True
        for t in range(T):
            env.steps(t, policy=POLICY_UNIFORM, features=False)
        env.profile = False
        tm = env.event_times_us()
        scores.append(tm["sense"]["avg_us"] + tm["fuse"]["avg_us"])
    print("scores", [round(x, 1) for x in scores], file=sys.stderr)
    scratch = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")

    def mark(n):
        for _ in range(n):
            env.ctx.call("ippm_stream_copy", _ffi.ptr(scratch), _ffi.ptr(scratch[32 << 20:]), 32 << 20, env.stream)

    for which, n in ((min, 1), (max, 2)):
        i = scores.index(which(scores))
        env._use_arena(arenas[i])
        env.reset(ids)
        for t in range(T):
            env.steps(t, policy=POLICY_UNIFORM, features=False)
        env.reset(ids)
        torch.cuda.synchronize()
        mark(n)
        for t in range(T):
            env.steps(t, policy=POLICY_UNIFORM, features=False)
        torch.cuda.synchronize()


main()
