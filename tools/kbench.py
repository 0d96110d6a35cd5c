#!/usr/bin/env python
"""Per-kernel timing of the env-step kernels on a realistic mid-episode state (HIP events on the launch stream).

    python tools/kbench.py [--envs 1024] [--reps 20]

Prints one line per kernel: average microseconds, algorithmic bytes per launch, achieved GB/s.  Used to choose
launch geometry (IPPM_SPLIT_K3/K4/K5) and to compare kernel variants; not part of the product path."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import torch  # noqa: E402

from ippmarl.params import grid256_params  # noqa: E402
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM  # noqa: E402


def timed(fn, reps):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--tag", default="")
    args = ap.parse_args()
    env = VecEnv(grid256_params(), args.envs)
    E = env.E
    out = {"tag": args.tag, "env": {k: v for k, v in os.environ.items() if k.startswith("IPPM_")}}
    # run a few full steps per t, timing each kernel at every step of one episode
    acc = {}
    env.reset(torch.arange(1, E + 1))
    for t in range(env.d.budget + 1):
        snap = {k: getattr(env, k).clone() for k in ("local", "glob", "ws", "sums", "pos", "rect", "code", "comm")}

        def restore():
            for k, v in snap.items():
                getattr(env, k).copy_(v)

        def k_comm():
            env.ctx.call("ippm_comm_matrix", env._p(env.episode), env._p(env.pos), env._p(env.comm_range), None, env._p(env.comm),
                         t, E, env.stream)

        def k4():
            env.ctx.call("ippm_fuse_local", env._p(env.local), env._p(env.code), env._p(env.rect), env._p(env.pos), env._p(env.comm),
                         env._p(env.ws), -1, E, env.stream)

        def k5():
            env.ctx.call("ippm_fuse_global_reward", env._p(env.glob), env._p(env.code), env._p(env.rect), env._p(env.pos),
                         env._p(env.ws), env._p(env.sums), env._p(env.reward), E, env.stream)

        def k3():
            env.sense(stage=t + 1)

        # back-to-back repetitions inside one event pair: host launch gaps are hidden behind queued work.  Repeating K4/K5
        # re-fuses the same measurements (values saturate, the work per launch stays the same).
        for name, fn in (("K4_fuse_local", k4), ("K5_fuse_global", k5), ("K3_sense", k3)):
            restore()
            k_comm()
            fn()
            torch.cuda.synchronize()
            env.counters(reset=True)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) * 1e3 / args.reps
            c = env.counters()
            cells = {"K4_fuse_local": (c["fuse_local_cells"], c["fuse_local_ops"]), "K5_fuse_global": (c["fuse_global_cells"], c["fuse_global_ops"]),
                     "K3_sense": (c["sense_cells"], 0)}[name]
            bytes_ = (cells[0] * (10 if name == "K3_sense" else 8) + cells[1]) / args.reps
            acc.setdefault(name, []).append((us, bytes_))
        restore()
        env.build_observations(t, features=False)
        env.steps(t, policy=POLICY_UNIFORM, features=False)
    for name, rows in acc.items():
        us = sum(r[0] for r in rows) / len(rows)
        by = sum(r[1] for r in rows) / len(rows)
        out[name] = {"avg_us": round(us, 1), "MB": round(by / 1e6, 1), "GBps": round(by / us / 1e3, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
