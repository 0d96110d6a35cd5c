"""Is the run-to-run spread of the fusion kernel (81 vs 86 us at config 2) a property of the process (clocks) or of where the
env's buffers landed?  Several VecEnv instances in ONE process, stepped alternately with kernel timing on; prints per instance
the kernels' average durations and the device addresses of the big buffers.
    python tools/placement_probe.py [instances] [rounds]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
sys.path.insert(0, ROOT)
from bench import bench_params  # noqa: E402
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM  # noqa: E402


class A:
    envs, agents, grid, actions, terrain = 1024, 4, 256, None, "random_field"


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    params = bench_params(A)
    envs = []
    pad = []
    for i in range(k):
        envs.append(VecEnv(params, 1024, philox_seed=3, terrain="random_field", track_area=False))
        pad.append(torch.empty((1 + 3 * i) * 1024 * 1024 + 4096 * i, dtype=torch.uint8, device="cuda"))   # shifts the next instance
    for i, env in enumerate(envs):
        print(i, {n: hex(getattr(env, n).data_ptr()) for n in ("local", "glob", "code", "truth", "ws", "work")})
    T = envs[0].d.budget + 1
    from ippmarl import _ffi
    scratch = torch.empty_like(envs[0].local)

    def copy_rate(env, t):   # GB/s of a 16 B / lane streaming copy out of tensor t
        n = t.numel() * t.element_size()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        env.ctx.call("ippm_stream_copy", env._p(t), _ffi.ptr(scratch), n, env.stream)
        a.record()
        for _ in range(5):
            env.ctx.call("ippm_stream_copy", env._p(t), _ffi.ptr(scratch), n, env.stream)
        b.record()
        torch.cuda.synchronize()
        return round(5 * 2 * n / (a.elapsed_time(b) * 1e-3) / 1e9)

    for i, env in enumerate(envs):
        print(i, "stream copy GB/s out of local / glob:", copy_rate(env, env.local), copy_rate(env, env.glob))
    last = {}
    for r in range(rounds):
        for i, env in enumerate(envs):
            env.reset(list(range(1 + r * 1024, 1 + (r + 1) * 1024)))
            for t in range(T):   # warm
                env.steps(t, policy=POLICY_UNIFORM, features=False)
            env.reset(list(range(1 + r * 1024, 1 + (r + 1) * 1024)))
            env.profile = True
            for rep in range(3):
                for t in range(T):
                    env.steps(t, policy=POLICY_UNIFORM, features=False)
                env.reset(list(range(1 + r * 1024, 1 + (r + 1) * 1024)))
            env.profile = False
            tm = env.event_times_us()
            print(f"round {r} instance {i}:", {c: (round(v["avg_us"], 1), round(v["min_us"], 1)) for c, v in tm.items() if c in ("sense", "fuse", "plan")})
            last[i] = tm["fuse"]["avg_us"]
    # the slowest instance: move one buffer at a time to a fresh allocation and time again
    slow = max(last, key=last.get)
    fast = min(last.values())
    print("slowest", slow, round(last[slow], 1), "fastest", round(fast, 1))
    env = envs[slow]

    def measure(tag):
        ids = list(range(1, 1025))
        env.reset(ids)
        env.profile = True
        for rep in range(3):
            for t in range(T):
                env.steps(t, policy=POLICY_UNIFORM, features=False)
            env.reset(ids)
        env.profile = False
        tm = env.event_times_us()
        print(f"  {tag}:", {c: (round(v["avg_us"], 1), round(v["min_us"], 1)) for c, v in tm.items() if c in ("sense", "fuse", "plan")})

    measure("as is")
    for name in ("local", "glob", "code", "truth", "ws", "work"):
        old = getattr(env, name)
        hold = torch.empty(5 * 1024 * 1024 + 8192, dtype=torch.uint8, device="cuda")   # keeps the allocator from handing the old block back
        new = torch.empty_like(old)
        new.copy_(old)
        setattr(env, name, new)
        measure(f"{name} moved {hex(old.data_ptr())} -> {hex(new.data_ptr())}")
        pad.append(old)
        pad.append(hold)


main()
