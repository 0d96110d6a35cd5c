"""Round 5 experiment: the batch as TWO half-batches on two streams, so that the latency-bound plan kernel of one half runs beside the
bandwidth-bound fusion / K3 of the other.  Same total work (1024 envs), episodes in lock step, resets included.
    python tools/two_streams_probe.py [envs] [parts]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ipp-marl_amd"))
sys.path.insert(0, ROOT)
from bench import bench_params  # noqa: E402
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM  # noqa: E402


class A:
    agents, grid, actions, terrain = 4, 256, None, "random_field"


def run(E, parts, steps=240, warm=48, stagger=False):
    envs, streams = [], []
    for k in range(parts):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            env = VecEnv(bench_params(A), E // parts, philox_seed=3, terrain="random_field", track_area=False)
            env.tune_placement(12)
        envs.append(env)
        streams.append(s)
    T = envs[0].d.budget + 1
    wave = [0]

    def reset():
        for k, (env, s) in enumerate(zip(envs, streams)):
            with torch.cuda.stream(s):
                base = 1 + wave[0] * E + k * (E // parts)
                env.reset(list(range(base, base + E // parts)))
        wave[0] += 1

    reset()
    t = 0
    if stagger:     # part k runs k * T / parts steps ahead: the parts' resets fall on different steps
        ts = [0] * parts
        waves = [1] * parts
        for k, (env, s) in enumerate(zip(envs, streams)):
            with torch.cuda.stream(s):
                for _ in range(k * T // parts):
                    env.steps(ts[k], policy=POLICY_UNIFORM, features=False)
                    ts[k] += 1

    def loop(n):
        nonlocal t
        for _ in range(n):
            if stagger:
                for k, (env, s) in enumerate(zip(envs, streams)):
                    with torch.cuda.stream(s):
                        env.steps(ts[k], policy=POLICY_UNIFORM, features=False)
                        ts[k] += 1
                        if ts[k] == T:
                            base = 1 + waves[k] * E + k * (E // parts)
                            env.reset(list(range(base, base + E // parts)))
                            waves[k] += 1
                            ts[k] = 0
                continue
            for env, s in zip(envs, streams):
                with torch.cuda.stream(s):
                    env.steps(t, policy=POLICY_UNIFORM, features=False)
            t += 1
            if t == T:
                reset()
                t = 0

    loop(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{E} envs as {parts} part(s) on {parts} stream(s){', episodes staggered' if stagger else ''}: {1e3 * dt / steps:.4f} ms per step, {E * 4 * steps / dt / 1e6:.2f} M agent-env steps/s", flush=True)


E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for parts, stagger in ([(int(sys.argv[2]), False)] if len(sys.argv) > 2 else [(1, False), (2, False), (2, True), (3, False), (3, True), (2, True), (2, False), (1, False)]):
    run(E if parts != 3 else 1023, parts, stagger=stagger)
