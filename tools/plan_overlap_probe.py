"""Round 5 TIMING-ONLY probe (results are not checked: the deferred-clamp flags race): what would the step cost if the plan kernel of
step t + 1 ran beside K3 of step t -- within one stream's sequence, and on top of two sub-batches on two streams?
    stream A:  fuse(t) -> K3(t) ................ -> fuse(t+1) -> K3(t+1)
    stream B:            plan(t+1) [after fuse(t)] ---^
    python tools/plan_overlap_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ipp-marl_amd"))
sys.path.insert(0, ROOT)
from bench import bench_params  # noqa: E402
from ippmarl import _ffi  # noqa: E402
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM  # noqa: E402


class A:
    agents, grid, actions, terrain = 4, 256, None, "random_field"


FLAGS = _ffi.STEP_COMM | _ffi.STEP_GLOBAL | _ffi.STEP_MOVE


def run(E, parts, overlap, steps=240, warm=48):
    envs, main, side = [], [], []
    for k in range(parts):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            env = VecEnv(bench_params(A), E // parts, philox_seed=3, terrain="random_field", track_area=False)
            env.tune_placement(12)
        envs.append(env); main.append(s); side.append(torch.cuda.Stream())
    T = envs[0].d.budget + 1
    wave = [0]

    def reset():
        for k, env in enumerate(envs):
            with torch.cuda.stream(main[k]):
                base = 1 + wave[0] * E + k * (E // parts)
                env.reset(list(range(base, base + E // parts)))
        wave[0] += 1

    reset()
    t = [0]

    def loop(n):
        for _ in range(n):
            for k, env in enumerate(envs):
                if not overlap:
                    with torch.cuda.stream(main[k]):
                        env.steps(t[0], policy=POLICY_UNIFORM, features=False)
                    continue
                # the plan of this step was launched on the side stream while the previous step's K3 ran (first step: now)
                with torch.cuda.stream(main[k]):
                    if getattr(env, "_planned", None) is None:
                        env._plan_step(t[0], FLAGS, None, POLICY_UNIFORM, None, None)
                    else:
                        main[k].wait_event(env._planned)
                    env._fuse_step()
                    fused = torch.cuda.Event(); fused.record(main[k])
                    env._sense(stage=t[0] + 1, close_step=True)
                nxt = t[0] + 1
                if nxt < T:
                    with torch.cuda.stream(side[k]):
                        side[k].wait_event(fused)
                        env._plan_step(nxt, FLAGS, None, POLICY_UNIFORM, None, None)      # beside K3 of this step (RACE: timing only)
                        env._planned = torch.cuda.Event(); env._planned.record(side[k])
                else:
                    env._planned = None
            t[0] += 1
            if t[0] == T:
                for k in range(parts):
                    main[k].wait_stream(side[k])
                reset()
                t[0] = 0

    loop(warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{E} envs, {parts} stream pair(s), plan(t+1) beside K3(t): {overlap}: {1e3 * dt / steps:.4f} ms per step", flush=True)


for parts, ov in [(1, False), (1, True), (2, False), (2, True), (1, True), (2, True), (1, False), (2, False)]:
    run(1024, parts, ov)
