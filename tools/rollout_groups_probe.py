#!/usr/bin/env python
"""Does a learned-policy rollout overlap when the envs are split into G groups, each stepping on its own stream (one group's
actor forward pass beside another group's env kernels)?  python tools/rollout_groups_probe.py [ENVS=1024]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ipp-marl_amd"))
import torch  # noqa: E402

from ippmarl.networks import ActorNetwork  # noqa: E402
from ippmarl.params import grid256_params  # noqa: E402
from ippmarl.vec_env import VecEnv, POLICY_SAMPLE  # noqa: E402

E = int(os.environ.get("ENVS", 1024))
params = grid256_params()
actor = ActorNetwork(params).to("cuda:0")


def rollout(envs, streams, first):
    main = torch.cuda.current_stream()
    T = envs[0].d.budget + 1
    lo = first
    for env, s in zip(envs, streams):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            env.reset(torch.arange(lo, lo + env.E))
        lo += env.E
    for t in range(T):
        for env, s in zip(envs, streams):
            with torch.cuda.stream(s):
                obs = env.build_observations(t)
                with torch.no_grad():
                    probs, _ = actor(obs.view(env.E * env.d.n_agents, 11, 11, 7), 0.1)
                env.steps(t, policy=POLICY_SAMPLE, probs=probs.view(env.E, env.d.n_agents, -1))
    for s in streams:
        main.wait_stream(s)


for G in (1, 2, 4):
    envs = [VecEnv(params, E // G, terrain="random_field") for _ in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)] if G > 1 else [torch.cuda.current_stream()]
    rollout(envs, streams, 1)
    rollout(envs, streams, 5000)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(3):
        rollout(envs, streams, 10000 + r * 5000)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"groups {G}: rollout {dt * 1e3:.1f} ms = {E * 4 * 15 / dt / 1e6:.2f} M agent-env steps/s")
    del envs
    torch.cuda.empty_cache()
