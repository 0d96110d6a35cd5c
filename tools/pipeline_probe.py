"""G independent env groups, each stepping serially on its own stream (no cross-queue joins inside a step), launched
round-robin from one host thread: do the idle phases of one group hide behind the big kernels of the others?"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import torch
from ippmarl.params import grid256_params
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM

TOTAL = 1024
for G in (1, 2, 4):
    for overlap in (True, False):
        envs = [VecEnv(grid256_params(), TOTAL // G, terrain="random_field") for _ in range(G)]
        streams = [torch.cuda.Stream() for _ in range(G)]
        T = envs[0].d.budget + 1
        for g, env in enumerate(envs):
            env.overlap = overlap
            with torch.cuda.stream(streams[g]):
                env.reset(torch.arange(1, env.E + 1) + g * env.E)
                env.capture_step_graphs(POLICY_UNIFORM)
        torch.cuda.synchronize()
        wave = 1

        def run(steps):
            global wave
            for s in range(steps):
                t = s % T
                for g, env in enumerate(envs):
                    with torch.cuda.stream(streams[g]):
                        if t == 0 and s > 0:
                            env.reset(torch.arange(1, env.E + 1) + g * env.E + wave * TOTAL)
                        env.step_graphed(t)
                if t == T - 1:
                    wave += 1
        run(30)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(150)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"groups": G, "side_stream_k5": overlap, "us_per_step": round(dt / 150 * 1e6, 1),
                          "Msteps_per_s": round(TOTAL * 4 * 150 / dt / 1e6, 2)}))
        del envs
        torch.cuda.empty_cache()
