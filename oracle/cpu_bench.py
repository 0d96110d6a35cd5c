#!/usr/bin/env python
"""CPU baseline leg of bench.py: the NumPy oracle stepping the env-only workload on host cores.

TEST INFRASTRUCTURE (see ipp_oracle.py header): only bench.py's ``cpu_baseline`` leg runs this, as the thing timed BESIDE
the GPU path, never as part of it.

    python oracle/cpu_bench.py --envs 8 --seconds 10 --first-episode 1 [--agents 4 --number 30 --terrain random_field]

steps ``--envs`` environments one after the other (episode by episode, one NumPy process = one core) for ``--seconds`` and
prints {"agent_env_steps": n, "seconds": s}.  bench.py starts one such process per core for the many-env figure.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")


def run(params, n_envs: int, seconds: float, first_episode: int, terrain: str, seed: int = 3):
    import ipp_oracle as O

    d = O.Derived(params)
    steps, t0 = 0, time.perf_counter()
    episode = first_episode
    while time.perf_counter() - t0 < seconds:
        for _ in range(n_envs):   # one wave of n_envs episodes, env after env (the reference's own execution model)
            holder = {}

            def correctness(i, s, shape, episode=episode, holder=holder):
                pos = holder["ep"].agents[i]["position"]
                _, fc = O.project_field_of_view(d, pos)
                return O.philox_correctness(seed, episode, i, s, fc, d.gy, O.noise_of_altitude(pos[2]))

            truth = (O.grf_field(d.gx, d.gy, episode, float(params["sensor"]["simulation"]["cluster_radius"]))
                     if terrain == "random_field" else None)
            ep = O.OracleEpisode(params, episode, correctness,
                                 lambda i, t, m, o, episode=episode: O.uniform_valid_action(O.philox_action_word(seed, episode, i, t), m),
                                 build_features=False, truth=truth)
            holder["ep"] = ep
            for t in range(d.budget + 1):
                ep.step(t)
                steps += d.n_agents
                if time.perf_counter() - t0 > seconds:
                    break
            episode += 1
            if time.perf_counter() - t0 > seconds:
                break
    return steps, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--first-episode", type=int, default=1)
    ap.add_argument("--agents", type=int, default=4)
    ap.add_argument("--number", type=int, default=30, help="sensor.pixel.number_x/y: 15/30/60/120 -> 128/256/512/1024 cells")
    ap.add_argument("--terrain", default="random_field")
    args = ap.parse_args()
    from configs import make_params

    params = make_params("c2", experiment__missions__n_agents=args.agents, sensor__pixel__number_x=args.number,
                         sensor__pixel__number_y=args.number)
    steps, dt = run(params, args.envs, args.seconds, args.first_episode, args.terrain)
    print(json.dumps({"agent_env_steps": steps, "seconds": dt}))


if __name__ == "__main__":
    main()
