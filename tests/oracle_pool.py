"""Oracle episodes of a batch, one process per env (TEST INFRASTRUCTURE: the oracle is the checker, never the product).

The parity tests compare a GPU batch of E envs with E independent oracle episodes; at BASELINE config 4 / 5 shapes one oracle
episode takes 6 - 50 s of NumPy on one core, so the episodes of a batch run side by side on the host's cores.  Workers are
spawned (not forked: the parent holds a HIP context) and import only NumPy and the oracle."""
from __future__ import annotations

import os
import sys
from concurrent.futures import ProcessPoolExecutor
import multiprocessing as mp

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle",):
    path = os.path.join(ROOT, sub)
    if path not in sys.path:
        sys.path.insert(0, path)


def philox_episode(params, episode, seed, truth=None):
    """One episode of the oracle under the production randomness (Philox flips / uniform valid actions / comm draws keyed by
    (seed; episode, agent, step), mirrored from the device): (per-step log, final local maps [n, gx, gy], final global map)."""
    import ipp_oracle as O
    d = O.Derived(params)
    holder = {}

    def correctness(i, s, shape):
        ag_pos = holder["ep"].agents[i]["position"]
        _, fc = O.project_field_of_view(d, ag_pos)
        return O.philox_correctness(seed, episode, i, s, fc, d.gy, O.noise_of_altitude(ag_pos[2]))

    def choose(i, t, mask, obs):
        return O.uniform_valid_action(O.philox_action_word(seed, episode, i, t), mask)

    ep = O.OracleEpisode(params, episode, correctness, choose, comm_draw=lambda i, j, t: O.philox_comm_draw(seed, episode, i, j, t),
                         build_features=True, exact=True, truth=truth)
    holder["ep"] = ep
    log = ep.run()
    return log, np.array([a["local_map"] for a in ep.agents]), np.array(ep.global_map)


def _job(args):
    return philox_episode(*args)


def philox_episodes(params, episodes, seed, truths=None, min_parallel_cells=1 << 19):
    """[philox_episode(...)] for every episode of a batch; in worker processes when the grid is large enough to pay for them."""
    truths = [None] * len(episodes) if truths is None else list(truths)
    # (params: one dict for the batch, or one per episode -- mixed team sizes: env e is a run with its own n_agents)
    per_env = list(params) if isinstance(params, (list, tuple)) else [params] * len(episodes)
    jobs = [(pr, int(ep), seed, tr) for pr, ep, tr in zip(per_env, episodes, truths)]
    import ipp_oracle as O
    d = O.Derived(max(per_env, key=lambda pr: pr["experiment"]["missions"]["n_agents"]))
    if len(jobs) < 2 or d.gx * d.gy * d.n_agents < min_parallel_cells:
        return [_job(j) for j in jobs]
    workers = min(len(jobs), max(1, (os.cpu_count() or 2) - 1))
    env_keep = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    os.environ.update(OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")   # inherited by the spawned workers
    try:
        with ProcessPoolExecutor(max_workers=workers, mp_context=mp.get_context("spawn")) as pool:
            return list(pool.map(_job, jobs))
    finally:
        for k, v in env_keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
