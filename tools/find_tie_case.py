"""Searches (CPU, oracle only) for Philox-mode episodes in which an 11 x 11 area average that decides a class weight of a feature
plane lands ON a threshold (0.499 / 0.501 up to float64 rounding): the cases tests/test_hip_env_parity.py pins the tie rule on.
    python tools/find_tie_case.py [pixels] [first_seed] [n]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ipp_oracle as O  # noqa: E402
from configs import make_params  # noqa: E402

px = int(sys.argv[1]) if len(sys.argv) > 1 else 17
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
params = make_params("small", sensor__pixel__number_x=px, sensor__pixel__number_y=px)
d = O.Derived(params)
for k in range(n):
    seed, episode = seed0 + k, 11 + 7 * k

    def correctness(i, s, shape):
        pos = ep.agents[i]["position"]
        _, fc = O.project_field_of_view(d, pos)
        return O.philox_correctness(seed, episode, i, s, fc, d.gy, O.noise_of_altitude(pos[2]))

    ep = O.OracleEpisode(params, episode, correctness,
                         lambda i, t, m, o: O.uniform_valid_action(O.philox_action_word(seed, episode, i, t), m),
                         comm_draw=lambda i, j, t: O.philox_comm_draw(seed, episode, i, j, t), build_features=True, exact=True)
    hits = []
    for rec in ep.run():
        for key in ("decide_local", "decide_fp"):
            for i, v in enumerate(rec[key]):
                dist = np.minimum(np.abs(v - 0.499), np.abs(v - 0.501))
                if dist.min() < 1e-12:
                    hits.append((rec["t"], key, i, float(dist.min())))
        dist = np.minimum(np.abs(rec["decide_global"] - 0.499), np.abs(rec["decide_global"] - 0.501))
        if dist.min() < 1e-12:
            hits.append((rec["t"], "decide_global", -1, float(dist.min())))
    print(f"px={px} seed={seed} episode={episode}: {len(hits)} ties {hits[:4]}", flush=True)
