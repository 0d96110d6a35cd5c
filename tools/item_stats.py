"""What the tile fusion's work list looks like at a shape: items per env, and per number of ops that meet an item (na) the share
of items, of lane-loads, and how full the items are (lane-loads / (64 x slots)).  Reads the list the plan kernel wrote.
    python tools/item_stats.py [--envs 64 --agents 16 --grid 1024 --actions 27 --episode-comm-range] [--steps 10]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ipp-marl_amd"))
sys.path.insert(0, ROOT)
from bench import bench_params  # noqa: E402
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM  # noqa: E402


def slots_of(na):
    return 4 if na <= 4 else (2 if na <= 10 else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--agents", type=int, default=4)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--actions", type=int, default=None)
    ap.add_argument("--episode-comm-range", action="store_true")
    ap.add_argument("--comm-range", type=float, default=None)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--format", default=os.environ.get("IPPM_ITEM_FORMAT", "auto"), help="rows (round 4: x0 | rows << 16) or runs (round 5: start | count << 16)")
    a = ap.parse_args()
    a.terrain = "split"
    env = VecEnv(bench_params(a), a.envs, philox_seed=3, terrain="split", track_area=False)
    env.reset(list(range(1, a.envs + 1)))
    E, N = env.E, env.d.n_agents
    ops, G = N + 1, (env.d.grid_y + 3) // 4
    cap_items = None
    agg = {}
    per_env = []
    for t in range(a.steps):
        env.steps(t, policy=POLICY_UNIFORM, features=False)
        w = env.work.cpu().numpy()
        base = (E + 3) & ~3
        if cap_items is None:
            cap_items = (len(w) - base) // (4 * E)
        for e in range(E):
            n = int(w[e]) & 0x0FFFFFFF
            per_env.append(n)
            it = w[base + e * cap_items * 4: base + (e * cap_items + n) * 4].reshape(-1, 4).astype(np.int64)
            mask = it[:, 3] & 0x00FFFFFF
            na = np.array([bin(int(m)).count("1") for m in mask])
            slot = (it[:, 3] >> 24) & 0xFF
            W = (it[:, 2] >> 16) & 0xFFFF
            if a.format == "runs" or (a.format == "auto" and env.ctx.lib.ippm_version() >= 500):
                loads = (it[:, 0] >> 16) & 0xFFFF
            else:
                loads = ((it[:, 1] >> 16) & 0xFFFF) * W
            for k in np.unique(na):
                sel = na == k
                s = agg.setdefault(int(k), [0, 0, 0, 0])
                s[0] += int(sel.sum())
                s[1] += int(loads[sel].sum())
                s[2] += int(sel.sum()) * 64 * slots_of(int(k))
                s[3] += int((slot[sel] == N).sum())
    per_env = np.array(per_env)
    print(f"items per env and step: mean {per_env.mean():.0f}  min {per_env.min()}  max {per_env.max()}  (list capacity {cap_items})")
    tot_items = sum(v[0] for v in agg.values())
    tot_loads = sum(v[1] for v in agg.values())
    print(" na   items%  lane-loads%  fill   global-map items%   mean lane-loads per item")
    for k in sorted(agg):
        n, l, c, g = agg[k]
        print(f"{k:3d}  {100 * n / tot_items:6.1f}  {100 * l / tot_loads:10.1f}  {l / c:5.2f}  {100 * g / n:8.1f}  {l / n:10.1f}")
    print(f"all: fill {tot_loads / sum(v[2] for v in agg.values()):.3f}; {tot_loads / len(per_env) * 16 * 4 / 1e6:.1f} MB of map cells per env and step")


main()
