"""A/B of library knobs (environment variables read at ippm_ctx_create) or variant libraries on ONE allocation of the hot planes:
every setting gets its own VecEnv, all of them use the first one's arena (so the allocation lottery of DESIGN section 2 cannot
decide the comparison), and timed episodes alternate between the settings.
    python tools/ab_knobs.py [--rounds 4] [--envs 1024 --agents 4 --grid 256] "" "IPPM_NO_HTAB=1" "IPPM_TILE_WAVES=48" ...
A setting is a space-separated list of NAME=VALUE (empty string = defaults).  Prints avg / min us per launch of the step's kernels."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
sys.path.insert(0, ROOT)
from bench import bench_params  # noqa: E402
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--agents", type=int, default=4)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--actions", type=int, default=None)
    ap.add_argument("--episode-comm-range", action="store_true")
    ap.add_argument("--team-sizes", default=None, help="e.g. 2,4,8,16: env e flies team_sizes[e % len] of the --agents UAVs")
    ap.add_argument("--tracked", action="store_true", help="the training sequence: area sums tracked, two plan launches per step")
    ap.add_argument("--draws", type=int, default=8, help="placement search on the first env before the comparison")
    ap.add_argument("settings", nargs="+")
    a = ap.parse_args()
    a.terrain = "random_field"
    pattern = [int(v) for v in a.team_sizes.split(",")] if a.team_sizes else None
    teams = [pattern[e % len(pattern)] for e in range(a.envs)] if pattern else None
    envs = []
    for setting in a.settings:
        pairs = [kv.split("=", 1) for kv in setting.split()]
        saved = {k: os.environ.get(k) for k, _ in pairs}
        for k, v in pairs:
            os.environ[k] = v
        envs.append(VecEnv(bench_params(a), a.envs, philox_seed=3, terrain="random_field", track_area=a.tracked, team_sizes=teams))
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if a.draws > 1:
        print("placement", envs[0].tune_placement(a.draws), flush=True)
    for e in envs[1:]:
        e._use_arena(envs[0]._arena)
    T = envs[0].d.budget + 1
    ids = list(range(1, a.envs + 1))

    def episode(env, timed):
        env._boxes_valid = False
        env.reset(ids)
        env.profile = timed
        for t in range(T):
            if a.tracked:
                env.build_observations(t, features=False)
            env.steps(t, policy=POLICY_UNIFORM, features=False)
        env.reset(ids)
        env.profile = False

    for env in envs:
        episode(env, False)
    acc = [dict() for _ in envs]
    for r in range(a.rounds):
        for k, env in enumerate(envs):
            episode(env, True)
            for cls, rec in env.event_times_us().items():
                s = acc[k].setdefault(cls, [0.0, 0, 1e30])
                s[0] += rec["avg_us"] * rec["launches"]
                s[1] += rec["launches"]
                s[2] = min(s[2], rec["min_us"])
    # the states must agree whatever the knobs (same episodes): a knob that changes results is a bug, not a tuning
    ref = envs[0]
    for k, env in enumerate(envs[1:], 1):
        same = torch.equal(ref.pos, env.pos) and torch.equal(ref.action, env.action)
        print(f"setting {k} vs 0: positions/actions equal {same}; max |reward diff| {float((ref.reward - env.reward).abs().max()):.3e}")
    for setting, s in zip(a.settings, acc):
        line = "  ".join(f"{cls} {v[0] / max(v[1], 1):6.1f} (min {v[2]:5.1f})" for cls, v in sorted(s.items()) if cls in ("sense", "fuse", "plan", "reset_maps", "terrain"))
        print(f"[{setting or 'defaults':40s}] {line}", flush=True)


main()
