"""A/B of the two fusion kernels on identical episodes: VecEnv(track_area=False) takes the one-trip tile form (fuse_tiles.hip),
VecEnv(track_area=True) the row walker (fuse.hip).  Maps must agree bit for bit, rewards to float64 summation order.
    python tools/tiles_ab.py [config] [envs] [episodes]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
from configs import make_params  # noqa: E402
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
E = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n_ep = int(sys.argv[3]) if len(sys.argv) > 3 else 2
params = make_params(name)
a = VecEnv(params, E, philox_seed=11, track_area=False)
b = VecEnv(params, E, philox_seed=11, track_area=True)
bad = 0
for ep in range(n_ep):
    ids = list(range(1 + ep * E, 1 + (ep + 1) * E))
    a.reset(ids)
    b.reset(ids)
    for t in range(a.d.budget + 1):
        ra, _, _ = a.steps(t, policy=POLICY_UNIFORM, features=False)
        b.build_observations(t)
        rb, _, _ = b.steps(t, policy=POLICY_UNIFORM)
        torch.cuda.synchronize()
        for nm, x, y in (("local", a.local, b.local), ("glob", a.glob, b.glob), ("pos", a.pos, b.pos)):
            if not torch.equal(x, y):
                d = (x != y).nonzero()
                print(f"ep {ep} t {t}: {nm} differs in {len(d)} cells; first {d[:6].tolist()}")
                i = tuple(d[0].tolist())
                print("   tiles", float(x[i]), "walker", float(y[i]))
                bad += 1
        if not torch.allclose(ra, rb, rtol=1e-6, atol=1e-7):
            k = int((ra - rb).abs().sum(1).argmax())
            print(f"ep {ep} t {t}: reward differs, worst env {k}: tiles {ra[k].tolist()} walker {rb[k].tolist()}")
            bad += 1
        if bad > 6:
            break
    if bad > 6:
        break
ca, cb = a.counters(), b.counters()
print("counters tiles ", ca)
print("counters walker", cb)
print("A/B", "FAILED" if bad else "OK", f"({name}, {E} envs, {n_ep} episodes)")
sys.exit(1 if bad else 0)
