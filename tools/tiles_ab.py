"""A/B of the fusion / sensing kernel forms on identical episodes: a = one-trip tile items, no area sums (the env-only step);
b = tile items with the area sums tracked (rollouts that build network inputs); c = the row walker and k_sense_update with the
area sums tracked (IPPM_NO_TILES=1 IPPM_K3_CLASSIC=1: round 2's kernels, still the path of prior != 0.5 and narrow grids).
Maps must agree bit for bit, rewards to float64 summation order, the area sums of b and c to 3e-7 of an area average.
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
os.environ["IPPM_NO_TILES"] = "1"
os.environ["IPPM_K3_CLASSIC"] = "1"    # (read at every ippm_sense_step: set before c's first launch, cleared around a's and b's)
c = VecEnv(params, E, philox_seed=11, track_area=True)
del os.environ["IPPM_NO_TILES"]
del os.environ["IPPM_K3_CLASSIC"]
bad = 0


def classic(on):
    if on:
        os.environ["IPPM_K3_CLASSIC"] = "1"
    else:
        os.environ.pop("IPPM_K3_CLASSIC", None)


G = float(a.d.grid_x * a.d.grid_y)
for ep in range(n_ep):
    ids = list(range(1 + ep * E, 1 + (ep + 1) * E))
    a.reset(ids)
    b.reset(ids)
    classic(True)
    c.reset(ids)
    classic(False)
    for t in range(a.d.budget + 1):
        ra, _, _ = a.steps(t, policy=POLICY_UNIFORM, features=False)
        b.build_observations(t)
        rb, _, _ = b.steps(t, policy=POLICY_UNIFORM)
        classic(True)
        c.build_observations(t)
        rc, _, _ = c.steps(t, policy=POLICY_UNIFORM)
        classic(False)
        torch.cuda.synchronize()
        da = float((b.area - c.area).abs().max()) / G
        if da > 3e-7 or not torch.allclose(rb, rc, rtol=1e-6, atol=1e-7) or not torch.allclose(b.obs, c.obs, rtol=1e-5, atol=2e-6):
            print(f"ep {ep} t {t}: tracked tiles vs walker: area averages differ by {da:.2e}, rewards by {float((rb - rc).abs().max()):.2e}, "
                  f"observations by {float((b.obs - c.obs).abs().max()):.2e}")
            bad += 1
        for nm, x, y in (("local", a.local, b.local), ("glob", a.glob, b.glob), ("pos", a.pos, b.pos),
                         ("local (walker)", a.local, c.local), ("glob (walker)", a.glob, c.glob)):
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
ca, cb, cc = a.counters(), b.counters(), c.counters()
print("counters tiles          ", ca)
print("counters tiles, tracked ", cb)
print("counters walker, tracked", cc)
print("A/B", "FAILED" if bad else "OK", f"({name}, {E} envs, {n_ep} episodes)")
sys.exit(1 if bad else 0)
