import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("tests", "oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import ipp_oracle as O
from configs import make_params
from ippmarl.vec_env import VecEnv, POLICY_EXPLICIT
over = {'experiment__missions__n_agents': 11, 'experiment__constraints__num_actions': 27, 'experiment__uav__communication_range': 15, 'experiment__uav__failure_rate': 0.0, 'experiment__uav__fix_range': False, 'sensor__pixel__number_x': 14, 'sensor__pixel__number_y': 14}
params = make_params("small", **over)
d = O.Derived(params); d.exact = True
env = VecEnv(params, 3, philox_seed=184610582780 & 0xFFFFFFFF)
eps = np.arange(11, 14); env.reset(eps)
print("grid", d.gx, d.gy)
for t in range(4):
    env.build_observations(t, features=False)
    lo = env.local.cpu().numpy(); local = 1.0 / (1.0 + np.exp(-lo.astype(np.float64)))
    pos = env.pos.cpu().numpy()
    acts = env.ig_actions(communication=True)
    gains = env.ig_gains.cpu().numpy()
    for e in range(env.E):
        prior = []
        for i in range(d.n_agents):
            m = O.apply_collision_mask(d, pos[e, i], O.action_mask(d, pos[e, i]), prior)
            ap, g = O.ig_individual(d, pos[e, i], m, local[e, i].astype(np.float64))
            bad = np.nonzero(~np.isclose(gains[e, i], g, rtol=1e-5, atol=1e-9))[0]
            for a in bad:
                newpos = O.action_to_position(d, pos[e, i], int(a))
                full, fc = O.project_field_of_view(d, newpos)
                print("t", t, "e", e, "i", i, "a", a, "got", gains[e, i, a], "want", g[a], "pos", pos[e, i], "->", newpos, "rect", fc)
                yu, yd, xl, xr = fc
                sec = lo[e, i][xl:xr, yu:yd]
                print("   section shape", sec.shape, "logodds min/max", sec.min(), sec.max(), "n nonzero", (sec != 0).sum(), "vals", np.unique(np.round(sec, 4))[:12])
            prior.append(pos[e, i])
    env.steps(t, policy=POLICY_EXPLICIT, actions=acts, features=False)
