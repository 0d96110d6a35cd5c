import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import torch
from ippmarl.params import grid256_params
from configs import make_params
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM
name, E, track, terrain = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
params = grid256_params() if name == "c2" else make_params(name)
env = VecEnv(params, E, track_area=bool(track), terrain=terrain)
env.reset(torch.arange(1, E + 1))
env.capture_step_graphs(POLICY_UNIFORM)
for ep in range(4):
    for t in range(env.d.budget + 1):
        env.step_graphed(t)
    env.reset(torch.arange(1, E + 1) + 1000 * (ep + 1))
torch.cuda.synchronize()
print("ok", name, E, track, terrain, float(env.reward.sum()))
