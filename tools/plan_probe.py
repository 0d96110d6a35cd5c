"""k_plan_step with one flag combination, 60 launches (for rocprofv3 --kernel-trace): python tools/plan_probe.py <flags> [work 0/1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ipp-marl_amd"))
import torch
from ippmarl import _ffi
from ippmarl.params import grid256_params
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM
flags, use_work = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 1
env = VecEnv(grid256_params(), int(os.environ.get("ENVS", 1024)), track_area=False)
env.reset(torch.arange(1, env.E + 1))
for t in range(5):
    env.steps(t, policy=POLICY_UNIFORM, features=False)
torch.cuda.synchronize()
pos0 = env.pos.clone()
for i in range(60):
    env.pos.copy_(pos0)
    env.ctx.call("ippm_plan_step", env._p(env.episode), env._p(env.pos), env._p(env.comm_range), None, env._p(env.comm), env._p(env.rect),
                 env._p(env.ws), 5, flags, None, None, POLICY_UNIFORM, env._p(env.mask), env._p(env.action),
                 env._p(env.fault), env._p(env.rect_next), env._p(env.work) if use_work else None, env.E, env.stream)
torch.cuda.synchronize()
print("done", flags, use_work)
