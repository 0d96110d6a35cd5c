"""Times the stages of the device terrain generator (E=1024, 256x256)."""
import json, sys, os
sys.path[:0] = [os.path.join(os.path.dirname(__file__), "..", "ipp-marl_amd")]
import torch
from ippmarl import _ffi
from ippmarl.params import grid256_params
from ippmarl.vec_env import VecEnv

env = VecEnv(grid256_params(), 1024, terrain="random_field")
env.reset(torch.arange(1, 1025))
g = env._field
E, gx, gy = 1024, env.d.grid_x, env.d.grid_y

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

out = {}
out["generate_us"] = timeit(lambda: g.generate(env.episode, env.truth, env.stream))
out["reset_split_us"] = timeit(lambda: env.reset(torch.arange(1, 1025), terrain="split"))
out["reset_field_us"] = timeit(lambda: env.reset(torch.arange(1, 1025)))
print(json.dumps(out))
