import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ipp-marl_amd")); sys.path.insert(0, ROOT)
from bench import bench_params
from ippmarl.vec_env import VecEnv
class A: agents, grid, actions, terrain = 4, 256, None, "random_field"
env = VecEnv(bench_params(A), int(sys.argv[1]) if len(sys.argv) > 1 else 512, philox_seed=3, terrain="random_field", track_area=False)
r = env.tune_placement(24)
print("JUMP_GB", os.environ.get("IPPM_PLACEMENT_JUMP_GB", "12"), "NO_EARLY", os.environ.get("IPPM_PLACEMENT_NO_EARLY"), r["stopped"], "jumps", r["jumps"], r["map_kernels_us_per_step"], flush=True)
