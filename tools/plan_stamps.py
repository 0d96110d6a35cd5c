"""Phase stamps of k_plan_step (variant library built with -DIPPM_PLAN_STAMPS): IPPMARL_LIB=.../libippmarl_stamps.so python tools/plan_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ipp-marl_amd"))
sys.path.insert(0, ROOT)
from bench import bench_params  # noqa: E402
from ippmarl.vec_env import VecEnv, POLICY_UNIFORM  # noqa: E402


class A:   # python tools/plan_stamps.py [envs agents grid [actions [episode_comm_range]]]
    envs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    agents = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    grid = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    actions = int(sys.argv[4]) if len(sys.argv) > 4 else None
    episode_comm_range = len(sys.argv) > 5 and sys.argv[5] not in ("0", "")


env = VecEnv(bench_params(A), A.envs, philox_seed=3, terrain="split", track_area=False)
env.reset(list(range(1, A.envs + 1)))
names = ["start", "loaded", "plans done", "builder start", "builder done", "K1 start", "K1 done", "comm done"]
for t in range(8):
    env.steps(t, policy=POLICY_UNIFORM, features=False)
    raw = np.zeros(512, dtype=np.uint64)
    env.ctx.lib.ippm_debug_raw_counters(env.ctx.handle, C.c_void_p(raw.ctypes.data))
    st = raw.reshape(64, 8)[:, 7].astype(np.int64)
    t0 = st[0]
    line = []
    for wv in (0, 1, 2, 3, 4, 5, 6, 7):   # 6 / 7: wavefront 0 of the last / the middle env; 2 .. 5: builders
        ks = {0: (0, 1, 7, 2, 3, 4), 6: (0, 1, 7, 2, 4), 7: (0, 1, 7, 2, 4), 1: (0, 5, 6)}.get(wv, (0, 3, 4))
        line.append(f"w{wv}: " + " ".join(f"{names[k]}={(st[wv * 8 + k] - t0) / 100.0:.1f}us" for k in ks if st[wv * 8 + k]))
    print(f"t={t}  " + " | ".join(line))
