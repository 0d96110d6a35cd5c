#!/usr/bin/env python
"""K4/K5 timing with non-overlapping vs overlapping footprints (is the generic multi-op path the slow one?)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("oracle", "ipp-marl_amd"):
    sys.path.insert(0, os.path.join(ROOT, sub))
import torch
from ippmarl.params import grid256_params
from ippmarl.vec_env import VecEnv

E = 1024
env = VecEnv(grid256_params(), E)
layouts = {"disjoint": [[10, 10, 15], [10, 40, 15], [40, 10, 15], [40, 40, 15]],
           "disjoint10": [[10, 10, 10], [10, 40, 10], [40, 10, 10], [40, 40, 10]],
           "disjoint5": [[10, 10, 5], [10, 40, 5], [40, 10, 5], [40, 40, 5]],
           "near5": [[25, 25, 5], [25, 30, 5], [30, 25, 5], [30, 30, 5]],
           "stacked": [[25, 25, 15], [25, 30, 15], [30, 25, 15], [30, 30, 15]],
           "random": None}
for name, lay in layouts.items():
    sp = None if lay is None else torch.tensor([lay] * E, dtype=torch.int32)
    env.reset(torch.arange(1, E + 1), start_positions=sp)
    env.comm_matrix(0)
    res = {}
    for kname, fn in (("K4", lambda: env.fuse_local()), ("K5", lambda: env._launch_k5(env.stream))):
        fn(); torch.cuda.synchronize()
        env.counters(reset=True)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(6):
            fn()
        b.record(); torch.cuda.synchronize()
        c = env.counters()
        cells = (c["fuse_local_cells"] if kname == "K4" else c["fuse_global_cells"]) / 6
        res[kname] = {"us": round(a.elapsed_time(b) * 1e3 / 6, 1), "Mcells": round(cells / 1e6, 2)}
    print(name, json.dumps(res))
