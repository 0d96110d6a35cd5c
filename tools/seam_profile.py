"""cProfile of the drop-in seam (EpisodeGenerator.execute over the reference-shaped objects, one env): where the host time of an
agent-env step goes.    python tools/seam_profile.py [episodes]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ipp-marl_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ippmarl.batch_memory import BatchMemory  # noqa: E402
from ippmarl.coma_wrapper import COMAWrapper  # noqa: E402
from ippmarl.mapping.grid_maps import GridMap  # noqa: E402
from ippmarl.missions.episode_generator import EpisodeGenerator  # noqa: E402
from ippmarl.params import grid256_params  # noqa: E402
from ippmarl.sensors import Sensor  # noqa: E402
from ippmarl.sensors.models import SensorModel  # noqa: E402

episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 5
params = grid256_params(experiment__missions__n_agents=4)
np.random.seed(7)
torch.manual_seed(7)
wrapper = COMAWrapper(params, None)
grid_map = GridMap(params)
gen = EpisodeGenerator(params, None, grid_map, Sensor(SensorModel(), grid_map))
gen.execute(1, BatchMemory(params, wrapper), wrapper, "train")
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for ep in range(2, 2 + episodes):
    gen.execute(ep, BatchMemory(params, wrapper), wrapper, "train")
torch.cuda.synchronize()
pr.disable()
dt = time.perf_counter() - t0
print(f"{episodes * 60 / dt:.0f} agent-env steps/s under the profiler, {1e3 * dt / episodes:.1f} ms per episode")
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(25)
