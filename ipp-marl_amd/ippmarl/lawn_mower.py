"""Fixed coverage-path comparison mission with the reference's class name and interface (lawn_mower.py:23-315):
``LawnMower(params, writer, num_episode).execute() -> (0, entropies, f1s)``.

Eight platforms fly fixed 15-waypoint boustrophedon paths at ``experiment.baselines.lawnmower.altitude`` and update ONE
shared map in turn, whatever ``n_agents`` says (the reference hard-codes the eight paths for its 50 m x 50 m world:
two row-wise and two column-wise sweeps, each listed twice).  Sensing (K3) and the metrics run on the GPU."""
from __future__ import annotations

from typing import Dict, List

import numpy as np

from ._mission import MissionMetrics, save_mission_numbers  # noqa: F401
from .agent.state_space import AgentStateSpace
from .coma_wrapper import COMAWrapper, ReplayHooks
from .mapping.grid_maps import GridMap
from .mapping.mappings import Mapping
from .sensors import Sensor
from .sensors.models import SensorModel


def sweep(fixed_start: int, along_x: bool, altitude: int) -> np.ndarray:
    """out along one lane from 10 m to 40 m in 5 m hops, one hop sideways, back along the lane 10 m further over: 15 cells."""
    run = list(range(10, 45, 5))
    a = [(v, fixed_start) for v in run] + [(40, fixed_start + 5)] + [(v, fixed_start + 10) for v in reversed(run)]
    return np.array([[p, q, altitude] if along_x else [q, p, altitude] for p, q in a])


def coverage_paths(altitude: int) -> List[np.ndarray]:
    """The reference's positions1..8 (lawn_mower.py:46-203)."""
    four = [sweep(10, True, altitude), sweep(30, True, altitude), sweep(10, False, altitude), sweep(30, False, altitude)]
    return four + [p.copy() for p in four]


class LawnMower(MissionMetrics):
    def __init__(self, params: Dict, writer, num_episode):
        self.params = params
        self.n_agents = params["experiment"]["missions"]["n_agents"]
        self.x_dim = params["environment"]["x_dim"]
        self.y_dim = params["environment"]["y_dim"]
        self.altitude = params["experiment"]["baselines"]["lawnmower"]["altitude"]
        self.budget = params["experiment"]["constraints"]["budget"]
        self.coma_wrapper = COMAWrapper(params, writer)
        self.grid_map = GridMap(params)
        self.sensor = Sensor(SensorModel(), self.grid_map)
        self.mapping = Mapping(self.grid_map, self.sensor, params, num_episode)
        self.agent_state_space = AgentStateSpace(params)
        self.map = self.mapping.init_priors()
        self.replay = None
        self.f1_bracket = []

    def execute(self):
        hooks = self.replay or ReplayHooks()
        paths = coverage_paths(self.altitude)
        shared = self._shared_map()
        self.mapping.engine.set_local(0, np.asarray(self.map))
        entropy, f1 = self._metrics(shared)
        entropies, f1s, rewards = [entropy], [f1], []
        for idx in range(len(paths[0])):
            for k, path in enumerate(paths):
                self._shared_sense(path[idx], hooks.correctness(k, idx))
            entropy, f1 = self._metrics(shared)
            entropies.append(entropy)
            f1s.append(f1)
            rewards.append(0)
        self.map = self.mapping.engine.get_local(0)
        return sum(rewards), entropies, f1s


def main(config_path=None, out_path="lawnmower_f1.json"):
    from .params import load_params
    params = load_params(config_path)
    trials = params["experiment"]["baselines"]["lawnmower"]["trials"]
    budget = params["experiment"]["constraints"]["budget"]
    entropies_list, f1_list = [], []
    for trial in range(1, trials + 1):
        _, entropies, f1s = LawnMower(params, None, trial).execute()
        entropies_list.append(entropies)
        f1_list.append(f1s)
    return save_mission_numbers(entropies_list, f1_list, trials, budget, out_path)


if __name__ == "__main__":
    main()
