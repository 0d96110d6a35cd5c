"""Random-walk comparison mission with the reference's class name and interface (random_baseline.py:24-124):
``RandomBaseline(params, writer, num_episode).execute() -> (0, entropies, f1s)``.

All platforms update ONE shared map directly (no fusion step, no collision mask): t = 0 senses at the seeded start
cells, every later step draws an action uniformly from the boundary mask.  Sensing (K3) and both metrics run on the GPU;
the action draw stays ``torch.multinomial`` on the host as in the reference (its stream is unpinned anyway)."""
from __future__ import annotations

from typing import Dict

import torch

from ._mission import MissionMetrics, save_mission_numbers  # noqa: F401  (re-exported like the reference module)
from .agent.agent import Agent
from .agent.state_space import AgentStateSpace
from .coma_wrapper import COMAWrapper, ReplayHooks
from .mapping.grid_maps import GridMap
from .mapping.mappings import Mapping
from .sensors import Sensor
from .sensors.models import SensorModel


class RandomBaseline(MissionMetrics):
    def __init__(self, params: Dict, writer, num_episode):
        self.params = params
        self.num_episode = num_episode
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
        self.replay = None      # optional ReplayHooks(correctness=(agent, stage) -> draws, action=(agent, t) -> index)
        self.f1_bracket = []

    def execute(self):
        hooks = self.replay or ReplayHooks()
        agents = [Agent(self.coma_wrapper.actor_network, self.params, self.mapping, i, self.agent_state_space)
                  for i in range(self.n_agents)]
        self.agents = agents
        shared = self._shared_map()            # engine slot 0 holds the one map (prior at this point)
        entropy, f1 = self._metrics(shared)
        entropies, f1s, rewards = [entropy], [f1], []
        for t in range(self.budget + 1):
            for i, agent in enumerate(agents):
                if t == 0:
                    agent.position = self.agent_state_space.get_random_agent_state(i, self.num_episode)
                else:
                    action_mask, _ = agent.action_space.get_action_mask(agent.position)
                    action = hooks.action(i, t)
                    if action is None:
                        action = torch.multinomial(torch.tensor(action_mask).float(), 1, replacement=True).item()
                    agent.position = agent.action_space.action_to_position(agent.position, int(action))
                self._shared_sense(agent.position, hooks.correctness(i, t))
            entropy, f1 = self._metrics(shared)
            entropies.append(entropy)
            f1s.append(f1)
            rewards.append(0)
        self.map = self.mapping.engine.get_local(0)
        return sum(rewards), entropies, f1s


def main(config_path=None, out_path="random_f1.json"):
    from .params import load_params
    params = load_params(config_path)
    n_episodes = params["experiment"]["baselines"]["random"]["n_episodes"]
    budget = params["experiment"]["constraints"]["budget"]
    entropies_list, f1_list = [], []
    for episode in range(1, n_episodes + 1):
        _, entropies, f1s = RandomBaseline(params, None, episode).execute()
        entropies_list.append(entropies)
        f1_list.append(f1s)
    return save_mission_numbers(entropies_list, f1_list, n_episodes, budget, out_path)


if __name__ == "__main__":
    main()
